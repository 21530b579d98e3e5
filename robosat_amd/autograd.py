"""Train-mode forward + hand-scheduled backward of the U-Net as ONE ``torch.autograd.Function``.

PyTorch's role here is bookkeeping only: it sees a single differentiable node ``logits = f(x, *parameters)``, calls our
``backward`` when the reference-style training loop does ``loss.backward()`` (robosat/tools/train.py:186) and hands
the returned gradients to ``Adam``.  Every FLOP of the 61 convolutions, 53 BatchNorms, pools and the head -- forward,
data-gradient and weight-gradient -- runs in the HIP kernels behind ``robosat_amd.ops``.

Backward structure (no tensor is ever upsampled or concatenated in memory, in either direction):
  * weight gradients: ``rs_conv2d_wgrad`` re-reads the forward input through the forward gather;
  * data gradients:   ``rs_conv2d_fwd`` on dy with tap-flipped, transposed weights (``ups=2`` = stride-2 adjoint);
    the ReLU backward of the producing layer and the residual/skip gradient sums ride in that kernel's epilogue;
  * decoder: the data gradient lands at the upsampled resolution and ``rs_upsample2x_bwd`` folds 2x2 sums + the
    ``torch.cat`` split + the ReLU masks into one pass.
"""

import os

import torch

from . import ops


_SIDE = {}


def _side_stream(device):
    """One extra HIP stream per device for the weight-gradient kernels: a layer's wgrad and dgrad both only read dy, so
    the wgrad (and its split reduction) runs beside the dgrad -> BatchNorm chain that is the backward's critical path and
    fills the CUs the short encoder kernels leave idle."""

    if torch.device(device).type != "cuda":
        return None  # (host tensors -- the 8-rank gloo test of the arena / reducer bookkeeping, tests/dp_worker.py: no streams to order)
    if os.environ.get("ROBOSAT_WGRAD_STREAM", "1") == "0":  # measurement knob: serial backward (clean per-kernel timings)
        return torch.cuda.current_stream(device)
    s = _SIDE.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE[device] = s
    return s


class GradArena:
    """All parameter gradients of one backward pass live in ONE flat fp32 buffer, carved sequentially in the order
    the backward produces them (head, decoder, layer4 ... stem).  The kernels write straight into their slice, so a
    finished prefix of the buffer is a ready-made all-reduce bucket: ``flush()`` hands the newly completed range to the
    data-parallel reducer (``robosat_amd.parallel.GradReducer``), which sums it over RCCL while the rest of the
    backward keeps computing -- no flatten/unflatten copies, no per-tensor collectives."""

    # measurement hook (tests/dp_worker.py): a list -> every arena appends a dict: `events` = (start, main_done) timing events
    # recorded on the main stream when the backward starts and right before its one join with the side stream at the end;
    # `joins` = how often the main stream was made to wait for the side stream; `flushes` = per bucket, whether the collective
    # was issued with the side stream current (i.e. ordered after it, not after the main stream)
    TRACE = None

    def __init__(self, params, device, reducer=None):
        # sized for the parameters that receive a gradient: `params` excludes resnet.fc (never used by UNet.forward).
        # Slices are 16-byte aligned; with a reducer the <= 3 floats of padding between them travel over the wire, so the
        # buffer is zero-filled then (uninitialised padding could hold NaNs) -- a 150 MB memset, ~30 us.
        total = sum((p.numel() + 3) // 4 * 4 for p in params if p.requires_grad)
        alloc = torch.zeros if reducer is not None else torch.empty
        self.flat = alloc(total, device=device, dtype=torch.float32)
        self.side = _side_stream(device)
        if self.side is not None:
            self.flat.record_stream(self.side)  # written by the wgrad kernels on the side stream
        self.off = 0
        self.sent = 0
        self.reducer = reducer
        self.grads = {}
        self._t0, self.joins, self.flushes = None, 0, []
        if GradArena.TRACE is not None and self.side is not None:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(torch.cuda.current_stream(device))

    def take(self, param, shape):
        """Next 16-byte aligned slice, shaped ``shape`` (the kernel's layout, e.g. KRSC), registered for ``param``."""

        n = 1
        for d in shape:
            n *= d
        assert n == param.numel() and self.off + n <= self.flat.numel()
        v = self.flat[self.off:self.off + n].view(shape)
        self.off += (n + 3) // 4 * 4
        self.grads[param] = v
        return v

    def conv(self, conv):
        """Slice for a conv weight gradient in KRSC; autograd receives the logical [Cout,Cin,kh,kw] view of it."""

        w = conv.weight
        v = self.take(w, (w.shape[0], w.shape[2], w.shape[3], w.shape[1]))
        self.grads[w] = v.permute(0, 3, 1, 2)
        return v

    def wgrad(self, fn, *tensors):
        """Run ``fn`` (weight-gradient launches reading ``tensors``) on the side stream, after everything enqueued so far."""

        if self.side is None:
            fn()
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()
        for t in tensors:
            if t is not None:
                t.record_stream(self.side)  # the caching allocator must not recycle them under the side stream

    def join(self):
        self.joins += 1
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def flush(self):
        """Hand the range completed since the last flush to the reducer -- ordered after the SIDE stream.

        A bucket is complete when (a) the side stream has finished its weight gradients and (b) the main stream has written
        its BatchNorm / bias gradients: the side stream is made to wait for an event recorded on the main stream now (b), it
        is itself the stream the weight gradients run on (a), and the collective is issued with the side stream current, so
        RCCL's communicator stream waits for exactly that.  The main stream -- the backward's critical path -- neither
        waits for the side stream (which trails it by milliseconds) nor for the wire: it goes on with the next layer's data
        gradient.  (Round 2 joined the side stream into the main stream at each of the five flushes.)"""

        if self.reducer is not None and self.off > self.sent:
            if self.side is None:  # host tensors: nothing to order
                self.flushes.append(False)
                self.reducer.reduce_async(self.flat[self.sent:self.off])
                self.sent = self.off
                return
            main = torch.cuda.current_stream()
            if self.side != main:
                ev = torch.cuda.Event()
                ev.record(main)
                self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                self.flushes.append(torch.cuda.current_stream() == self.side)
                self.reducer.reduce_async(self.flat[self.sent:self.off])
            self.sent = self.off

    def finish(self):
        self.flush()
        done = None
        if self._t0 is not None:  # (before the waits below: gloo's wait blocks the HOST, which would delay this record)
            done = torch.cuda.Event(enable_timing=True)
            done.record(torch.cuda.current_stream())
        if self.reducer is not None:
            if self.side is None:
                self.reducer.wait()
            else:
                with torch.cuda.stream(self.side):
                    self.reducer.wait()  # the side stream waits for the collectives (+ the bf16 wire's casts back run on it)
        self.join()  # once per step: the optimizer (main stream) needs every gradient
        if done is not None:
            GradArena.TRACE.append({"events": (self._t0, done), "joins": self.joins, "flushes": list(self.flushes)})


class _Tape:
    """Forward state kept for the backward pass."""

    def __init__(self):
        self.blocks = []  # one dict per bottleneck, forward order
        self.t = {}


def _bn_train(bn, y, residual=None, relu=True, partial=None, want_bits=False):
    """Train-mode BatchNorm (+ residual, ReLU).  ``partial``: the statistics' partial sums when the producing convolution
    already reduced them in its epilogue (``ops.conv2d_bnstats``) -- otherwise a read pass over ``y`` computes them.
    ``want_bits``: also returns z's ReLU mask as one bit per element (None where the layer's width has no bit form): what the
    data-gradient epilogue that ends at this layer reads instead of z itself."""

    if partial is None:
        mean, invstd, scale, shift = ops.bn_train_stats(
            y, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked)
    else:
        mean, invstd, scale, shift = ops.bn_finalize_stats(
            partial, y.numel() // y.shape[-1], bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, bn.running_mean,
            bn.running_var, bn.num_batches_tracked)
    bn._folded = None
    if want_bits:
        if relu and ops.bn_bits_ok(y.shape[-1]):
            z, bits = ops.bn_apply(y, scale, shift, residual=residual, relu=relu, want_bits=True)
        else:
            z, bits = ops.bn_apply(y, scale, shift, residual=residual, relu=relu), None
        return z, (mean, invstd), bits
    z = ops.bn_apply(y, scale, shift, residual=residual, relu=relu)
    return z, (mean, invstd)


def _conv_bn(bn, src, w, stride=1, pad=0, residual=None, relu=True, conv=None):
    """conv -> train-mode BatchNorm (-> + residual -> ReLU) with the statistics fused into the convolution: returns
    (y, z, (mean, invstd), ReLU mask bits of z or None).  ``conv``: the module, for the stride-1 3x3 layers of the fp32 path, which
    run the Winograd F(2x2, 3x3) form with the same statistics epilogue (4/9 of the multiply-adds)."""

    if (conv is not None and src.dtype == torch.float32 and stride == 1 and pad == 1 and ops.wino33_ok(src, conv.cout)
            and os.environ.get("ROBOSAT_WINO33_STATS", "1") != "0"):
        y, partial = ops.conv2d_wino33_bnstats(src, conv.wino33())
    else:
        y, partial = ops.conv2d_bnstats(src, w, stride=stride, pad=pad)
    z, st, bits = _bn_train(bn, y, residual=residual, relu=relu, partial=partial, want_bits=True)
    return y, z, st, bits


def _prefetch_decoder_weights(net, dt, device):
    """The decoder's derived weights for this step -- the phase-form filters of center / dec0..dec4 and the weights of their
    4x4 / stride-2 data gradients, 12 small packing launches -- issued on the side stream, where they run beside the stem
    and the encoder instead of in front of each decoder layer (forward) and each data gradient (backward).  Returns the
    event the main stream waits for before the first decoder layer (None when the side stream is switched off)."""

    main = torch.cuda.current_stream(device)
    side = _side_stream(device)
    if side == main:
        return None
    ready = torch.cuda.Event()
    ready.record(main)  # (the optimizer step that wrote the master weights is on the main stream)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        packed = []
        for blk in (net.center, net.dec0, net.dec1, net.dec2, net.dec3, net.dec4):
            conv = blk.block.block
            packed += [conv.phase(dt), conv.dgrad_phase(dt)]
            # fp32: the Winograd forms' transformed filters, for the layers whose geometry took those forms in the previous step (the
            # first step packs them where they are used)
            if dt == torch.float32 and getattr(conv, "_phase_wino", None) is not None:
                packed.append(conv.phase_wino())
            if dt == torch.float32 and getattr(conv, "_dgrad_phase_wino", None) is not None:
                packed.append(conv.dgrad_phase_wino())
    done = torch.cuda.Event()
    done.record(side)
    for w in packed:
        w.record_stream(main)  # allocated under the side stream, read by the main stream's kernels
    return done


def _forward(net, x, tape, backward=False):
    from .unet import _bump_generation

    _bump_generation()  # a training forward re-derives every compute copy of the weights (see unet._GENERATION)
    decoder_weights_ready = _prefetch_decoder_weights(net, net.compute_dtype, x.device)
    r = net.resnet
    t = tape.t
    dt = net.compute_dtype  # fp32 or bf16 activations (the image is cast on upload)
    if dt == torch.bfloat16:
        net.prep_bf16_weights()  # one launch: bf16 casts + data-gradient layouts of the encoder weights for this step
    elif backward:
        net.prep_f32_weights()  # one launch: the fp32 data-gradient layouts (only a step that has a backward needs them)
    x4 = ops.nchw_to_nhwc4(x, dt)
    t["x4"] = x4
    if dt == torch.bfloat16:
        y0 = ops.stem_conv_bf16(x4, ops.pack_stem_weight(r.conv1.krsc(), dt))
    else:
        y0 = ops.conv2d(x4, ops.pack_stem_weight(r.conv1.krsc()), stride=2, pad=3, stem=7, bands=net.in_channels)
    z0, st0 = _bn_train(r.bn1, y0)
    p0, am0 = ops.maxpool2d(z0, 3, 2, 1, want_argmax=True)
    t.update(y0=y0, st0=st0, z0=z0, am0=am0)

    h = p0
    enc = []
    for layer in net._blocks():
        for blk in layer:
            rec = {"blk": blk, "h": h}
            y1, z1, rec["st1"], rec["b1"] = _conv_bn(blk.bn1, h, blk.conv1.krsc(dt))
            y2, z2, rec["st2"], rec["b2"] = _conv_bn(blk.bn2, z1, blk.conv2.krsc(dt), stride=blk.stride, pad=1, conv=blk.conv2)
            if blk.downsample is not None:
                yd, idt, rec["std"], _ = _conv_bn(blk.downsample[1], h, blk.downsample[0].krsc(dt), stride=blk.stride, relu=False)
                rec["yd"] = yd
            else:
                idt = h
            y3, z3, rec["st3"], rec["b3"] = _conv_bn(blk.bn3, z2, blk.conv3.krsc(dt), residual=idt, relu=True)
            rec.update(y1=y1, z1=z1, y2=y2, z2=z2, y3=y3, z3=z3)
            tape.blocks.append(rec)
            h = z3
        enc.append(h)
    enc1, enc2, enc3, enc4 = enc

    def up(block, skip, prev=None):  # DecoderBlock on the source grid: Winograd form in fp32, phase form in bf16
        from .unet import decoder_block

        return decoder_block(block.block.block, skip, prev, dt)

    pooled, amc = ops.maxpool2d(enc4, 2, 2, 0, want_argmax=True)
    if decoder_weights_ready is not None:
        torch.cuda.current_stream(x.device).wait_event(decoder_weights_ready)
    center = up(net.center, pooled)
    dec0 = up(net.dec0, enc4, center)
    dec1 = up(net.dec1, enc3, dec0)
    dec2 = up(net.dec2, enc2, dec1)
    dec3 = up(net.dec3, enc1, dec2)
    dec4 = up(net.dec4, dec3)
    if dt == torch.float32 and ops.wino33_ok(dec4, net.dec5.block.cout) and os.environ.get("ROBOSAT_WINO33_BWD", "1") != "0":
        dec5 = ops.conv2d_wino33(dec4, net.dec5.block.wino33(), relu=True)  # (as the eval forward runs it)
    else:
        dec5 = ops.conv2d(dec4, net.dec5.block.krsc(dt), pad=1, relu=True)
    t.update(enc=enc, pooled=pooled, amc=amc, center=center, dec0=dec0, dec1=dec1, dec2=dec2, dec3=dec3, dec4=dec4, dec5=dec5)
    wf = net.final.weight.detach().reshape(net.num_classes, -1)
    return ops.final_conv1x1(dec5, wf, net.final.bias.detach())


def _dgrad(dy, conv, out_hw, residual=None, relu_mask=None):
    """Data gradient of ``conv`` (a parameter holder with .k/.stride/.padding) evaluated at dy (fp32 or bf16)."""

    wd = conv.dgrad_weight(dy.dtype)
    return ops.conv2d(dy, wd, ups=2 if conv.stride == 2 else 0, pad=conv.k - 1 - conv.padding, out_hw=out_hw,
                      residual=residual, relu_mask=relu_mask)


def _backward(net, tape, dlogits, arena):
    t = tape.t
    enc1, enc2, enc3, enc4 = t["enc"]

    def bn_grads(bn):
        return {"dgamma": arena.take(bn.weight, (bn.num_features,)), "dbeta": arena.take(bn.bias, (bn.num_features,))}

    # ---- head + decoder -----------------------------------------------------------------------------------
    wf = net.final.weight.detach().reshape(net.num_classes, -1)
    dwf = arena.take(net.final.weight, tuple(wf.shape))
    arena.grads[net.final.weight] = dwf.view(net.num_classes, -1, 1, 1)
    d5, _, _ = ops.final_conv1x1_bwd(t["dec5"], wf, dlogits, relu_mask=True, dw=dwf,
                                     db=arena.take(net.final.bias, (net.num_classes,)))

    c5 = net.dec5.block
    w5 = arena.conv(c5)
    arena.wgrad(lambda: ops.conv2d_wgrad(d5, t["dec4"], 3, 3, pad=1, out=w5), d5, t["dec4"])
    if d5.dtype == torch.float32 and ops.wino33_dgrad_ok(d5, c5.cin):  # fp32: the Winograd F(2x2, 3x3) form (4/9 of the multiply-adds)
        d4, _ = ops.conv2d_wino33_dgrad(d5, c5.dgrad_wino33(), relu_mask=t["dec4"])
    else:
        d4 = ops.conv2d(d5, c5.dgrad_weight(d5.dtype), pad=1, relu_mask=t["dec4"])
    del d5

    def up_bwd(block, dz, skip, prev, mask_skip, mask_prev, skip_grad_out=None):
        """dz = gradient at the block's conv output (ReLU already applied).  Returns (d skip, d prev)."""
        conv = block.block.block
        wo = arena.conv(conv)
        arena.wgrad(lambda: ops.conv2d_wgrad(dz, skip, 3, 3, src2=prev, ups=1, pad=1, out=wo), dz, skip, prev)
        # phase form: the gradient wrt the pre-upsample tensors is ONE 4x4 / stride-2 convolution over dz (the 2x2 sum of
        # interpolate's backward is folded into pre-summed taps): 4/9 of the MACs, output already at source resolution
        hw = (skip.shape[1], skip.shape[2])
        c1 = skip.shape[3]
        c2 = 0 if prev is None else prev.shape[3]
        if dz.dtype == torch.float32 and ops.wino_dgrad_ok(dz.shape[0], hw[0], hw[1], c1, c2, dz.shape[3]):
            # fp32: the same gradient through the Winograd F(2x2, 2x2) machinery of the forward (9/16 of the 4x4 form's multiply-adds)
            u = conv.dgrad_phase_wino()
            if prev is not None and skip_grad_out is None and c1 % 64 == 0:
                return ops.conv2d_dgrad_phase_wino(dz, u, c1, c2, mask1=mask_skip, mask2=mask_prev, split=True)
            if prev is None and skip_grad_out is None:
                return ops.conv2d_dgrad_phase_wino(dz, u, c1, 0, mask1=mask_skip)[0], None
        wd = conv.dgrad_phase(dz.dtype)  # (packed beside the forward: _prefetch_decoder_weights)
        if prev is None and skip_grad_out is None:  # single source: the ReLU mask rides in the epilogue, nothing to split
            return ops.conv2d(dz, wd, stride=2, pad=1, out_hw=hw, relu_mask=mask_skip, alg_scale=2.25), None
        if prev is not None and skip_grad_out is None and c1 % 128 == 0:  # torch.cat's backward fused into the store (two destinations)
            return ops.conv2d_split(dz, wd, c1, stride=2, pad=1, out_hw=hw, mask1=mask_skip, mask2=mask_prev, alg_scale=2.25)
        dsrc = ops.conv2d(dz, wd, stride=2, pad=1, out_hw=hw, alg_scale=2.25)
        return ops.cat_split_bwd(dsrc, c1, 0 if prev is None else prev.shape[3], mask1=mask_skip, mask2=mask_prev,
                                 out1=skip_grad_out)

    d3, _ = up_bwd(net.dec4, d4, t["dec3"], None, t["dec3"], None)
    del d4
    g_enc1, d2 = up_bwd(net.dec3, d3, enc1, t["dec2"], None, t["dec2"])
    del d3
    g_enc2, d1 = up_bwd(net.dec2, d2, enc2, t["dec1"], None, t["dec1"])
    del d2
    g_enc3, d0 = up_bwd(net.dec1, d1, enc3, t["dec0"], None, t["dec0"])
    del d1
    g_enc4, dcen = up_bwd(net.dec0, d0, enc4, t["center"], None, t["center"])
    del d0
    dpooled, _ = up_bwd(net.center, dcen, t["pooled"], None, None, None)
    del dcen
    g = ops.maxpool2d_bwd(dpooled, t["amc"], tuple(enc4.shape), 2, 2, 0, out=g_enc4)  # accumulates into g_enc4
    del dpooled
    arena.flush()  # bucket 1: head + decoder (13.8 M gradients)

    # ---- encoder, last bottleneck first ----------------------------------------------------------------------
    # gradient a block's INPUT additionally receives from the decoder skip (only the first block of layers 2..4)
    r = net.resnet
    skip_grad = {id(r.layer2[0]): g_enc1, id(r.layer3[0]): g_enc2, id(r.layer4[0]): g_enc3}
    layer_heads = {id(r.layer1[0]), id(r.layer2[0]), id(r.layer3[0]), id(r.layer4[0])}

    def dgrad_into_bn(dy, conv, out_hw, y, st, z, bits, residual=None):
        """Gradient at the OUTPUT of a BatchNorm+ReLU (z) from the convolution that consumed z, with the ReLU mask (z's sign:
        its bit form when the forward wrote one) and BatchNorm's two backward reductions done in the convolution's
        epilogue: returns (g, partial)."""
        if (dy.dtype == torch.float32 and conv.k == 3 and conv.stride == 1 and conv.padding == 1 and residual is None
                and ops.wino33_dgrad_ok(dy, conv.cin)):
            # fp32 stride-1 3x3 (Bottleneck.conv2): the Winograd F(2x2, 3x3) form with the same epilogue (4/9 of the multiply-adds)
            return ops.conv2d_wino33_dgrad(dy, conv.dgrad_wino33(), relu_mask=z if bits is None else None, relu_mask_bits=bits,
                                           bn=(y, st[0], st[1]))
        wd = conv.dgrad_weight(dy.dtype)
        return ops.conv2d_dgrad_bnstats(dy, wd, out_hw, y, st[0], st[1], ups=2 if conv.stride == 2 else 0,
                                        pad=conv.k - 1 - conv.padding, residual=residual,
                                        relu_mask=z if bits is None else None, relu_mask_bits=bits)

    g_partial = None  # partial sums riding with g when a fused dgrad produced it
    blocks = tape.blocks
    for bi in range(len(blocks) - 1, -1, -1):
        rec = blocks[bi]
        blk, h = rec["blk"], rec["h"]
        hw_in = (h.shape[1], h.shape[2])
        if g_partial is None:  # g arrives unmasked from the decoder: the two-pass kernel masks, reduces and applies
            dy3, _, _, gm = ops.bn_bwd(g, rec["z3"], rec["y3"], rec["st3"][0], rec["st3"][1], blk.bn3.weight.detach(),
                                       want_masked=True, **bn_grads(blk.bn3))
        else:  # g is already masked (it IS the residual-branch gradient) and its reductions are done
            dy3, _, _ = ops.bn_bwd_from_partials(g, rec["y3"], rec["st3"][0], rec["st3"][1], blk.bn3.weight.detach(), g_partial,
                                                 **bn_grads(blk.bn3))
            gm = g
        del g
        w3 = arena.conv(blk.conv3)
        arena.wgrad(lambda: ops.conv2d_wgrad(dy3, rec["z2"], 1, 1, out=w3), dy3, rec["z2"])
        g2, p2 = dgrad_into_bn(dy3, blk.conv3, (rec["z2"].shape[1], rec["z2"].shape[2]), rec["y2"], rec["st2"], rec["z2"], rec["b2"])
        del dy3
        dy2, _, _ = ops.bn_bwd_from_partials(g2, rec["y2"], rec["st2"][0], rec["st2"][1], blk.bn2.weight.detach(), p2,
                                             **bn_grads(blk.bn2))
        del g2
        w2 = arena.conv(blk.conv2)
        arena.wgrad(lambda: ops.conv2d_wgrad(dy2, rec["z1"], 3, 3, stride=blk.stride, pad=1, out=w2), dy2, rec["z1"])
        h1, w1_ = rec["z1"].shape[1], rec["z1"].shape[2]
        if (blk.conv2.stride == 2 and blk.conv2.k == 3 and blk.conv2.padding == 1 and h1 == 2 * dy2.shape[1] and w1_ == 2 * dy2.shape[2]
                and os.environ.get("ROBOSAT_S2_DGRAD", "1") != "0" and os.environ.get("ROBOSAT_S2_DGRAD_3X3", "1") != "0"):
            # 3x3 / stride 2: on each input parity the gradient is a 2x2 convolution over dy -- the phase form's geometry (16, in the
            # fp32 Winograd form 9, multiply-adds per four pixels against the zero-insertion launch's 36); BatchNorm's reductions
            # then take their own pass (bn_bwd) instead of riding in that launch's epilogue
            if dy2.dtype == torch.float32 and ops.wino_ok(dy2, None, blk.conv2.cin):
                g1 = ops.conv2d_phase_wino(dy2, blk.conv2.dgrad_s2_phase(torch.float32, wino=True))
                zm = rec["z1"]
            else:
                g1 = ops.conv2d_phase(dy2, blk.conv2.dgrad_s2_phase(dy2.dtype), relu_mask=rec["z1"])
                zm = None
            del dy2
            dy1, _, _ = ops.bn_bwd(g1, zm, rec["y1"], rec["st1"][0], rec["st1"][1], blk.bn1.weight.detach(), **bn_grads(blk.bn1))
        else:
            g1, p1 = dgrad_into_bn(dy2, blk.conv2, (h1, w1_), rec["y1"], rec["st1"], rec["z1"], rec["b1"])
            del dy2
            dy1, _, _ = ops.bn_bwd_from_partials(g1, rec["y1"], rec["st1"][0], rec["st1"][1], blk.bn1.weight.detach(), p1,
                                                 **bn_grads(blk.bn1))
        del g1
        w1 = arena.conv(blk.conv1)
        arena.wgrad(lambda: ops.conv2d_wgrad(dy1, h, 1, 1, out=w1), dy1, h)
        extra = skip_grad.get(id(blk))
        if blk.downsample is not None:
            dconv, dbn = blk.downsample[0], blk.downsample[1]
            dyd, _, _ = ops.bn_bwd(gm, None, rec["yd"], rec["std"][0], rec["std"][1], dbn.weight.detach(), **bn_grads(dbn))
            wdn = arena.conv(dconv)
            arena.wgrad(lambda: ops.conv2d_wgrad(dyd, h, 1, 1, stride=blk.stride, out=wdn), dyd, h)
            if (extra is not None and dconv.k == 1 and dconv.stride == 2 and dconv.padding == 0 and hw_in[0] % 2 == 0 and hw_in[1] % 2 == 0
                    and os.environ.get("ROBOSAT_S2_DGRAD", "1") != "0"):
                # 1x1 / stride 2: the transposed product on the low-resolution grid, added onto the even positions of the skip
                # branch's gradient in place (as a zero-insertion convolution three of four GEMM rows are zeros)
                res = ops.scatter_add_stride2(ops.conv2d(dyd, dconv.dgrad_weight(dyd.dtype)), extra)
            else:
                res = _dgrad(dyd, dconv, hw_in, residual=extra)
            del dyd
        else:
            assert extra is None
            res = gm
        if bi > 0:  # h is the previous bottleneck's z3 = relu(bn3(y3) + identity): fuse its mask + reductions
            prev = blocks[bi - 1]
            g, g_partial = dgrad_into_bn(dy1, blk.conv1, hw_in, prev["y3"], prev["st3"], prev["z3"], prev["b3"], residual=res)
        else:  # h is the stem's pooled output
            g, g_partial = _dgrad(dy1, blk.conv1, hw_in, residual=res), None
        del dy1, res, gm
        if id(blk) in layer_heads:
            arena.flush()  # one bucket per finished ResNet layer

    # ---- stem -------------------------------------------------------------------------------------------------
    dz0 = ops.maxpool2d_bwd(g, t["am0"], tuple(t["z0"].shape), 3, 2, 1)
    dy0, _, _ = ops.bn_bwd(dz0, t["z0"], t["y0"], t["st0"][0], t["st0"][1], r.bn1.weight.detach(), **bn_grads(r.bn1))
    if dy0.dtype == torch.bfloat16:
        dwp = ops.stem_conv_wgrad_bf16(dy0, t["x4"])
    else:
        dwp = ops.conv2d_wgrad(dy0, t["x4"], 7, 7, stride=2, pad=3, stem=7)
    ops.unpack_stem_weight(dwp, 7, net.in_channels, out=arena.conv(r.conv1))
    arena.finish()
    return arena.grads


class _UNetTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        if not x.is_cuda:
            raise RuntimeError("robosat_amd.UNet runs on the MI355X only (got a {} tensor); there is no CPU fallback".format(x.device))
        tape = _Tape()
        logits = _forward(net, x.detach().float().contiguous(), tape, backward=True)
        ctx.net, ctx.tape, ctx.params = net, tape, params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        # a fresh arena per step: autograd ADOPTS the returned views as .grad (no 150 MB clone: nothing else references
        # them once this function returns), so the memory must not be reused
        net = ctx.net
        unused = {id(p) for p in net.resnet.fc.parameters()}  # in the state dict and in Adam, never in the graph
        arena = GradArena([p for p in ctx.params if id(p) not in unused], dlogits.device, getattr(net, "grad_reducer", None))
        grads = _backward(net, ctx.tape, dlogits.contiguous(), arena)
        ctx.tape = None
        out = tuple(grads.pop(p, None) for p in ctx.params)
        grads.clear()
        return (None, None) + out


def unet_train_forward(net, x):
    assert x.size(1) == net.in_channels
    params = tuple(net.parameters())
    return _UNetTrainFn.apply(net, x, *params)
