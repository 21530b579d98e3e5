"""Thin tensor-level wrappers over the C ABI (``include/robosat_hip.h``).

Tensors are torch CUDA(=HIP) fp32 tensors used purely as device memory: activations are contiguous ``[N,H,W,C]``
(NHWC), convolution weights contiguous ``[Cout,kh,kw,Cin]`` (KRSC).  Work is enqueued on torch's current stream.
Nothing here computes with torch; a CPU tensor is an error (there is no CPU path).
"""

import ctypes
import os

import torch

from . import _lib
from ._lib import RS_BF16, RS_F32, ConvDesc, check

BF16 = torch.bfloat16


def _dt(t):
    """dtype code of an activation tensor for the ``*_dt`` entry points."""

    if t.dtype == torch.float32:
        return RS_F32
    if t.dtype == torch.bfloat16:
        return RS_BF16
    raise TypeError("robosat_amd: activations are fp32 or bf16, got {}".format(t.dtype))


def _dev(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("robosat_amd: `{}` is on {} -- the hot path only runs on the MI355X (no CPU fallback)".format(name, t.device))
    if t.dtype != dtype:
        raise TypeError("robosat_amd: `{}` must be {}, got {}".format(name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("robosat_amd: `{}` must be contiguous".format(name))
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# When set to a list, every conv launch is bracketed by HIP events on the launch stream and
# (kernel name, algorithmic flops, start event, end event) is appended: bench.py's roofline leg.
PROFILE = None


def _record(name, flops, shape, ev0, ev1, nbytes, executed=None):
    """One roofline record: ALGORITHMIC flops / bytes of the launch (reference shapes) and the flops
    the kernel actually executes (smaller for the phase-form decoder kernels: 4/9)."""

    PROFILE.append((name, flops, shape, ev0, ev1, nbytes, flops if executed is None else executed))


def conv_flops(d):
    """Algorithmic FLOPs of one launch: 2*N*Cout*Cin*kh*kw*Ho*Wo on the reference's shapes."""

    cin = 3 if d.stem else d.C1 + d.C2  # the stem's 4th (zero) band and 8th (zero) tap are not algorithmic work
    return 2.0 * d.N * d.Cout * cin * d.kh * d.kw * d.Ho * d.Wo


def conv_bytes(d, esize, epilogue_tensors=0):
    """Algorithmic HBM bytes of one launch: every tensor once -- the PRE-upsample sources, the
    weights, the output, and the ``epilogue_tensors`` output-shaped operands the fused epilogue reads (residual, ReLU
    mask, the BatchNorm input of the fused backward statistics)."""

    cin = 4 if d.stem else d.C1 + d.C2
    return esize * (d.N * d.Hs * d.Ws * cin + d.Cout * d.kh * d.kw * cin + (1 + epilogue_tensors) * d.N * d.Ho * d.Wo * d.Cout)


def conv_desc(src1, weight, src2=None, ups=0, stride=1, pad=0, relu=False, stem=0, out_hw=None, bands=4):
    n, hs, ws, c1 = src1.shape
    cout, kh, kw_, _ = weight.shape
    c2 = 0 if src2 is None else src2.shape[3]
    kw = int(stem) if stem else kw_  # packed stem weights are [Cout][kh][8][4]; `stem` carries the true kw (7)
    if out_hw is None:
        assert ups in (0, 1)
        hv, wv = (hs * 2, ws * 2) if ups == 1 else (hs, ws)
        out_hw = ((hv + 2 * pad - kh) // stride + 1, (wv + 2 * pad - kw) // stride + 1)
    # rs_conv_desc.stem: 1 = the packed stem layout; 3 = ... and the caller vouches that the 4th band of the input and the
    # filter's c = 3 entries are zeros (an RGB image through nchw_to_nhwc4 / pack_stem_weight): the kernel skips them
    return ConvDesc(n, hs, ws, c1, c2, ups, kh, kw, stride, pad, out_hw[0], out_hw[1], cout, int(relu),
                    (3 if bands <= 3 else 1) if stem else 0)


def conv2d(src1, weight, src2=None, ups=0, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False,
           stem=0, out_hw=None, out=None, relu_mask=None, alg_scale=1.0, bands=4):
    """``rs_conv2d_fwd``: out = relu?(conv(gather(src1|src2)) * scale + shift + residual).

    ``stem``: 0, or the true filter width (7) when ``weight`` is the packed ``[Cout,kh,8,4]`` stem filter; ``bands``: 3 when
    the image had three bands (its 4th NHWC4 channel and the packed filter's 4th entries are zeros and are skipped).
    ``relu_mask``: tensor shaped like ``out``; the result is zeroed where it is <= 0 (fused ReLU backward)."""

    d = conv_desc(src1, weight, src2, ups, stride, pad, relu, stem, out_hw, bands)
    act = src1.dtype  # fp32: exact-fp32 MFMA kernels; bf16: bf16 operands, fp32 accumulation
    bf = act == BF16
    if bf and stem:
        raise ValueError("the 7x7 stem runs on the fp32 kernel (cast happens at the stem max-pool)")
    if out is None:
        out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=src1.device, dtype=act)
    if src2 is not None:
        assert src2.shape[:3] == src1.shape[:3]
    if not stem:
        assert weight.shape[3] == d.C1 + d.C2, "weight Cin {} != {}+{}".format(weight.shape[3], d.C1, d.C2)
    if residual is not None:
        assert residual.shape == out.shape
    if relu_mask is not None:
        assert relu_mask.shape == out.shape
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    fn = _lib.lib().rs_conv2d_fwd_bf16 if bf else _lib.lib().rs_conv2d_fwd
    rc = fn(
        ctypes.byref(d), _dev(src1, "src1", act), _dev(src2, "src2", act), _dev(weight, "weight", act), _dev(scale, "scale"),
        _dev(shift, "shift"), _dev(residual, "residual", act), _dev(relu_mask, "relu_mask", act), _dev(out, "out", act), _stream(),
    )
    check(rc, "rs_conv2d_fwd_bf16" if bf else "rs_conv2d_fwd")
    if PROFILE is not None:
        ev1.record()
        name = conv_tile_name(d, bf, plain=relu_mask is None)
        if alg_scale != 1.0 and not name.startswith(("conv_thin", "conv_halo")):  # phase-form data gradient: 16 taps at source resolution stand for 9 at the upsampled one
            name = name.replace("<", "<dgrad4x4,")
        _record(name, conv_flops(d) * alg_scale, (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                conv_bytes(d, 2 if bf else 4, (residual is not None) + (relu_mask is not None)), conv_flops(d))
    return out


def conv2d_bnstats(src1, weight, src2=None, ups=0, stride=1, pad=0):
    """Train-mode ``conv -> BatchNorm`` front half: the raw convolution output plus the per-tile partial sums of its
    BatchNorm statistics, produced in the convolution's epilogue (``rs_conv2d_fwd_bnstats_dt``).
    Returns (out, partial [tiles,2,Cout] fp32)."""

    d = conv_desc(src1, weight, src2, ups, stride, pad, False, 0, None)
    act = src1.dtype
    lib = _lib.lib()
    out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=src1.device, dtype=act)
    rows = lib.rs_conv2d_bnstats_rows_dt(ctypes.byref(d), _dt(src1))
    if rows <= 0:
        raise ValueError("rs_conv2d_bnstats_rows: invalid arguments")
    partial = torch.empty((rows, 2, d.Cout), device=src1.device, dtype=torch.float32)
    assert weight.shape[3] == d.C1 + d.C2
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.rs_conv2d_fwd_bnstats_dt(ctypes.byref(d), _dt(src1), _dev(src1, "src1", act), _dev(src2, "src2", act),
                                      _dev(weight, "weight", act), _dev(out, "out", act), _dev(partial, "partial"), _stream())
    check(rc, "rs_conv2d_fwd_bnstats_dt")
    if PROFILE is not None:
        ev1.record()
        bf = act == BF16
        _record(conv_tile_name(d, bf, plain=False), conv_flops(d), (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                        conv_bytes(d, 2 if bf else 4))
    return out, partial


def conv2d_dgrad_bnstats(dy, wd, out_hw, bn_y, bn_mean, bn_invstd, ups=0, pad=0, residual=None, relu_mask=None,
                         relu_mask_bits=None, stride=1):
    """Data-gradient convolution whose output g is the gradient at a BatchNorm+ReLU output: ``rs_conv2d_fwd`` semantics
    (``wd`` = packed dgrad weights, optional residual, relu_mask = z) + per-tile partial sums of BatchNorm's two backward
    reductions (``rs_conv2d_dgrad_bnstats_dt``).  ``relu_mask_bits`` (from ``bn_apply(..., want_bits=True)``) replaces
    ``relu_mask`` by one bit per element (``rs_conv2d_dgrad_bnstats_bits_dt``).  Returns (g, partial [tiles,2,C])."""

    d = conv_desc(dy, wd, None, ups, stride, pad, False, 0, out_hw)
    act = dy.dtype
    lib = _lib.lib()
    out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=dy.device, dtype=act)
    assert bn_y.shape == out.shape
    rows = lib.rs_conv2d_bnstats_rows_dt(ctypes.byref(d), _dt(dy))
    if rows <= 0:
        raise ValueError("rs_conv2d_bnstats_rows: invalid arguments")
    partial = torch.empty((rows, 2, d.Cout), device=dy.device, dtype=torch.float32)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if relu_mask_bits is not None:
        assert relu_mask_bits.numel() * 8 == out.numel(), "one mask bit per output element"
        rc = lib.rs_conv2d_dgrad_bnstats_bits_dt(
            ctypes.byref(d), _dt(dy), _dev(dy, "dy", act), _dev(wd, "weight", act), _dev(residual, "residual", act),
            _dev(relu_mask_bits, "relu_mask_bits", torch.uint8), _dev(bn_y, "bn_y", act), _dev(bn_mean, "bn_mean"),
            _dev(bn_invstd, "bn_invstd"), _dev(out, "out", act), _dev(partial, "partial"), _stream())
        check(rc, "rs_conv2d_dgrad_bnstats_bits_dt")
    else:
        rc = lib.rs_conv2d_dgrad_bnstats_dt(
            ctypes.byref(d), _dt(dy), _dev(dy, "dy", act), _dev(wd, "weight", act), _dev(residual, "residual", act),
            _dev(relu_mask, "relu_mask", act), _dev(bn_y, "bn_y", act), _dev(bn_mean, "bn_mean"), _dev(bn_invstd, "bn_invstd"),
            _dev(out, "out", act), _dev(partial, "partial"), _stream())
        check(rc, "rs_conv2d_dgrad_bnstats_dt")
    if PROFILE is not None:
        ev1.record()
        bf = act == BF16
        _record(conv_tile_name(d, bf, plain=False), conv_flops(d), (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                        conv_bytes(d, 2 if bf else 4, 1 + (residual is not None) + (relu_mask is not None))
                        + (out.numel() // 8 if relu_mask_bits is not None else 0))
    return out, partial


def bn_bwd_from_partials(g, y, mean, invstd, gamma, partial, dgamma=None, dbeta=None):
    """BatchNorm backward from ``conv2d_dgrad_bnstats``'s partial sums: returns (dy, dgamma, dbeta); ``g`` is already
    masked by the ReLU, so it is also the gradient of a residual branch."""

    c = y.shape[-1]
    m = y.numel() // c
    dy = torch.empty_like(y)
    if dgamma is None:
        dgamma = torch.empty(c, device=y.device, dtype=torch.float32)
    if dbeta is None:
        dbeta = torch.empty(c, device=y.device, dtype=torch.float32)
    t = y.dtype
    rc = _lib.lib().rs_bn_bwd_from_partials_dt(
        _dev(g, "g", t), _dev(y, "y", t), _dev(mean, "mean"), _dev(invstd, "invstd"), _dev(gamma, "gamma"), _dev(dy, "dy", t),
        _dev(dgamma, "dgamma"), _dev(dbeta, "dbeta"), _dev(partial, "partial"), partial.shape[0], _dt(y), m, c,
        _workspace(64 * 2 * c * 8 + 3 * c * 4, y.device), _stream())
    check(rc, "rs_bn_bwd_from_partials_dt")
    return dy, dgamma, dbeta


def bn_finalize_stats(partial, m, gamma, beta, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """(mean, invstd, scale, shift) from ``conv2d_bnstats``'s partial sums; updates the running buffers like bn_train_stats."""

    rows, _, c = partial.shape
    mean, invstd, scale, shift = (torch.empty(c, device=partial.device, dtype=torch.float32) for _ in range(4))
    rc = _lib.lib().rs_bn_finalize_stats(
        _dev(partial, "partial"), rows, m, c, ctypes.c_float(eps), ctypes.c_float(momentum), _dev(gamma, "gamma"),
        _dev(beta, "beta"), _dev(mean, "mean"), _dev(invstd, "invstd"), _dev(scale, "scale"), _dev(shift, "shift"),
        _dev(running_mean, "running_mean"), _dev(running_var, "running_var"),
        _dev(num_batches_tracked, "num_batches_tracked", torch.int64), _workspace(64 * 2 * c * 8, partial.device), _stream())
    check(rc, "rs_bn_finalize_stats")
    return mean, invstd, scale, shift


def pack_phase_weight(w_krsc, dtype=torch.float32):
    """fp32 KRSC [Cout,3,3,Cin] -> [4,Cout,2,2,Cin] in ``dtype``: the four parity-specific 2x2 filters that a 3x3 / pad-1
    convolution over a nearest-x2 upsampled input collapses to (``rs_pack_phase_weight_dt``)."""

    cout, kh, kw, cin = w_krsc.shape
    assert kh == 3 and kw == 3
    out = torch.empty((4, cout, 2, 2, cin), device=w_krsc.device, dtype=dtype)
    check(_lib.lib().rs_pack_phase_weight_dt(_dev(w_krsc, "w"), _dev(out, "out", dtype), RS_BF16 if dtype == BF16 else RS_F32,
                                             cout, cin, _stream()), "rs_pack_phase_weight_dt")
    return out


def pack_s2_dgrad_phase_weight(w_krsc, dtype=torch.float32):
    """fp32 KRSC [Cout,3,3,Cin] of a 3x3 / stride-2 / pad-1 convolution -> [4,Cin,2,2,Cout] in ``dtype``: the phase pack of its DATA
    gradient (``rs_pack_s2_dgrad_phase_weight_dt``) -- ``conv2d_phase(dy, pack)`` is the gradient wrt the input (even sizes)."""

    cout, kh, kw, cin = w_krsc.shape
    assert kh == 3 and kw == 3
    out = torch.empty((4, cin, 2, 2, cout), device=w_krsc.device, dtype=dtype)
    check(_lib.lib().rs_pack_s2_dgrad_phase_weight_dt(_dev(w_krsc, "w"), _dev(out, "out", dtype), RS_BF16 if dtype == BF16 else RS_F32,
                                                      cout, cin, _stream()), "rs_pack_s2_dgrad_phase_weight_dt")
    return out


def pack_dgrad_phase_weight(w_krsc, dtype=torch.float32):
    """fp32 KRSC [Cout,3,3,Cin] -> [Cin,4,4,Cout] in ``dtype``: weights of the 4x4 / stride-2 convolution over dz that is
    the data gradient of DecoderBlock wrt its pre-upsample input (``rs_pack_dgrad_phase_weight_dt``)."""

    cout, kh, kw, cin = w_krsc.shape
    assert kh == 3 and kw == 3
    out = torch.empty((cin, 4, 4, cout), device=w_krsc.device, dtype=dtype)
    # two coalesced steps: LDS-tiled transpose to [Cin,3,3,Cout] (rs_pack_dgrad_weight), then the tap sums along Cout
    wt = pack_dgrad_weight(w_krsc, torch.float32)
    check(_lib.lib().rs_combine_dgrad_phase_weight_dt(_dev(wt, "wt"), _dev(out, "out", dtype),
                                                      RS_BF16 if dtype == BF16 else RS_F32, cout, cin, _stream()),
          "rs_combine_dgrad_phase_weight_dt")
    return out


def conv2d_split(src, weight, c1, stride=1, pad=0, out_hw=None, mask1=None, mask2=None, alg_scale=1.0):
    """``rs_conv2d_fwd_split_dt``: one convolution whose output channels [0, c1) and [c1, Cout) land in two tensors (the
    backward of torch.cat fused into the store), each with its optional ReLU mask.  Returns (out1, out2)."""

    d = conv_desc(src, weight, None, 0, stride, pad, False, 0, out_hw)
    act = src.dtype
    c2 = d.Cout - c1
    out1 = torch.empty((d.N, d.Ho, d.Wo, c1), device=src.device, dtype=act)
    out2 = torch.empty((d.N, d.Ho, d.Wo, c2), device=src.device, dtype=act)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_fwd_split_dt(ctypes.byref(d), _dt(src), _dev(src, "src", act), _dev(weight, "weight", act),
                                           _dev(out1, "out1", act), _dev(mask1, "mask1", act), _dev(out2, "out2", act),
                                           _dev(mask2, "mask2", act), c1, _stream())
    check(rc, "rs_conv2d_fwd_split_dt")
    if PROFILE is not None:
        ev1.record()
        bf = act == BF16
        name = conv_tile_name(d, bf, plain=False)
        if alg_scale != 1.0 and not name.startswith("conv_halo"):
            name = name.replace("<", "<dgrad4x4,")
        _record(name, conv_flops(d) * alg_scale, (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                conv_bytes(d, 2 if bf else 4, ((mask1 is not None) * c1 + (mask2 is not None) * (d.Cout - c1)) / d.Cout),
                conv_flops(d))
    return out1, out2


def cat_split_bwd(dcat, c1, c2=0, mask1=None, mask2=None, out1=None):
    """dcat [N,H,W,C1+C2] -> (d1 [N,H,W,C1], d2 [N,H,W,C2] or None): the torch.cat split + ReLU masks; ``out1`` given =>
    accumulate into it."""

    n, h, w, ct = dcat.shape
    assert ct == c1 + c2
    t = dcat.dtype
    acc = out1 is not None
    d1 = out1 if acc else torch.empty((n, h, w, c1), device=dcat.device, dtype=t)
    d2 = torch.empty((n, h, w, c2), device=dcat.device, dtype=t) if c2 else None
    rc = _lib.lib().rs_cat_split_bwd_dt(_dev(dcat, "dcat", t), _dev(d1, "d1", t), _dev(d2, "d2", t), _dev(mask1, "mask1", t),
                                        _dev(mask2, "mask2", t), _dt(dcat), n, h, w, c1, c2, int(acc), _stream())
    check(rc, "rs_cat_split_bwd_dt")
    return d1, d2


def conv2d_phase(src1, weight_phase, src2=None, scale=None, shift=None, residual=None, relu=False, relu_mask=None):
    """DecoderBlock in phase form: relu?(conv3x3(interpolate(cat[src1, src2], x2 nearest), pad 1)) computed as four 2x2
    convolutions on the source grid (``rs_conv2d_fwd_phase_dt``); ``weight_phase`` from ``pack_phase_weight``."""

    n, hs, ws, c1 = src1.shape
    c2 = 0 if src2 is None else src2.shape[3]
    cout = weight_phase.shape[1]
    assert tuple(weight_phase.shape) == (4, cout, 2, 2, c1 + c2)
    if src2 is not None and tuple(src2.shape[:3]) != (n, hs, ws):
        # what torch.cat raises in the reference (unet.py:134-137) when a skip and the decoder tensor disagree
        raise RuntimeError("Sizes of tensors must match except in dimension 1: skip {} vs decoder {}".format(
            tuple(src1.shape[:3]), tuple(src2.shape[:3])))
    act = src1.dtype
    d = ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, int(relu), 0)
    out = torch.empty((n, 2 * hs, 2 * ws, cout), device=src1.device, dtype=act)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_fwd_phase_dt(
        ctypes.byref(d), _dt(src1), _dev(src1, "src1", act), _dev(src2, "src2", act), _dev(weight_phase, "weight", act),
        _dev(scale, "scale"), _dev(shift, "shift"), _dev(residual, "residual", act), _dev(relu_mask, "relu_mask", act),
        _dev(out, "out", act), _stream())
    check(rc, "rs_conv2d_fwd_phase_dt")
    if PROFILE is not None:
        ev1.record()
        bf = act == BF16
        _record(conv_tile_name(d, bf, phase=True), conv_flops(d),
                (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                conv_bytes(d, 2 if bf else 4, (residual is not None) + (relu_mask is not None)), conv_flops(d) * 4.0 / 9.0)
    return out


def _phase_desc(src1, src2, cout, relu):
    n, hs, ws, c1 = src1.shape
    c2 = 0 if src2 is None else src2.shape[3]
    return ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, int(relu), 0)


def wino_ok(src1, src2, cout, force=False):
    """Whether the fp32 Winograd form of DecoderBlock (``conv2d_phase_wino``) should run this layer
    (``rs_conv2d_phase_wino_ok``: it can, and the launch is large enough to fill the chip); ``force``: whether it CAN (the
    parity tests reach the kernel with small problems).  ROBOSAT_WINOGRAD=0 switches it off (A/B measurements: the generic
    phase kernel then runs every layer)."""

    import os

    if src1.dtype != torch.float32 or (not force and os.environ.get("ROBOSAT_WINOGRAD", "1") == "0"):
        return False
    rc = _lib.lib().rs_conv2d_phase_wino_ok(ctypes.byref(_phase_desc(src1, src2, cout, False)))
    return rc != 0 if force else rc == 1


def pack_wino_phase_weight(weight_phase):
    """Phase pack [4,Cout,2,2,Cin] fp32 -> U = G g G^T, [4,9,Cout,Cin]: the transformed filters of the Winograd form."""

    _, cout, _, _, cin = weight_phase.shape
    u = torch.empty((4, 9, cout, cin), device=weight_phase.device, dtype=torch.float32)
    check(_lib.lib().rs_pack_wino_phase_weight(_dev(weight_phase, "w_phase"), _dev(u, "u"), cout, cin, _stream()),
          "rs_pack_wino_phase_weight")
    return u


def conv2d_phase_wino(src1, u, src2=None, relu=False):
    """DecoderBlock in fp32 as a Winograd F(2x2, 2x2) convolution on the phase form (``rs_conv2d_fwd_phase_wino``): same
    result as ``conv2d_phase`` up to fp32 summation order, 9/16 of its multiply-adds."""

    n, hs, ws, c1 = src1.shape
    c2 = 0 if src2 is None else src2.shape[3]
    cout = u.shape[2]
    assert tuple(u.shape) == (4, 9, cout, c1 + c2)
    if src2 is not None and tuple(src2.shape[:3]) != (n, hs, ws):
        raise RuntimeError("Sizes of tensors must match except in dimension 1: skip {} vs decoder {}".format(
            tuple(src1.shape[:3]), tuple(src2.shape[:3])))
    d = _phase_desc(src1, src2, cout, relu)
    out = torch.empty((n, 2 * hs, 2 * ws, cout), device=src1.device, dtype=torch.float32)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_fwd_phase_wino(ctypes.byref(d), _dev(src1, "src1"), _dev(src2, "src2"), _dev(u, "u"), _dev(out, "out"),
                                             _stream())
    check(rc, "rs_conv2d_fwd_phase_wino")
    if PROFILE is not None:
        ev1.record()
        name = _lib.lib().rs_conv2d_phase_wino_name(ctypes.byref(d)).decode()
        # executed: 9 multiply-adds per 2x2 outputs of a parity = 1/4 of the reference-shape count (the phase form: 4/9)
        _record(name, conv_flops(d), (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1, conv_bytes(d, 4),
                conv_flops(d) * 0.25)
    return out


def wino_dgrad_ok(n, hs, ws, c1, c2, cout):
    """Whether the fp32 DecoderBlock (sources [n,hs,ws,c1(+c2)] -> cout at 2 hs x 2 ws) can take its data gradient through the Winograd
    form (``rs_conv2d_dgrad_phase_wino_ok``: geometry only; ROBOSAT_WINO_DGRAD=0 keeps the 4x4 / stride-2 kernel for A/B runs)."""

    import os

    if os.environ.get("ROBOSAT_WINO_DGRAD", "1") == "0":
        return False
    d = ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, 0, 0)
    return _lib.lib().rs_conv2d_dgrad_phase_wino_ok(ctypes.byref(d)) == 1


def pack_wino_dgrad_weight(wd4x4):
    """[Cin,4,4,Cout] fp32 (``pack_dgrad_phase_weight``) -> the transformed data-gradient filters [4,9,Cin,Cout]
    (``rs_pack_wino_dgrad_weight``)."""

    cin, kh, kw, cout = wd4x4.shape
    assert kh == 4 and kw == 4 and wd4x4.dtype == torch.float32
    u = torch.empty((4, 9, cin, cout), device=wd4x4.device, dtype=torch.float32)
    check(_lib.lib().rs_pack_wino_dgrad_weight(_dev(wd4x4, "wd"), _dev(u, "u"), cin, cout, _stream()), "rs_pack_wino_dgrad_weight")
    return u


def conv2d_dgrad_phase_wino(dz, u, c1, c2=0, mask1=None, mask2=None, split=False):
    """Data gradient of the fp32 DecoderBlock wrt cat[skip, prev] at source resolution (``rs_conv2d_dgrad_phase_wino``): ``dz``
    [N,2Hs,2Ws,Cout], ``u`` from ``pack_wino_dgrad_weight``.  ``split``: two tensors ([..,c1], [..,c2]) with their optional ReLU masks
    (torch.cat's backward fused into the store); else one tensor [..,c1+c2] with ``mask1``.  Returns (d1, d2 | None)."""

    n, ho, wo, cout = dz.shape
    hs, ws = ho // 2, wo // 2
    assert tuple(u.shape) == (4, 9, c1 + c2, cout) and dz.dtype == torch.float32
    d = ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, ho, wo, cout, 0, 0)
    if split:
        out1 = torch.empty((n, hs, ws, c1), device=dz.device, dtype=torch.float32)
        out2 = torch.empty((n, hs, ws, c2), device=dz.device, dtype=torch.float32)
    else:
        out1, out2 = torch.empty((n, hs, ws, c1 + c2), device=dz.device, dtype=torch.float32), None
        assert mask2 is None
    for m, o in ((mask1, out1), (mask2, out2)):
        assert m is None or tuple(m.shape) == tuple(o.shape)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_dgrad_phase_wino(ctypes.byref(d), _dev(dz, "dz"), _dev(u, "u"), _dev(out1, "out"), _dev(mask1, "mask"),
                                               _dev(out2, "out2"), _dev(mask2, "mask2"), c1 if split else 0, _stream())
    check(rc, "rs_conv2d_dgrad_phase_wino")
    if PROFILE is not None:
        ev1.record()
        # algorithmic: the reference's 3x3 at the upsampled resolution; the 4x4 / stride-2 form executes 16 taps at source resolution,
        # this form 9/16 of those
        alg = 2.0 * n * ho * wo * cout * (c1 + c2) * 9
        ex4 = 2.0 * n * hs * ws * cout * (c1 + c2) * 16
        nbytes = 4 * (n * ho * wo * cout + 16 * cout * (c1 + c2) + n * hs * ws * (c1 + c2) * (1 + (mask1 is not None) * (c1 if split else c1 + c2) / (c1 + c2)
                                                                                            + (mask2 is not None) * c2 / (c1 + c2)))
        _record(_lib.lib().rs_conv2d_dgrad_phase_wino_name(ctypes.byref(d)).decode(), alg, (cout, c1 + c2, 4, 2, 0, hs, ws), ev0, ev1, nbytes, ex4 * 9.0 / 16.0)
    return out1, out2


def _conv33_desc(src, cout, relu):
    n, h, w, c = src.shape
    return ConvDesc(n, h, w, c, 0, 0, 3, 3, 1, 1, h, w, cout, int(relu), 0)


def wino33_ok(src, cout):
    """Whether the fp32 Winograd F(2x2, 3x3) kernel runs a stride-1 3x3 / pad-1 convolution of ``src`` to ``cout`` channels
    (``rs_conv2d_wino33_ok``: the layer's geometry only, never the batch size); ROBOSAT_WINOGRAD=0 switches it off."""

    import os

    if src.dtype != torch.float32 or os.environ.get("ROBOSAT_WINOGRAD", "1") == "0":
        return False
    return _lib.lib().rs_conv2d_wino33_ok(ctypes.byref(_conv33_desc(src, cout, False))) == 1


def pack_wino33_weight(w_krsc):
    """fp32 KRSC [Cout,3,3,Cin] -> U = G g G^T, [16,Cout,Cin]: the transformed filters of the F(2x2, 3x3) form."""

    cout, kh, kw, cin = w_krsc.shape
    assert kh == 3 and kw == 3
    u = torch.empty((16, cout, cin), device=w_krsc.device, dtype=torch.float32)
    check(_lib.lib().rs_pack_wino33_weight(_dev(w_krsc, "w"), _dev(u, "u"), cout, cin, _stream()), "rs_pack_wino33_weight")
    return u


def conv2d_wino33(src, u, scale=None, shift=None, relu=False):
    """relu?(conv3x3(src, pad 1) * scale + shift) in fp32 as a Winograd F(2x2, 3x3) convolution (``rs_conv2d_fwd_wino33``):
    the eval-mode Bottleneck conv2 / dec5 of the predict path; 4/9 of the direct form's multiply-adds."""

    n, h, w, c = src.shape
    cout = u.shape[1]
    assert tuple(u.shape) == (16, cout, c)
    d = _conv33_desc(src, cout, relu)
    out = torch.empty((n, h, w, cout), device=src.device, dtype=torch.float32)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_fwd_wino33(ctypes.byref(d), _dev(src, "src"), _dev(u, "u"), _dev(scale, "scale"), _dev(shift, "shift"),
                                         _dev(out, "out"), _stream())
    check(rc, "rs_conv2d_fwd_wino33")
    if PROFILE is not None:
        ev1.record()
        name = _lib.lib().rs_conv2d_wino33_name(ctypes.byref(d)).decode()
        _record(name, conv_flops(d), (d.C1, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1, conv_bytes(d, 4), conv_flops(d) * 4.0 / 9.0)
    return out


def conv2d_wino33_bnstats(src, u):
    """Train-mode ``conv3x3 -> BatchNorm`` front half in the fp32 Winograd form (``rs_conv2d_fwd_wino33_stats``): the raw output and
    the per-block partial sums of its statistics, as ``conv2d_bnstats`` returns them.  ROBOSAT_WINO33_STATS=0 at the call site keeps
    the generic kernel (A/B runs)."""

    n, h, w, c = src.shape
    cout = u.shape[1]
    assert tuple(u.shape) == (16, cout, c) and src.dtype == torch.float32
    d = _conv33_desc(src, cout, False)
    lib = _lib.lib()
    rows = lib.rs_conv2d_wino33_stats_rows(ctypes.byref(d))
    if rows <= 0:
        raise ValueError("rs_conv2d_wino33_stats_rows: invalid arguments")
    out = torch.empty((n, h, w, cout), device=src.device, dtype=torch.float32)
    partial = torch.empty((rows, 2, cout), device=src.device, dtype=torch.float32)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.rs_conv2d_fwd_wino33_stats(ctypes.byref(d), _dev(src, "src"), _dev(u, "u"), _dev(out, "out"), _dev(partial, "partial"), _stream()),
          "rs_conv2d_fwd_wino33_stats")
    if PROFILE is not None:
        ev1.record()
        _record(lib.rs_conv2d_wino33_name(ctypes.byref(d)).decode().replace("<3x3,", "<3x3+stats,"), conv_flops(d), (d.C1, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                conv_bytes(d, 4), conv_flops(d) * 4.0 / 9.0)
    return out, partial


def wino33_dgrad_ok(dy, cout):
    """Whether the fp32 data gradient of a stride-1 3x3 / pad-1 convolution (dy [n,h,w,c] -> [n,h,w,cout]) takes the Winograd form
    (``rs_conv2d_dgrad_wino33``: ``wino33_ok`` on the gradient's convolution, cout % 32 == 0); ROBOSAT_WINO33_BWD=0 keeps the generic
    kernel (A/B runs)."""

    import os

    if os.environ.get("ROBOSAT_WINO33_BWD", "1") == "0" or cout % 32:
        return False
    return wino33_ok(dy, cout)


def conv2d_wino33_dgrad(dy, u, relu_mask=None, relu_mask_bits=None, bn=None):
    """``rs_conv2d_dgrad_wino33``: the data gradient of a stride-1 3x3 convolution in the fp32 Winograd form -- ``u`` =
    ``pack_wino33_weight(pack_dgrad_weight(w))`` -- masked by the ReLU it arrives at (``relu_mask``: the forward activation;
    ``relu_mask_bits``: its bit form) and, with ``bn`` = (y, mean, invstd), with the partial sums of BatchNorm's two backward reductions,
    as ``conv2d_dgrad_bnstats`` returns them.  Returns (g, partial or None)."""

    n, h, w, c = dy.shape
    cout = u.shape[1]
    assert tuple(u.shape) == (16, cout, c) and dy.dtype == torch.float32
    d = _conv33_desc(dy, cout, False)
    lib = _lib.lib()
    out = torch.empty((n, h, w, cout), device=dy.device, dtype=torch.float32)
    partial = None
    y = mean = invstd = None
    if bn is not None:
        y, mean, invstd = bn
        assert y.shape == out.shape
        rows = lib.rs_conv2d_wino33_stats_rows(ctypes.byref(d))
        if rows <= 0:
            raise ValueError("rs_conv2d_wino33_stats_rows: invalid arguments")
        partial = torch.empty((rows, 2, cout), device=dy.device, dtype=torch.float32)
    if relu_mask is not None:
        assert relu_mask.shape == out.shape
    if relu_mask_bits is not None:
        assert relu_mask_bits.numel() * 8 == out.numel(), "one mask bit per output element"
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.rs_conv2d_dgrad_wino33(ctypes.byref(d), _dev(dy, "dy"), _dev(u, "u"), _dev(relu_mask, "relu_mask"),
                                     _dev(relu_mask_bits, "relu_mask_bits", torch.uint8), _dev(y, "bn_y"), _dev(mean, "bn_mean"),
                                     _dev(invstd, "bn_invstd"), _dev(out, "out"), _dev(partial, "partial"), _stream()),
          "rs_conv2d_dgrad_wino33")
    if PROFILE is not None:
        ev1.record()
        _record(lib.rs_conv2d_wino33_name(ctypes.byref(d)).decode().replace("<3x3,", "<3x3+bwd,"), conv_flops(d),
                (d.C1, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                conv_bytes(d, 4, 1 + (relu_mask is not None) + (bn is not None)) + (out.numel() // 8 if relu_mask_bits is not None else 0),
                conv_flops(d) * 4.0 / 9.0)
    return out, partial


def wino33_head_ok(src, cout, classes):
    """Whether dec5 + ``self.final`` run as one launch (``rs_conv2d_wino33_head_ok``: the Winograd 3x3 form on a 32-cout
    layer, <= 8 classes); ROBOSAT_FUSED_HEAD=0 keeps the two launches (A/B runs)."""

    import os

    if not wino33_ok(src, cout) or os.environ.get("ROBOSAT_FUSED_HEAD", "1") == "0":
        return False
    return _lib.lib().rs_conv2d_wino33_head_ok(ctypes.byref(_conv33_desc(src, cout, True)), int(classes)) == 1


def conv2d_wino33_head(src, u, final_w, final_b, mode="logits", overlap=0, relu=True):
    """``final(relu(conv3x3(src, pad 1)))`` in one launch (``rs_conv2d_fwd_wino33_head``; reference unet.py:139-141) and, by
    ``mode``, what the ``final_conv1x1*`` functions return: "logits" / "softmax" -> fp32 NCHW [N,C,H,W]; "quantize" ->
    uint8 quantised probabilities of the crop without the ``overlap`` border; "argmax" -> uint8 [N,H,W] class indices."""

    n, h, w, c = src.shape
    cout, classes = u.shape[1], final_w.shape[0]
    assert tuple(u.shape) == (16, cout, c) and tuple(final_w.shape) == (classes, cout)
    d = _conv33_desc(src, cout, relu)
    m = {"logits": 0, "softmax": 1, "quantize": 2, "argmax": 3}[mode]
    out = qout = anchors = None
    if m <= 1:
        out = torch.empty((n, classes, h, w), device=src.device, dtype=torch.float32)
    elif m == 2:
        shape = (n, h - 2 * overlap, w - 2 * overlap) + ((classes - 1,) if classes > 2 else ())
        qout = torch.empty(shape, device=src.device, dtype=torch.uint8)
        anchors = _anchors(src.device)
    else:
        qout = torch.empty((n, h, w), device=src.device, dtype=torch.uint8)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_fwd_wino33_head(
        ctypes.byref(d), _dev(src, "src"), _dev(u, "u"), None, None, _dev(final_w, "final_w"), _dev(final_b, "final_b"), classes, m,
        _dev(anchors, "anchors", torch.float64), int(overlap), _dev(out, "out"), _dev(qout, "qout", torch.uint8), _stream())
    check(rc, "rs_conv2d_fwd_wino33_head")
    if PROFILE is not None:
        ev1.record()
        fl = conv_flops(d) + 2.0 * n * h * w * cout * classes  # (the 1x1 rides along: < 1 % of the launch)
        nbytes = 4 * (n * h * w * c + 9 * cout * c + classes * cout) + (4 * classes if m <= 1 else 1) * n * h * w
        _record(_lib.lib().rs_conv2d_wino33_head_name().decode(), fl, (d.C1, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1,
                nbytes, conv_flops(d) * 4.0 / 9.0 + 2.0 * n * h * w * cout * classes)
    return out if m <= 1 else qout


def bottleneck_tail_ok(x, w3, w1):
    """Whether ``bottleneck_tail`` can run these operands: fp32, layer1's widths (64 -> 256 -> 64), whole 32-pixel sub-tiles."""

    m = x.numel() // x.shape[-1]
    return (x.dtype == torch.float32 and x.shape[-1] == 64 and tuple(w3.shape) == (256, 1, 1, 64) and tuple(w1.shape) == (64, 1, 1, 256)
            and m % 32 == 0 and os.environ.get("ROBOSAT_TAIL_FUSE", "1") != "0")


def bottleneck_tail(x, w3, scale3, shift3, identity, w1, scale1, shift1):
    """``rs_bottleneck_tail_f32``: out = relu(conv1x1(x; w3) * scale3 + shift3 + identity) and z = relu(conv1x1(out; w1) * scale1 +
    shift1) in one launch -- the last convolution of a layer1 Bottleneck and the first of the next (eval mode, fp32).  Returns (out, z)."""

    n, h, w, c1 = x.shape
    cm, c2 = w3.shape[0], w1.shape[0]
    assert identity.shape == (n, h, w, cm) and w1.shape[3] == cm
    out = torch.empty((n, h, w, cm), device=x.device, dtype=torch.float32)
    z = torch.empty((n, h, w, c2), device=x.device, dtype=torch.float32)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_bottleneck_tail_f32(_dev(x, "x"), _dev(w3, "w3"), _dev(scale3, "scale3"), _dev(shift3, "shift3"),
                                           _dev(identity, "identity"), _dev(w1, "w1"), _dev(scale1, "scale1"), _dev(shift1, "shift1"),
                                           _dev(out, "out"), _dev(z, "z"), n * h * w, c1, cm, c2, _stream())
    check(rc, "rs_bottleneck_tail_f32")
    if PROFILE is not None:
        ev1.record()
        m = n * h * w
        fl = 2.0 * m * cm * (c1 + c2)
        _record("bottleneck_tail_f32", fl, (c1, cm, 1, 1, 0, h, w), ev0, ev1, 4 * (m * (c1 + 2 * cm + c2) + cm * (c1 + c2)), fl)
    return out, z


def conv1x1_wave_ok(x, w):
    """Whether ``conv1x1_wave`` can run: fp32, 64 -> 256, 1x1, whole 32-pixel sub-tiles (layer1's downsample branch)."""

    m = x.numel() // x.shape[-1]
    return (x.dtype == torch.float32 and x.shape[-1] == 64 and tuple(w.shape) == (256, 1, 1, 64) and m % 32 == 0
            and os.environ.get("ROBOSAT_TAIL_FUSE", "1") != "0")


def conv1x1_wave(x, w, scale, shift, residual=None, relu=False):
    """``rs_conv1x1_wave_f32``: [relu](conv1x1(x; w) * scale + shift [+ residual]) on the fused tail's first stage alone."""

    n, h, wd, c1 = x.shape
    cout = w.shape[0]
    out = torch.empty((n, h, wd, cout), device=x.device, dtype=torch.float32)
    if residual is not None:
        assert residual.shape == out.shape
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv1x1_wave_f32(_dev(x, "x"), _dev(w, "w"), _dev(scale, "scale"), _dev(shift, "shift"), _dev(residual, "residual"),
                                        int(relu), _dev(out, "out"), n * h * wd, c1, cout, _stream())
    check(rc, "rs_conv1x1_wave_f32")
    if PROFILE is not None:
        ev1.record()
        m = n * h * wd
        fl = 2.0 * m * cout * c1
        _record("conv1x1_wave_f32", fl, (c1, cout, 1, 1, 0, h, wd), ev0, ev1, 4 * (m * (c1 + cout * (2 if residual is not None else 1)) + cout * c1), fl)
    return out


def conv_tile_name(d, bf16=False, phase=False, plain=True):
    """Report name of the kernel a convolution launch runs, 1:1 with the launched symbol:
    ``conv_igemm_<f32|bf16><[phase,]BMxBN,r<row bytes>>`` (or the stem kernel).  ``plain=False``: a launch with fused
    BatchNorm statistics / a ReLU mask (never the plain-epilogue 1x1 kernel of conv1x1_ew_f32.hip)."""

    lib = _lib.lib()
    if d.stem:
        return lib.rs_conv2d_tile_name(lib.rs_conv2d_tile(ctypes.byref(d))).decode()
    tile, rowb = ctypes.c_int(0), ctypes.c_int(0)
    form = int(bool(phase)) | (0 if plain else 2)  # (the library answers for the epilogue kind: no knob is touched around the query)
    check(lib.rs_conv2d_config(ctypes.byref(d), 2 if bf16 else 4, form, ctypes.byref(tile), ctypes.byref(rowb)), "rs_conv2d_config")
    base = (lib.rs_conv2d_tile_name_bf16 if bf16 else lib.rs_conv2d_tile_name)(tile.value).decode()
    if tile.value == TILES["thin"]:  # conv_thin_bf16.hip: named by the form it computes
        return "{}<{}>".format(base, "phase" if phase else ("dgrad4x4" if d.kh == 4 else "3x3"))
    if tile.value == TILES["halo"]:  # halo-once forms of the kernel: form + patch pixels x N tile (`rowb` carries the N tile)
        return "{}<{},{}x{}>".format(base, "phase" if phase else ("dgrad4x4" if d.kh == 4 else "3x3"),
                                     512 if rowb.value & 0x1000 else 256, rowb.value & 0xFFF)
    return base.replace("<", "<phase," if phase else "<").replace(">", ",r{}>".format(rowb.value))


TILES = {"128x128": 0, "128x64": 1, "128x32": 2, "64x64": 3, "256x128": 5, "256x256": 6, "thin": 7, "halo": 8}


class tuning:
    """``with ops.tuning(tile="256x256", rowb=128): ...`` -- force the convolution dispatcher (``rs_conv2d_set_tuning``) for
    the launches inside: the parity tests use it to reach every kernel symbol with small problems, the layer benchmarks
    for A/B runs.  Process-global; restores the measured heuristics on exit."""

    def __init__(self, tile=None, rowb=0):
        self.tile = -1 if tile is None else (TILES[tile] if isinstance(tile, str) else int(tile))
        self.rowb = int(rowb)

    def __enter__(self):
        check(_lib.lib().rs_conv2d_set_tuning(self.tile, self.rowb), "rs_conv2d_set_tuning")
        return self

    def __exit__(self, *exc):
        check(_lib.lib().rs_conv2d_set_tuning(-1, 0), "rs_conv2d_set_tuning")
        return False


def get_knob(name):
    """Current value of one of the library's measurement switches (``rs_get_knob``; the table is in csrc/knobs.hip)."""

    v = ctypes.c_int(0)
    check(_lib.lib().rs_get_knob(name.encode(), ctypes.byref(v)), "rs_get_knob({})".format(name))
    return v.value


def set_knob(name, value):
    """``rs_set_knob`` (process-global; prefer ``with ops.knob(...)`` where the old value should come back)."""

    check(_lib.lib().rs_set_knob(name.encode(), int(value)), "rs_set_knob({})".format(name))


class knob:
    """``with ops.knob("conv1x1_ew", 0): ...`` -- set a measurement switch of the library (``rs_set_knob``) for the launches
    inside and restore it on exit.  Process-global, like ``ops.tuning``: for tests and A/B scripts, not a tuning API."""

    def __init__(self, name, value):
        self.name, self.value, self.old = name, int(value), None

    def __enter__(self):
        self.old = get_knob(self.name)
        check(_lib.lib().rs_set_knob(self.name.encode(), self.value), "rs_set_knob({})".format(self.name))
        return self

    def __exit__(self, *exc):
        check(_lib.lib().rs_set_knob(self.name.encode(), self.old), "rs_set_knob({})".format(self.name))
        return False


def cast_bf16(t):
    """fp32 -> bf16 copy (round to nearest even); used for the per-step compute copies of the fp32 master weights."""

    out = torch.empty(t.shape, device=t.device, dtype=BF16)
    check(_lib.lib().rs_cast_f32_to_bf16(_dev(t, "src"), _dev(out, "dst", BF16), t.numel(), _stream()), "rs_cast_f32_to_bf16")
    return out


def cast_bf16_scaled(t, scale):
    """bf16(t * scale): a gradient bucket on its way to a bf16 exchange, already divided by the world size."""

    out = torch.empty(t.shape, device=t.device, dtype=BF16)
    check(_lib.lib().rs_cast_f32_to_bf16_scaled(_dev(t, "src"), _dev(out, "dst", BF16), t.numel(), ctypes.c_float(scale), _stream()),
          "rs_cast_f32_to_bf16_scaled")
    return out


def cast_f32_scaled(src_bf16, dst_f32, scale):
    """dst (fp32, in place) = float(src bf16) * scale: the way back from a bf16 gradient exchange (``GradReducer``)."""

    assert src_bf16.numel() == dst_f32.numel()
    check(_lib.lib().rs_cast_bf16_to_f32_scaled(_dev(src_bf16, "src", BF16), _dev(dst_f32, "dst"), src_bf16.numel(),
                                                 ctypes.c_float(scale), _stream()), "rs_cast_bf16_to_f32_scaled")
    return dst_f32


class _WPrepItem(ctypes.Structure):  # rs_wprep_item (include/robosat_hip.h)
    _fields_ = [("w", ctypes.c_void_p), ("cast", ctypes.c_void_p), ("dgrad", ctypes.c_void_p), ("Cout", ctypes.c_int),
                ("taps", ctypes.c_int), ("Cin", ctypes.c_int), ("tile_begin", ctypes.c_int)]


class WeightPrep:
    """One launch (``rs_weight_prep_bf16``) that refreshes the bf16 compute copies of a fixed set of fp32 KRSC weights:
    the bf16 KRSC cast and the transposed, tap-flipped data-gradient layout of each -- bit-identical to ``cast_bf16`` /
    ``pack_dgrad_weight(w, bfloat16)`` per tensor.  The destination buffers and the device-side item table are allocated
    once; ``run()`` is what a training step calls after the optimizer moved the master weights."""

    def __init__(self, weights_krsc, want_dgrad=True, dtype=BF16):
        """``dtype=torch.float32`` (round 5, ``rs_weight_prep_f32``): the fp32 training step's data-gradient layouts only -- the
        fp32 KRSC master is its own compute copy, so ``cast`` stays empty."""

        assert weights_krsc and all(w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4 for w in weights_krsc)
        assert dtype == BF16 or want_dgrad
        dev = weights_krsc[0].device
        self.dtype = dtype
        self.weights = list(weights_krsc)
        self.cast = [torch.empty(w.shape, device=dev, dtype=BF16) if dtype == BF16 else None for w in self.weights]
        self.dgrad = [torch.empty((w.shape[3], w.shape[1], w.shape[2], w.shape[0]), device=dev, dtype=dtype) if want_dgrad else None
                      for w in self.weights]
        items = (_WPrepItem * len(self.weights))()
        tiles = 0
        for i, w in enumerate(self.weights):
            cout, kh, kw, cin = w.shape
            items[i] = _WPrepItem(_dev(w, "w").value, _dev(self.cast[i], "cast", BF16).value if dtype == BF16 else None,
                                  _dev(self.dgrad[i], "dgrad", dtype).value if want_dgrad else None, cout, kh * kw, cin, tiles)
            tiles += kh * kw * ((cin + 31) // 32) * ((cout + 31) // 32)
        self.tiles = tiles
        self.table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        self.ptrs = tuple(w.data_ptr() for w in self.weights)

    def run(self):
        fn = _lib.lib().rs_weight_prep_bf16 if self.dtype == BF16 else _lib.lib().rs_weight_prep_f32
        check(fn(_dev(self.table, "items", torch.uint8), len(self.weights), self.tiles, _stream()), "rs_weight_prep")


def pack_stem_weight(w_krsc, dtype=torch.float32):
    """fp32 [Cout,kh,kw<=8,Cin<=4] -> [Cout,kh,8,4] (zero padded) in ``dtype``."""

    cout, kh, kw, cin = w_krsc.shape
    out = torch.empty((cout, kh, 8, 4), device=w_krsc.device, dtype=dtype)
    fn = _lib.lib().rs_pack_stem_weight_bf16 if dtype == BF16 else _lib.lib().rs_pack_stem_weight
    check(fn(_dev(w_krsc, "w"), _dev(out, "out", dtype), cout, kh, kw, cin, _stream()), "rs_pack_stem_weight")
    return out


def nchw_to_nhwc4(x, dtype=torch.float32):
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, 4), device=x.device, dtype=dtype)
    fn = _lib.lib().rs_nchw_to_nhwc4_bf16 if dtype == BF16 else _lib.lib().rs_nchw_to_nhwc4
    check(fn(_dev(x, "x"), _dev(out, "out", dtype), n, c, h, w, _stream()), "rs_nchw_to_nhwc4")
    return out


def stem_conv_bf16(x4, w_packed, scale=None, shift=None, relu=False):
    """resnet.conv1 (7x7/2, pad 3) on NHWC4 bf16 input with packed bf16 weights [64,7,8,4] -> [N,H/2,W/2,64] bf16."""

    n, h, w, _ = x4.shape
    out = torch.empty((n, h // 2, w // 2, 64), device=x4.device, dtype=BF16)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_stem_conv_fwd_bf16(_dev(x4, "x4", BF16), _dev(w_packed, "w", BF16), _dev(scale, "scale"),
                                          _dev(shift, "shift"), _dev(out, "out", BF16), n, h, w, int(relu), _stream())
    check(rc, "rs_stem_conv_fwd_bf16")
    if PROFILE is not None:
        ev1.record()
        flops = 2.0 * n * 64 * 3 * 49 * (h // 2) * (w // 2)
        _record("stem_conv_bf16", flops, (3, 64, 7, 2, 0, h // 2, w // 2), ev0, ev1, 2 * (x4.numel() + out.numel()))
    return out


def wgrad_kernel_name(d, form=None):
    """The bf16 weight-gradient kernel ``rs_conv2d_wgrad_bf16`` launches for ``d``, named like its instantiation (rocprofv3
    shows ``conv_wgrad_bf16<128, 128, 2, 2, 64, false>`` for ``conv_wgrad_bf16<128x128>``; ``phase`` = the last flag)."""

    lib = _lib.lib()
    if form is None:
        form = lib.rs_conv2d_wgrad_bf16_form(ctypes.byref(d))
    if form == 1:  # all-taps kernel: instantiated per input-channel slab (32 / 64 / 128) and for the fused x2 upsample
        return "conv_wgrad_thin_bf16<{}{}>".format(min(d.C1, 128), ",ups" if d.ups else "")
    t = lib.rs_conv2d_wgrad_bf16_tile(ctypes.byref(d))
    if t <= 0:
        raise ValueError("rs_conv2d_wgrad_bf16_tile: invalid arguments")
    tile = "{}x{}".format(t >> 16, t & 255) + ("+{}x{}".format(t >> 16, (t >> 8) & 255) if (t >> 8) & 255 else "")
    # phase form: its 128 x 128 launch is conv_wgrad_phase4_bf16 (one dz plane x four source offsets per block) unless knob wgrad_phase4 = 0
    phase = "" if form != 2 else ("phase4," if (t >> 16, t & 255) == (128, 128) and get_knob("wgrad_phase4") else "phase,")
    return "conv_wgrad_bf16<{}{}>".format(phase, tile)


def stem_conv_wgrad_bf16(dy, x4):
    """Packed fp32 gradient [64,7,8,4] of the stem filter from bf16 dy [N,H/2,W/2,64] and the NHWC4 bf16 input."""

    n, h, w, _ = x4.shape
    lib = _lib.lib()
    dw = torch.empty((64, 7, 8, 4), device=dy.device, dtype=torch.float32)
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.rs_stem_conv_wgrad_bf16(_dev(dy, "dy", BF16), _dev(x4, "x4", BF16), _dev(dw, "dw"), n, h, w,
                                     _workspace(lib.rs_stem_conv_wgrad_bf16_workspace_bytes(n, h, w), dy.device), _stream())
    check(rc, "rs_stem_conv_wgrad_bf16")
    if PROFILE is not None:
        ev1.record()
        flops = 2.0 * n * 64 * 3 * 49 * (h // 2) * (w // 2)
        _record("stem_wgrad_bf16", flops, (3, 64, 7, 2, 0, h // 2, w // 2), ev0, ev1, 2 * (x4.numel() + dy.numel()))
    return dw


def u8_to_nhwc4_norm(img, mean, std):
    """uint8 HWC tiles [N,H,W,C] -> normalised NHWC4 fp32: ToTensor + Normalize (tools/predict.py:71) on the device."""

    n, h, w, c = img.shape
    assert len(mean) == c and len(std) == c
    out = torch.empty((n, h, w, 4), device=img.device, dtype=torch.float32)
    fm, fs = (ctypes.c_float * c)(*mean), (ctypes.c_float * c)(*std)
    check(_lib.lib().rs_u8_to_nhwc4_norm(_dev(img, "img", torch.uint8), _dev(out, "out"), fm, fs, n, h, w, c, _stream()),
          "rs_u8_to_nhwc4_norm")
    return out


_ANCHORS = {}


def _anchors(device):
    import numpy as np

    anchors = _ANCHORS.get(device)
    if anchors is None:
        anchors = torch.from_numpy(np.linspace(0, 1, 256)).to(device)  # numpy's own float64 anchors
        _ANCHORS[device] = anchors
    return anchors


def final_conv1x1_quantize(x, w, bias, overlap):
    """self.final + softmax + crop of the `overlap` border + np.digitize(p_c, linspace(0,1,256)).astype(uint8) of every
    non-background class (tools/predict.py:87,96-103) in one kernel: uint8 [N, H-2*overlap, W-2*overlap] for a binary
    model (the reference's case, byte for byte), [N, H', W', C-1] for C > 2 classes."""

    n, h, wd, cin = x.shape
    c = w.shape[0]
    shape = (n, h - 2 * overlap, wd - 2 * overlap) + ((c - 1,) if c > 2 else ())
    out = torch.empty(shape, device=x.device, dtype=torch.uint8)
    rc = _lib.lib().rs_final_conv1x1_quantize_dt(_dev(x, "x", x.dtype), _dt(x), _dev(w, "w"), _dev(bias, "bias"),
                                                 _dev(_anchors(x.device), "anchors", torch.float64), _dev(out, "out", torch.uint8),
                                                 n, h, wd, cin, c, overlap, _stream())
    check(rc, "rs_final_conv1x1_quantize_dt")
    return out


def final_conv1x1_argmax(x, w, bias):
    """self.final + argmax over the classes: uint8 [N,H,W] class indices (tools/serve.py:160-164)."""

    n, h, wd, cin = x.shape
    out = torch.empty((n, h, wd), device=x.device, dtype=torch.uint8)
    rc = _lib.lib().rs_final_conv1x1_argmax_dt(_dev(x, "x", x.dtype), _dt(x), _dev(w, "w"), _dev(bias, "bias"),
                                               _dev(out, "out", torch.uint8), n, h, wd, cin, w.shape[0], _stream())
    check(rc, "rs_final_conv1x1_argmax_dt")
    return out


def maxpool2d(x, k, stride, pad, want_argmax=False, out_dtype=None):
    """``out_dtype`` (default: x.dtype): torch.bfloat16 on an fp32 input is the precision boundary of the bf16 path."""

    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((n, ho, wo, c), device=x.device, dtype=out_dtype or x.dtype)
    amax = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_argmax else None
    rc = _lib.lib().rs_maxpool2d_fwd_dt(_dev(x, "x", x.dtype), _dt(x), _dev(out, "out", out.dtype), _dt(out),
                                        _dev(amax, "argmax", torch.uint8), n, h, w, c, k, stride, pad, ho, wo, _stream())
    check(rc, "rs_maxpool2d_fwd_dt")
    return (out, amax) if want_argmax else out


def bn_fold(gamma, beta, mean, var, eps):
    c = gamma.numel()
    scale = torch.empty(c, device=gamma.device, dtype=torch.float32)
    shift = torch.empty(c, device=gamma.device, dtype=torch.float32)
    rc = _lib.lib().rs_bn_fold(_dev(gamma, "gamma"), _dev(beta, "beta"), _dev(mean, "mean"), _dev(var, "var"),
                               ctypes.c_float(eps), _dev(scale, "scale"), _dev(shift, "shift"), c, _stream())
    check(rc, "rs_bn_fold")
    return scale, shift


def final_conv1x1(x, w, bias, softmax=False):
    """x [N,H,W,Cin] NHWC, w [C,Cin] -> NCHW [N,C,H,W] logits (or probabilities if ``softmax``)."""

    n, h, wd, cin = x.shape
    c = w.shape[0]
    out = torch.empty((n, c, h, wd), device=x.device, dtype=torch.float32)
    rc = _lib.lib().rs_final_conv1x1_dt(_dev(x, "x", x.dtype), _dt(x), _dev(w, "w"), _dev(bias, "bias"), _dev(out, "out"), n, h,
                                        wd, cin, c, int(softmax), _stream())
    check(rc, "rs_final_conv1x1_dt")
    return out


# ------------------------------------------------------------------------------------------------------------------
# training path
# ------------------------------------------------------------------------------------------------------------------

_WORKSPACE = {}


def _workspace(nbytes, device):
    """A scratch buffer per (device, stream), grown on demand: users on one stream serialise; the weight-gradient side
    stream of the backward pass (robosat_amd.autograd) gets its own."""

    if nbytes < 0:
        raise ValueError("workspace query failed (RS_EINVAL)")
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), device=device, dtype=torch.uint8)
        _WORKSPACE[key] = ws
    return ctypes.c_void_p(ws.data_ptr())


def pack_dgrad_weight(w_krsc, dtype=torch.float32):
    """fp32 [Cout,kh,kw,Cin] -> [Cin,kh,kw,Cout] in ``dtype``, taps flipped (weights of the data-gradient convolution)."""

    cout, kh, kw, cin = w_krsc.shape
    out = torch.empty((cin, kh, kw, cout), device=w_krsc.device, dtype=dtype)
    fn = _lib.lib().rs_pack_dgrad_weight_bf16 if dtype == BF16 else _lib.lib().rs_pack_dgrad_weight
    check(fn(_dev(w_krsc, "w"), _dev(out, "out", dtype), cout, kh, kw, cin, _stream()), "rs_pack_dgrad_weight")
    return out


def unpack_stem_weight(packed, kw, cin, out=None):
    cout, kh = packed.shape[:2]
    if out is None:
        out = torch.empty((cout, kh, kw, cin), device=packed.device, dtype=torch.float32)
    assert tuple(out.shape) == (cout, kh, kw, cin)
    check(_lib.lib().rs_unpack_stem_weight(_dev(packed, "packed"), _dev(out, "out"), cout, kh, kw, cin, _stream()),
          "rs_unpack_stem_weight")
    return out


def conv2d_wgrad(dy, src1, kh, kw, src2=None, ups=0, stride=1, pad=0, stem=0, out=None):
    """``rs_conv2d_wgrad``: KRSC filter gradient [Cout,kh,kw,Cin] (packed [Cout,kh,8,4] for the stem)."""

    n, ho, wo, cout = dy.shape
    _, hs, ws, c1 = src1.shape
    c2 = 0 if src2 is None else src2.shape[3]
    d = ConvDesc(n, hs, ws, c1, c2, ups, kh, kw, stride, pad, ho, wo, cout, 0, int(bool(stem)))
    lib = _lib.lib()
    act = dy.dtype
    bf = act == BF16
    shape = (cout, kh, 8, 4) if stem else (cout, kh, kw, c1 + c2)
    dw = out if out is not None else torch.empty(shape, device=dy.device, dtype=torch.float32)
    assert tuple(dw.shape) == shape
    wsb = (lib.rs_conv2d_wgrad_bf16_workspace_bytes if bf else lib.rs_conv2d_wgrad_workspace_bytes)(ctypes.byref(d))
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    fn = lib.rs_conv2d_wgrad_bf16 if bf else lib.rs_conv2d_wgrad
    rc = fn(ctypes.byref(d), _dev(dy, "dy", act), _dev(src1, "src1", act), _dev(src2, "src2", act), _dev(dw, "dw"),
            _workspace(wsb, dy.device), _stream())
    check(rc, "rs_conv2d_wgrad_bf16" if bf else "rs_conv2d_wgrad")
    if PROFILE is not None:
        ev1.record()
        es = 2 if bf else 4
        nbytes = es * (d.N * d.Ho * d.Wo * d.Cout + d.N * d.Hs * d.Ws * (4 if stem else d.C1 + d.C2)) + 4 * dw.numel()
        form = lib.rs_conv2d_wgrad_bf16_form(ctypes.byref(d)) if bf else 0
        if not bf and not stem:
            # conv_wgrad.hip: 2 = the fp32 phase form of DecoderBlock (16 / 36 of the MACs), 3 = that in the Winograd domain (9 / 36),
            # 4 = a stride-1 3x3 convolution in the Winograd domain of F(2x2, 3x3) (16 / 36)
            form = lib.rs_conv2d_wgrad_form(ctypes.byref(d))
        # fp32: the LDS-DMA kernel (conv_wgrad_f32_dma.hip) for everything but the packed stem, unless knob wgrad_f32_dma = 0
        name = wgrad_kernel_name(d, form) if bf else ("conv_wgrad_f32" if stem or get_knob("wgrad_f32_dma") == 0 else
                                                      "conv_wgrad_wino_f32" if form == 3 else "conv_wgrad_wino33_f32" if form == 4 else
                                                      "conv_wgrad_f32_dma")
        _record(name, conv_flops(d), (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1, nbytes,
                conv_flops(d) * (0.25 if form == 3 else 4.0 / 9.0 if form in (2, 4) else 1.0))
    return dw


def _bn_ws(m, c, device):
    return _workspace(_lib.lib().rs_bn_workspace_bytes(m, c) + 3 * c * 4, device)


def bn_train_stats(y, gamma, beta, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """Batch statistics of y [N,H,W,C]; returns (mean, invstd, scale, shift) and updates the running buffers."""

    c = y.shape[-1]
    m = y.numel() // c
    mean, invstd, scale, shift = (torch.empty(c, device=y.device, dtype=torch.float32) for _ in range(4))
    rc = _lib.lib().rs_bn_train_stats_dt(
        _dev(y, "y", y.dtype), _dt(y), m, c, ctypes.c_float(eps), ctypes.c_float(momentum), _dev(gamma, "gamma"), _dev(beta, "beta"),
        _dev(mean, "mean"), _dev(invstd, "invstd"), _dev(scale, "scale"), _dev(shift, "shift"),
        _dev(running_mean, "running_mean"), _dev(running_var, "running_var"),
        _dev(num_batches_tracked, "num_batches_tracked", torch.int64), _bn_ws(m, c, y.device), _stream())
    check(rc, "rs_bn_train_stats")
    return mean, invstd, scale, shift


def bn_bits_ok(c):
    """Channel counts for which ``bn_apply(..., want_bits=True)`` is available (the streaming form: C divides 2048)."""

    return 8 <= c <= 2048 and 2048 % c == 0


def bn_apply(y, scale, shift, residual=None, relu=False, want_bits=False):
    """z = relu?(y * scale + shift (+ residual)).  ``want_bits``: returns (z, bits) with the ReLU mask of z as one bit per
    element (uint8 [numel/8]; bit e of byte i: element 8*i + e is > 0) for ``conv2d_dgrad_bnstats(relu_mask_bits=...)``."""

    c = y.shape[-1]
    out = torch.empty_like(y)
    bits = torch.empty(y.numel() // 8, device=y.device, dtype=torch.uint8) if want_bits else None
    rc = _lib.lib().rs_bn_apply_bits_dt(_dev(y, "y", y.dtype), _dev(scale, "scale"), _dev(shift, "shift"),
                                        _dev(residual, "residual", y.dtype), _dev(out, "out", y.dtype),
                                        _dev(bits, "bits", torch.uint8), _dt(y), y.numel() // c, c, int(relu), _stream())
    check(rc, "rs_bn_apply_bits_dt")
    return (out, bits) if want_bits else out


def bn_bwd(dz, zmask, y, mean, invstd, gamma, want_masked=False, dgamma=None, dbeta=None):
    """Returns (dy, dgamma, dbeta[, dmasked])."""

    c = y.shape[-1]
    m = y.numel() // c
    dy = torch.empty_like(y)
    dmasked = torch.empty_like(y) if want_masked else None
    if dgamma is None:
        dgamma = torch.empty(c, device=y.device, dtype=torch.float32)
    if dbeta is None:
        dbeta = torch.empty(c, device=y.device, dtype=torch.float32)
    t = y.dtype
    rc = _lib.lib().rs_bn_bwd_dt(_dev(dz, "dz", t), _dev(zmask, "zmask", t), _dev(y, "y", t), _dev(mean, "mean"),
                                 _dev(invstd, "invstd"), _dev(gamma, "gamma"), _dev(dy, "dy", t), _dev(dmasked, "dmasked", t),
                                 _dev(dgamma, "dgamma"), _dev(dbeta, "dbeta"), _dt(y), m, c, _bn_ws(m, c, y.device), _stream())
    check(rc, "rs_bn_bwd")
    return (dy, dgamma, dbeta, dmasked) if want_masked else (dy, dgamma, dbeta)


def scatter_add_stride2(t, out):
    """``out[:, ::2, ::2, :] += t`` in place (``rs_scatter_add_stride2_dt``): the data gradient of a 1x1 / stride-2 convolution is its
    transposed product on the low-resolution grid (``t``), landing on the even positions of the input grid."""

    n, hs, ws, c = t.shape
    assert out.shape[0] == n and out.shape[3] == c and out.dtype == t.dtype and out.is_contiguous()
    check(_lib.lib().rs_scatter_add_stride2_dt(_dev(t, "t", t.dtype), _dev(out, "out", t.dtype), _dt(t), n, hs, ws, out.shape[1], out.shape[2], c,
                                               _stream()), "rs_scatter_add_stride2_dt")
    return out


def maxpool2d_bwd(dy, argmax, in_shape, k, stride, pad, out=None, out_dtype=None):
    """Gradient wrt the pooling input [N,H,W,C]; ``out`` given => accumulate into it.  ``out_dtype`` fp32 on a bf16 dy
    is the precision boundary of the bf16 path (stem pool)."""

    n, h, w, c = in_shape
    ho, wo = dy.shape[1:3]
    acc = out is not None
    if out is None:
        out = torch.empty(in_shape, device=dy.device, dtype=out_dtype or dy.dtype)
    rc = _lib.lib().rs_maxpool2d_bwd_dt(_dev(dy, "dy", dy.dtype), _dt(dy), _dev(argmax, "argmax", torch.uint8),
                                        _dev(out, "dx", out.dtype), _dt(out), n, h, w, c, k, stride, pad, ho, wo, int(acc),
                                        _stream())
    check(rc, "rs_maxpool2d_bwd")
    return out


def upsample2x_bwd(dup, c1, c2=0, mask1=None, mask2=None, out1=None):
    """dup [N,2H,2W,C1+C2] -> (d1 [N,H,W,C1], d2 [N,H,W,C2] or None); ``out1`` given => accumulate into it."""

    n, h2, w2, ct = dup.shape
    assert ct == c1 + c2 and h2 % 2 == 0 and w2 % 2 == 0
    h, w = h2 // 2, w2 // 2
    acc = out1 is not None
    t = dup.dtype
    d1 = out1 if acc else torch.empty((n, h, w, c1), device=dup.device, dtype=t)
    d2 = torch.empty((n, h, w, c2), device=dup.device, dtype=t) if c2 else None
    rc = _lib.lib().rs_upsample2x_bwd_dt(_dev(dup, "dup", t), _dev(d1, "d1", t), _dev(d2, "d2", t), _dev(mask1, "mask1", t),
                                         _dev(mask2, "mask2", t), _dt(dup), n, h, w, c1, c2, int(acc), _stream())
    check(rc, "rs_upsample2x_bwd")
    return d1, d2


def final_conv1x1_bwd(x, w, dlogits, relu_mask=True, dw=None, db=None):
    """x [N,H,W,Cin] NHWC, w [C,Cin], dlogits NCHW -> (dx NHWC, dw [C,Cin], db [C])."""

    n, h, wd, cin = x.shape
    c = w.shape[0]
    lib = _lib.lib()
    dx = torch.empty_like(x)
    if dw is None:
        dw = torch.empty((c, cin), device=x.device, dtype=torch.float32)
    if db is None:
        db = torch.empty(c, device=x.device, dtype=torch.float32)
    ws = _workspace(lib.rs_final_conv1x1_bwd_workspace_bytes(cin, c), x.device)
    rc = lib.rs_final_conv1x1_bwd_dt(_dev(x, "x", x.dtype), _dev(w, "w"), _dev(dlogits, "dlogits"), _dev(dx, "dx", x.dtype),
                                     _dev(dw, "dw"), _dev(db, "db"), _dt(x), n, h, wd, cin, c, int(relu_mask), ws, _stream())
    check(rc, "rs_final_conv1x1_bwd")
    return dx, dw, db


# ------------------------------------------------------------------------------------------------------------------
# losses / metrics (NCHW logits, int64 targets)
# ------------------------------------------------------------------------------------------------------------------

NLL_CROSS_ENTROPY, NLL_FOCAL = 0, 1


def nll_loss_fwd(logits, targets, weight, mode, gamma=2.0):
    n, c, h, w = logits.shape
    lib = _lib.lib()
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    stats = torch.empty(2, device=logits.device, dtype=torch.float32)
    rc = lib.rs_nll_loss_fwd(_dev(logits, "logits"), _dev(targets, "targets", torch.int64), _dev(weight, "weight"),
                             _dev(loss, "loss"), _dev(stats, "stats"), n, c, h, w, mode, ctypes.c_float(gamma),
                             _workspace(lib.rs_nll_loss_workspace_bytes(), logits.device), _stream())
    check(rc, "rs_nll_loss_fwd")
    return loss, stats


def nll_loss_bwd(logits, targets, weight, stats, grad_out, mode, gamma=2.0):
    n, c, h, w = logits.shape
    dlogits = torch.empty_like(logits)
    rc = _lib.lib().rs_nll_loss_bwd(_dev(logits, "logits"), _dev(targets, "targets", torch.int64), _dev(weight, "weight"),
                                    _dev(stats, "stats"), _dev(grad_out, "grad_out"), _dev(dlogits, "dlogits"), n, c, h, w,
                                    mode, ctypes.c_float(gamma), _stream())
    check(rc, "rs_nll_loss_bwd")
    return dlogits


def miou_loss_fwd(logits, targets, weight):
    n, c, h, w = logits.shape
    lib = _lib.lib()
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    stats = torch.empty(5 + 2 * n * c, device=logits.device, dtype=torch.float32)
    rc = lib.rs_miou_loss_fwd(_dev(logits, "logits"), _dev(targets, "targets", torch.int64), _dev(weight, "weight"),
                              _dev(loss, "loss"), _dev(stats, "stats"), n, c, h, w,
                              _workspace(lib.rs_miou_loss_workspace_bytes(n, c), logits.device), _stream())
    check(rc, "rs_miou_loss_fwd")
    return loss, stats


def miou_loss_bwd(logits, targets, weight, stats, grad_out):
    n, c, h, w = logits.shape
    dlogits = torch.empty_like(logits)
    rc = _lib.lib().rs_miou_loss_bwd(_dev(logits, "logits"), _dev(targets, "targets", torch.int64), _dev(weight, "weight"),
                                     _dev(stats, "stats"), _dev(grad_out, "grad_out"), _dev(dlogits, "dlogits"), n, c, h, w,
                                     _stream())
    check(rc, "rs_miou_loss_bwd")
    return dlogits


def lovasz_fwd(logits, targets, want_grad=True):
    """Returns (loss, d loss / d logits for grad_out = 1 or None)."""

    n, c, h, w = logits.shape
    lib = _lib.lib()
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    rc = lib.rs_lovasz_fwd(_dev(logits, "logits"), _dev(targets, "targets", torch.int64), _dev(loss, "loss"),
                           _dev(grad, "grad"), n, c, h, w,
                           _workspace(lib.rs_lovasz_workspace_bytes(n, c, h, w), logits.device), _stream())
    check(rc, "rs_lovasz_fwd")
    return loss, grad


def scale_by_scalar(src, scalar):
    out = torch.empty_like(src)
    check(_lib.lib().rs_scale_by_scalar(_dev(src, "src"), _dev(scalar, "scalar"), _dev(out, "out"), src.numel(), _stream()),
          "rs_scale_by_scalar")
    return out


def confusion_counts(scores, targets, counts):
    """counts (uint64-as-int64 [4] device tensor) += (tn, fn, fp, tp) of the whole batch (reference naming)."""

    n, c, h, w = scores.shape
    rc = _lib.lib().rs_confusion_counts(_dev(scores, "scores"), _dev(targets, "targets", torch.int64),
                                        _dev(counts, "counts", torch.int64), n, c, h, w, _stream())
    check(rc, "rs_confusion_counts")
    return counts


def confusion_matrix(scores, targets, counts):
    """counts (int64 [C*C] device tensor, row = actual, column = predicted) += the batch's confusion matrix."""

    n, c, h, w = scores.shape
    assert counts.numel() == c * c
    rc = _lib.lib().rs_confusion_matrix(_dev(scores, "scores"), _dev(targets, "targets", torch.int64),
                                        _dev(counts, "counts", torch.int64), n, c, h, w, _stream())
    check(rc, "rs_confusion_matrix")
    return counts


def label_histogram_u8(labels, counts256):
    """counts256 (int64 [256] device tensor) += np.bincount(labels) of a uint8 device tensor (tools/weights.py:41-47)."""

    rc = _lib.lib().rs_label_histogram_u8(_dev(labels, "labels", torch.uint8), labels.numel(), _dev(counts256, "counts", torch.int64),
                                          _stream())
    check(rc, "rs_label_histogram_u8")
    return counts256


def softvote_masks(quantized, weights=None):
    """quantized uint8 [K, P] (binary models) or [K, P, C-1] -> uint8 [P] class indices: the weighted soft vote of
    tools/masks.py:42-84 over K models' probability bytes."""

    if quantized.dim() == 2:
        quantized = quantized.unsqueeze(-1)
    k, p, cq = quantized.shape
    out = torch.empty(p, device=quantized.device, dtype=torch.uint8)
    wt = None if weights is None else torch.as_tensor(list(weights), dtype=torch.float64).to(quantized.device)
    rc = _lib.lib().rs_softvote_masks(_dev(quantized, "quantized", torch.uint8), _dev(wt, "weights", torch.float64),
                                      _dev(_anchors(quantized.device), "anchors", torch.float64), _dev(out, "out", torch.uint8),
                                      k, p, cq + 1, _stream())
    check(rc, "rs_softvote_masks")
    return out


def augment_tiles(images, masks, index, op, mean, std):
    """Batch from a decoded-tile cache: images uint8 [T,S,S,C], masks uint8 [T,S,S] or None, index / op int32 [N] device
    tensors -> (images fp32 NCHW [N,C,S,S], masks int64 [N,S,S] or None): flip / rot90 / ToTensor / Normalize in one pass."""

    t, s, s2, c = images.shape
    assert s == s2, "square tiles (a 90-degree rotation must keep the shape)"
    n = index.numel()
    out = torch.empty((n, c, s, s), device=images.device, dtype=torch.float32)
    om = torch.empty((n, s, s), device=images.device, dtype=torch.int64) if masks is not None else None
    fm, fs = (ctypes.c_float * c)(*mean), (ctypes.c_float * c)(*std)
    rc = _lib.lib().rs_augment_tiles(_dev(images, "images", torch.uint8), _dev(masks, "masks", torch.uint8),
                                     _dev(index, "index", torch.int32), _dev(op, "op", torch.int32), fm, fs, _dev(out, "out"),
                                     _dev(om, "out_masks", torch.int64), n, s, c, _stream())
    check(rc, "rs_augment_tiles")
    return out, om
