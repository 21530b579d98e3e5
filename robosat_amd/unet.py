"""MI355X-native U-Net with a ResNet-50 encoder: drop-in for the reference's ``robosat.unet`` module.

Same public surface as reference ``robosat/unet.py`` -- ``UNet(num_classes, num_filters=32, pretrained=True)``,
``forward(x[N,3,H,W]) -> logits[N,C,H,W]``, the same assertion on ``H, W % 32`` (unet.py:119-120) and the same
sub-module names, hence the same 329 state-dict keys in the same order (``resnet.conv1.weight`` ...
``final.bias``), so checkpoints written by either side load in the other.

What differs is everything underneath: the modules below only HOLD parameters.  All arithmetic of the forward (and
backward) pass runs in the hand-written gfx950 kernels of ``librobosat_hip.so`` (``include/robosat_hip.h``):
NHWC activations, KRSC weights (conv parameters are kept in ``torch.channels_last`` memory format, which *is* KRSC,
so no per-step weight transform exists), nearest-x2 upsample and skip concatenation folded into the convolution's
gather, eval-mode BatchNorm + residual + ReLU folded into the convolution epilogue.  There is no CPU path: calling the
model on a CPU tensor raises.
"""

import math
import os
import warnings

import torch
import torch.nn as nn

from . import ops

# torchvision 0.3.0 caches the ImageNet weights under this name (reference docker/Dockerfile.*:21)
_RESNET50_FILE = "resnet50-19c8e357.pth"


class _Conv(nn.Module):
    """Parameter holder for a bias-free/biased convolution; weight is [Cout,Cin,kh,kw] in channels_last (= KRSC)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, init="default"):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, k, stride, padding
        w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
        fan_in, fan_out = cin * k * k, cout * k * k
        if init == "resnet":  # torchvision: kaiming_normal_(mode="fan_out", nonlinearity="relu")
            nn.init.normal_(w, 0.0, math.sqrt(2.0 / fan_out))
        else:  # nn.Conv2d default: kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            nn.init.uniform_(w, -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in)))
        else:
            self.register_parameter("bias", None)

    def krsc(self, dtype=torch.float32):
        """The weight as a contiguous [Cout,kh,kw,Cin] view (no copy while the parameter stays channels_last).

        ``dtype=torch.bfloat16`` returns the bf16 compute copy of the fp32 master weight (cast on the device, cached
        until the parameter is modified -- i.e. re-cast once per optimizer step)."""

        w = self.weight.detach().permute(0, 2, 3, 1)
        w = w if w.is_contiguous() else w.contiguous()
        if dtype == torch.float32:
            return w
        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0])
        c = getattr(self, "_bf16", None)
        if c is None or c[0] != key:
            c = (key, ops.cast_bf16(w))
            self._bf16 = c
        return c[1]

    def dgrad_weight(self, dtype=torch.float32):
        """Data-gradient weights [Cin,kh,kw,Cout], taps flipped, in ``dtype``: the copy ``UNet.prep_bf16_weights`` made for
        the current parameter version if there is one, else packed now (``rs_pack_dgrad_weight[_bf16]``)."""

        c = getattr(self, "_dgrad" if dtype == torch.bfloat16 else "_dgrad_f32", None)
        if c is not None and c[0] == (self.weight.data_ptr(), self.weight._version, _GENERATION[0]):
            return c[1]
        return ops.pack_dgrad_weight(self.krsc(), dtype)

    def phase(self, dtype=torch.float32):
        """The four parity-specific 2x2 filters of this 3x3 convolution behind a nearest-x2 upsample (DecoderBlock),
        packed on the device from the fp32 master and cached until the parameter is modified."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0], dtype)
        c = getattr(self, "_phase", None)
        if c is None or c[0] != key:
            c = (key, ops.pack_phase_weight(self.krsc(), dtype))
            self._phase = c
        return c[1]

    def phase_wino(self):
        """The transformed filters U = G g G^T of the fp32 Winograd form of DecoderBlock (``rs_pack_wino_phase_weight`` on
        the phase pack), cached like ``phase``."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0])
        c = getattr(self, "_phase_wino", None)
        if c is None or c[0] != key:
            c = (key, ops.pack_wino_phase_weight(self.phase(torch.float32)))
            self._phase_wino = c
        return c[1]

    def wino33(self):
        """The transformed filters U = G g G^T of the fp32 Winograd F(2x2, 3x3) form (``rs_pack_wino33_weight``), cached like
        ``phase``."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0])
        c = getattr(self, "_wino33", None)
        if c is None or c[0] != key:
            c = (key, ops.pack_wino33_weight(self.krsc()))
            self._wino33 = c
        return c[1]

    def dgrad_wino33(self):
        """The transformed filters of the fp32 Winograd F(2x2, 3x3) form of this layer's DATA gradient (``rs_pack_wino33_weight`` on
        ``dgrad_weight``), cached like ``phase``."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0])
        c = getattr(self, "_dgrad_wino33", None)
        if c is None or c[0] != key:
            c = (key, ops.pack_wino33_weight(self.dgrad_weight(torch.float32)))
            self._dgrad_wino33 = c
        return c[1]

    def dgrad_s2_phase(self, dtype=torch.float32, wino=False):
        """The phase pack of this 3x3 / stride-2 convolution's DATA gradient (``rs_pack_s2_dgrad_phase_weight_dt``; ``wino``: its fp32
        Winograd transform, ``rs_pack_wino_phase_weight``), cached like ``phase``."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0], dtype, wino)
        c = getattr(self, "_dgrad_s2_phase", None)
        if c is None or c[0] != key:
            pack = ops.pack_s2_dgrad_phase_weight(self.krsc(), dtype)
            c = (key, ops.pack_wino_phase_weight(pack) if wino else pack)
            self._dgrad_s2_phase = c
        return c[1]

    def dgrad_phase(self, dtype=torch.float32):
        """Weights of the phase form's data gradient (one 4x4 / stride-2 convolution over dz, ``rs_pack_dgrad_phase_weight_dt``),
        cached like ``phase``."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0], dtype)
        c = getattr(self, "_dgrad_phase", None)
        if c is None or c[0] != key:
            c = (key, ops.pack_dgrad_phase_weight(self.krsc(), dtype))
            self._dgrad_phase = c
        return c[1]

    def dgrad_phase_wino(self):
        """The transformed filters of the fp32 Winograd form of DecoderBlock's data gradient (``rs_pack_wino_dgrad_weight`` on
        ``dgrad_phase``), cached like ``phase``."""

        key = (self.weight.data_ptr(), self.weight._version, _GENERATION[0])
        c = getattr(self, "_dgrad_phase_wino", None)
        if c is None or c[0] != key:
            c = (key, ops.pack_wino_dgrad_weight(self.dgrad_phase(torch.float32)))
            self._dgrad_phase_wino = c
        return c[1]

    def extra_repr(self):
        return "{}, {}, kernel_size={}, stride={}, padding={}".format(self.cin, self.cout, self.k, self.stride, self.padding)


# Derived tensors (bf16 casts, packed filters, folded BatchNorm, captured graphs) are cached per (data_ptr, _version) of their
# sources AND the generation below.  torch's version counters catch ordinary in-place updates (Adam's default multi-tensor
# step, load_state_dict, copy_), but not every writer bumps them -- ``torch.optim.Adam(fused=True)`` does not -- so the
# generation moves whenever stale copies could otherwise survive: at every training forward (the weights are re-derived once
# per step anyway) and at every ``train()`` / ``eval()`` switch (validation after fused steps).  ``UNet.invalidate_caches()``
# is the explicit form for anything else that writes through ``.data``.
_GENERATION = [0]


def _bump_generation():
    _GENERATION[0] += 1


class _BatchNorm(nn.Module):
    """Parameter/buffer holder with nn.BatchNorm2d's names and defaults (eps 1e-5, momentum 0.1)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._folded = None

    def folded(self):
        """Eval-mode scale/shift for the conv epilogue; cached until a parameter or buffer is modified."""

        ts = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((t.data_ptr(), t._version) for t in ts) + (_GENERATION[0],)
        if self._folded is None or self._folded[0] != key:
            scale, shift = ops.bn_fold(self.weight.detach(), self.bias.detach(), self.running_mean, self.running_var, self.eps)
            self._folded = (key, scale, shift)
        return self._folded[1], self._folded[2]

    def extra_repr(self):
        return "{}, eps={}, momentum={}".format(self.num_features, self.eps, self.momentum)


class _Linear(nn.Module):
    """``resnet.fc``: never used by UNet.forward, but part of every reference checkpoint (unet.py:94)."""

    def __init__(self, cin, cout):
        super().__init__()
        bound = 1.0 / math.sqrt(cin)
        self.weight = nn.Parameter(torch.empty(cout, cin).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))


class _Bottleneck(nn.Module):
    """torchvision-0.3.0 Bottleneck parameters: 1x1 -> 3x3 (stride) -> 1x1 (x4); attribute order = key order."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = _Conv(inplanes, planes, 1, init="resnet")
        self.bn1 = _BatchNorm(planes)
        self.conv2 = _Conv(planes, planes, 3, stride=stride, padding=1, init="resnet")
        self.bn2 = _BatchNorm(planes)
        self.conv3 = _Conv(planes, planes * 4, 1, init="resnet")
        self.bn3 = _BatchNorm(planes * 4)
        self.downsample = downsample
        self.stride = stride


class _ResNet50(nn.Module):
    def __init__(self, in_channels=3):
        super().__init__()
        self.conv1 = _Conv(in_channels, 64, 7, stride=2, padding=3, init="resnet")
        self.bn1 = _BatchNorm(64)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
            layer = []
            for b in range(blocks):
                s = stride if b == 0 else 1
                down = None
                if b == 0 and (s != 1 or inplanes != planes * 4):
                    down = nn.Sequential(_Conv(inplanes, planes * 4, 1, stride=s, init="resnet"), _BatchNorm(planes * 4))
                layer.append(_Bottleneck(inplanes, planes, s, down))
                inplanes = planes * 4
            setattr(self, "layer{}".format(i + 1), nn.Sequential(*layer))
        self.fc = _Linear(2048, 1000)


class ConvRelu(nn.Module):
    """3x3 convolution (padding 1, no bias) + ReLU; parameters under ``block`` (reference unet.py:18-44)."""

    def __init__(self, num_in, num_out):
        super().__init__()
        self.block = _Conv(num_in, num_out, 3, padding=1)


class DecoderBlock(nn.Module):
    """Nearest x2 upsample then ``ConvRelu``; parameters under ``block.block`` (reference unet.py:47-73)."""

    def __init__(self, num_in, num_out):
        super().__init__()
        self.block = ConvRelu(num_in, num_out)


def conv3x3_eval(conv, x, dt, scale=None, shift=None):
    """relu(conv3x3(x, pad 1) * scale + shift) of the eval-mode forward: a stride-1 Bottleneck conv2 with its folded BatchNorm,
    or dec5.  fp32 and large enough: the Winograd F(2x2, 3x3) kernel; else the generic implicit GEMM."""

    if dt == torch.float32 and conv.stride == 1 and ops.wino33_ok(x, conv.cout):
        return ops.conv2d_wino33(x, conv.wino33(), scale=scale, shift=shift, relu=True)
    return ops.conv2d(x, conv.krsc(dt), stride=conv.stride, pad=1, scale=scale, shift=shift, relu=True)


def decoder_block(conv, skip, prev, dt):
    """relu(conv3x3(interpolate(cat[skip, prev], x2 nearest), pad 1)) -- reference unet.py:63-73 -- on the source grid: in
    fp32 as the Winograd F(2x2, 2x2) form of the four parity convolutions where the layer qualifies (9/16 of the phase form's
    multiply-adds; the fp32 matrix cores are what bounds these layers), else the phase form itself (bf16; tiny layers)."""

    if dt == torch.float32 and ops.wino_ok(skip, prev, conv.cout):
        return ops.conv2d_phase_wino(skip, conv.phase_wino(), src2=prev, relu=True)
    return ops.conv2d_phase(skip, conv.phase(dt), src2=prev, relu=True)


def _find_pretrained():
    cands = [os.environ.get("ROBOSAT_RESNET50_WEIGHTS")]
    home = os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch"))
    cands += [os.path.join(home, "checkpoints", _RESNET50_FILE), os.path.join(home, "hub", "checkpoints", _RESNET50_FILE)]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    return None


class UNet(nn.Module):
    """ResNet-50-encoder U-Net ("AlbuNet"), computed by hand-written gfx950 kernels.

    Args (reference unet.py:82): ``num_classes``; ``num_filters`` (32); ``pretrained`` -- load ImageNet encoder
    weights from ``$ROBOSAT_RESNET50_WEIGHTS`` or torch's cache (there is no network access here; if no file is found
    the encoder keeps its random init and a warning is issued).  Extensions: ``in_channels`` (default 3, up to 4 bands)
    and ``compute_dtype`` -- ``torch.float32`` (default: exact-fp32 MFMA kernels, the parity path) or
    ``torch.bfloat16`` (BASELINE configs[2]: bf16 activations and MFMA operands, fp32 accumulation, fp32 master
    weights / statistics / gradients / logits; the 7x7 stem stays fp32).  Parameters and checkpoints are fp32 either
    way."""

    def __init__(self, num_classes, num_filters=32, pretrained=True, in_channels=3, compute_dtype=torch.float32):
        super().__init__()
        assert 1 <= in_channels <= 4, "the stem kernel packs up to 4 input bands"
        nf = num_filters
        self.num_classes, self.in_channels = num_classes, in_channels
        self.set_compute_dtype(compute_dtype)

        self.resnet = _ResNet50(in_channels)

        self.center = DecoderBlock(2048, nf * 8)

        self.dec0 = DecoderBlock(2048 + nf * 8, nf * 8)
        self.dec1 = DecoderBlock(1024 + nf * 8, nf * 8)
        self.dec2 = DecoderBlock(512 + nf * 8, nf * 2)
        self.dec3 = DecoderBlock(256 + nf * 2, nf * 2 * 2)
        self.dec4 = DecoderBlock(nf * 2 * 2, nf)
        self.dec5 = ConvRelu(nf, nf)

        self.final = _Conv(nf, num_classes, 1, bias=True)

        if pretrained:
            path = _find_pretrained()
            if path is None:
                if pretrained == "require":  # `rs train` without a checkpoint: the reference ALWAYS starts from ImageNet weights
                    raise FileNotFoundError(
                        "robosat_amd.UNet: no local {} found (set $ROBOSAT_RESNET50_WEIGHTS or put it in torch's checkpoint "
                        "cache; there is no network to download it), and training from a random encoder was not asked for "
                        "([model] pretrained = false)".format(_RESNET50_FILE))
                warnings.warn("robosat_amd.UNet(pretrained=True): no local {} found; encoder stays randomly initialised".format(_RESNET50_FILE))
            else:
                self.load_pretrained_encoder(torch.load(path, map_location="cpu"))

    def load_pretrained_encoder(self, state):
        """Load a torchvision ``resnet50`` state dict into the encoder.  The file torchvision 0.3.0 downloads
        (resnet50-19c8e357.pth, reference unet.py:94) predates BatchNorm's ``num_batches_tracked`` buffer;
        ``nn.BatchNorm2d`` tolerates that through its version-2 loading shim, which these parameter holders replicate:
        the missing counters stay at 0 -- and nothing else may be missing or unexpected."""

        state = dict(state)
        if self.in_channels != 3:
            state.pop("conv1.weight", None)  # a 4-band stem keeps its own initialisation
        result = self.resnet.load_state_dict(state, strict=False)
        missing = [k for k in result.missing_keys if not k.endswith(".num_batches_tracked")]
        if self.in_channels != 3:
            missing = [k for k in missing if k != "conv1.weight"]
        if missing or result.unexpected_keys:
            raise RuntimeError("pretrained resnet50 state dict does not fit: missing {}, unexpected {}".format(
                missing, list(result.unexpected_keys)))

    # -- plumbing -----------------------------------------------------------------------------------------------

    def set_compute_dtype(self, dtype):
        if isinstance(dtype, str):
            dtype = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16,
                     "bfloat16": torch.bfloat16}.get(dtype.lower())
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_state_list", None)  # (.to() may replace the parameter tensors)
        self.__dict__.pop("_graphs", None)
        for m in self.modules():  # .to()/.cuda() may drop the KRSC layout of size-1-dim weights: restore it
            if isinstance(m, _Conv) and not m.weight.is_contiguous(memory_format=torch.channels_last):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            if isinstance(m, _BatchNorm):
                m._folded = None
        return out

    def invalidate_caches(self):
        """Drop every derived copy of the parameters / buffers (bf16 casts, packed filters, folded BatchNorm, captured
        graphs).  Needed only after a write torch's version counters do not see (e.g. through ``.data``); optimizer steps
        -- fused ones included --, ``load_state_dict`` and ``train()`` / ``eval()`` switches are handled."""

        _bump_generation()
        self.__dict__.pop("_graphs", None)
        return self

    def train(self, mode=True):
        self.invalidate_caches()
        return super().train(mode)

    def _blocks(self):
        r = self.resnet
        return [list(r.layer1), list(r.layer2), list(r.layer3), list(r.layer4)]

    def prep_bf16_weights(self):
        """Refresh, in ONE launch, the bf16 compute copies (KRSC cast + data-gradient layout) of the convolutions that
        use them in a bf16 training step -- the 52 encoder convolutions and dec5 -- if any master weight changed since
        the last refresh (i.e. once per optimizer step).  ``_Conv.krsc(bf16)`` / ``_Conv.dgrad_weight(bf16)`` then hit
        these copies; without this call they cast / pack per tensor as before (105 small launches per step)."""

        self._prep_weights(torch.bfloat16)

    def prep_f32_weights(self):
        """The fp32 training step's twin (round 5): the data-gradient layouts of the same 53 convolutions in one launch
        (``rs_weight_prep_f32``) instead of one ``rs_pack_dgrad_weight`` per convolution in front of its data gradient
        (59 launches of ~5 us on the backward's main stream)."""

        self._prep_weights(torch.float32)

    def _prep_weights(self, dt):
        convs = []
        for layer in self._blocks():
            for blk in layer:
                convs += [blk.conv1, blk.conv2, blk.conv3] + ([blk.downsample[0]] if blk.downsample is not None else [])
        convs.append(self.dec5.block)
        keys = [(c.weight.data_ptr(), c.weight._version, _GENERATION[0]) for c in convs]
        slot = "_wprep" if dt == torch.bfloat16 else "_wprep_f32"
        st = getattr(self, slot, None)
        if st is not None and st[1] == keys:
            return
        ws = [c.krsc() for c in convs]
        if st is None or st[0].ptrs != tuple(w.data_ptr() for w in ws):  # first use, or the parameters moved (.to(), load)
            st = [ops.WeightPrep(ws, dtype=dt), None]
        st[0].run()
        st[1] = keys
        setattr(self, slot, st)
        for c, k, cast, dg in zip(convs, keys, st[0].cast, st[0].dgrad):
            if dt == torch.bfloat16:
                c._bf16 = (k, cast)
                c._dgrad = (k, dg)
            else:
                c._dgrad_f32 = (k, dg)

    # -- hipGraph replay of the eval forward (latency path) -------------------------------------------------------

    def _graph_replay(self, kind, fn, x):
        """Run ``fn(x)`` through a captured hipGraph (one per input shape / dtype / entry point).

        A small-batch forward is ~70 launches of 5-100 us kernels: the host (Python + ctypes, ~10 us per launch) is then
        slower than the GPU, and ``rs serve``'s single-tile latency is launch-bound.  Capturing the launch sequence once
        and replaying it removes the host from the loop.  The graph bakes in device addresses, so it is keyed on the input
        signature AND on the version counters of every parameter / buffer: any in-place update (an optimizer step, a
        ``load_state_dict``) drops the stale graphs.  Returns a fresh tensor (the graph's static output is reused)."""

        sig = (kind, tuple(x.shape), x.dtype, x.device, self.compute_dtype,
               tuple(t._version for t in self._state_tensors()), tuple(t.data_ptr() for t in self._state_tensors()[:4]),
               _GENERATION[0])
        cache = self.__dict__.setdefault("_graphs", {})
        entry = cache.get(sig)
        if entry is None:
            if len(cache) >= 8:
                cache.clear()
            static_x = x.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):  # warm-up off the capture: fills the weight / folded-BN caches, the workspaces
                fn(static_x)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = fn(static_x)
            entry = (graph, static_x, static_out)
            cache[sig] = entry
        graph, static_x, static_out = entry
        static_x.copy_(x)
        graph.replay()
        return static_out.clone()

    def _state_tensors(self):
        st = self.__dict__.get("_state_list")
        if st is None:
            st = [p for p in self.parameters()] + [b for b in self.buffers()]
            self.__dict__["_state_list"] = st
        return st

    @staticmethod
    def _want_graph(x):
        """Graphs for the latency regime (a few tiles per call); ROBOSAT_GRAPHS=0|1 overrides."""

        forced = os.environ.get("ROBOSAT_GRAPHS")
        if forced is not None:
            return forced == "1"
        return x.shape[0] * x.shape[1] * x.shape[2] <= 2 * 512 * 512

    # -- forward ------------------------------------------------------------------------------------------------

    def forward(self, x):
        """Logits [N,num_classes,H,W] (NCHW fp32), as reference unet.py:110-141."""

        size = x.size()
        assert size[-1] % 32 == 0 and size[-2] % 32 == 0, "image resolution has to be divisible by 32 for resnet"
        if self.training and torch.is_grad_enabled():
            from .autograd import unet_train_forward

            return unet_train_forward(self, x)
        if self.training:
            # train mode without autograd (e.g. a forward under torch.no_grad() to refresh BatchNorm statistics): the
            # training forward -- batch statistics, running buffers updated -- with its tape thrown away
            from .autograd import _Tape, _forward

            if not x.is_cuda:
                raise RuntimeError("robosat_amd.UNet runs on the MI355X only (got a {} tensor); there is no CPU fallback".format(x.device))
            assert x.size(1) == self.in_channels
            with torch.no_grad():
                return _forward(self, x.detach().float().contiguous(), _Tape())
        return self._forward_eval(x, softmax=False)

    @torch.no_grad()
    def predict_probs(self, x):
        """softmax(forward(x), dim=1) with the softmax fused into the last kernel (tools/predict.py:84-87)."""

        size = x.size()
        assert size[-1] % 32 == 0 and size[-2] % 32 == 0, "image resolution has to be divisible by 32 for resnet"
        assert not self.training, "predict_probs is an eval-mode call"
        return self._forward_eval(x, softmax=True)

    @torch.no_grad()
    def predict_quantized(self, images_u8, overlap=0, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """The whole device side of ``rs predict`` for a batch of decoded tiles:
        uint8 HWC ``[N,H,W,C]`` in -> ToTensor + Normalize -> U-Net -> softmax -> crop of the ``overlap`` border ->
        ``np.digitize(p_foreground, np.linspace(0,1,256)).astype(uint8)`` out, ``[N,H-2*overlap,W-2*overlap]`` uint8:
        exactly the bytes the reference writes into its probability PNGs (tools/predict.py:71-103), 1 byte per pixel each
        way over PCIe instead of 12 in / 8 out.  Models with more than two classes (which the reference's predict tool
        asserts away) return ``[N,H',W',C-1]``: the same encoding for every non-background class."""

        assert not self.training, "predict_quantized is an eval-mode call"
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.size(3) == self.in_channels
        assert images_u8.size(1) % 32 == 0 and images_u8.size(2) % 32 == 0, "image resolution has to be divisible by 32 for resnet"
        if not images_u8.is_cuda:
            raise RuntimeError("robosat_amd.UNet runs on the MI355X only (got a {} tensor); there is no CPU fallback".format(images_u8.device))
        mean, std = list(mean)[:self.in_channels], list(std)[:self.in_channels]

        def run(u8):
            x4 = ops.u8_to_nhwc4_norm(u8, mean, std)
            if self.compute_dtype == torch.bfloat16:
                x4 = x4.to(torch.bfloat16)  # (device-side cast of the 4-channel image; the stem then runs in bf16)
            return self._forward_eval(None, softmax=False, x4=x4, quantize_overlap=overlap)

        images_u8 = images_u8.contiguous()
        if self._want_graph(images_u8):
            return self._graph_replay(("quantized", overlap, tuple(mean), tuple(std)), run, images_u8)
        return run(images_u8)

    @torch.no_grad()
    def predict_classes(self, images_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """The device side of ``rs serve``'s ``Predictor.segment`` (reference tools/serve.py:149-164): uint8 HWC tiles
        ``[N,H,W,C]`` in -> ToTensor + Normalize -> U-Net -> ``argmax`` over the class logits -> uint8 ``[N,H,W]`` class
        indices out (``self.final`` and the argmax are one kernel; the logits never reach HBM)."""

        assert not self.training, "predict_classes is an eval-mode call"
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.size(3) == self.in_channels
        assert images_u8.size(1) % 32 == 0 and images_u8.size(2) % 32 == 0, "image resolution has to be divisible by 32 for resnet"
        if not images_u8.is_cuda:
            raise RuntimeError("robosat_amd.UNet runs on the MI355X only (got a {} tensor); there is no CPU fallback".format(images_u8.device))
        mean, std = list(mean)[:self.in_channels], list(std)[:self.in_channels]

        def run(u8):
            x4 = ops.u8_to_nhwc4_norm(u8, mean, std)
            if self.compute_dtype == torch.bfloat16:
                x4 = x4.to(torch.bfloat16)
            return self._forward_eval(None, softmax=False, x4=x4, argmax=True)

        images_u8 = images_u8.contiguous()
        if self._want_graph(images_u8):
            return self._graph_replay(("classes", tuple(mean), tuple(std)), run, images_u8)
        return run(images_u8)

    def _forward_eval(self, x, softmax, x4=None, quantize_overlap=None, argmax=False):
        r = self.resnet
        dt = self.compute_dtype  # activations after the stem pool (the stem itself always runs in fp32)
        if x4 is None:
            if not x.is_cuda:
                raise RuntimeError("robosat_amd.UNet runs on the MI355X only (got a {} tensor); there is no CPU fallback".format(x.device))
            assert x.size(1) == self.in_channels
            x = x.detach().float().contiguous()
            x4 = ops.nchw_to_nhwc4(x, dt)
        h = x4
        sc, sh = r.bn1.folded()
        if dt == torch.bfloat16:
            h = ops.stem_conv_bf16(h, ops.pack_stem_weight(r.conv1.krsc(), dt), scale=sc, shift=sh, relu=True)
        else:
            h = ops.conv2d(h, ops.pack_stem_weight(r.conv1.krsc()), stride=2, pad=3, scale=sc, shift=sh, relu=True, stem=7,
                           bands=self.in_channels)
        h = ops.maxpool2d(h, 3, 2, 1)

        enc = []
        for layer in self._blocks():
            blocks = list(layer)
            ahead = None  # this block's conv1 -> bn1 -> ReLU output when the previous block's fused tail already produced it
            for bi, blk in enumerate(blocks):
                if ahead is None:
                    sc, sh = blk.bn1.folded()
                    o = ops.conv2d(h, blk.conv1.krsc(dt), scale=sc, shift=sh, relu=True)
                else:
                    o, ahead = ahead, None
                sc, sh = blk.bn2.folded()
                o = conv3x3_eval(blk.conv2, o, dt, scale=sc, shift=sh)
                if blk.downsample is not None:
                    sc, sh = blk.downsample[1].folded()
                    dw = blk.downsample[0].krsc(dt)
                    if blk.stride == 1 and ops.conv1x1_wave_ok(h, dw):  # (layer1: 64 -> 256 on the fused tail's first stage)
                        idt = ops.conv1x1_wave(h, dw, sc, sh)
                    else:
                        idt = ops.conv2d(h, dw, stride=blk.stride, scale=sc, shift=sh)
                else:
                    idt = h
                sc, sh = blk.bn3.folded()
                nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
                if (nxt is not None and nxt.conv1.k == 1 and nxt.conv1.stride == 1 and dt == torch.float32
                        and ops.bottleneck_tail_ok(o, blk.conv3.krsc(dt), nxt.conv1.krsc(dt))):
                    # layer1: conv3 -> bn3 -> + identity -> ReLU and the NEXT block's conv1 -> bn1 -> ReLU in one launch (the second
                    # product reads the first one's accumulators: the 256-channel tensor is written once and not read back)
                    sc1, sh1 = nxt.bn1.folded()
                    h, ahead = ops.bottleneck_tail(o, blk.conv3.krsc(dt), sc, sh, idt, nxt.conv1.krsc(dt), sc1, sh1)
                else:  # (layer1's last conv3 stays on conv1x1_ew_f32: with a residual the wave form measured 140 against 127 us)
                    h = ops.conv2d(o, blk.conv3.krsc(dt), scale=sc, shift=sh, residual=idt, relu=True)
            enc.append(h)
        enc1, enc2, enc3, enc4 = enc

        def up(block, skip, prev=None):  # DecoderBlock in phase form: four 2x2 convolutions on the source grid
            return decoder_block(block.block.block, skip, prev, dt)

        center = up(self.center, ops.maxpool2d(enc4, 2, 2, 0))
        dec0 = up(self.dec0, enc4, center)
        dec1 = up(self.dec1, enc3, dec0)
        dec2 = up(self.dec2, enc2, dec1)
        dec3 = up(self.dec3, enc1, dec2)
        dec4 = up(self.dec4, dec3)
        wf = self.final.weight.detach().reshape(self.num_classes, -1)
        if dt == torch.float32 and ops.wino33_head_ok(dec4, self.dec5.block.cout, self.num_classes):
            # dec5 + final (+ softmax / quantise / argmax) in one launch: dec5's output never leaves the CU (unet.py:139-141)
            mode = "argmax" if argmax else ("quantize" if quantize_overlap is not None else ("softmax" if softmax else "logits"))
            return ops.conv2d_wino33_head(dec4, self.dec5.block.wino33(), wf, self.final.bias.detach(), mode,
                                          overlap=quantize_overlap or 0)
        dec5 = conv3x3_eval(self.dec5.block, dec4, dt)
        if argmax:
            return ops.final_conv1x1_argmax(dec5, wf, self.final.bias.detach())
        if quantize_overlap is not None:
            return ops.final_conv1x1_quantize(dec5, wf, self.final.bias.detach(), quantize_overlap)
        return ops.final_conv1x1(dec5, wf, self.final.bias.detach(), softmax=softmax)
