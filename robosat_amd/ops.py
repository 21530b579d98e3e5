"""Thin tensor-level wrappers over the C ABI (``include/robosat_hip.h``).

Tensors are torch CUDA(=HIP) fp32 tensors used purely as device memory: activations are contiguous ``[N,H,W,C]``
(NHWC), convolution weights contiguous ``[Cout,kh,kw,Cin]`` (KRSC).  Work is enqueued on torch's current stream.
Nothing here computes with torch; a CPU tensor is an error (there is no CPU path).
"""

import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check


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


def conv_flops(d):
    """Algorithmic FLOPs of one launch: 2*N*Cout*Cin*kh*kw*Ho*Wo on the reference's shapes."""

    cin = 3 if d.stem else d.C1 + d.C2  # the stem's 4th (zero) band and 8th (zero) tap are not algorithmic work
    return 2.0 * d.N * d.Cout * cin * d.kh * d.kw * d.Ho * d.Wo


def conv_desc(src1, weight, src2=None, ups=0, stride=1, pad=0, relu=False, stem=0, out_hw=None):
    n, hs, ws, c1 = src1.shape
    cout, kh, kw_, _ = weight.shape
    c2 = 0 if src2 is None else src2.shape[3]
    kw = int(stem) if stem else kw_  # packed stem weights are [Cout][kh][8][4]; `stem` carries the true kw (7)
    if out_hw is None:
        assert ups in (0, 1)
        hv, wv = (hs * 2, ws * 2) if ups == 1 else (hs, ws)
        out_hw = ((hv + 2 * pad - kh) // stride + 1, (wv + 2 * pad - kw) // stride + 1)
    return ConvDesc(n, hs, ws, c1, c2, ups, kh, kw, stride, pad, out_hw[0], out_hw[1], cout, int(relu), int(bool(stem)))


def conv2d(src1, weight, src2=None, ups=0, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False,
           stem=0, out_hw=None, out=None):
    """``rs_conv2d_fwd``: out = relu?(conv(gather(src1|src2)) * scale + shift + residual).

    ``stem``: 0, or the true filter width (7) when ``weight`` is the packed ``[Cout,kh,8,4]`` stem filter."""

    d = conv_desc(src1, weight, src2, ups, stride, pad, relu, stem, out_hw)
    if out is None:
        out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=src1.device, dtype=torch.float32)
    if src2 is not None:
        assert src2.shape[:3] == src1.shape[:3]
    if not stem:
        assert weight.shape[3] == d.C1 + d.C2, "weight Cin {} != {}+{}".format(weight.shape[3], d.C1, d.C2)
    if residual is not None:
        assert residual.shape == out.shape
    if PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().rs_conv2d_fwd(
        ctypes.byref(d), _dev(src1, "src1"), _dev(src2, "src2"), _dev(weight, "weight"), _dev(scale, "scale"),
        _dev(shift, "shift"), _dev(residual, "residual"), _dev(out, "out"), _stream(),
    )
    check(rc, "rs_conv2d_fwd")
    if PROFILE is not None:
        ev1.record()
        PROFILE.append((conv_tile_name(d), conv_flops(d), (d.C1 + d.C2, d.Cout, d.kh, d.stride, d.ups, d.Ho, d.Wo), ev0, ev1))
    return out


def conv_tile_name(d):
    lib = _lib.lib()
    return lib.rs_conv2d_tile_name(lib.rs_conv2d_tile(ctypes.byref(d))).decode()


def pack_stem_weight(w_krsc):
    """[Cout,kh,kw<=8,Cin<=4] -> [Cout,kh,8,4] (zero padded)."""

    cout, kh, kw, cin = w_krsc.shape
    out = torch.empty((cout, kh, 8, 4), device=w_krsc.device, dtype=torch.float32)
    check(_lib.lib().rs_pack_stem_weight(_dev(w_krsc, "w"), _dev(out, "out"), cout, kh, kw, cin, _stream()),
          "rs_pack_stem_weight")
    return out


def nchw_to_nhwc4(x):
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, 4), device=x.device, dtype=torch.float32)
    check(_lib.lib().rs_nchw_to_nhwc4(_dev(x, "x"), _dev(out, "out"), n, c, h, w, _stream()), "rs_nchw_to_nhwc4")
    return out


def maxpool2d(x, k, stride, pad, want_argmax=False):
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    out = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
    amax = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_argmax else None
    rc = _lib.lib().rs_maxpool2d_fwd(_dev(x, "x"), _dev(out, "out"), _dev(amax, "argmax", torch.uint8), n, h, w, c, k,
                                     stride, pad, ho, wo, _stream())
    check(rc, "rs_maxpool2d_fwd")
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
    rc = _lib.lib().rs_final_conv1x1(_dev(x, "x"), _dev(w, "w"), _dev(bias, "bias"), _dev(out, "out"), n, h, wd, cin, c,
                                     int(softmax), _stream())
    check(rc, "rs_final_conv1x1")
    return out
