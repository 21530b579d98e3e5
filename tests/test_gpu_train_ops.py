"""Parity of the training-path kernels on the MI355X against PyTorch autograd on the host CPU (fp32), op by op."""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import robosat_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def krsc(w):
    return w.permute(0, 2, 3, 1).contiguous().to(DEV)


def close(got, want, tol=3e-4, what=""):
    scale = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, "{} max abs err {} (scale {})".format(what, err, scale)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


WGRAD = [
    # name, N, Cin, H, W, Cout, k, stride, pad   -> wgrad tile
    ("128x128", 2, 128, 20, 20, 128, 3, 1, 1),
    ("128x64", 2, 64, 24, 24, 256, 1, 1, 0),
    ("64x128", 2, 128, 16, 16, 64, 3, 1, 1),
    ("64x64_s2", 2, 64, 22, 18, 64, 3, 2, 1),
    ("32x128", 1, 128, 32, 32, 32, 3, 1, 1),
    ("32x32", 2, 32, 40, 40, 32, 3, 1, 1),
    ("1x1_s2", 2, 256, 16, 16, 512, 1, 2, 0),
    ("split_big", 2, 32, 128, 128, 32, 3, 1, 1),
]


@pytest.mark.parametrize("case", WGRAD, ids=[c[0] for c in WGRAD])
def test_wgrad_and_dgrad(case):
    from robosat_amd import ops

    _, n, cin, h, w, cout, k, stride, pad = case
    x = rnd(n, cin, h, w, seed=1).requires_grad_(True)
    wt = (rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    gy = rnd(*y.shape, seed=3)
    y.backward(gy)
    dw = ops.conv2d_wgrad(nhwc(gy), nhwc(x.detach()), k, k, stride=stride, pad=pad)
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, what="wgrad")
    wd = ops.pack_dgrad_weight(krsc(wt.detach()))
    dx = ops.conv2d(nhwc(gy), wd, ups=2 if stride == 2 else 0, pad=k - 1 - pad, out_hw=(h, w))
    close(nchw(dx), x.grad, what="dgrad")


@pytest.mark.parametrize("case", WGRAD, ids=[c[0] for c in WGRAD])
def test_wgrad_fp32_lds_dma_kernel_vs_register_staged_kernel(case):
    """conv_wgrad_f32_dma.hip (round 5: both operands HBM -> LDS by LDS-DMA as they lie, one dword per MFMA operand; what every
    non-stem fp32 weight gradient runs) against autograd AND against conv_wgrad.hip's register-staged kernel on the same
    launch (knob ``wgrad_f32_dma`` = 0): every tile variant, stride 2, a reduction that is not a multiple of the 32-pixel
    chunk, many splits.  The two pair the pixels of a k-step differently: equal to fp32 summation-order noise."""
    from robosat_amd import ops

    _, n, cin, h, w, cout, k, stride, pad = case
    x = rnd(n, cin, h, w, seed=11).requires_grad_(True)
    wt = (rnd(cout, cin, k, k, seed=12) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    gy = rnd(*y.shape, seed=13)
    y.backward(gy)
    args = (nhwc(gy), nhwc(x.detach()), k, k)
    assert ops.get_knob("wgrad_f32_dma") != 0  # the shipped setting
    dma = ops.conv2d_wgrad(*args, stride=stride, pad=pad)
    with ops.knob("wgrad_f32_dma", 0):
        staged = ops.conv2d_wgrad(*args, stride=stride, pad=pad)
    close(dma.permute(0, 3, 1, 2).cpu(), wt.grad, what="LDS-DMA wgrad vs autograd")
    close(staged.permute(0, 3, 1, 2).cpu(), wt.grad, what="register-staged wgrad vs autograd")
    assert float((dma - staged).abs().max()) <= 2e-5 * float(staged.abs().max())


@pytest.mark.parametrize("c1,c2,cout", [(256, 64, 128), (128, 0, 32), (512, 256, 64)])
def test_decoder_block_backward(c1, c2, cout):
    """wgrad through the fused upsample+concat gather, dgrad + 2x2 sum + split + ReLU masks."""
    from robosat_amd import ops

    n, h, w = 2, 8, 12
    a = rnd(n, c1, h, w, seed=4).requires_grad_(True)
    b = F.relu(rnd(n, c2, h, w, seed=5)).requires_grad_(True) if c2 else None
    wt = (rnd(cout, c1 + c2, 3, 3, seed=6) * 0.03).requires_grad_(True)
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = rnd(*y.shape, seed=7)
    y.backward(gy)
    dw = ops.conv2d_wgrad(nhwc(gy), nhwc(a.detach()), 3, 3, src2=nhwc(b.detach()) if c2 else None, ups=1, pad=1)
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, what="wgrad")
    dup = ops.conv2d(nhwc(gy), ops.pack_dgrad_weight(krsc(wt.detach())), pad=1)
    mask2 = nhwc(b.detach()) if c2 else None
    d1, d2 = ops.upsample2x_bwd(dup, c1, c2, mask2=mask2)
    close(nchw(d1), a.grad, what="d skip")
    if c2:
        close(nchw(d2), b.grad * (b.detach() > 0), what="d prev (masked)")
    acc = torch.ones_like(d1)
    d1b, _ = ops.upsample2x_bwd(dup, c1, c2, mask2=mask2, out1=acc)
    close(nchw(d1b), a.grad + 1.0, what="accumulate")


@pytest.mark.parametrize("n,c1,c2,cout,h,w", [(3, 64, 32, 64, 9, 13), (1, 128, 0, 128, 33, 20), (5, 32, 0, 32, 16, 16)])
def test_decoder_wgrad_fp32_phase_form_vs_autograd_and_direct_form(n, c1, c2, cout, h, w):
    """The fp32 weight gradient of DecoderBlock in phase form (16 parity / offset reductions over the SOURCE pixels + a
    combine: 16/36 of the multiply-adds, conv_wgrad.hip) against autograd on the reference formulation (unet.py:63-73) and
    against the direct form (nine taps over the upsampled pixels) on the same launch: odd sizes, splits that straddle images,
    two sources."""
    from robosat_amd import ops

    a = rnd(n, c1, h, w, seed=41).requires_grad_(True)
    b = rnd(n, c2, h, w, seed=42).requires_grad_(True) if c2 else None
    wt = (rnd(cout, c1 + c2, 3, 3, seed=43) * 0.05).requires_grad_(True)
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = rnd(*y.shape, seed=44)
    y.backward(gy)
    args = (nhwc(gy), nhwc(a.detach()), 3, 3)
    kw = dict(src2=nhwc(b.detach()) if c2 else None, ups=1, pad=1)
    dw = ops.conv2d_wgrad(*args, **kw)
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, what="phase-form wgrad")
    with ops.knob("wgrad_f32_phase", 0):
        direct = ops.conv2d_wgrad(*args, **kw)
    with ops.knob("wgrad_f32_dma", 0):  # the register-staged kernel's phase form (round 4) on the same launch
        staged = ops.conv2d_wgrad(*args, **kw)
    assert float((dw - staged).abs().max()) <= 2e-5 * float(staged.abs().max())
    close(direct.permute(0, 3, 1, 2).cpu(), wt.grad, what="direct-form wgrad")
    assert float((dw - direct).abs().max()) <= 1e-4 * float(direct.abs().max())


@pytest.mark.parametrize("n,c1,c2,cout,h,w", [(2, 64, 0, 64, 8, 8), (1, 128, 64, 32, 9, 13), (3, 64, 64, 64, 16, 12), (2, 192, 128, 128, 7, 5), (1, 64, 0, 96, 33, 20)])
def test_decoder_wgrad_fp32_winograd_domain_vs_autograd_and_phase_form(n, c1, c2, cout, h, w):
    """DecoderBlock's fp32 weight gradient in the Winograd domain of the forward's F(2x2, 2x2) form (conv_wgrad_wino_f32.hip, round 6:
    dU = sum over tiles of (A dY A^T) (.) (B^T d B), dg = G^T dU G; 9/16 of the phase form's multiply-adds) against autograd on the
    reference formulation (unet.py:63-73) and against the phase form on the same launch: odd sizes (half tiles at the edge), two
    sources, both block shapes (64 and 32 couts), splits that straddle images."""
    import ctypes

    from robosat_amd import _lib, ops

    a = rnd(n, c1, h, w, seed=51).requires_grad_(True)
    b = rnd(n, c2, h, w, seed=52).requires_grad_(True) if c2 else None
    wt = (rnd(cout, c1 + c2, 3, 3, seed=53) * 0.05).requires_grad_(True)
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = rnd(*y.shape, seed=54)
    y.backward(gy)
    args = (nhwc(gy), nhwc(a.detach()), 3, 3)
    kw = dict(src2=nhwc(b.detach()) if c2 else None, ups=1, pad=1)
    d = ops.ConvDesc(n, h, w, c1, c2, 1, 3, 3, 1, 1, 2 * h, 2 * w, cout, 0, 0)
    assert ops.get_knob("wgrad_f32_wino") == 1 and _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d)) == 3  # the shipped setting
    ops.PROFILE = []
    try:
        dw = ops.conv2d_wgrad(*args, **kw)
        assert [r[0] for r in ops.PROFILE] == ["conv_wgrad_wino_f32"], ops.PROFILE
    finally:
        ops.PROFILE = None
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, what="Winograd-domain wgrad")
    with ops.knob("wgrad_f32_wino", 0):
        assert _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d)) == 2
        phase = ops.conv2d_wgrad(*args, **kw)
    assert float((dw - phase).abs().max()) <= 2e-5 * float(phase.abs().max())
    with ops.knob("wgrad_f32_wino_blocks", 64):  # another split of the tile sequence: same sums up to fp32 order
        few = ops.conv2d_wgrad(*args, **kw)
    assert float((dw - few).abs().max()) <= 2e-5 * float(phase.abs().max())
    assert torch.equal(dw, ops.conv2d_wgrad(*args, **kw))  # deterministic: no atomics, splits summed in order


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 8, 8), (1, 32, 32, 9, 13), (3, 64, 128, 16, 12), (2, 128, 64, 7, 5), (1, 64, 96, 33, 20), (2, 64, 64, 34, 50)])
def test_wgrad_fp32_3x3_winograd_domain_vs_autograd_and_nine_taps(n, cin, cout, h, w):
    """The fp32 weight gradient of a stride-1 3x3 / pad-1 convolution (torchvision Bottleneck.conv2 under tools/train.py:186) in the
    Winograd domain of F(2x2, 3x3) (conv_wgrad_wino33_f32.hip, round 6: dU = sum over tiles of (A dY A^T) (.) (B^T d B), dg = G^T dU G;
    16/36 of the multiply-adds) against autograd and against the nine-tap kernel on the same launch: odd sizes (half tiles, rows that
    end inside a chunk of eight tiles), more than one chunk per row, both block shapes, splits that straddle images."""
    import ctypes

    from robosat_amd import _lib, ops

    a = rnd(n, cin, h, w, seed=61)
    wt = (rnd(cout, cin, 3, 3, seed=63) * 0.05).requires_grad_(True)
    y = F.conv2d(a, wt, padding=1)
    gy = rnd(*y.shape, seed=64)
    y.backward(gy)
    args = (nhwc(gy), nhwc(a), 3, 3)
    d = ops.ConvDesc(n, h, w, cin, 0, 0, 3, 3, 1, 1, h, w, cout, 0, 0)
    assert ops.get_knob("wgrad_f32_wino33") == 1 and _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d)) == 4  # the shipped setting
    ops.PROFILE = []
    try:
        dw = ops.conv2d_wgrad(*args, pad=1)
        assert [r[0] for r in ops.PROFILE] == ["conv_wgrad_wino33_f32"], ops.PROFILE
    finally:
        ops.PROFILE = None
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, what="Winograd-domain wgrad")
    with ops.knob("wgrad_f32_wino33", 0):
        assert _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d)) == 0
        nine = ops.conv2d_wgrad(*args, pad=1)
    assert float((dw - nine).abs().max()) <= 2e-5 * float(nine.abs().max())
    with ops.knob("wgrad_f32_wino33_blocks", 16):  # another split of the chunk sequence: same sums up to fp32 order
        few = ops.conv2d_wgrad(*args, pad=1)
    assert float((dw - few).abs().max()) <= 2e-5 * float(nine.abs().max())
    assert torch.equal(dw, ops.conv2d_wgrad(*args, pad=1))  # deterministic: no atomics, splits summed in order


@pytest.mark.parametrize("n,cin,cout,h,w,mask", [(2, 64, 64, 32, 32, "bits"), (1, 128, 128, 17, 15, "z"), (3, 256, 256, 16, 16, "bits"), (2, 32, 32, 24, 40, "z"),
                                                   (2, 64, 128, 16, 24, "none")])
def test_wino33_data_gradient_with_relu_mask_and_bn_partials_vs_the_generic_kernel_and_autograd(n, cin, cout, h, w, mask):
    """The fp32 data gradient of a stride-1 3x3 convolution in the Winograd F(2x2, 3x3) form (rs_conv2d_dgrad_wino33, round 6) with the
    epilogue of the generic kernel's EPI_BWD: ReLU mask (tensor or bits) and BatchNorm's two backward partial sums -- against
    rs_conv2d_dgrad_bnstats[_bits]_dt on the same inputs (masked gradient; column sums of the partial rows) and against autograd of
    relu(bn(y)) -> conv3x3 (torchvision Bottleneck.conv2 behind bn1 + ReLU, tools/train.py:186).  `cin` = the forward layer's input
    channels = the gradient's output channels."""
    from robosat_amd import ops

    # forward: y1 [n,cin,h,w] -> bn (batch statistics) -> relu -> z1 -> conv3x3 (cin -> cout) -> y2 ; gradient gy2 given
    y1 = rnd(n, cin, h, w, seed=71)
    wt = rnd(cout, cin, 3, 3, seed=72) * (2.0 / (cin * 9)) ** 0.5
    gy2 = rnd(n, cout, h, w, seed=73)
    mean = y1.mean((0, 2, 3))
    invstd = 1 / torch.sqrt(y1.var((0, 2, 3), unbiased=False) + 1e-5)
    z1 = torch.relu((y1 - mean[None, :, None, None]) * invstd[None, :, None, None]).requires_grad_(True)
    F.conv2d(z1, wt, padding=1).backward(gy2)
    g_ref = z1.grad * (z1.detach() > 0) if mask != "none" else z1.grad  # the masked gradient at the BatchNorm+ReLU output
    xhat = (y1 - mean[None, :, None, None]) * invstd[None, :, None, None]
    dy, yd, zd = nhwc(gy2), nhwc(y1), nhwc(z1.detach())
    md, isd = mean.to(DEV), invstd.to(DEV)
    wd = ops.pack_dgrad_weight(krsc(wt))
    u = ops.pack_wino33_weight(wd)
    assert ops.wino33_dgrad_ok(dy, cin)
    bits = None
    if mask == "bits":
        _, bits = ops.bn_apply(yd, isd, -md * isd, relu=True, want_bits=True)
    kw = dict(relu_mask=zd if mask == "z" else None, relu_mask_bits=bits)
    g, part = ops.conv2d_wino33_dgrad(dy, u, bn=(yd, md, isd), **kw)
    gg, partg = ops.conv2d_dgrad_bnstats(dy, wd, (h, w), yd, md, isd, pad=1, **kw)
    close(nchw(g), g_ref, 2e-5, "masked gradient")
    assert float((g - gg).abs().max()) <= 2e-5 * float(gg.abs().max())
    assert torch.equal(g == 0, gg == 0) or mask == "none"  # the same elements masked
    s, sg = part.double().sum(0).cpu(), partg.double().sum(0).cpu()
    close(s[0], g_ref.double().sum((0, 2, 3)), 1e-4, "sum g")
    close(s[1], (g_ref.double() * xhat.double()).sum((0, 2, 3)), 1e-4, "sum g * xhat")
    assert float((s - sg).abs().max()) <= 1e-5 * float(sg.abs().max())
    g2, none = ops.conv2d_wino33_dgrad(dy, u, **kw)  # without the BatchNorm half (ConvRelu: dec5's gradient)
    assert none is None and torch.equal(g2, g)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_stride2_1x1_data_gradient_as_a_low_resolution_product_plus_scatter_vs_the_zero_insertion_form_and_autograd(dtype):
    """Bottleneck.downsample[0] (1x1 / stride 2) under loss.backward(): its data gradient as rs_conv2d_fwd of the transposed filters on
    the low-resolution grid + rs_scatter_add_stride2_dt onto the gradient the tensor already has (round 6) -- against the zero-insertion
    launch (ups = 2, residual) it replaces (fp32: bit for bit) and against autograd."""
    from robosat_amd import ops

    act = torch.bfloat16 if dtype == "bf16" else torch.float32
    n, cin, cout, h, w = 2, 64, 128, 16, 24
    x = rnd(n, cin, h, w, seed=81).requires_grad_(True)
    wt = rnd(cout, cin, 1, 1, seed=82) * 0.1
    if dtype == "bf16":
        wt = wt.to(act).float()
    gy = rnd(n, cout, h // 2, w // 2, seed=83)
    if dtype == "bf16":
        gy = gy.to(act).float()
    extra = rnd(n, cin, h, w, seed=84)
    if dtype == "bf16":
        extra = extra.to(act).float()
    F.conv2d(x, wt, stride=2).backward(gy)
    want = x.grad + extra
    wd = ops.pack_dgrad_weight(krsc(wt), act)
    dyd, ex = nhwc(gy).to(act), nhwc(extra).to(act)
    old = ops.conv2d(dyd, wd, ups=2, pad=0, out_hw=(h, w), residual=ex)
    new = ops.scatter_add_stride2(ops.conv2d(dyd, wd), ex.clone())
    tol = 2e-2 if dtype == "bf16" else 2e-5
    close(nchw(new.float()), want, tol, "scatter form vs autograd")
    if dtype == "fp32":
        assert torch.equal(new, old)
    else:
        assert float((new.float() - old.float()).abs().max()) <= 2e-2 * float(old.float().abs().max())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 32, 32), (1, 128, 128, 16, 24), (3, 64, 32, 8, 12)])
def test_stride2_3x3_data_gradient_in_phase_form_vs_the_zero_insertion_form_and_autograd(dtype, n, cin, cout, h, w):
    """Bottleneck.conv2 of layer2..layer4's first block (3x3 / stride 2 / pad 1) under loss.backward(): its data gradient as a phase-form
    convolution over dy with rs_pack_s2_dgrad_phase_weight_dt's pack (round 6), in fp32 also in that form's Winograd kernel where the
    layer qualifies -- against the zero-insertion launch (ups = 2) it replaces and against autograd; with the ReLU mask in the
    epilogue as the bf16 step uses it."""
    from robosat_amd import ops

    act = torch.bfloat16 if dtype == "bf16" else torch.float32
    q = (lambda t: t.to(act).float()) if dtype == "bf16" else (lambda t: t)
    x = rnd(n, cin, h, w, seed=91).requires_grad_(True)
    wt = q(rnd(cout, cin, 3, 3, seed=92) * (2.0 / (cin * 9)) ** 0.5)
    gy = q(rnd(n, cout, h // 2, w // 2, seed=93))
    z = q(rnd(n, cin, h, w, seed=94))
    F.conv2d(x, wt, stride=2, padding=1).backward(gy)
    dyd, zd = nhwc(gy).to(act), nhwc(z).to(act)
    pack = ops.pack_s2_dgrad_phase_weight(krsc(wt), act)
    got = ops.conv2d_phase(dyd, pack)
    old = ops.conv2d(dyd, ops.pack_dgrad_weight(krsc(wt), act), ups=2, pad=1, out_hw=(h, w))
    tol = 2e-2 if dtype == "bf16" else 2e-5
    close(nchw(got.float()), x.grad, tol, "phase form vs autograd")
    assert float((got.float() - old.float()).abs().max()) <= tol * float(old.float().abs().max())
    masked = ops.conv2d_phase(dyd, pack, relu_mask=zd)
    assert torch.equal(masked, torch.where(zd > 0, got, torch.zeros_like(got)))
    if dtype == "fp32" and ops.wino_ok(dyd, None, cin, force=True):
        wino = ops.conv2d_phase_wino(dyd, ops.pack_wino_phase_weight(pack))
        close(nchw(wino), x.grad, 2e-5, "Winograd phase form vs autograd")


def test_stem_wgrad():
    from robosat_amd import ops

    for cin in (3, 4):
        x = rnd(2, cin, 64, 96, seed=8)
        wt = (rnd(64, cin, 7, 7, seed=9) * 0.1).requires_grad_(True)
        y = F.conv2d(x, wt, stride=2, padding=3)
        gy = rnd(*y.shape, seed=10)
        y.backward(gy)
        x4 = ops.nchw_to_nhwc4(x.to(DEV))
        dwp = ops.conv2d_wgrad(nhwc(gy), x4, 7, 7, stride=2, pad=3, stem=7)
        dw = ops.unpack_stem_weight(dwp, 7, cin)
        close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, what="stem wgrad")


@pytest.mark.parametrize("c,hw,res,relu", [(64, 24, False, True), (256, 10, True, True), (320, 6, False, False), (2048, 2, True, True)])
def test_batchnorm_train(c, hw, res, relu):
    from robosat_amd import ops

    n = 3
    y = (rnd(n, c, hw, hw, seed=11) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.rand(c) + 0.5).requires_grad_(True)
    beta = (rnd(c, seed=12) * 0.1).requires_grad_(True)
    rm, rv = rnd(c, seed=13) * 0.1, torch.rand(c) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    resid = rnd(n, c, hw, hw, seed=14).requires_grad_(True) if res else None
    out = F.batch_norm(y, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if res:
        out = out + resid
    z = F.relu(out) if relu else out
    gz = rnd(*z.shape, seed=15)
    z.backward(gz)

    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    drm, drv = rm0.to(DEV), rv0.to(DEV)
    yd = nhwc(y.detach())
    mean, invstd, scale, shift = ops.bn_train_stats(yd, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5, 0.1, drm, drv, nbt)
    zd = ops.bn_apply(yd, scale, shift, residual=nhwc(resid.detach()) if res else None, relu=relu)
    close(nchw(zd), z.detach(), what="bn fwd")
    close(drm.cpu(), rm, 1e-5, "running_mean")
    close(drv.cpu(), rv, 1e-5, "running_var")
    assert int(nbt.item()) == 1
    dy, dg, db, gm = ops.bn_bwd(nhwc(gz), zd if relu else None, yd, mean, invstd, gamma.detach().to(DEV), want_masked=True)
    close(nchw(dy), y.grad, 1e-3, "bn dy")
    close(dg.cpu(), gamma.grad, 1e-3, "dgamma")
    close(db.cpu(), beta.grad, 1e-3, "dbeta")
    if res:
        close(nchw(gm), resid.grad, what="residual grad")


@pytest.mark.parametrize("rows,c", [(8, 64), (300, 64), (1000, 256), (4097, 128), (700, 4096)],
                         ids=["one_slice", "few_slices", "two_level", "64_slices", "no_counters"])
def test_batchnorm_from_tile_partials(rows, c):
    """The one-launch reduce + finalize (last-arriving block finalizes; C > 2048 falls back to two launches) against a
    float64 evaluation of the same per-tile partial sums: forward statistics (torch.nn.BatchNorm2d train mode) and the
    backward coefficients, twice in a row so that a counter left dirty by the first launch would show."""
    from robosat_amd import ops

    m = rows * 128
    g = torch.Generator().manual_seed(rows + c)
    partial = torch.stack([torch.randn(rows, c, generator=g) * 5 + 3, torch.rand(rows, c, generator=g) * 300 + 200], 1)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    p64 = partial.double().sum(0)
    mu = p64[0] / m
    var = (p64[1] / m - mu * mu).clamp_min(0)
    inv = 1 / torch.sqrt(var + 1e-5)
    for _ in range(2):
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.int64, device=DEV)
        mean, invstd, scale, shift = ops.bn_finalize_stats(partial.to(DEV), m, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rm, rv, nbt)
        close(mean.cpu().double(), mu, 1e-6, "mean")
        close(invstd.cpu().double(), inv, 1e-6, "invstd")
        close(scale.cpu().double(), gamma.double() * inv, 1e-6, "scale")
        close(shift.cpu().double(), beta.double() - mu * gamma.double() * inv, 1e-5, "shift")
        close(rm.cpu().double(), 0.1 * mu, 1e-6, "running_mean")
        close(rv.cpu().double(), 0.9 + 0.1 * var * m / (m - 1), 1e-6, "running_var")
        assert int(nbt) == 1
    # backward: dy = k1*g - k2 - k3*(y - mean) with k1 = gamma*invstd, k2 = k1*sum(g)/M, k3 = k1*invstd*sum(g*xhat)/M
    hw = 16
    gd, yd = torch.randn(1, hw, hw, c, generator=g).to(DEV), torch.randn(1, hw, hw, c, generator=g).to(DEV)
    mean_d, inv_d = torch.randn(c, generator=g).to(DEV), (torch.rand(c, generator=g) + 0.5).to(DEV)
    for _ in range(2):
        dy, dgamma, dbeta = ops.bn_bwd_from_partials(gd, yd, mean_d, inv_d, gamma.to(DEV), partial.to(DEV))
        close(dbeta.cpu().double(), p64[0], 1e-6, "dbeta")
        close(dgamma.cpu().double(), p64[1], 1e-6, "dgamma")
        # (the coefficients divide by M = pixels of g, here hw*hw: the partial sums are synthetic, the formula is what is checked)
        mm = hw * hw
        k1 = gamma.double() * inv_d.cpu().double()
        want = k1 * gd.cpu().double() - k1 * p64[0] / mm - k1 * inv_d.cpu().double() * p64[1] / mm * (yd.cpu().double() - mean_d.cpu().double())
        close(dy.cpu().double(), want, 1e-5, "dy")


@pytest.mark.parametrize("k,s,p", [(3, 2, 1), (2, 2, 0)])
def test_maxpool_backward(k, s, p):
    from robosat_amd import ops

    x = F.relu(rnd(2, 64, 18, 22, seed=16)).requires_grad_(True)  # many exact-zero ties, as after a ReLU
    y = F.max_pool2d(x, k, s, p)
    gy = rnd(*y.shape, seed=17)
    y.backward(gy)
    _, amax = ops.maxpool2d(nhwc(x.detach()), k, s, p, want_argmax=True)
    dx = ops.maxpool2d_bwd(nhwc(gy), amax, (2, 18, 22, 64), k, s, p)
    close(nchw(dx), x.grad, 1e-6)
    dx2 = ops.maxpool2d_bwd(nhwc(gy), amax, (2, 18, 22, 64), k, s, p, out=torch.ones_like(dx))
    close(nchw(dx2), x.grad + 1.0, 1e-6)


@pytest.mark.parametrize("ncls", [2, 4])
def test_final_conv_backward(ncls):
    from robosat_amd import ops

    x = F.relu(rnd(2, 32, 24, 40, seed=18)).requires_grad_(True)
    w = (rnd(ncls, 32, 1, 1, seed=19) * 0.3).requires_grad_(True)
    b = rnd(ncls, seed=20).requires_grad_(True)
    pre = rnd(2, 32, 24, 40, seed=18).requires_grad_(True)  # the same values before the ReLU
    y = F.conv2d(F.relu(pre), w, b)
    gy = rnd(*y.shape, seed=21)
    y.backward(gy)
    dx, dw, db = ops.final_conv1x1_bwd(nhwc(x.detach()), w.detach().view(ncls, 32).to(DEV), gy.to(DEV), relu_mask=True)
    close(nchw(dx), pre.grad, what="dx through relu")
    close(dw.cpu(), w.grad.view(ncls, 32), what="dw")
    close(db.cpu(), b.grad, what="db")


@pytest.mark.parametrize("tag", ["c2", "c4"])
@pytest.mark.parametrize("name", ["CrossEntropy", "Focal", "mIoU", "Lovasz"])
def test_losses_match_reference_golden(golden_dir, tag, name):
    from robosat_amd import losses

    g = np.load(os.path.join(golden_dir, "losses.npz"))
    logits = torch.from_numpy(g[tag + "_logits"]).to(DEV).requires_grad_(True)
    targets = torch.from_numpy(g[tag + "_targets"]).to(DEV)
    weight = torch.from_numpy(g[tag + "_weight"])
    crit = {"CrossEntropy": lambda: losses.CrossEntropyLoss2d(weight=weight), "Focal": lambda: losses.FocalLoss2d(weight=weight),
            "mIoU": lambda: losses.mIoULoss2d(weight=weight), "Lovasz": lambda: losses.LovaszLoss2d()}[name]().to(DEV)
    loss = crit(logits, targets)
    (loss * 1.5).backward()  # a non-unit upstream gradient
    want = float(g["{}_{}_loss".format(tag, name)])
    assert abs(loss.item() - want) <= 2e-5 * max(1.0, abs(want)), (loss.item(), want)
    close(logits.grad.cpu() / 1.5, torch.from_numpy(g["{}_{}_grad".format(tag, name)]), 2e-4, name + " grad")


@pytest.mark.parametrize("noise,margin,branch", [(0.05, 0.1, "nll"), (0.5, 3.0, "miou"), (1.0, 2.0, "miou")])
def test_miou_both_branches_vs_oracle(noise, margin, branch):
    """Near-uniform predictions -> the NLL branch (~log 3 > miou); confident, mostly right predictions -> the soft-IoU
    branch.  Either way value and gradient must follow the branch the reference's max() picks."""
    from robosat_amd import losses
    from oracle import seeded
    import torch.nn.functional as F

    n, c, h, w = 2, 3, 32, 64
    targets = seeded.synthetic_targets(n, c, h, w, 5)
    logits = rnd(n, c, h, w, seed=23) * noise + margin * R.onehot(targets, c)
    weight = torch.ones(3)
    nll = float(F.nll_loss(F.log_softmax(logits, 1), targets, weight=weight))
    assert (abs(float(R.miou2d(logits, targets, weight=weight)) - nll) < 1e-9) == (branch == "nll")  # the case is what it claims
    ref_in = logits.clone().requires_grad_(True)
    want = R.miou2d(ref_in, targets, weight=weight)
    want.backward()
    got_in = logits.to(DEV).requires_grad_(True)
    got = losses.mIoULoss2d(weight=weight).to(DEV)(got_in, targets.to(DEV))
    got.backward()
    assert abs(got.item() - want.item()) <= 2e-5 * max(1.0, abs(want.item())), (got.item(), want.item())
    close(got_in.grad.cpu(), ref_in.grad, 1e-3, "miou grad")


@pytest.mark.parametrize("n,c,h,w", [(3, 2, 128, 128), (2, 4, 64, 96), (1, 3, 48, 80),
                                     (2, 3, 7, 9),        # H * W = 63: the scalar (unaligned) forms of the key / scan kernels
                                     (2, 2, 512, 512),    # 524 288 keys per image: what BASELINE configs[2] sorts (x 32 images)
                                     (1, 4, 512, 512),    # 1 048 576 keys per image: configs[4]
                                     (3, 3, 256, 320),    # 245 760 keys: not a multiple of the 8 192-key digit tile x blocks
                                     (16, 2, 512, 512)])  # 8.4 M keys in all: the 4 096-key digit tiles of the large sorts (round 5)
def test_lovasz_vs_oracle(n, c, h, w):
    """Multi-block radix sort / segmented scan against the CPU oracle (reference losses.py:96-119), from one-block images up
    to the key counts the benchmark runs -- the 8 192-element LDS digit tiles, the multi-block scan carry and the segment
    offsets across > 64 blocks per image are only reached at the full sizes -- with the same bars at every size."""
    from robosat_amd import losses
    from oracle import seeded

    logits = rnd(n, c, h, w, seed=22) * 3
    targets = seeded.synthetic_targets(n, c, h if h % 8 == 0 else h, w, 3)
    ref_in = logits.clone().requires_grad_(True)
    want = R.lovasz2d(ref_in, targets)
    want.backward()
    got_in = logits.to(DEV).requires_grad_(True)
    got = losses.LovaszLoss2d()(got_in, targets.to(DEV))
    got.backward()
    assert abs(got.item() - want.item()) <= 2e-5 * max(1.0, abs(want.item())), (got.item(), want.item())
    # Equal errors may be sorted in any order (torch's CPU sort is not stable at the large sizes): gradients are compared per
    # group of equal errors -- elementwise wherever an error is unique -- see tests/lovasz_check.py.  The Jaccard deltas are
    # differences of nearly equal fp32 numbers: 1 ulp of jac (3e-8) is ~4e-4 of a delta at 32k keys.
    from lovasz_check import assert_lovasz_grad_close

    rel = assert_lovasz_grad_close(logits, targets, got_in.grad.cpu(), ref_in.grad, 2e-3)
    print("lovasz {}: loss {} oracle {}, gradient max err over tie groups {:.2e} of the largest entry".format(
        (n, c, h, w), got.item(), want.item(), rel))


def test_metrics_match_reference_golden(golden_dir):
    from robosat_amd.metrics import Metrics

    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    m = Metrics(range(2))
    scores, actual = torch.from_numpy(g["scores"]).to(DEV), torch.from_numpy(g["actual"]).to(DEV)
    m.add(actual[0], scores[0])  # reference-style single sample
    m.add_batch(actual[1:], scores[1:])
    assert [m.tn, m.fn, m.fp, m.tp] == g["counts"].tolist()
    assert np.allclose([m.get_miou(), m.get_fg_iou(), m.get_mcc()], g["scores3"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c1,c2,cout,h,w", [(64, 32, 64, 10, 14), (128, 0, 32, 16, 8), (256, 64, 128, 8, 8), (64, 64, 128, 5, 3),
                                            (256, 64, 32, 128, 128)])  # last: 320 couts -> ragged third 128-wide N tile
def test_decoder_dgrad_phase_form(c1, c2, cout, h, w, dtype):
    """d loss / d (skip, prev) of relu-less DecoderBlock conv3x3(interpolate(cat[skip, prev], x2)) == one 4x4 / stride-2
    convolution over dz with rs_pack_dgrad_phase_weight_dt weights, then the cat split (vs autograd)."""
    from robosat_amd import ops

    n = 2
    bf = dtype == torch.bfloat16
    rq = (lambda t: t.to(dtype).float()) if bf else (lambda t: t)
    a = rnd(n, c1, h, w, seed=1).requires_grad_(True)
    b = rnd(n, c2, h, w, seed=2).requires_grad_(True) if c2 else None
    wt = rnd(cout, c1 + c2, 3, 3, seed=3) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    dz = rq(rnd(*y.shape, seed=4))
    y.backward(dz)
    wd = ops.pack_dgrad_phase_weight(krsc(wt), dtype)
    assert tuple(wd.shape) == (c1 + c2, 4, 4, cout)
    dsrc = ops.conv2d(nhwc(dz).to(dtype), wd, stride=2, pad=1, out_hw=(h, w))
    m1 = rq(rnd(n, c1, h, w, seed=5))
    base = rq(rnd(n, c1, h, w, seed=6))
    acc = nhwc(base).to(dtype)
    d1, d2 = ops.cat_split_bwd(dsrc, c1, c2, mask1=nhwc(m1).to(dtype), out1=acc)
    tol = 3e-2 if bf else 3e-4
    close(nchw(d1.float()), base + a.grad * (m1 > 0), tol, "d skip")
    if c2:
        close(nchw(d2.float()), b.grad, tol, "d prev")
        if c1 % 64 == 0 and cout % 32 == 0:  # the same split fused into the convolution's store
            m2 = rq(rnd(n, c2, h, w, seed=7))
            try:
                e1, e2 = ops.conv2d_split(nhwc(dz).to(dtype), wd, c1, stride=2, pad=1, out_hw=(h, w), mask1=nhwc(m1).to(dtype),
                                          mask2=nhwc(m2).to(dtype))
            except ValueError:
                return  # csplit not a multiple of the tile's cout width for this shape
            close(nchw(e1.float()), a.grad * (m1 > 0), tol, "fused d skip")
            close(nchw(e2.float()), b.grad * (m2 > 0), tol, "fused d prev")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 64, 24, 20, 256, 1, 0), (3, 128, 16, 16, 128, 3, 1), (1, 512, 9, 7, 2048, 1, 0)],
                         ids=["conv1_dgrad_128x128", "conv2_dgrad", "ragged_rows"])
def test_relu_mask_as_bits(case, dtype):
    """The ReLU mask of a BatchNorm output as one bit per element: `bn_apply(want_bits=True)` writes exactly (z > 0), and the
    data-gradient epilogue reading the bits (rs_conv2d_dgrad_bnstats_bits_dt) returns, bit for bit, what it returns reading z
    (rs_conv2d_dgrad_bnstats_dt) -- output and both BatchNorm partial sums."""
    from robosat_amd import ops

    n, cin, h, w, cout, k, pad = case
    y = rnd(n, h, w, cout, seed=31).to(DEV).to(dtype)
    res = rnd(n, h, w, cout, seed=32).to(DEV).to(dtype)
    scale, shift = (torch.rand(cout) + 0.5).to(DEV), (rnd(cout, seed=33) * 0.3).to(DEV)
    z, bits = ops.bn_apply(y, scale, shift, residual=res, relu=True, want_bits=True)
    z_plain = ops.bn_apply(y, scale, shift, residual=res, relu=True)
    assert torch.equal(z, z_plain)
    want = (z.float() > 0).reshape(-1, 8).to(torch.uint8)
    want = (want << torch.arange(8, device=DEV, dtype=torch.uint8)).sum(1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert 0.2 < float((z > 0).float().mean()) < 0.8

    dy = (rnd(n, h, w, cin, seed=34) * 0.1).to(DEV).to(dtype)
    wd = (rnd(cout, k, k, cin, seed=35) * 0.05).to(DEV).to(dtype)  # (already in data-gradient layout: any weights do)
    bn_y = rnd(n, h, w, cout, seed=36).to(DEV).to(dtype)
    mean, inv = rnd(cout, seed=37).to(DEV), (torch.rand(cout) + 0.5).to(DEV)
    rgrad = rnd(n, h, w, cout, seed=38).to(DEV).to(dtype)
    g0, p0 = ops.conv2d_dgrad_bnstats(dy, wd, (h, w), bn_y, mean, inv, pad=pad, residual=rgrad, relu_mask=z)
    g1, p1 = ops.conv2d_dgrad_bnstats(dy, wd, (h, w), bn_y, mean, inv, pad=pad, residual=rgrad, relu_mask_bits=bits)
    assert torch.equal(g0, g1) and torch.equal(p0, p1)
    assert float((g1 == 0).float().mean()) > 0.2  # the mask did something
    with pytest.raises(Exception):
        ops.bn_apply(y[..., :24].contiguous(), scale[:24], shift[:24], relu=True, want_bits=True)  # 24 does not divide 2048


@pytest.mark.parametrize("n,c1,c2,cout,hs,ws,masks", [
    (2, 128, 0, 32, 16, 16, True),      # dec4's class: one source, one mask, two chunks per parity plane
    (1, 256, 64, 128, 16, 32, True),    # dec3's class: two destinations, a mask each, two patches side by side
    (2, 64, 64, 64, 24, 20, False),     # ragged patches (12 x 10 tiles), no masks
    (1, 128, 64, 64, 17, 15, True),     # odd sizes: the last tile row / column hangs over the image
    (3, 64, 0, 32, 31, 33, False),      # blocks that straddle images
    (2, 192, 128, 64, 16, 16, True),    # csplit = 192: a multiple of 64, not of 128
])
def test_wino_dgrad_fp32_vs_autograd_and_the_4x4_kernel(n, c1, c2, cout, hs, ws, masks):
    """The fp32 DecoderBlock's data gradient in the Winograd form (rs_conv2d_dgrad_phase_wino, round 6: four parity planes of dz
    accumulated into one source-resolution tile, 9/16 of the 4x4 / stride-2 kernel's multiply-adds) against autograd of the reference
    formulation (unet.py:63-73; torch.cat's backward and the ReLU masks of the layers below fused into the store) and against the
    generic kernel on the same launch -- exact-fp32 MFMA both, only the summation order differs.  Also: the rule is geometry only and
    it is what the fp32 training step runs."""
    from robosat_amd import ops

    a = rnd(n, c1, hs, ws, seed=1).requires_grad_(True)
    b = rnd(n, c2, hs, ws, seed=2).requires_grad_(True) if c2 else None
    wt = rnd(cout, c1 + c2, 3, 3, seed=3) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    y = F.conv2d(F.interpolate(torch.cat([a, b], 1) if c2 else a, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = rnd(*y.shape, seed=4)
    y.backward(gy)
    m1 = rnd(n, c1, hs, ws, seed=5) if masks else None
    m2 = rnd(n, c2, hs, ws, seed=6) if (masks and c2) else None
    want1 = a.grad * (m1 > 0) if masks else a.grad
    want2 = (b.grad * (m2 > 0) if masks else b.grad) if c2 else None
    dev = lambda t: None if t is None else nhwc(t)
    wd = ops.pack_dgrad_phase_weight(krsc(wt))
    u = ops.pack_wino_dgrad_weight(wd)
    dz = nhwc(gy)
    assert ops.wino_dgrad_ok(n, hs, ws, c1, c2, cout) and ops.wino_dgrad_ok(7 * n, hs, ws, c1, c2, cout)  # (never the batch size)
    assert not ops.wino_dgrad_ok(n, 8, 8, c1, c2, cout)  # (fewer than 8 tiles per image side: the 4x4 kernel keeps it -- `center`)
    if c2:
        g1, g2 = ops.conv2d_dgrad_phase_wino(dz, u, c1, c2, mask1=dev(m1), mask2=dev(m2), split=True)
        r1, r2 = ops.conv2d_split(dz, wd, c1, stride=2, pad=1, out_hw=(hs, ws), mask1=dev(m1), mask2=dev(m2))
    else:
        g1, g2 = ops.conv2d_dgrad_phase_wino(dz, u, c1, 0, mask1=dev(m1))
        r1, r2 = ops.conv2d(dz, wd, stride=2, pad=1, out_hw=(hs, ws), relu_mask=dev(m1)), None
    for got, ref, want, what in ((g1, r1, want1, "d skip"), (g2, r2, want2, "d prev")):
        if got is None:
            continue
        close(nchw(got), want, 2e-5, what + " vs autograd")
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), what + " vs the 4x4 / stride-2 kernel"


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 32, 32), (1, 128, 128, 17, 15), (3, 256, 256, 16, 16), (2, 64, 128, 24, 40), (5, 32, 32, 16, 16)])
def test_wino33_train_forward_with_statistics_vs_torch_and_the_generic_kernel(n, cin, cout, h, w):
    """The stride-1 3x3 convolutions of the fp32 TRAIN forward in the Winograd F(2x2, 3x3) form (rs_conv2d_fwd_wino33_stats, round 6):
    raw output and the per-block partial sums of BatchNorm's statistics -- against F.conv2d, against the generic kernel's output and
    against the statistics finalized from either kernel's partial rows (torchvision Bottleneck.conv2 -> bn2, tools/train.py:169)."""
    from robosat_amd import ops

    x = rnd(n, cin, h, w, seed=11)
    wt = rnd(cout, cin, 3, 3, seed=12) * (2.0 / (cin * 9)) ** 0.5
    want = F.conv2d(x, wt, padding=1)
    xd, wk = nhwc(x), krsc(wt)
    assert ops.wino33_ok(xd, cout)
    y, part = ops.conv2d_wino33_bnstats(xd, ops.pack_wino33_weight(wk))
    yg, partg = ops.conv2d_bnstats(xd, wk, pad=1)
    close(nchw(y), want, 2e-5, "raw output")
    assert float((y - yg).abs().max()) <= 2e-5 * float(yg.abs().max())
    m = n * h * w
    s = part.double().sum(0).cpu()
    close(s[0] / m, want.double().mean((0, 2, 3)), 1e-5, "mean from the partial rows")
    close(s[1] / m, (want.double() ** 2).mean((0, 2, 3)), 1e-5, "mean of squares from the partial rows")
    sg = partg.double().sum(0).cpu()
    assert float((s - sg).abs().max()) <= 1e-5 * float(sg.abs().max())
    g, b = torch.rand(cout) + 0.5, rnd(cout, seed=13)
    st = ops.bn_finalize_stats(part, m, g.to(DEV), b.to(DEV), 1e-5, 0.1)
    close(st[0].cpu(), want.mean((0, 2, 3)), 1e-5, "mean")
    close(st[1].cpu(), 1 / torch.sqrt(want.var((0, 2, 3), unbiased=False) + 1e-5), 1e-4, "invstd")
