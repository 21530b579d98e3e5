"""bf16 path (BASELINE configs[2]: rs train bf16) on the MI355X, through the C ABI.

Per-kernel parity: operands are rounded to bf16 ON THE HOST and the reference is plain PyTorch fp32 on those rounded
values, so what is left is (a) accumulation order (fp32 accumulators: ~1e-6) and (b) the single bf16 rounding of a
bf16 OUTPUT (half an ulp = 2^-9 relative).  Tolerances: 1e-2 of the output scale for bf16 outputs (2^-8 = 3.9e-3 per
element plus headroom for cancellation), 2e-3 for fp32 outputs (weight gradients, statistics).

Whole network: bf16 is not expected to meet the fp32 bar (1e-3 on probabilities); the tolerances below are what 60
stacked bf16 layers measurably deliver and are asserted so that a broken kernel (not rounding) trips them."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import robosat_ref as R, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def q(t):  # round to bf16, keep fp32 storage (host reference operand)
    return t.to(BF).float()


def nhwc(t, dtype=BF):  # NCHW cpu fp32 -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


def krsc(w, dtype=BF):
    return w.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def close(got, want, tol, what=""):
    scale = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, "{} max abs err {} (scale {})".format(what, err, scale)


TOL_BF, TOL_F32 = 1e-2, 2e-3

CONV = [
    # name,             N, Cin, H,  W,  Cout, k, stride, pad   (tile / K-chunk the C ABI picks)
    ("3x3_c32_tail", 2, 32, 17, 19, 32, 3, 1, 1),     # 128x32, KC 32, M tail
    ("3x3_c64_small", 2, 64, 16, 16, 64, 3, 1, 1),    # 64x64, KC 64
    ("3x3_c64_s2", 2, 64, 18, 22, 64, 3, 2, 1),       # stride 2
    ("3x3_c32_c64_big", 2, 32, 192, 192, 64, 3, 1, 1),  # 128x64, KC 32
    ("3x3_c64_c128_big", 1, 64, 256, 256, 128, 3, 1, 1),  # 128x128, KC 64
    ("3x3_c32_c128_big", 1, 32, 256, 256, 128, 3, 1, 1),  # 128x128, KC 32
    ("1x1_c256", 2, 64, 20, 12, 256, 1, 1, 0),
    ("1x1_s2", 2, 128, 20, 12, 256, 1, 2, 0),
    ("3x3_c96in", 1, 96, 9, 33, 160, 3, 1, 1),        # Cin 96 -> KC 32; Cout 160 -> 128x32 tiles
    ("1x1_c2048", 1, 2048, 16, 16, 512, 1, 1, 0),     # long K
]


@pytest.mark.parametrize("case", CONV, ids=[c[0] for c in CONV])
def test_conv_bf16(case):
    from robosat_amd import ops

    _, n, cin, h, w, cout, k, stride, pad = case
    x, wt = q(rnd(n, cin, h, w, seed=1)), q(rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5)
    want = F.conv2d(x, wt, stride=stride, padding=pad)
    got = ops.conv2d(nhwc(x), krsc(wt), stride=stride, pad=pad)
    assert got.dtype == BF
    close(nchw(got), want, TOL_BF)


def test_conv_bf16_epilogue_and_mask():
    from robosat_amd import ops

    n, cin, h, w, cout = 2, 64, 24, 24, 128
    x, wt = q(rnd(n, cin, h, w, seed=3)), q(rnd(cout, cin, 3, 3, seed=4) * 0.06)
    sc, sh, res = rnd(cout, seed=5), rnd(cout, seed=6), q(rnd(n, cout, h, w, seed=7))
    mask = q(rnd(n, cout, h, w, seed=8))
    base = F.conv2d(x, wt, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res
    got = ops.conv2d(nhwc(x), krsc(wt), pad=1, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res), relu=True)
    close(nchw(got), F.relu(base), TOL_BF, "relu")
    got = ops.conv2d(nhwc(x), krsc(wt), pad=1, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res), relu_mask=nhwc(mask))
    close(nchw(got), base * (mask > 0), TOL_BF, "mask")


@pytest.mark.parametrize("c1,c2,cout,h,w", [(64, 32, 64, 10, 14), (32, 0, 32, 16, 16), (256, 64, 128, 8, 8), (128, 64, 64, 32, 32)])
def test_conv_bf16_upsample_concat(c1, c2, cout, h, w):
    """DecoderBlock: conv3x3(interpolate(cat[skip, prev], x2 nearest)) + ReLU (reference unet.py:73,134-137)."""
    from robosat_amd import ops

    n = 2
    a = q(rnd(n, c1, h, w, seed=8))
    b = q(rnd(n, c2, h, w, seed=9)) if c2 else None
    wt = q(rnd(cout, c1 + c2, 3, 3, seed=10) * (2.0 / ((c1 + c2) * 9)) ** 0.5)
    cat = torch.cat([a, b], 1) if c2 else a
    want = F.relu(F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1))
    got = ops.conv2d(nhwc(a), krsc(wt), src2=nhwc(b) if c2 else None, ups=1, pad=1, relu=True)
    close(nchw(got), want, TOL_BF)


@pytest.mark.parametrize("k,pad,hin,stride", [(3, 1, 16, 2), (1, 0, 16, 2), (3, 1, 15, 2), (3, 1, 20, 1)])
def test_dgrad_bf16(k, pad, hin, stride):
    """rs_pack_dgrad_weight_bf16 + the ups=2 / ups=0 gather == data gradient of the convolution (vs autograd)."""
    from robosat_amd import ops

    n, cin, cout = 2, 64, 96
    x = rnd(n, cin, hin, hin, seed=11).requires_grad_(True)
    wt = q(rnd(cout, cin, k, k, seed=12) * 0.05)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    gy = q(rnd(*y.shape, seed=13))
    y.backward(gy)
    wd = ops.pack_dgrad_weight(krsc(wt, torch.float32), BF)
    assert wd.dtype == BF and tuple(wd.shape) == (cin, k, k, cout)
    got = ops.conv2d(nhwc(gy), wd, ups=2 if stride == 2 else 0, pad=k - 1 - pad, out_hw=(hin, hin))
    close(nchw(got), x.grad, TOL_BF)


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
    ("tail", 1, 64, 13, 11, 128, 3, 1, 1),  # M = 143: not a multiple of the 64-pixel chunk
    ("256x128", 2, 128, 20, 20, 256, 3, 1, 1),
    ("256x128_1x1_tail", 1, 256, 15, 23, 512, 1, 1, 0),  # two cout tiles x two cin tiles, M = 345
    # Cout = 32 3x3 layers with Ho, Wo multiples of 8 take the all-taps-per-block kernel (conv_wgrad_thin_bf16.hip):
    # "32x128", "32x32", "split_big" above; these keep the generic 32-wide tiles covered and add Cin = 64
    ("32x128_generic", 1, 128, 36, 36, 32, 3, 1, 1),
    ("32x32_generic", 2, 32, 36, 44, 32, 3, 1, 1),
    ("thin_c64", 3, 64, 24, 40, 32, 3, 1, 1),
    ("thin_c128_edge", 2, 128, 8, 8, 32, 3, 1, 1),  # one patch per image: every halo side is padding
    # the all-taps kernel on wider layers (round 2): grid = patch runs x 32-cout tiles x 128-cin slabs
    ("thin_wide_c64", 4, 64, 128, 128, 64, 3, 1, 1),
    ("thin_wide_c128_co64", 4, 128, 128, 128, 64, 3, 1, 1),
    ("thin_wide_c256", 2, 256, 64, 64, 256, 3, 1, 1),
]


@pytest.mark.parametrize("case", WGRAD, ids=[c[0] for c in WGRAD])
def test_wgrad_bf16(case):
    from robosat_amd import ops

    _, n, cin, h, w, cout, k, stride, pad = case
    x = q(rnd(n, cin, h, w, seed=1))
    wt = (rnd(cout, cin, k, k, seed=2) * 0.05).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    gy = q(rnd(*y.shape, seed=3))
    y.backward(gy)
    dw = ops.conv2d_wgrad(nhwc(gy), nhwc(x), k, k, stride=stride, pad=pad)
    assert dw.dtype == torch.float32
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32)
    # the instantiation the case is named after is the one that ran (what bench.py reports per symbol)
    from robosat_amd import _lib
    expected = {"128x128": "conv_wgrad_bf16<128x128>", "128x64": "conv_wgrad_bf16<128x64>", "64x128": "conv_wgrad_bf16<64x128>",
                "64x64_s2": "conv_wgrad_bf16<64x64>", "256x128": "conv_wgrad_bf16<256x128>",
                "256x128_1x1_tail": "conv_wgrad_bf16<256x128>", "1x1_s2": "conv_wgrad_bf16<256x128>", "32x128_generic": "conv_wgrad_bf16<32x128>",
                "32x32_generic": "conv_wgrad_bf16<32x32>", "thin_c64": "conv_wgrad_thin_bf16<64>", "32x32": "conv_wgrad_thin_bf16<32>",
                "thin_c128_edge": "conv_wgrad_thin_bf16<128>", "thin_wide_c64": "conv_wgrad_thin_bf16<64>",
                "thin_wide_c128_co64": "conv_wgrad_thin_bf16<128>", "thin_wide_c256": "conv_wgrad_thin_bf16<128>"}
    if case[0] in expected:
        d = _lib.ConvDesc(n, h, w, cin, 0, 0, k, k, stride, pad, y.shape[2], y.shape[3], cout, 0, 0)
        assert ops.wgrad_kernel_name(d) == expected[case[0]]


@pytest.mark.parametrize("cin,h,w", [(128, 16, 12), (32, 4, 8), (64, 12, 12)])
def test_wgrad_bf16_thin_upsample(cin, h, w):
    """dec4-shaped: conv3x3(interpolate(x, x2 nearest)) with Cout = 32 -> thin kernel with the source-resolution halo."""
    from robosat_amd import ops

    n, cout = 2, 32
    a = q(rnd(n, cin, h, w, seed=4))
    wt = (rnd(cout, cin, 3, 3, seed=6) * 0.05).requires_grad_(True)
    y = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = q(rnd(*y.shape, seed=7))
    y.backward(gy)
    dw = ops.conv2d_wgrad(nhwc(gy), nhwc(a), 3, 3, ups=1, pad=1)
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32)


@pytest.mark.parametrize("n,c1,c2,cout,h,w", [(2, 128, 64, 64, 12, 10), (1, 256, 0, 128, 9, 7), (3, 64, 64, 256, 5, 16), (2, 256, 64, 128, 16, 16),
                                              (1, 512, 256, 64, 6, 8)])
def test_wgrad_bf16_upsample_concat(n, c1, c2, cout, h, w):
    """DecoderBlock filter gradient; takes the phase form (16 parity/offset reductions over SOURCE pixels + combine)."""
    from robosat_amd import ops

    a = q(rnd(n, c1, h, w, seed=4))
    b = q(rnd(n, c2, h, w, seed=5)) if c2 else None
    wt = (rnd(cout, c1 + c2, 3, 3, seed=6) * 0.05).requires_grad_(True)
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = q(rnd(*y.shape, seed=7))
    y.backward(gy)
    dw = ops.conv2d_wgrad(nhwc(gy), nhwc(a), 3, 3, src2=nhwc(b) if c2 else None, ups=1, pad=1)
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32)
    from robosat_amd import _lib
    d = _lib.ConvDesc(n, h, w, c1, c2, 1, 3, 3, 1, 1, 2 * h, 2 * w, cout, 0, 0)
    expected = {(128, 64, 64): "conv_wgrad_bf16<phase,64x128+64x64>", (256, 0, 128): "conv_wgrad_bf16<phase,128x128>",
                (64, 64, 256): "conv_wgrad_bf16<phase,128x64>", (256, 64, 128): "conv_wgrad_bf16<phase,128x128+128x64>",
                (512, 256, 64): "conv_wgrad_bf16<phase,64x128>"}
    # (the 128 x 128 launches of the phase form run conv_wgrad_phase4_bf16 by default: one dz plane x four source offsets per block)
    assert ops.wgrad_kernel_name(d) == expected[(c1, c2, cout)].replace("phase,128x128", "phase4,128x128")


@pytest.mark.parametrize("n,c1,c2,cout,h,w", [(1, 256, 0, 128, 9, 7), (3, 256, 128, 256, 9, 13), (2, 256, 64, 128, 16, 16), (4, 128, 0, 128, 32, 32)])
def test_wgrad_bf16_phase_form_four_offsets_per_block(n, c1, c2, cout, h, w):
    """conv_wgrad_phase4_bf16 (round 5: a block owns one dz parity plane and all four of its source offsets -- the dz tile is
    fetched once for four products; knob ``wgrad_phase4`` = 1, the default again since round 6) against autograd on the reference
    formulation (unet.py:63-73) and against the block-per-pair kernel on the same launch (knob = 0): ragged
    32-pixel chunks, splits that straddle images, two sources incl. a narrower second one, image borders on every side."""
    from robosat_amd import _lib, ops

    a = q(rnd(n, c1, h, w, seed=24))
    b = q(rnd(n, c2, h, w, seed=25)) if c2 else None
    wt = (rnd(cout, c1 + c2, 3, 3, seed=26) * 0.05).requires_grad_(True)
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = q(rnd(*y.shape, seed=27))
    y.backward(gy)
    args = (nhwc(gy), nhwc(a), 3, 3)
    kw = dict(src2=nhwc(b) if c2 else None, ups=1, pad=1)
    assert ops.get_knob("wgrad_phase4") == 1  # (round 5 withdrew it; the defect was its gather table's barrier: profiles/r06/dma_order.txt)
    new = ops.conv2d_wgrad(*args, **kw)
    d = _lib.ConvDesc(n, h, w, c1, c2, 1, 3, 3, 1, 1, 2 * h, 2 * w, cout, 0, 0)
    assert "phase4" in ops.wgrad_kernel_name(d) or cout % 128 or c1 % 128
    with ops.knob("wgrad_phase4", 0):
        old = ops.conv2d_wgrad(*args, **kw)
    close(new.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32)
    close(old.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32)
    assert float((new - old).abs().max()) <= 2e-5 * float(old.abs().max())  # (exact bf16 products, fp32 sums in another order)


@pytest.mark.parametrize("n,cin,cout,k,stride,h,w", [(3, 256, 512, 1, 1, 20, 24), (2, 128, 128, 3, 2, 18, 22), (5, 64, 256, 1, 1, 16, 16),
                                                     (2, 192, 64, 1, 1, 9, 7), (1, 64, 64, 1, 1, 33, 5), (2, 128, 64, 1, 1, 12, 12)])
def test_wgrad_bf16_ring_of_three_chunk_buffers(n, cin, cout, k, stride, h, w):
    """conv_wgrad_bf16<.., RING = 3> (round 5: two chunks in flight behind the one being multiplied, counted vmcnt waits; the default,
    knob ``wgrad_ring``) against autograd, and BIT-IDENTICAL to the two-buffer pipeline at the same split count (the same chunks
    are summed in the same order): every tile width, ragged last chunks, splits shorter than the ring.  (Beside an LDS-using
    neighbour: tests/test_gpu_race_screen.py.)"""
    from robosat_amd import ops

    x = q(rnd(n, cin, h, w, seed=31))
    wt = (rnd(cout, cin, k, k, seed=32) * 0.05).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=k // 2)
    gy = q(rnd(*y.shape, seed=33))
    y.backward(gy)
    assert ops.get_knob("wgrad_ring") == 3 and ops.get_knob("wgrad_blocks") == 96
    for blocks in (96, 4096):  # (the ring's target, and splits of a chunk or two: shorter than the ring)
        with ops.knob("wgrad_blocks", blocks):
            new = ops.conv2d_wgrad(nhwc(gy), nhwc(x), k, k, stride=stride, pad=k // 2)
            with ops.knob("wgrad_ring", 2):
                old = ops.conv2d_wgrad(nhwc(gy), nhwc(x), k, k, stride=stride, pad=k // 2)
        close(new.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32)
        assert torch.equal(new, old), blocks


def test_bn_train_bf16():
    from robosat_amd import ops

    n, c, h, w = 4, 64, 24, 20
    y = q(rnd(n, c, h, w, seed=1) * 2 + 0.5)
    res = q(rnd(n, c, h, w, seed=2))
    gamma, beta = rnd(c, seed=3).abs() + 0.5, rnd(c, seed=4)
    yt = y.clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.relu(F.batch_norm(yt, None, None, gt, bt, training=True, eps=1e-5) + res)
    dz = q(rnd(n, c, h, w, seed=5))
    z.backward(dz)

    yd = nhwc(y)
    mean, invstd, scale, shift = ops.bn_train_stats(yd, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1)
    close(mean.cpu(), y.mean((0, 2, 3)), 1e-5, "mean")
    close(invstd.cpu(), 1 / torch.sqrt(y.var((0, 2, 3), unbiased=False) + 1e-5), 1e-4, "invstd")
    zd = ops.bn_apply(yd, scale, shift, residual=nhwc(res), relu=True)
    assert zd.dtype == BF
    close(nchw(zd), z.detach(), TOL_BF, "z")
    # backward takes the bf16-rounded z as ReLU mask: use the reference's own z (same sign pattern except at rounding
    # ties to zero, which the tolerance absorbs)
    dy, dgamma, dbeta, dm = ops.bn_bwd(nhwc(dz), nhwc(z.detach()), yd, mean, invstd, gamma.to(DEV), want_masked=True)
    close(dgamma.cpu(), gt.grad, TOL_F32, "dgamma")
    close(dbeta.cpu(), bt.grad, TOL_F32, "dbeta")
    close(nchw(dy), yt.grad, TOL_BF, "dy")
    close(nchw(dm), dz * (z.detach() > 0), TOL_BF, "dmasked")


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("case", [(2, 64, 24, 20, 128, 3, 1, 1), (3, 128, 16, 16, 64, 1, 2, 0), (1, 32, 37, 19, 32, 3, 1, 1)],
                         ids=["128x128", "64x64_s2", "128x32_tail"])
def test_conv_with_fused_bn_statistics(case, dtype):
    """rs_conv2d_fwd_bnstats_dt + rs_bn_finalize_stats == rs_conv2d_fwd followed by rs_bn_train_stats (same stored
    output, statistics of the stored values), and both match torch's batch statistics."""
    from robosat_amd import ops

    n, cin, h, w, cout, k, stride, pad = case
    x = q(rnd(n, cin, h, w, seed=1)) if dtype == BF else rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5
    wt = q(wt) if dtype == BF else wt
    gamma, beta = rnd(cout, seed=3).abs() + 0.5, rnd(cout, seed=4)
    xd, wd = nhwc(x, dtype), krsc(wt, dtype)
    y_ref = ops.conv2d(xd, wd, stride=stride, pad=pad)
    y, partial = ops.conv2d_bnstats(xd, wd, stride=stride, pad=pad)
    assert torch.equal(y, y_ref)
    m = y.numel() // cout
    rm, rv, nbt = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    got = ops.bn_finalize_stats(partial, m, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rm, rv, nbt)
    rm2, rv2, nbt2 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    want = ops.bn_train_stats(y_ref, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rm2, rv2, nbt2)
    for a, b, name in zip(got, want, ("mean", "invstd", "scale", "shift")):
        close(a.cpu(), b.cpu(), 2e-5, name)
    close(rm.cpu(), rm2.cpu(), 2e-5, "running_mean")
    close(rv.cpu(), rv2.cpu(), 2e-5, "running_var")
    assert int(nbt) == 1 == int(nbt2)
    yf = nchw(y)
    close(got[0].cpu(), yf.mean((0, 2, 3)), 1e-4, "mean vs torch")
    close(got[1].cpu(), 1 / torch.sqrt(yf.var((0, 2, 3), unbiased=False) + 1e-5), 1e-4, "invstd vs torch")


@pytest.mark.parametrize("cin,n,h,w", [(3, 2, 64, 96), (4, 1, 128, 64), (3, 3, 32, 32)])
def test_stem_bf16(cin, n, h, w):
    """resnet.conv1 7x7/2 in bf16: forward (+ folded-BN/ReLU epilogue) and weight gradient vs fp32 PyTorch on bf16-rounded
    operands (reference unet.py:122)."""
    from robosat_amd import ops

    x = q(rnd(n, cin, h, w, seed=14))
    wt = q(rnd(64, cin, 7, 7, seed=15) * 0.1).requires_grad_(True)
    y = F.conv2d(x, wt, stride=2, padding=3)
    x4 = ops.nchw_to_nhwc4(x.to(DEV), BF)
    assert x4.dtype == BF and x4.shape == (n, h, w, 4)
    assert torch.equal(x4[..., :cin].float().cpu(), x.permute(0, 2, 3, 1))
    wp = ops.pack_stem_weight(krsc(wt.detach(), torch.float32), BF)
    got = ops.stem_conv_bf16(x4, wp)
    close(nchw(got), y.detach(), TOL_BF, "stem fwd")
    sc, sh = rnd(64, seed=16), rnd(64, seed=17)
    got2 = ops.stem_conv_bf16(x4, wp, scale=sc.to(DEV), shift=sh.to(DEV), relu=True)
    close(nchw(got2), F.relu(y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), TOL_BF, "stem fwd epilogue")
    gy = q(rnd(*y.shape, seed=18))
    y.backward(gy)
    dwp = ops.stem_conv_wgrad_bf16(nhwc(gy), x4)
    dw = ops.unpack_stem_weight(dwp, 7, cin)
    close(dw.permute(0, 3, 1, 2).cpu(), wt.grad, TOL_F32, "stem wgrad")


def test_pool_upsample_final_bf16():
    from robosat_amd import ops

    # stem pool: fp32 in, bf16 out; backward bf16 dy -> fp32 dx
    x = rnd(2, 64, 18, 22, seed=16)
    xt = x.clone().requires_grad_(True)
    want = F.max_pool2d(xt, 3, 2, 1)
    got, amax = ops.maxpool2d(nhwc(x, torch.float32), 3, 2, 1, want_argmax=True, out_dtype=BF)
    assert got.dtype == BF
    assert torch.equal(nchw(got), q(want.detach()))
    dy = q(rnd(*want.shape, seed=17))
    want.backward(dy)
    dx = ops.maxpool2d_bwd(nhwc(dy), amax, (2, 18, 22, 64), 3, 2, 1, out_dtype=torch.float32)
    assert dx.dtype == torch.float32
    close(nchw(dx), xt.grad, 1e-6, "maxpool bwd")
    # bf16 -> bf16 pool (center), accumulate form
    xb = q(x)
    got2, amax2 = ops.maxpool2d(nhwc(xb), 2, 2, 0, want_argmax=True)
    assert torch.equal(nchw(got2), F.max_pool2d(xb, 2, 2, 0))
    base = q(rnd(2, 64, 18, 22, seed=18))
    xt2 = xb.clone().requires_grad_(True)
    w2 = F.max_pool2d(xt2, 2, 2, 0)
    dy2 = q(rnd(*w2.shape, seed=19))
    w2.backward(dy2)
    acc = nhwc(base)
    ops.maxpool2d_bwd(nhwc(dy2), amax2, (2, 18, 22, 64), 2, 2, 0, out=acc)
    close(nchw(acc), base + xt2.grad, TOL_BF, "maxpool bwd acc")

    # upsample backward + cat split + masks
    n, c1, c2, h, w = 2, 64, 32, 6, 10
    dup = q(rnd(n, c1 + c2, 2 * h, 2 * w, seed=20))
    m1, m2 = q(rnd(n, c1, h, w, seed=21)), q(rnd(n, c2, h, w, seed=22))
    s = dup.view(n, c1 + c2, h, 2, w, 2).sum((3, 5))
    d1, d2 = ops.upsample2x_bwd(nhwc(dup), c1, c2, mask1=nhwc(m1), mask2=nhwc(m2))
    close(nchw(d1), s[:, :c1] * (m1 > 0), TOL_BF, "d1")
    close(nchw(d2), s[:, c1:] * (m2 > 0), TOL_BF, "d2")

    # final 1x1 (+ softmax) and its backward
    for ncls in (2, 4):
        xf = q(rnd(2, 32, 32, 40, seed=23).abs())
        wf, bias = rnd(ncls, 32, 1, 1, seed=24) * 0.3, rnd(ncls, seed=25)
        xt3, wt3, bt3 = xf.clone().requires_grad_(True), wf.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        logits = F.conv2d(xt3, wt3, bt3)
        gotl = ops.final_conv1x1(nhwc(xf), wf.view(ncls, 32).to(DEV), bias.to(DEV))
        close(gotl.cpu(), logits.detach(), 1e-5, "final")
        gl = rnd(*logits.shape, seed=26)
        logits.backward(gl)
        dxf, dwf, dbf = ops.final_conv1x1_bwd(nhwc(xf), wf.view(ncls, 32).to(DEV), gl.to(DEV), relu_mask=True)
        assert dxf.dtype == BF
        close(nchw(dxf), xt3.grad * (xf > 0), TOL_BF, "final dx")
        close(dwf.cpu().view_as(wf), wt3.grad, TOL_F32, "final dw")
        close(dbf.cpu(), bt3.grad, TOL_F32, "final db")


def _pair(num_classes, seed):
    from robosat_amd.unet import UNet

    ref = R.UNetRef(num_classes)
    sd = seeded.seeded_state_dict(ref.state_dict(), seed)
    ref.load_state_dict(sd)
    net = UNet(num_classes, pretrained=False, compute_dtype=BF)
    net.load_state_dict(sd)
    return ref, net.to(DEV)


def test_unet_bf16_predict_vs_oracle():
    """bf16 predict: probabilities within 3e-2 of the fp32 CPU oracle and >= 99.5 % identical decisions (the fp32 path
    is the one held to the 1e-3 bar, tests/test_gpu_unet.py)."""

    ref, net = _pair(2, 11)
    x = seeded.synthetic_images(2, 3, 256, 256, seed=5)
    want = R.predict_probs(ref.eval(), x)
    got = net.eval().predict_probs(x.to(DEV)).cpu()
    assert got.dtype == torch.float32
    err = float((got - want).abs().max())
    agree = float((got.argmax(1) == want.argmax(1)).float().mean())
    print("bf16 predict: max|dprob| {:.3e}, argmax agreement {:.4f}".format(err, agree))
    assert err <= 3e-2
    assert agree >= 0.995


class _RoundBF16(torch.autograd.Function):
    """x -> bf16 -> fp32 in both directions: what storing an activation / its gradient as bf16 does."""

    @staticmethod
    def forward(ctx, x):
        return x.to(BF).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(BF).float()


def _oracle_grads(loss_name, x, t, wts, emulate_bf16):
    """fp32 CPU oracle train step; with ``emulate_bf16`` every activation the bf16 path stores (conv / BN / ReLU / pool
    outputs, and the uploaded image) is rounded to bf16, forward and backward: the CALIBRATION of what bf16 storage alone does
    to the gradients of this network at this batch size."""

    ref = R.UNetRef(2)
    ref.load_state_dict(seeded.seeded_state_dict(ref.state_dict(), 2))
    ref.train()
    if emulate_bf16:
        for name, m in ref.named_modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.ReLU, torch.nn.MaxPool2d)) and name != "final":
                m.register_forward_hook(lambda mod, inp, out: _RoundBF16.apply(out))
        x = x.to(BF).float()  # the image is cast on upload
    out = ref(x)
    rl = R.cross_entropy2d(out, t, weight=wts) if loss_name == "CrossEntropy" else R.lovasz2d(out, t)
    rl.backward()
    return ref, rl.item(), {n: p.grad for n, p in ref.named_parameters()}


def _cosines(got, want):
    out = {}
    for name, w in want.items():
        if w is None or float(w.norm()) < 1e-7:
            continue
        g = got[name].float().cpu()
        out[name] = float((g * w).sum() / (g.norm() * w.norm() + 1e-30))
    return out


def test_unet_bf16_predict_across_batch_sizes():
    """the fp32 path is batch-invariant bit for bit (tests/test_gpu_configs.py); the bf16 path is NOT promised to be --
    its dispatcher picks the halo-once forms (K accumulated chunk-major) or the implicit GEMM (tap-major) by how many blocks a
    launch has, which includes the batch -- so a tile's bf16 probabilities may differ by rounding with the batch it travels in.
    What is promised: within the bf16 bar of the oracle at every batch size, and within one rounding step of each other."""

    ref, net = _pair(2, 11)
    net = net.eval()
    x = seeded.synthetic_images(12, 3, 256, 256, seed=6)
    want = R.predict_probs(ref.eval(), x[:3])
    together = net.predict_probs(x.to(DEV)).cpu()  # 12 tiles: enough blocks for the halo forms on the 3x3 layers
    alone = torch.cat([net.predict_probs(x[i:i + 1].to(DEV)).cpu() for i in range(3)])
    assert float((together[:3] - want).abs().max()) <= 3e-2 and float((alone - want).abs().max()) <= 3e-2
    d = float((together[:3] - alone).abs().max())
    agree = float((together[:3].argmax(1) == alone.argmax(1)).float().mean())
    print("bf16 predict, a tile alone vs in a batch of 12: max|dprob| {:.3e}, argmax agreement {:.5f}".format(d, agree))
    assert d <= 2e-2 and agree >= 0.998


@pytest.mark.parametrize("loss_name", ["CrossEntropy", "Lovasz"])
def test_unet_bf16_train_step_vs_oracle(loss_name):
    """One bf16 training step vs the fp32 CPU oracle on the same seeded weights / batch.

    Loss within 2 %.  Gradients: bf16 storage of ~100 stacked activations perturbs the encoder's gradients visibly at
    this tiny batch (32-sample BatchNorm statistics in layer4), so the bar is CALIBRATED: the same oracle with bf16
    rounding emulated at every stored activation gives (measured) mean cosine ~0.94 / worst ~0.88 against fp32; the HIP
    path must be as close to fp32 as that emulation is (mean within 0.03, worst within 0.08), and >= 0.98 on the decoder /
    head, whose gradients have crossed few bf16 layers.  BatchNorm running statistics within 1e-2."""
    from robosat_amd import losses

    x = seeded.synthetic_images(2, 3, 128, 128, 2)
    t = seeded.synthetic_targets(2, 2, 128, 128, 2)
    wts = torch.tensor([1.6248, 5.762827])
    ref, rloss, rgrads = _oracle_grads(loss_name, x, t, wts, emulate_bf16=False)
    _, eloss, egrads = _oracle_grads(loss_name, x, t, wts, emulate_bf16=True)
    ecos = _cosines(egrads, rgrads)
    e_mean, e_worst = sum(ecos.values()) / len(ecos), min(ecos.values())

    _, net = _pair(2, 2)
    net.train()
    crit = (losses.CrossEntropyLoss2d(weight=wts) if loss_name == "CrossEntropy" else losses.LovaszLoss2d()).to(DEV)
    logits = net(x.to(DEV))
    assert logits.dtype == torch.float32
    loss = crit(logits, t.to(DEV))
    loss.backward()
    print(loss_name, "bf16 loss", loss.item(), "oracle fp32", rloss, "oracle bf16-emulated", eloss)
    assert abs(loss.item() - rloss) <= 2e-2 * max(1.0, abs(rloss))

    grads = {}
    for name, p in net.named_parameters():
        if rgrads[name] is None:
            assert p.grad is None, name
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
            grads[name] = p.grad
    cos = _cosines(grads, rgrads)
    mean, worst = sum(cos.values()) / len(cos), min(cos.values())
    print("gradient cosine vs fp32 oracle: HIP bf16 mean {:.4f} worst {:.4f} | bf16-emulated oracle mean {:.4f} worst {:.4f} ({} tensors)".format(
        mean, worst, e_mean, e_worst, len(cos)))
    assert len(cos) >= 160
    assert mean >= e_mean - 0.03 and worst >= e_worst - 0.08
    for name, c in cos.items():
        if name.startswith(("dec", "center", "final")):
            assert c >= 0.98, (name, c)
        w = rgrads[name]
        if w.dim() == 4:
            gn, wn = float(grads[name].float().norm()), float(w.norm())
            assert abs(gn - wn) <= 0.15 * wn, (name, gn, wn)
    rb = dict(ref.named_buffers())
    for name, b in net.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            want = rb[name]
            assert float((b.cpu() - want).abs().max()) <= 1e-2 * max(1.0, float(want.abs().max())), name

def test_weight_prep_multi_tensor_is_bit_identical():
    """rs_weight_prep_bf16 (one launch for many weights) == rs_cast_f32_to_bf16 + rs_pack_dgrad_weight_bf16 per tensor, and
    UNet.prep_bf16_weights() re-runs exactly when a master weight changes."""
    from robosat_amd import ops
    from robosat_amd.unet import UNet

    shapes = [(64, 1, 1, 64), (64, 3, 3, 64), (256, 1, 1, 64), (96, 3, 3, 40), (33, 1, 1, 70), (512, 3, 3, 512)]
    ws = [rnd(*s, seed=40 + i).to(DEV) for i, s in enumerate(shapes)]
    prep = ops.WeightPrep(ws)
    prep.run()
    for w, c, d in zip(ws, prep.cast, prep.dgrad):
        assert torch.equal(c, ops.cast_bf16(w))
        assert torch.equal(d, ops.pack_dgrad_weight(w, torch.bfloat16))

    net = UNet(2, pretrained=False, compute_dtype=torch.bfloat16).to(DEV)
    net.prep_bf16_weights()
    c = net.resnet.layer2[1].conv2
    first = c.krsc(torch.bfloat16)
    assert first is net._wprep[0].cast[net._wprep[0].ptrs.index(c.krsc().data_ptr())]
    assert torch.equal(first, ops.cast_bf16(c.krsc())) and torch.equal(c.dgrad_weight(torch.bfloat16), ops.pack_dgrad_weight(c.krsc(), torch.bfloat16))
    with torch.no_grad():
        c.weight.mul_(1.5)  # what an optimizer step does: in place, version bump
    assert c.krsc(torch.bfloat16) is not first or not torch.equal(c.krsc(torch.bfloat16), first)  # stale copy is not served
    net.prep_bf16_weights()
    assert torch.equal(c.krsc(torch.bfloat16), ops.cast_bf16(c.krsc()))
    assert torch.equal(c.dgrad_weight(torch.bfloat16), ops.pack_dgrad_weight(c.krsc(), torch.bfloat16))


def test_weight_prep_fp32_twin_is_bit_identical():
    """rs_weight_prep_f32 (round 5: the fp32 training step's data-gradient layouts in one launch) == rs_pack_dgrad_weight per
    tensor; UNet.prep_f32_weights() serves them through _Conv.dgrad_weight(float32) until a master weight changes."""
    from robosat_amd import ops
    from robosat_amd.unet import UNet

    shapes = [(64, 1, 1, 64), (64, 3, 3, 64), (256, 1, 1, 64), (96, 3, 3, 40), (33, 1, 1, 70), (512, 3, 3, 512)]
    ws = [rnd(*s, seed=60 + i).to(DEV) for i, s in enumerate(shapes)]
    prep = ops.WeightPrep(ws, dtype=torch.float32)
    prep.run()
    for w, c, d in zip(ws, prep.cast, prep.dgrad):
        assert c is None and d.dtype == torch.float32
        assert torch.equal(d, ops.pack_dgrad_weight(w, torch.float32))

    net = UNet(2, pretrained=False).to(DEV)
    c = net.resnet.layer3[2].conv2
    assert getattr(c, "_dgrad_f32", None) is None
    net.prep_f32_weights()
    served = c.dgrad_weight(torch.float32)
    assert served is net._wprep_f32[0].dgrad[net._wprep_f32[0].ptrs.index(c.krsc().data_ptr())]
    assert torch.equal(served, ops.pack_dgrad_weight(c.krsc(), torch.float32))
    with torch.no_grad():
        c.weight.mul_(1.5)
    assert torch.equal(c.dgrad_weight(torch.float32), ops.pack_dgrad_weight(c.krsc(), torch.float32))  # stale copy is not served
    net.prep_f32_weights()
    assert torch.equal(c.dgrad_weight(torch.float32), ops.pack_dgrad_weight(c.krsc(), torch.float32))
