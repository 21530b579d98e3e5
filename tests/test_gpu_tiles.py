"""Every convolution symbol the benchmark launches, reached on purpose.

The dispatcher (``pick_tile`` / ``pick_rowb``) routes by problem size: the 8-wave 256x256 bf16 tile needs >= 384 blocks,
the 128-byte rows a long K, ... -- sizes a CPU reference cannot check in seconds.  ``rs_conv2d_set_tuning`` forces tile and
K-chunk row size, so each (tile, rows, form) symbol is compared with plain PyTorch fp32 on problems of a few thousand
pixels, including M tails, ragged N, both concat sources, every epilogue option, the phase form (DecoderBlock), its
4x4 / stride-2 data gradient and the two-destination store.  ``test_bench_symbols_are_covered`` closes the loop: every
kernel name a full-size ``bench.py`` run reports must be in the set these tests exercise.

Reference semantics: robosat/unet.py:32-44 (ConvRelu), :63-73 (DecoderBlock), :134-137 (torch.cat) and their autograd."""

import json
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TILES = ["128x128", "128x64", "128x32", "64x64", "256x128", "256x256"]
ROWS = [64, 128]
# symbols these tests launch (and assert they launched): conv_igemm_<f32|bf16><[phase,|dgrad4x4,]tile,r<rows>>
COVERED = set()
for _dt in ("f32", "bf16"):
    for _t in TILES:
        if _t == "256x256" and _dt == "f32":
            continue
        for _r in ROWS:
            for _form in ("", "phase,", "dgrad4x4,"):
                COVERED.add("conv_igemm_{}<{}{},r{}>".format(_dt, _form, _t, _r))
# non-implicit-GEMM symbols of the step have their own parity tests (test_gpu_ops / test_gpu_train_ops / test_gpu_bf16)
COVERED |= {"conv_thin_bf16<3x3>", "conv_thin_bf16<phase>", "conv_thin_bf16<dgrad4x4>"}  # test_thin_* below
# layer1's fused Bottleneck tail (conv3 + identity + ReLU -> the next block's conv1): tests/test_gpu_ops.py::test_bottleneck_tail_*
COVERED |= {"bottleneck_tail_f32", "conv1x1_wave_f32"}
COVERED |= {"stem_conv_f32<128x64>", "stem_conv_bf16", "stem_wgrad_bf16", "conv_wgrad_f32", "conv_wgrad_f32_dma", "conv_wgrad_wino_f32", "conv_wgrad_wino33_f32"}
# fp32 1x1 launches with K <= 64 (layer1) take the epilogue-wave kernel by rule: test_epilogue_wave_1x1_kernel_* below
COVERED |= {"conv1x1_ew_f32<128x64,r64>"}
# the fp32 DecoderBlock data gradient in the Winograd form (round 6): tests/test_gpu_train_ops.py::test_wino_dgrad_*
COVERED |= {"conv_wino_f32<dgrad4x4,p8,128x64>", "conv_wino_f32<dgrad4x4,p8,64x64>"}
# the fp32 train forward's stride-1 3x3 convolutions in the Winograd form with BatchNorm's partial sums (round 6):
# tests/test_gpu_train_ops.py::test_wino33_train_forward_with_statistics_*
COVERED |= {"conv_wino_f32<3x3+stats,p8,64x32>", "conv_wino_f32<3x3+stats,p8,128x16>"}
# ... and their data gradients in the same form (ReLU mask + BatchNorm backward partial sums): test_wino33_data_gradient_*
COVERED |= {"conv_wino_f32<3x3+bwd,p8,64x32>"}
# all-taps weight gradient: per input-channel slab, with / without the fused upsample (test_wgrad_bf16 "thin_*" cases assert the
# names; test_wgrad_bf16_thin_upsample runs the ",ups" forms)
COVERED |= {"conv_wgrad_thin_bf16<{}>".format(t) for t in ("32", "64", "128", "32,ups", "64,ups", "128,ups")}
# bf16 weight-gradient tiles: tests/test_gpu_bf16.py::test_wgrad_bf16 / test_wgrad_bf16_upsample_concat assert, case by case,
# that these are the instantiations they ran (the last one is dec3's per-source split)
COVERED |= {"conv_wgrad_bf16<{}>".format(t) for t in ("256x128", "128x128", "128x64", "64x128", "64x64", "32x128", "32x32")}
COVERED |= {"conv_wgrad_bf16<phase,{}>".format(t) for t in ("128x128", "128x64", "64x128", "64x128+64x64", "128x128+128x64")}
# ... and, with knob wgrad_phase4 = 1, the 128 x 128 launches as one dz plane x four source offsets per block (test_wgrad_bf16_phase_form_four_offsets_per_block)
COVERED |= {"conv_wgrad_bf16<phase4,128x128>", "conv_wgrad_bf16<phase4,128x128+128x64>"}


# halo-once forms of the kernel (bf16): form x N tile; tests/test_gpu_tiles.py::test_halo_*
COVERED |= {"conv_halo_bf16<{},256x{}>".format(f, bn) for f in ("3x3", "phase", "dgrad4x4") for bn in (128, 64)}
COVERED |= {"conv_halo_bf16<{},512x128>".format(f) for f in ("3x3", "phase", "dgrad4x4")}  # 16 x 32 patches, 32-channel chunks


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def prep(t, dtype):  # operand as the kernel sees it, in fp32 on the host
    return t.to(BF).float() if dtype == BF else t


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def nchw(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


def krsc(w, dtype):
    return w.permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)


def close(got, want, dtype, what=""):
    tol = 1e-2 if dtype == BF else 2e-4
    scale = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, "{} max abs err {} (scale {})".format(what, err, scale)


def combos():
    out = []
    for dtype in (torch.float32, BF):
        for tile in TILES:
            if tile == "256x256" and dtype != BF:
                continue
            for rowb in ROWS:
                out.append(pytest.param(dtype, tile, rowb, id="{}-{}-r{}".format("bf16" if dtype == BF else "f32", tile, rowb)))
    return out


def launched(d_args, dtype, phase, tile, rowb, dgrad=False):
    """Name of the symbol the forced dispatcher reports for this launch; asserts it is the one meant."""
    from robosat_amd import ops

    d = ops.conv_desc(*d_args[0], **d_args[1]) if not phase else d_args
    name = ops.conv_tile_name(d, dtype == BF, phase=phase)
    if dgrad:
        name = name.replace("<", "<dgrad4x4,")
    want = "conv_igemm_{}<{}{},r{}>".format("bf16" if dtype == BF else "f32", "phase," if phase else ("dgrad4x4," if dgrad else ""), tile, rowb)
    assert name == want, (name, want)
    assert name in COVERED
    return name


@pytest.mark.parametrize("dtype,tile,rowb", combos())
def test_forced_tile_plain_conv_all_epilogues(dtype, tile, rowb):
    """3x3 / pad 1 over two concatenated sources, M = 2*23*21 = 966 (tail in every tile height), Cout = 256; epilogue
    scale/shift + residual + ReLU, then the ReLU-mask form; then the fused BatchNorm statistics where the tile has them."""
    from robosat_amd import ops

    n, c1, c2, h, w, cout = 2, 128, 64, 23, 21, 256
    a, b = prep(rnd(n, c1, h, w, seed=1), dtype), prep(rnd(n, c2, h, w, seed=2), dtype)
    wt = prep(rnd(cout, c1 + c2, 3, 3, seed=3) * (2.0 / ((c1 + c2) * 9)) ** 0.5, dtype)
    sc, sh = rnd(cout, seed=4), rnd(cout, seed=5)
    res, mask = prep(rnd(n, cout, h, w, seed=6), dtype), prep(rnd(n, cout, h, w, seed=7), dtype)
    base = F.conv2d(torch.cat([a, b], 1), wt, padding=1)
    full = base * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res
    ad, bd, wd = nhwc(a, dtype), nhwc(b, dtype), krsc(wt, dtype)
    with ops.tuning(tile, rowb):
        launched(((ad, wd), dict(src2=bd, pad=1)), dtype, False, tile, rowb)
        got = ops.conv2d(ad, wd, src2=bd, pad=1, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res, dtype), relu=True)
        close(nchw(got), F.relu(full), dtype, "relu")
        got = ops.conv2d(ad, wd, src2=bd, pad=1, relu_mask=nhwc(mask, dtype))
        close(nchw(got), base * (mask > 0), dtype, "mask")
        if tile != "256x256":  # the statistics' block reduction is laid out for the 4-wave tiles
            y, partial = ops.conv2d_bnstats(ad, wd, src2=bd, pad=1)
            close(nchw(y), base, dtype, "bnstats y")
            yf = y.float()
            s = partial.sum(0).cpu()
            want0, want1 = yf.sum((0, 1, 2)).cpu(), (yf * yf).sum((0, 1, 2)).cpu()
            assert float((s[0] - want0).abs().max()) <= 1e-3 * float(want0.abs().max() + 1)
            assert float((s[1] - want1).abs().max()) <= 1e-3 * float(want1.abs().max() + 1)


@pytest.mark.parametrize("dtype,tile,rowb", combos())
def test_forced_tile_strided_and_1x1(dtype, tile, rowb):
    """3x3 stride 2 (odd input) and a 1x1 with K = 64 -- one chunk of 128-byte bf16 rows: the shortest K loop there is."""
    from robosat_amd import ops

    n, cin, h, w, cout = 3, 64, 19, 17, 256
    x = prep(rnd(n, cin, h, w, seed=8), dtype)
    w3 = prep(rnd(cout, cin, 3, 3, seed=9) * 0.05, dtype)
    w1 = prep(rnd(cout, cin, 1, 1, seed=10) * 0.1, dtype)
    xd = nhwc(x, dtype)
    with ops.tuning(tile, rowb):
        close(nchw(ops.conv2d(xd, krsc(w3, dtype), stride=2, pad=1)), F.conv2d(x, w3, stride=2, padding=1), dtype, "3x3 s2")
        close(nchw(ops.conv2d(xd, krsc(w1, dtype))), F.conv2d(x, w1), dtype, "1x1")
        close(nchw(ops.conv2d(xd, krsc(w1, dtype), stride=2)), F.conv2d(x, w1, stride=2), dtype, "1x1 s2")


@pytest.mark.parametrize("dtype,tile,rowb", combos())
def test_forced_tile_phase_form_and_its_gradients(dtype, tile, rowb):
    """DecoderBlock (unet.py:63-73) in phase form on cat[skip, prev], its 4x4/stride-2 data gradient, and that gradient
    with torch.cat's backward fused into the store (two destinations + ReLU masks) -- against autograd."""
    from robosat_amd import _lib, ops

    n, c1, c2, hs, ws, cout = 2, 256, 256, 11, 13, 256
    a = prep(rnd(n, c1, hs, ws, seed=11), dtype).requires_grad_(True)
    b = prep(rnd(n, c2, hs, ws, seed=12), dtype).requires_grad_(True)
    wt = rnd(cout, c1 + c2, 3, 3, seed=13) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    # the phase / dgrad packs sum fp32 taps and round once: the host reference uses the fp32 weights with rounded activations
    y = F.relu(F.conv2d(F.interpolate(torch.cat([a, b], 1), scale_factor=2, mode="nearest"), wt, padding=1))
    gy = prep(rnd(*y.shape, seed=14), dtype)
    y.backward(gy)
    w_krsc = krsc(wt, torch.float32)
    ad, bd = nhwc(a.detach(), dtype), nhwc(b.detach(), dtype)
    with ops.tuning(tile, rowb):
        d = _lib.ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, 1, 0)
        launched(d, dtype, True, tile, rowb)
        got = ops.conv2d_phase(ad, ops.pack_phase_weight(w_krsc, dtype), src2=bd, relu=True)
        tol_dtype = dtype
        close(nchw(got), y.detach(), tol_dtype, "phase fwd")
        # data gradient: dz = gy masked by the ReLU, one 4x4 / stride-2 convolution back to the source grid
        dz = (gy * (y.detach() > 0))
        wd = ops.pack_dgrad_phase_weight(w_krsc, dtype)
        dzd = nhwc(dz, dtype)
        launched(((dzd, wd), dict(stride=2, pad=1, out_hw=(hs, ws))), dtype, False, tile, rowb, dgrad=True)
        dsrc = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws))
        want = torch.cat([a.grad, b.grad], 1)
        close(nchw(dsrc), want, dtype, "dgrad4x4")
        m1, m2 = prep(rnd(n, c1, hs, ws, seed=15), dtype), prep(rnd(n, c2, hs, ws, seed=16), dtype)
        if c1 % int(tile.split("x")[1]) == 0:
            d1, d2 = ops.conv2d_split(dzd, wd, c1, stride=2, pad=1, out_hw=(hs, ws), mask1=nhwc(m1, dtype), mask2=nhwc(m2, dtype))
            close(nchw(d1), a.grad * (m1 > 0), dtype, "split d1")
            close(nchw(d2), b.grad * (m2 > 0), dtype, "split d2")


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("rowb", ROWS)
def test_ragged_last_n_tile_with_split_store(dtype, rowb):
    """Cout = 320 = 256 + 64 on 128-wide tiles (dec3's data gradient): the third N tile is half padding (weight rows past
    Cout read as zeros, the epilogue skips their columns) and the store splits at 256."""
    from robosat_amd import ops

    n, cin, h, w, c1, c2 = 2, 128, 24, 22, 256, 64
    x = prep(rnd(n, cin, h, w, seed=17), dtype)
    wt = prep(rnd(c1 + c2, cin, 3, 3, seed=18) * 0.04, dtype)
    want = F.conv2d(x, wt, padding=1)
    m1, m2 = prep(rnd(n, c1, h, w, seed=19), dtype), prep(rnd(n, c2, h, w, seed=20), dtype)
    xd, wd = nhwc(x, dtype), krsc(wt, dtype)
    with ops.tuning("128x128", rowb):
        got = ops.conv2d(xd, wd, pad=1)
        close(nchw(got), want, dtype, "ragged")
        d1, d2 = ops.conv2d_split(xd, wd, c1, pad=1, mask1=nhwc(m1, dtype), mask2=nhwc(m2, dtype))
        close(nchw(d1), want[:, :c1] * (m1 > 0), dtype, "ragged split 1")
        close(nchw(d2), want[:, c1:] * (m2 > 0), dtype, "ragged split 2")


THIN_SHAPES = [(2, 40, 72), (1, 16, 32), (5, 256, 256), (1, 24, 8)]  # edges in both directions; > 256 patches (persistent loop); tiny


@pytest.mark.parametrize("n,h,w", THIN_SHAPES)
def test_thin_3x3_c32_all_taps_kernel(n, h, w):
    """dec5 (ConvRelu 32 -> 32, unet.py:107,139) and its data gradient on the all-taps kernel (conv_thin_bf16.hip): ReLU and
    ReLU-mask epilogues vs fp32 PyTorch on the same bf16 operands, and vs the generic kernel forced on the same launch."""
    from robosat_amd import ops

    x = prep(rnd(n, 32, h, w, seed=31), BF)
    wt = prep(rnd(32, 32, 3, 3, seed=32) * 0.08, BF)
    mask = prep(rnd(n, 32, h, w, seed=33), BF)
    xd, wd, md = nhwc(x, BF), krsc(wt, BF), nhwc(mask, BF)
    assert ops.conv_tile_name(ops.conv_desc(xd, wd, pad=1), True) == "conv_thin_bf16<3x3>"
    want = F.conv2d(x, wt, padding=1)
    got = ops.conv2d(xd, wd, pad=1, relu=True)
    close(nchw(got), F.relu(want), BF, "relu")
    gotm = ops.conv2d(xd, wd, pad=1, relu_mask=md)
    close(nchw(gotm), want * (mask > 0), BF, "mask")
    plain = ops.conv2d(xd, wd, pad=1)
    with ops.tuning("128x32", 64):  # the generic implicit-GEMM kernel on the same launch: same values up to one bf16 rounding
        assert ops.conv_tile_name(ops.conv_desc(xd, wd, pad=1), True) == "conv_igemm_bf16<128x32,r64>"
        ref = ops.conv2d(xd, wd, pad=1)
    assert float((plain.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
    # a launch with a scale / residual epilogue is not the thin kernel's: it must fall through to the generic one
    sc = torch.rand(32, device=DEV) + 0.5
    close(nchw(ops.conv2d(xd, wd, pad=1, scale=sc, shift=torch.zeros(32, device=DEV))), want * sc.cpu().view(1, -1, 1, 1), BF, "scaled")


@pytest.mark.parametrize("n,hs,ws", [(2, 20, 36), (1, 8, 16), (5, 128, 128), (1, 12, 4)])
def test_thin_phase_c128_and_its_4x4_gradient(n, hs, ws):
    """dec4 (DecoderBlock 128 -> 32, unet.py:106,138): phase form with the four parities sharing one source halo, and the
    4x4 / stride-2 data gradient 32 -> 128 with the ReLU mask of dec3 -- against autograd on the reference formulation."""
    from robosat_amd import _lib, ops

    a = prep(rnd(n, 128, hs, ws, seed=34), BF).requires_grad_(True)
    wt = rnd(32, 128, 3, 3, seed=35) * (2.0 / (128 * 9)) ** 0.5
    y = F.relu(F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), wt, padding=1))
    gy = prep(rnd(*y.shape, seed=36), BF)
    y.backward(gy)
    w_krsc = krsc(wt, torch.float32)
    ad = nhwc(a.detach(), BF)
    d = _lib.ConvDesc(n, hs, ws, 128, 0, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, 32, 1, 0)
    assert ops.conv_tile_name(d, True, phase=True) == "conv_thin_bf16<phase>"
    got = ops.conv2d_phase(ad, ops.pack_phase_weight(w_krsc, BF), relu=True)
    close(nchw(got), y.detach(), BF, "phase fwd")
    dz = gy * (y.detach() > 0)
    wd = ops.pack_dgrad_phase_weight(w_krsc, BF)
    dzd = nhwc(dz, BF)
    assert ops.conv_tile_name(ops.conv_desc(dzd, wd, stride=2, pad=1, out_hw=(hs, ws)), True) == "conv_thin_bf16<dgrad4x4>"
    m = prep(rnd(n, 128, hs, ws, seed=37), BF)
    gsrc = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws), relu_mask=nhwc(m, BF), alg_scale=2.25)
    close(nchw(gsrc), a.grad * (m > 0), BF, "dgrad4x4 + mask")
    with ops.tuning("128x128", 64):
        ref = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws), relu_mask=nhwc(m, BF))
        refp = ops.conv2d_phase(ad, ops.pack_phase_weight(w_krsc, BF), relu=True)
    assert float((gsrc.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
    assert float((got.float() - refp.float()).abs().max()) <= 2 ** -7 * float(refp.float().abs().max())


def test_dispatcher_reaches_the_8_wave_tile_unforced():
    """One launch big enough for the heuristics themselves to pick 256x256 (>= 384 blocks, long K), checked on a strided
    sample of images against fp32 PyTorch: 1280 -> 256 3x3 at 32 x 56^2 = 100 352 pixels (392 blocks)."""
    from robosat_amd import ops

    n, cin, h, w, cout = 32, 1280, 56, 56, 256
    g = torch.Generator(device=DEV).manual_seed(21)
    xd = torch.randn(n, h, w, cin, device=DEV, generator=g).to(BF)
    wd = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) * 0.01).to(BF)
    d = ops.conv_desc(xd, wd, pad=1)
    assert ops.conv_tile_name(d, True) == "conv_igemm_bf16<256x256,r128>"
    got = ops.conv2d(xd, wd, pad=1)
    # reference on a sample: images 0 and 31, full (fp32 math on the same bf16 operands, on the host)
    for img in (0, n - 1):
        x = xd[img:img + 1].float().permute(0, 3, 1, 2).cpu()
        want = F.conv2d(x, wd.float().permute(0, 3, 1, 2).cpu(), padding=1)
        close(nchw(got[img:img + 1]), want, BF, "image {}".format(img))


# ---- halo-once forms (conv_igemm_dma_kernel.h, HALO template parameter) ------------------------------------------------------
def _halo_name(d, phase=False):
    from robosat_amd import ops

    name = ops.conv_tile_name(d, True, phase=phase)
    assert name in COVERED, name
    return name


def test_fused_statistics_on_a_4x4_stride2_bf16_launch_keep_the_implicit_gemm_kernel():
    """only the 3x3 halo form carries the statistics / into-BatchNorm epilogues.  A bf16 4x4 / stride-2 /
    pad-1 launch of the data-gradient halo form's geometry that asks for fused BatchNorm statistics must therefore keep the
    implicit-GEMM kernel -- and write every partial row it sized (rs_conv2d_bnstats_rows_dt and the launch agree)."""
    from robosat_amd import ops

    n, c, cout, hs = 2, 128, 128, 64
    x = prep(rnd(n, c, hs, hs, seed=71), BF)
    wt = prep(rnd(cout, c, 4, 4, seed=72) * (2.0 / (c * 16)) ** 0.5, BF)
    xd, wd = nhwc(x, BF), krsc(wt, BF)
    d = ops.conv_desc(xd, wd, stride=2, pad=1)
    base = F.conv2d(x, wt, stride=2, padding=1)
    with ops.knob("conv_halo_min", 1):  # (the unforced rule at any grid size: a plain launch of this shape takes the halo form ...)
        assert _halo_name(d) == "conv_halo_bf16<dgrad4x4,256x128>"
        close(nchw(ops.conv2d(xd, wd, stride=2, pad=1)), base, BF, "plain 4x4/s2 (halo form)")
        y, partial = ops.conv2d_bnstats(xd, wd, stride=2, pad=1)  # (... one with fused statistics does not)
        torch.cuda.synchronize()
    _fused_4x4_checks(ops, n, cout, hs, xd, wd, base, y, partial)


def _fused_4x4_checks(ops, n, cout, hs, xd, wd, base, y, partial):
    assert partial.shape[0] in (n * (hs // 2) * (hs // 2) // 128, n * (hs // 2) * (hs // 2) // 64)  # one row per implicit-GEMM M tile
    close(nchw(y), base, BF, "bnstats y")
    yf = y.float()
    s = partial.sum(0).cpu()
    want0, want1 = yf.sum((0, 1, 2)).cpu(), (yf * yf).sum((0, 1, 2)).cpu()
    assert torch.isfinite(partial).all()
    assert float((s[0] - want0).abs().max()) <= 1e-3 * float(want0.abs().max() + 1)
    assert float((s[1] - want1).abs().max()) <= 1e-3 * float(want1.abs().max() + 1)
    # the into-BatchNorm epilogue on the same geometry
    bn_y = prep(rnd(n, cout, hs // 2, hs // 2, seed=73), BF)
    mean, invstd = rnd(cout, seed=74) * 0.1, rnd(cout, seed=75).abs() + 0.5
    mask = prep(rnd(n, cout, hs // 2, hs // 2, seed=76), BF)
    with ops.knob("conv_halo_min", 1):
        g, part = ops.conv2d_dgrad_bnstats(xd, wd, (hs // 2, hs // 2), nhwc(bn_y, BF), mean.to(DEV), invstd.to(DEV), pad=1,
                                           relu_mask=nhwc(mask, BF), stride=2)
    close(nchw(g), base * (mask > 0), BF, "dgrad-into-bn g on 4x4/s2")
    gf = nchw(g)
    xhat = (bn_y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    ps = part.sum(0).cpu()
    w0, w1 = gf.sum((0, 2, 3)), (gf * xhat).sum((0, 2, 3))
    assert float((ps[0] - w0).abs().max()) <= 2e-3 * float(w0.abs().max() + 1)
    assert float((ps[1] - w1).abs().max()) <= 2e-3 * float(w1.abs().max() + 1)


@pytest.mark.parametrize("n,c,cout,h,w", [
    (2, 64, 64, 8, 32),      # one patch per image, one chunk, the 64-cout N tile (layer1's class)
    (3, 128, 128, 16, 64),   # 2 x 2 patches per image, two chunks: halo rows from the neighbouring patches, image borders
    (2, 192, 256, 24, 32),   # three chunks (odd count: both halo buffers end the loop), two N tiles
    (1, 64, 320, 8, 64),     # ragged last N tile
])
def test_halo_3x3_all_epilogues_vs_fp32_reference(n, c, cout, h, w):
    """3x3 / stride 1 / pad 1 through the halo-once form (Bottleneck.conv2 of layer1-3 in the bf16 train step and its data
    gradient): eval epilogue (scale / shift + residual + ReLU; ReLU mask), the train-mode statistics epilogue, the
    data-gradient-into-BatchNorm epilogue (bits mask + the two backward reductions) -- against plain PyTorch fp32 on the same
    bf16 operands, and against the implicit-GEMM kernel forced on the same launches."""
    from robosat_amd import ops

    x = prep(rnd(n, c, h, w, seed=61), BF)
    wt = prep(rnd(cout, c, 3, 3, seed=62) * (2.0 / (c * 9)) ** 0.5, BF)
    sc, sh = rnd(cout, seed=63).abs() + 0.5, rnd(cout, seed=64)
    res, mask = prep(rnd(n, cout, h, w, seed=65), BF), prep(rnd(n, cout, h, w, seed=66), BF)
    base = F.conv2d(x, wt, padding=1)
    xd, wd = nhwc(x, BF), krsc(wt, BF)
    with ops.tuning("halo"):
        d = ops.conv_desc(xd, wd, pad=1)
        assert _halo_name(d) == "conv_halo_bf16<3x3,256x{}>".format(128 if cout % 128 == 0 or cout > 128 else 64)
        got = ops.conv2d(xd, wd, pad=1, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res, BF), relu=True)
        close(nchw(got), F.relu(base * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res), BF, "relu")
        gotm = ops.conv2d(xd, wd, pad=1, relu_mask=nhwc(mask, BF))
        close(nchw(gotm), base * (mask > 0), BF, "mask")
        plain = ops.conv2d(xd, wd, pad=1)
        if cout % 64 == 0 and cout != 320:
            y, partial = ops.conv2d_bnstats(xd, wd, pad=1)
            assert partial.shape[0] == n * (h // 8) * (w // 32)  # one partial row per 8 x 32 patch
            close(nchw(y), base, BF, "bnstats y")
            yf = y.float()
            s = partial.sum(0).cpu()
            want0, want1 = yf.sum((0, 1, 2)).cpu(), (yf * yf).sum((0, 1, 2)).cpu()
            assert float((s[0] - want0).abs().max()) <= 1e-3 * float(want0.abs().max() + 1)
            assert float((s[1] - want1).abs().max()) <= 1e-3 * float(want1.abs().max() + 1)
            # data gradient into a BatchNorm: g = (conv(dy) + residual) * mask bits, partial sums of g and g * xhat
            bn_y = prep(rnd(n, cout, h, w, seed=67), BF)
            mean, invstd = rnd(cout, seed=68) * 0.1, rnd(cout, seed=69).abs() + 0.5
            z = nhwc(mask, BF)
            bits = torch.zeros(z.numel() // 8, dtype=torch.uint8, device=DEV)
            zb = (z.reshape(-1, 8) > 0).to(torch.uint8)
            for e in range(8):
                bits |= zb[:, e] << e
            g, part = ops.conv2d_dgrad_bnstats(xd, wd, (h, w), nhwc(bn_y, BF), mean.to(DEV), invstd.to(DEV), pad=1,
                                               residual=nhwc(res, BF), relu_mask_bits=bits)
            want_g = (base + res) * (mask > 0)
            close(nchw(g), want_g, BF, "dgrad-into-bn g")
            gf = nchw(g)
            xhat = (bn_y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
            ps = part.sum(0).cpu()
            w0, w1 = gf.sum((0, 2, 3)), (gf * xhat).sum((0, 2, 3))
            assert float((ps[0] - w0).abs().max()) <= 2e-3 * float(w0.abs().max() + 1)
            assert float((ps[1] - w1).abs().max()) <= 2e-3 * float(w1.abs().max() + 1)
    with ops.tuning("128x128" if cout % 128 == 0 or cout > 128 else "128x64", 128):  # the implicit-GEMM kernel: same values up to one bf16 rounding
        ref = ops.conv2d(xd, wd, pad=1)
    assert float((plain.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())


@pytest.mark.parametrize("n,c1,c2,cout,hs,ws", [
    (2, 64, 0, 64, 8, 32),       # one patch, one chunk, single source, 64-cout tile (dec2's class)
    (2, 128, 64, 128, 16, 32),   # two sources (three chunks), two patches per image: halo rows across patches + borders (dec3's class)
    (1, 256, 64, 256, 8, 64),    # five chunks, two N tiles, two patches side by side (dec1's class)
])
def test_halo_phase_form_and_its_gradient_vs_autograd(n, c1, c2, cout, hs, ws):
    """DecoderBlock (unet.py:63-73) in phase form through the halo-once kernel (one output parity per block, the 2x2 taps
    from one source halo) and its 4x4 / stride-2 data gradient as four parity-plane 2x2 convolutions, with torch.cat's
    backward fused into the store -- against autograd on the reference formulation and against the implicit-GEMM kernel."""
    from robosat_amd import _lib, ops

    a = prep(rnd(n, c1, hs, ws, seed=71), BF).requires_grad_(True)
    b = prep(rnd(n, c2, hs, ws, seed=72), BF).requires_grad_(True) if c2 else None
    src = torch.cat([a, b], 1) if c2 else a
    wt = rnd(cout, c1 + c2, 3, 3, seed=73) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    y = F.relu(F.conv2d(F.interpolate(src, scale_factor=2, mode="nearest"), wt, padding=1))
    gy = prep(rnd(*y.shape, seed=74), BF)
    y.backward(gy)
    w_krsc = krsc(wt, torch.float32)
    ad = nhwc(a.detach(), BF)
    bd = nhwc(b.detach(), BF) if c2 else None
    wp = ops.pack_phase_weight(w_krsc, BF)
    dz = gy * (y.detach() > 0)
    dzd = nhwc(dz, BF)
    wd = ops.pack_dgrad_phase_weight(w_krsc, BF)  # [Cin, 4, 4, Cout]: the gradient's "Cout" is c1 + c2
    want = torch.cat([a.grad, b.grad], 1) if c2 else a.grad
    bn_f = 128 if cout % 128 == 0 else 64
    cg = c1 + c2
    bn_g = 128 if (cg % 128 == 0 or (cg > 128 and -(-cg // 128) * 128 * 4 <= cg * 5)) else 64
    with ops.tuning("halo"):
        d = _lib.ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, 1, 0)
        assert _halo_name(d, phase=True) == "conv_halo_bf16<phase,256x{}>".format(bn_f)
        got = ops.conv2d_phase(ad, wp, src2=bd, relu=True)
        close(nchw(got), y.detach(), BF, "phase fwd")
        dd = ops.conv_desc(dzd, wd, stride=2, pad=1, out_hw=(hs, ws))
        assert _halo_name(dd) == "conv_halo_bf16<dgrad4x4,256x{}>".format(bn_g)
        dsrc = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws), alg_scale=2.25)
        close(nchw(dsrc), want, BF, "dgrad4x4")
        m1 = prep(rnd(n, c1, hs, ws, seed=75), BF)
        if c2 and c1 % bn_g == 0:
            m2 = prep(rnd(n, c2, hs, ws, seed=76), BF)
            d1, d2 = ops.conv2d_split(dzd, wd, c1, stride=2, pad=1, out_hw=(hs, ws), mask1=nhwc(m1, BF), mask2=nhwc(m2, BF), alg_scale=2.25)
            close(nchw(d1), a.grad * (m1 > 0), BF, "split d1")
            close(nchw(d2), b.grad * (m2 > 0), BF, "split d2")
        elif not c2:
            dm = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws), relu_mask=nhwc(m1, BF), alg_scale=2.25)
            close(nchw(dm), a.grad * (m1 > 0), BF, "dgrad4x4 + mask")
    with ops.tuning("128x128" if bn_f == 128 else "128x64", 128):
        refp = ops.conv2d_phase(ad, wp, src2=bd, relu=True)
    with ops.tuning("128x128" if bn_g == 128 else "128x64", 128):
        refg = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws))
    assert float((got.float() - refp.float()).abs().max()) <= 2 ** -7 * float(refp.float().abs().max())
    assert float((dsrc.float() - refg.float()).abs().max()) <= 2 ** -7 * float(refg.float().abs().max())


@pytest.mark.parametrize("n,c,cout,h,w", [
    (2, 96, 128, 16, 32),    # one 16 x 32 patch per image, three 32-channel chunks (96 is no multiple of 64: only this form can run it)
    (2, 128, 256, 32, 64),   # 2 x 2 patches, four chunks, two N tiles
])
def test_halo512_3x3_epilogues_vs_fp32_reference(n, c, cout, h, w):
    """The 512-pixel-patch variant (16 x 32 pixels, 32-channel chunks, wave tiles of 128 x 64) of the 3x3 halo form: eval
    epilogue, train-mode statistics, against plain PyTorch fp32 on the same bf16 operands."""
    from robosat_amd import ops

    x = prep(rnd(n, c, h, w, seed=81), BF)
    wt = prep(rnd(cout, c, 3, 3, seed=82) * (2.0 / (c * 9)) ** 0.5, BF)
    res, mask = prep(rnd(n, cout, h, w, seed=85), BF), prep(rnd(n, cout, h, w, seed=86), BF)
    base = F.conv2d(x, wt, padding=1)
    xd, wd = nhwc(x, BF), krsc(wt, BF)
    with ops.tuning("halo", 64):
        assert _halo_name(ops.conv_desc(xd, wd, pad=1)) == "conv_halo_bf16<3x3,512x128>"
        got = ops.conv2d(xd, wd, pad=1, residual=nhwc(res, BF), relu=True)
        close(nchw(got), F.relu(base + res), BF, "relu")
        gotm = ops.conv2d(xd, wd, pad=1, relu_mask=nhwc(mask, BF))
        close(nchw(gotm), base * (mask > 0), BF, "mask")
        y, partial = ops.conv2d_bnstats(xd, wd, pad=1)
        assert partial.shape[0] == n * (h // 16) * (w // 32)
        close(nchw(y), base, BF, "bnstats y")
        yf = y.float()
        s = partial.sum(0).cpu()
        want0, want1 = yf.sum((0, 1, 2)).cpu(), (yf * yf).sum((0, 1, 2)).cpu()
        assert float((s[0] - want0).abs().max()) <= 1e-3 * float(want0.abs().max() + 1)
        assert float((s[1] - want1).abs().max()) <= 1e-3 * float(want1.abs().max() + 1)


@pytest.mark.parametrize("n,c1,c2,cout,hs,ws", [
    (2, 128, 64, 128, 16, 32),   # dec3's class: two sources (4 + 2 chunks of 32 channels), one patch per image
    (1, 64, 0, 128, 32, 64),     # single source, 2 x 2 patches
])
def test_halo512_phase_form_and_its_gradient_vs_autograd(n, c1, c2, cout, hs, ws):
    """DecoderBlock phase form and its 4x4 / stride-2 data gradient on the 512-pixel-patch variant, against autograd."""
    from robosat_amd import _lib, ops

    a = prep(rnd(n, c1, hs, ws, seed=91), BF).requires_grad_(True)
    b = prep(rnd(n, c2, hs, ws, seed=92), BF).requires_grad_(True) if c2 else None
    src = torch.cat([a, b], 1) if c2 else a
    wt = rnd(cout, c1 + c2, 3, 3, seed=93) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    y = F.relu(F.conv2d(F.interpolate(src, scale_factor=2, mode="nearest"), wt, padding=1))
    gy = prep(rnd(*y.shape, seed=94), BF)
    y.backward(gy)
    w_krsc = krsc(wt, torch.float32)
    ad = nhwc(a.detach(), BF)
    bd = nhwc(b.detach(), BF) if c2 else None
    dzd = nhwc(gy * (y.detach() > 0), BF)
    wd = ops.pack_dgrad_phase_weight(w_krsc, BF)
    want = torch.cat([a.grad, b.grad], 1) if c2 else a.grad
    with ops.tuning("halo", 64):
        d = _lib.ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, 1, 0)
        assert _halo_name(d, phase=True) == "conv_halo_bf16<phase,512x128>"
        got = ops.conv2d_phase(ad, ops.pack_phase_weight(w_krsc, BF), src2=bd, relu=True)
        close(nchw(got), y.detach(), BF, "phase fwd")
        cg = c1 + c2
        if cg % 128 == 0 or (cg > 128 and -(-cg // 128) * 128 * 4 <= cg * 5):
            assert _halo_name(ops.conv_desc(dzd, wd, stride=2, pad=1, out_hw=(hs, ws))) == "conv_halo_bf16<dgrad4x4,512x128>"
        dsrc = ops.conv2d(dzd, wd, stride=2, pad=1, out_hw=(hs, ws), alg_scale=2.25)
        close(nchw(dsrc), want, BF, "dgrad4x4")


def test_halo_forms_are_what_the_bf16_train_step_runs_unforced():
    """At the benchmark's sizes (bs 32, 512^2) the dispatcher itself takes the halo forms for the layers they can tile:
    checked on one image-strided sample per layer against fp32 PyTorch (layer2's conv2 and dec3's DecoderBlock); and the
    layers where the implicit-GEMM tiles measured faster keep them (profiles/r04/halo_sweep_v2.txt)."""
    from robosat_amd import _lib, ops

    g = torch.Generator(device=DEV).manual_seed(22)
    n, c, h, w = 32, 128, 64, 64
    xd = torch.randn(n, h, w, c, device=DEV, generator=g).to(BF)
    wd = (torch.randn(c, 3, 3, c, device=DEV, generator=g) * 0.03).to(BF)
    assert ops.conv_tile_name(ops.conv_desc(xd, wd, pad=1), True) == "conv_halo_bf16<3x3,512x128>"  # (256 patches of 16 x 32: one per CU)
    got = ops.conv2d(xd, wd, pad=1)
    for img in (0, 17, n - 1):
        x = xd[img:img + 1].float().permute(0, 3, 1, 2).cpu()
        close(nchw(got[img:img + 1]), F.conv2d(x, wd.float().permute(0, 3, 1, 2).cpu(), padding=1), BF, "layer2 conv2 image {}".format(img))
    n, c1, c2, cout, hs, ws = 4, 256, 64, 128, 128, 128  # dec3's geometry, 4 images (16 384 patches-parities at bs 32; 2 048 here)
    a = torch.randn(n, hs, ws, c1, device=DEV, generator=g).to(BF)
    b = torch.randn(n, hs, ws, c2, device=DEV, generator=g).to(BF)
    wt = torch.randn(cout, c1 + c2, 3, 3, generator=torch.Generator().manual_seed(23)) * 0.02
    d = _lib.ConvDesc(n, hs, ws, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * ws, cout, 1, 0)
    assert ops.conv_tile_name(d, True, phase=True) == "conv_halo_bf16<phase,256x128>"  # (4 images: 128 patches of 16 x 32 < 256 -> the 8 x 32 patch)
    got = ops.conv2d_phase(a, ops.pack_phase_weight(krsc(wt, torch.float32), BF), src2=b, relu=True)
    src = torch.cat([a[:1], b[:1]], 3).float().permute(0, 3, 1, 2).cpu()
    want = F.relu(F.conv2d(F.interpolate(src, scale_factor=2, mode="nearest"), wt, padding=1))
    close(nchw(got[:1]), want, BF, "dec3 image 0")


def test_bench_symbols_are_covered(tmp_path):
    """Run the real benchmark (BASELINE configs[1] + the configs[2] train leg, 1 timed step each) and require every
    convolution symbol it reports to be one the tests above exercise; also that its in-line parity check ran."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    full = str(tmp_path / "bench_full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--train-steps", "2",
                        "--no-miou", "--full-json", full],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # stdout: the compact line (what the driver's 8 KB tail must hold whole) -- the train leg's numbers included
    compact = r.stdout.strip().splitlines()[-1]
    assert len(compact) <= 4096, len(compact)
    brief = json.loads(compact)
    assert brief["train"]["value"] > 0 and brief["train"]["ms_per_step"] > 0 and "frac" in brief["train"]["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(brief["roofline"])
    with open(full) as fp:  # the full record: per-kernel tables, every step time
        line = json.load(fp)
    assert line["train"]["value"] == brief["train"]["value"] and line["value"] == brief["value"]
    names = (set(line["roofline"]["per_kernel"]) | set(line["train"]["roofline"]["per_kernel"]) |
             set(line["legs"]["predict_fp32_bs32"]["roofline"]["per_kernel"]))
    assert names, line
    missing = sorted(n for n in names if n not in COVERED)
    assert not missing, missing
    assert line["parity"]["max_abs_vs_oracle"] <= 1e-3
    assert line["train"]["parity"]["max_abs_vs_oracle"] <= 5e-2
    # the other BASELINE configurations ride in the same line (configs[4]: 4 bands, 4 classes; fp32 training; configs[3])
    assert set(line["legs"]) == {"predict_fp32_bs32", "cfg5_train_bf16_4band_4class", "train_fp32_bs8", "cfg4_predict_fp32_1024_bs8"}
    # the north_star's "MFMA roofline on 3x3 conv" as one number per leg, and at bs 32
    for roof in (line["roofline"], line["train"]["roofline"], line["legs"]["predict_fp32_bs32"]["roofline"]):
        assert 0.0 < roof["conv3x3"]["frac"] <= 1.0 and roof["conv3x3"]["launches"] > 0
    assert 0.0 < brief["legs"]["predict_fp32_bs32"]["conv3x3"]["frac"] <= 1.0 and "conv3x3" in brief["roofline"]
    # roofline.traffic is printed only when the committed counter tables were taken on THESE kernel sources
    src = line["roofline"]["traffic_source"]
    assert (line["roofline"]["traffic"] is None) == bool(src["stale"] or line["roofline"]["kernel"] not in _pmc_names())
    assert all(leg["value"] > 0 for leg in line["legs"].values())
    assert line["legs"]["cfg5_train_bf16_4band_4class"]["config"]["bands"] == 4


def _pmc_names():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fp:
        return set(json.load(fp))


# ---- fp32 Winograd form of DecoderBlock (conv_wino_f32.hip) --------------------------------------------------------------------
WINO = {"conv_wino_f32<phase,p8,64x64>", "conv_wino_f32<phase,p8,128x64>", "conv_wino_f32<phase,p8,128x32>",
        "conv_wino_f32<phase,p4,64x64>", "conv_wino_f32<phase,p4,128x32>"}
COVERED |= WINO


@pytest.mark.parametrize("n,c1,c2,cout,h,w,want", [
    (2, 64, 32, 64, 16, 16, "p8,64x64"),     # one 8x8 tile patch per image, two sources
    (3, 48, 0, 128, 20, 36, "p8,64x64"),     # ragged patches (10 x 18 tiles), single source, 16-channel chunks that are not 32-multiples
    (2, 128, 0, 32, 32, 32, "p8,128x32"),    # dec4's shape class: 32 couts, two patches per block
    (5, 32, 32, 96, 8, 8, "p4,128x32"),      # 4x4-tile patches, 8 per block, blocks that straddle images, Cout % 64 != 0
    (3, 64, 0, 64, 9, 13, "p4,64x64"),       # odd sizes: ragged last tile row / column (positions past the image), center's class
    (1, 32, 16, 64, 64, 48, "p8,64x64"),     # many patches per image
    (8, 32, 0, 128, 64, 60, "p8,128x64"),    # enough work items for the 128-tile block (two patches per block, ragged last patch row)
])
def test_winograd_phase_form_vs_fp32_reference(n, c1, c2, cout, h, w, want):
    """relu(conv3x3(interpolate(cat[a, b], x2), pad 1)) through the Winograd F(2x2, 2x2) kernel against plain PyTorch fp32 --
    and against the generic phase kernel on the same launch; the transforms have 0 / +-1 coefficients, so the two agree to
    fp32 summation-order noise."""
    from robosat_amd import ops

    a = rnd(n, c1, h, w, seed=51)
    b = rnd(n, c2, h, w, seed=52) if c2 else None
    wt = rnd(cout, c1 + c2, 3, 3, seed=53) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    x = a if b is None else torch.cat([a, b], 1)
    ref = F.relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, padding=1))
    s1, s2 = nhwc(a, torch.float32), (nhwc(b, torch.float32) if b is not None else None)
    wp = ops.pack_phase_weight(krsc(wt, torch.float32), torch.float32)
    assert ops.wino_ok(s1, s2, cout, force=True)
    ops.PROFILE = []
    try:
        got = ops.conv2d_phase_wino(s1, ops.pack_wino_phase_weight(wp), src2=s2, relu=True)
        torch.cuda.synchronize()
        name = ops.PROFILE[0][0]
    finally:
        ops.PROFILE = None
    assert name == "conv_wino_f32<phase,{}>".format(want) and name in COVERED, name
    close(nchw(got), ref, torch.float32, "winograd vs fp32 reference")
    if c1 % 32 == 0 and c2 % 32 == 0:  # (the generic kernel's channel granularity)
        generic = ops.conv2d_phase(s1, wp, src2=s2, relu=True)
        assert float((got - generic).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # without the ReLU (negative outputs survive) and on a second call (no state left in the kernel)
    got2 = ops.conv2d_phase_wino(s1, ops.pack_wino_phase_weight(wp), src2=s2, relu=False)
    ref2 = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, padding=1)
    close(nchw(got2), ref2, torch.float32, "winograd, no relu")
    if want == "p8,128x64":  # the two block shapes of a 64-cout layer accumulate every output in the same order: bit-identical
        small = ops.conv2d_phase_wino(s1[:1], ops.pack_wino_phase_weight(wp), src2=None if s2 is None else s2[:1], relu=True)
        assert torch.equal(small[0], got[0])  # (N = 1: too few work items for the wide block -> the 64 x 64 one ran)


def test_winograd_declines_what_it_cannot_run():
    from robosat_amd import ops

    tiny = torch.zeros(1, 4, 4, 64, device=DEV)  # 2 x 2 tiles per image: the generic phase kernel's job
    assert not ops.wino_ok(tiny, None, 64, force=True)
    assert not ops.wino_ok(torch.zeros(1, 16, 16, 64, device=DEV, dtype=BF), None, 64, force=True)  # bf16 keeps the phase form
    small = torch.zeros(16, 8, 8, 64, device=DEV)  # runnable (4x4-tile patches), but left to the generic kernel: < 8 tiles per side
    assert ops.wino_ok(small, None, 64, force=True) and not ops.wino_ok(small, None, 64)
    # the choice is the layer's geometry, never the batch size: a tile's output must not depend on its batch neighbours
    assert ops.wino_ok(torch.zeros(1, 16, 16, 64, device=DEV), None, 64) and ops.wino_ok(torch.zeros(16, 16, 16, 64, device=DEV), None, 64)


# ---- fp32 Winograd F(2x2, 3x3): the eval-mode stride-1 3x3 convolutions (conv_wino33_f32.hip) ---------------------------------------
WINO33 = {"conv_wino_f32<3x3,p8,64x32>", "conv_wino_f32<3x3,p8,128x16>"}
COVERED |= WINO33


@pytest.mark.parametrize("n,cin,cout,h,w,want,epi", [
    (2, 64, 64, 16, 16, "64x32", "bn"),      # one 8x8 patch of tiles per image, folded BatchNorm + ReLU (Bottleneck.conv2)
    (3, 48, 96, 20, 36, "64x32", "bn"),      # ragged patches (10 x 18 tiles), 16-channel chunks that are not 32-multiples
    (2, 32, 32, 40, 24, "64x32", "relu"),    # dec5's shape class: two chunks, ReLU only
    (5, 64, 48, 15, 17, "128x16", "none"),   # odd sizes (last tile row / column past the image), Cout % 32 != 0, blocks that straddle images
    (1, 128, 64, 64, 48, "64x32", "bn"),     # many patches per image
])
def test_winograd_3x3_vs_fp32_reference(n, cin, cout, h, w, want, epi):
    """relu(conv3x3(x, pad 1) * scale + shift) through the Winograd F(2x2, 3x3) kernel against plain PyTorch fp32 and against
    the generic implicit-GEMM kernel on the same launch (2e-4 of the output scale, the generic kernel's own bar)."""
    from robosat_amd import ops

    x = rnd(n, cin, h, w, seed=61)
    wt = rnd(cout, cin, 3, 3, seed=62) * (2.0 / (cin * 9)) ** 0.5
    sc = (rnd(cout, seed=63) * 0.2 + 1.0) if epi == "bn" else None
    sh = rnd(cout, seed=64) * 0.3 if epi == "bn" else None
    ref = F.conv2d(x, wt, padding=1)
    if epi == "bn":
        ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if epi != "none":
        ref = F.relu(ref)
    s = nhwc(x, torch.float32)
    wk = krsc(wt, torch.float32)
    assert ops.wino33_ok(s, cout)
    dsc, dsh = (sc.to(DEV), sh.to(DEV)) if epi == "bn" else (None, None)
    ops.PROFILE = []
    try:
        got = ops.conv2d_wino33(s, ops.pack_wino33_weight(wk), scale=dsc, shift=dsh, relu=epi != "none")
        torch.cuda.synchronize()
        name = ops.PROFILE[0][0]
    finally:
        ops.PROFILE = None
    assert name == "conv_wino_f32<3x3,p8,{}>".format(want) and name in COVERED, name
    close(nchw(got), ref, torch.float32, "winograd 3x3 vs fp32 reference")
    if cin % 32 == 0 and cout % 32 == 0:
        generic = ops.conv2d(s, wk, pad=1, scale=dsc, shift=dsh, relu=epi != "none")
        assert float((got - generic).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))


COVERED |= {"conv_wino_f32<3x3+final,p8,64x32>"}


@pytest.mark.parametrize("n,h,w,classes", [(2, 32, 48, 2), (3, 17, 31, 4), (1, 64, 64, 5), (2, 40, 24, 8)])
def test_fused_dec5_final_head_vs_two_launches(n, h, w, classes):
    """dec5 + final (+ softmax / quantise / argmax) in one launch (conv_wino33_f32<.., HEAD>) against the two-launch form
    (Winograd dec5, then final_conv1x1*) and against plain PyTorch fp32: reference unet.py:139-141, tools/predict.py:87-103."""
    from robosat_amd import ops

    x = rnd(n, 32, h, w, seed=71)
    wt = rnd(32, 32, 3, 3, seed=72) * (2.0 / (32 * 9)) ** 0.5
    wf = rnd(classes, 32, seed=73) * 0.5
    bf = rnd(classes, seed=74) * 0.2
    s, u = nhwc(x, torch.float32), ops.pack_wino33_weight(krsc(wt, torch.float32))
    dwf, dbf = wf.to(DEV).contiguous(), bf.to(DEV).contiguous()
    assert ops.wino33_head_ok(s, 32, classes)
    ref_logits = F.conv2d(F.relu(F.conv2d(x, wt, padding=1)), wf.view(classes, 32, 1, 1), bf)
    dec5 = ops.conv2d_wino33(s, u, relu=True)
    ops.PROFILE = []
    try:
        logits = ops.conv2d_wino33_head(s, u, dwf, dbf, "logits")
        torch.cuda.synchronize()
        name = ops.PROFILE[0][0]
    finally:
        ops.PROFILE = None
    assert name == "conv_wino_f32<3x3+final,p8,64x32>" and name in COVERED
    two = ops.final_conv1x1(dec5, dwf, dbf, softmax=False)
    assert logits.shape == two.shape == (n, classes, h, w)
    scale = max(1.0, float(ref_logits.abs().max()))
    assert float((logits.cpu() - ref_logits).abs().max()) <= 2e-5 * scale
    assert float((logits - two).abs().max()) <= 5e-6 * scale  # (summation order over the 32 channels)
    probs = ops.conv2d_wino33_head(s, u, dwf, dbf, "softmax")
    assert float((probs - ops.final_conv1x1(dec5, dwf, dbf, softmax=True)).abs().max()) <= 5e-6
    assert float((probs.cpu() - torch.softmax(ref_logits, 1)).abs().max()) <= 2e-5
    # argmax: equal wherever the two best logits are not within rounding of each other
    am, am2 = ops.conv2d_wino33_head(s, u, dwf, dbf, "argmax"), ops.final_conv1x1_argmax(dec5, dwf, dbf)
    top2 = torch.topk(two, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * scale
    assert am.shape == (n, h, w) and bool((am == am2)[clear].all()) and float(clear.float().mean()) > 0.99
    # quantised probabilities of the crop: the same bytes except where a probability sits within rounding of a bin edge
    for ov in (0, 3):
        q, q2 = ops.conv2d_wino33_head(s, u, dwf, dbf, "quantize", overlap=ov), ops.final_conv1x1_quantize(dec5, dwf, dbf, ov)
        assert q.shape == q2.shape and q.dtype == torch.uint8
        diff = (q.int() - q2.int()).abs()
        diff = torch.minimum(diff, 256 - diff)  # (bin 256 wraps to 0)
        assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) <= 2e-3


def test_fused_head_is_what_the_fp32_predict_pass_runs():
    """UNet.eval() in fp32 ends in the fused launch (and ROBOSAT_FUSED_HEAD=0 restores the two launches): the probabilities of
    the two forms agree to fp32 rounding."""
    import os
    from robosat_amd import ops
    from robosat_amd.unet import UNet

    torch.manual_seed(3)
    net = UNet(3, pretrained=False).to(DEV).eval()
    x = torch.randn(2, 3, 128, 192, device=DEV)
    ops.PROFILE = []
    try:
        with torch.no_grad():
            fused = net.predict_probs(x)
        torch.cuda.synchronize()
        names = [r[0] for r in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert names[-1] == "conv_wino_f32<3x3+final,p8,64x32>", names[-3:]
    os.environ["ROBOSAT_FUSED_HEAD"] = "0"
    try:
        with torch.no_grad():
            two = net.predict_probs(x)
    finally:
        del os.environ["ROBOSAT_FUSED_HEAD"]
    assert fused.shape == two.shape and float((fused - two).abs().max()) <= 5e-6


def test_winograd_3x3_declines_what_it_cannot_run():
    from robosat_amd import ops

    assert not ops.wino33_ok(torch.zeros(4, 8, 8, 64, device=DEV), 64)          # < 8 tiles per side
    assert not ops.wino33_ok(torch.zeros(4, 32, 32, 16, device=DEV), 64)        # one chunk only
    assert not ops.wino33_ok(torch.zeros(4, 32, 32, 64, device=DEV, dtype=BF), 64)
    assert ops.wino33_ok(torch.zeros(1, 16, 16, 64, device=DEV), 64) and ops.wino33_ok(torch.zeros(16, 16, 16, 64, device=DEV), 64)


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,cin,cout,res", [(2, 16, 16, 64, 256, True), (3, 17, 13, 32, 64, False), (2, 32, 32, 512, 128, True)])
def test_epilogue_wave_1x1_kernel_is_bit_identical_to_the_generic_one(n, h, w, cin, cout, res):
    """conv1x1_ew_f32 (main loop and epilogue of an fp32 1x1 launch on separate waves of a persistent block; dispatched by
    rule for K <= 64, knob ``conv1x1_ew``) computes the generic kernel's K order and epilogue arithmetic: the same bits,
    ragged last pixel tile included (reference: Bottleneck conv1 / conv3 with the folded eval-mode BatchNorm,
    unet.py:94,122-130).  Also: the rule is geometry only (K <= 64 takes it, K = 512 does not) and the report name follows."""
    from robosat_amd import ops

    x = rnd(n, cin, h, w, seed=91)
    wt = rnd(cout, cin, 1, 1, seed=92) * (1.0 / cin) ** 0.5
    sc, sh = (torch.rand(cout) + 0.5).to(DEV), (rnd(cout, seed=93) * 0.1).to(DEV)
    r = nhwc(rnd(n, cout, h, w, seed=94), torch.float32) if res else None
    s, wk = nhwc(x, torch.float32), krsc(wt, torch.float32)
    d = ops.conv_desc(s, wk, pad=0)
    assert ops.get_knob("conv1x1_ew") == -1  # the shipped setting: by rule
    assert (ops.conv_tile_name(d) == "conv1x1_ew_f32<128x64,r64>") == (cin <= 64)
    with ops.knob("conv1x1_ew", 0):
        assert ops.conv_tile_name(d).startswith("conv_igemm_f32<")
        generic = ops.conv2d(s, wk, pad=0, scale=sc, shift=sh, residual=r, relu=True)
    with ops.knob("conv1x1_ew", 1):
        assert ops.conv_tile_name(d) == "conv1x1_ew_f32<128x64,r64>"
        ew = ops.conv2d(s, wk, pad=0, scale=sc, shift=sh, residual=r, relu=True)
    byrule = ops.conv2d(s, wk, pad=0, scale=sc, shift=sh, residual=r, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(byrule, generic)
    assert torch.equal(ew, generic)
    ref = F.conv2d(x, wt) * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)
    if res:
        ref = ref + r.cpu().permute(0, 3, 1, 2)
    assert float((ew.cpu().permute(0, 3, 1, 2) - F.relu(ref)).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
