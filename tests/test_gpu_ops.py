"""Per-kernel parity on the MI355X: every C-ABI op against plain PyTorch fp32 on the host CPU (the same ATen ops the
reference path dispatches).  fp32 MFMA is an exact-fp32 fma chain, so only the summation
order differs: tolerance 2e-4 relative to the output scale."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def nhwc(t):  # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().to(_dev())


def nchw(t):  # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def krsc(w):  # OIHW cpu -> KRSC gpu
    return w.permute(0, 2, 3, 1).contiguous().to(_dev())


def close(got, want, tol=2e-4):
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, "max abs err {} (scale {})".format(err, scale)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


CASES = [
    # name,            N, Cin, H,  W,  Cout, k, stride, pad  (tile the C ABI picks)
    ("3x3_c32_tail", 2, 32, 17, 19, 32, 3, 1, 1),  # 128x32, M tail
    ("3x3_c64_small", 2, 64, 16, 16, 64, 3, 1, 1),  # 64x64
    ("3x3_c64_s2", 2, 64, 18, 22, 64, 3, 2, 1),  # stride 2, 64x64
    ("3x3_c64_big", 2, 32, 192, 192, 64, 3, 1, 1),  # 128x64
    ("3x3_c128_big", 1, 32, 256, 256, 128, 3, 1, 1),  # 128x128
    ("1x1_c256", 2, 64, 20, 12, 256, 1, 1, 0),
    ("1x1_s2", 2, 128, 20, 12, 256, 1, 2, 0),
    ("3x3_c96in", 1, 96, 9, 33, 160, 3, 1, 1),  # Cout 160 -> 128x32 tiles
    ("3x3_c320_ragged", 1, 32, 256, 256, 320, 3, 1, 1),  # 128x128 with a ragged third N tile
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_plain(case):
    from robosat_amd import ops

    _, n, cin, h, w, cout, k, stride, pad = case
    x, wt = rnd(n, cin, h, w, seed=1), rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5
    want = F.conv2d(x, wt, stride=stride, padding=pad)
    got = nchw(ops.conv2d(nhwc(x), krsc(wt), stride=stride, pad=pad))
    close(got, want)


def test_conv_epilogue():
    from robosat_amd import ops

    n, cin, h, w, cout = 2, 64, 24, 24, 128
    x, wt = rnd(n, cin, h, w, seed=3), rnd(cout, cin, 3, 3, seed=4) * 0.06
    sc, sh, res = rnd(cout, seed=5), rnd(cout, seed=6), rnd(n, cout, h, w, seed=7)
    want = F.relu(F.conv2d(x, wt, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    got = nchw(ops.conv2d(nhwc(x), krsc(wt), pad=1, scale=sc.to(_dev()), shift=sh.to(_dev()), residual=nhwc(res), relu=True))
    close(got, want)


@pytest.mark.parametrize("c1,c2,cout,h,w", [(64, 32, 64, 10, 14), (32, 0, 32, 16, 16), (256, 64, 128, 8, 8)])
def test_conv_upsample_concat(c1, c2, cout, h, w):
    """DecoderBlock: conv3x3(interpolate(cat[skip, prev], x2 nearest)) + ReLU (reference unet.py:73,134-137)."""
    from robosat_amd import ops

    n = 2
    a = rnd(n, c1, h, w, seed=8)
    b = rnd(n, c2, h, w, seed=9) if c2 else None
    wt = rnd(cout, c1 + c2, 3, 3, seed=10) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    cat = torch.cat([a, b], 1) if c2 else a
    want = F.relu(F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1))
    got = nchw(ops.conv2d(nhwc(a), krsc(wt), src2=nhwc(b) if c2 else None, ups=1, pad=1, relu=True))
    close(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c1,c2,cout,h,w", [(64, 32, 64, 10, 14), (32, 0, 32, 16, 16), (256, 64, 128, 8, 8), (128, 0, 32, 32, 24),
                                            (64, 64, 128, 5, 3)])
def test_conv_phase_form_equals_upsample_conv(c1, c2, cout, h, w, dtype):
    """rs_conv2d_fwd_phase_dt (four parity 2x2 convolutions with pre-summed taps) == conv3x3(interpolate(cat, x2)) + ReLU
    (reference unet.py:73,134-137), including borders and the concat split."""
    from robosat_amd import ops

    n = 2
    bf = dtype == torch.bfloat16
    rq = (lambda t: t.to(dtype).float()) if bf else (lambda t: t)
    a = rq(rnd(n, c1, h, w, seed=8))
    b = rq(rnd(n, c2, h, w, seed=9)) if c2 else None
    wt = rnd(cout, c1 + c2, 3, 3, seed=10) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    cat = torch.cat([a, b], 1) if c2 else a
    want = F.relu(F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1))
    wp = ops.pack_phase_weight(krsc(wt), dtype)
    assert tuple(wp.shape) == (4, cout, 2, 2, c1 + c2)
    # the packed taps are sums of the original ones
    wsum = wt.sum((2, 3))
    assert float((wp.float().sum((2, 3)).cpu() - wsum.unsqueeze(0)).abs().max()) <= (5e-2 if bf else 1e-5) * float(wsum.abs().max())
    got = ops.conv2d_phase(nhwc(a).to(dtype), wp, src2=nhwc(b).to(dtype) if c2 else None, relu=True)
    assert got.dtype == dtype and tuple(got.shape) == (n, 2 * h, 2 * w, cout)
    close(nchw(got.float()), want, 2e-2 if bf else 2e-4)


@pytest.mark.parametrize("k,pad,hin", [(3, 1, 16), (1, 0, 16), (3, 1, 15)])
def test_conv_zero_insert_is_stride2_adjoint(k, pad, hin):
    """ups=2 gather == data gradient of a stride-2 convolution (checked against autograd)."""
    from robosat_amd import ops

    n, cin, cout = 2, 64, 96
    x = rnd(n, cin, hin, hin, seed=11).requires_grad_(True)
    wt = rnd(cout, cin, k, k, seed=12) * 0.05
    y = F.conv2d(x, wt, stride=2, padding=pad)
    gy = rnd(*y.shape, seed=13)
    y.backward(gy)
    # dgrad weights: [Cin][k][k][Cout], taps flipped
    wd = wt.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(_dev())
    got = nchw(ops.conv2d(nhwc(gy), wd, ups=2, pad=k - 1 - pad, out_hw=(hin, hin)))
    close(got, x.grad)


def test_stem():
    from robosat_amd import ops

    for cin in (3, 4):
        x, wt = rnd(2, cin, 64, 96, seed=14), rnd(64, cin, 7, 7, seed=15) * 0.1
        want = F.conv2d(x, wt, stride=2, padding=3)
        x4 = ops.nchw_to_nhwc4(x.to(_dev()))
        assert x4.shape == (2, 64, 96, 4)
        close(x4[..., :cin].cpu(), x.permute(0, 2, 3, 1))
        if cin == 3:
            assert float(x4[..., 3].abs().max()) == 0.0
        got = nchw(ops.conv2d(x4, ops.pack_stem_weight(krsc(wt)), stride=2, pad=3, stem=7))
        close(got, want)


STEM_CASES = [
    # name,         N, bands, H,   W,   what it covers (stem_f32.hip: 128-pixel tiles of one output row, persistent blocks)
    ("rgb_2tiles", 2, 3, 64, 512),     # Wo = 256: two full tiles per row, RGB form (the zero band skipped)
    ("rgb_ragged", 1, 3, 96, 336),     # Wo = 168: a 128-pixel and a 40-pixel tile per row
    ("rgb_narrow", 3, 3, 32, 32),      # Wo = 16: one ragged tile per row, images of 16 rows
    ("rgb_many", 5, 3, 256, 512),      # 1 280 tiles on <= 512 persistent blocks: several tiles per block, runs crossing images
    ("four_bands", 2, 4, 64, 352),     # four live bands (pixel stride 5), Wo = 176
    ("two_bands", 1, 2, 64, 256),      # fewer than three bands ride the RGB form (their missing bands are zeros)
]


@pytest.mark.parametrize("case", STEM_CASES, ids=[c[0] for c in STEM_CASES])
def test_stem_forms(case):
    """resnet.conv1 on the stem kernel against F.conv2d: raw (the train forward) and with the folded-BatchNorm + ReLU epilogue
    (predict); the RGB form must not read the 4th band -- it is filled with garbage here."""
    from robosat_amd import ops

    _, n, bands, h, w = case
    x, wt = rnd(n, bands, h, w, seed=21), rnd(64, bands, 7, 7, seed=22) * 0.1
    want = F.conv2d(x, wt, stride=2, padding=3)
    x4 = ops.nchw_to_nhwc4(x.to(_dev()))
    packed = ops.pack_stem_weight(krsc(wt))
    if bands <= 3:
        x4[..., 3] = 1e30  # (times the packed filter's zeros this would still be finite: the form is told to skip it, and must)
    got = nchw(ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=bands))
    close(got, want)
    sc, sh = rnd(64, seed=23), rnd(64, seed=24)
    got = nchw(ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=bands, scale=sc.to(_dev()), shift=sh.to(_dev()), relu=True))
    close(got, torch.relu(want * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))


def test_stem_is_batch_independent_and_deterministic():
    """A tile's stem output must not depend on its batch neighbours or on which persistent block computed it (bit for bit)."""
    from robosat_amd import ops

    x, wt = rnd(6, 3, 128, 256, seed=25), rnd(64, 3, 7, 7, seed=26) * 0.1
    packed = ops.pack_stem_weight(krsc(wt))
    x4 = ops.nchw_to_nhwc4(x.to(_dev()))
    full = ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=3)
    assert torch.equal(full, ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=3))
    for i in (0, 3, 5):
        assert torch.equal(full[i:i + 1], ops.conv2d(x4[i:i + 1].contiguous(), packed, stride=2, pad=3, stem=7, bands=3))
    # ... and the generic form (four live bands, the 4th all zeros) computes the same sums in another order
    close(nchw(ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=4)), nchw(full), 1e-5)


@pytest.mark.parametrize("n,h,w", [(2, 16, 16), (3, 8, 20), (1, 128, 128)])
def test_bottleneck_tail_vs_torch_and_vs_the_two_launches(n, h, w):
    """rs_bottleneck_tail_f32 (layer1: conv3 -> bn3 -> + identity -> ReLU, then the next block's conv1 -> bn1 -> ReLU, the second
    product fed from the first one's accumulator registers) against plain PyTorch on the host and against the two rs_conv2d_fwd
    launches it replaces."""
    from robosat_amd import ops

    x, idt = rnd(n, 64, h, w, seed=31), rnd(n, 256, h, w, seed=32)
    w3, w1 = rnd(256, 64, 1, 1, seed=33) * 0.15, rnd(64, 256, 1, 1, seed=34) * 0.08
    s3, t3, s1, t1 = rnd(256, seed=35) * 0.3 + 1.0, rnd(256, seed=36) * 0.2, rnd(64, seed=37) * 0.3 + 1.0, rnd(64, seed=38) * 0.2
    want_out = torch.relu(F.conv2d(x, w3) * s3.view(1, -1, 1, 1) + t3.view(1, -1, 1, 1) + idt)
    want_z = torch.relu(F.conv2d(want_out, w1) * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1))
    dev = _dev()
    xg, ig, w3g, w1g = nhwc(x), nhwc(idt), krsc(w3), krsc(w1)
    assert ops.bottleneck_tail_ok(xg, w3g, w1g)
    out, z = ops.bottleneck_tail(xg, w3g, s3.to(dev), t3.to(dev), ig, w1g, s1.to(dev), t1.to(dev))
    close(nchw(out), want_out)
    close(nchw(z), want_z)
    out2 = ops.conv2d(xg, w3g, scale=s3.to(dev), shift=t3.to(dev), residual=ig, relu=True)
    z2 = ops.conv2d(out2, w1g, scale=s1.to(dev), shift=t1.to(dev), relu=True)
    close(out.cpu(), out2.cpu(), 2e-6)  # (same products, same order in stage 1)
    close(z.cpu(), z2.cpu(), 1e-5)      # (stage 2 adds its 256 products in another order)
    # deterministic, and a pixel's result does not depend on its batch neighbours
    o3, z3 = ops.bottleneck_tail(xg, w3g, s3.to(dev), t3.to(dev), ig, w1g, s1.to(dev), t1.to(dev))
    assert torch.equal(out, o3) and torch.equal(z, z3)
    o1, z1 = ops.bottleneck_tail(xg[:1].contiguous(), w3g, s3.to(dev), t3.to(dev), ig[:1].contiguous(), w1g, s1.to(dev), t1.to(dev))
    assert torch.equal(out[:1], o1) and torch.equal(z[:1], z1)


@pytest.mark.parametrize("with_res", [False, True])
def test_conv1x1_wave_is_the_generic_launch(with_res):
    """rs_conv1x1_wave_f32 (the fused tail's first stage alone: layer1's downsample convolution and last conv3) against PyTorch and against
    rs_conv2d_fwd; its `out` equals the chained form's bit for bit (same kernel body)."""
    from robosat_amd import ops

    n, h, w = 2, 24, 16
    x, idt = rnd(n, 64, h, w, seed=41), rnd(n, 256, h, w, seed=42)
    w3, w1 = rnd(256, 64, 1, 1, seed=43) * 0.15, rnd(64, 256, 1, 1, seed=44) * 0.08
    sc, sh = rnd(256, seed=45) * 0.3 + 1.0, rnd(256, seed=46) * 0.2
    dev = _dev()
    want = F.conv2d(x, w3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if with_res:
        want = torch.relu(want + idt)
    xg, w3g, ig = nhwc(x), krsc(w3), nhwc(idt)
    assert ops.conv1x1_wave_ok(xg, w3g)
    got = ops.conv1x1_wave(xg, w3g, sc.to(dev), sh.to(dev), residual=ig if with_res else None, relu=with_res)
    close(nchw(got), want)
    ref = ops.conv2d(xg, w3g, scale=sc.to(dev), shift=sh.to(dev), residual=ig if with_res else None, relu=with_res)
    close(got.cpu(), ref.cpu(), 2e-6)
    if with_res:
        s1, t1 = rnd(64, seed=47) * 0.3 + 1.0, rnd(64, seed=48) * 0.2
        out, _ = ops.bottleneck_tail(xg, w3g, sc.to(dev), sh.to(dev), ig, krsc(w1), s1.to(dev), t1.to(dev))
        assert torch.equal(out, got)


def test_fused_tail_network_equals_unfused_network(monkeypatch):
    """The fp32 eval forward with layer1's fused tails against the same forward with them switched off (ROBOSAT_TAIL_FUSE=0)."""
    from oracle import robosat_ref as R, seeded
    from robosat_amd.unet import UNet

    net = UNet(2, pretrained=False)
    net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 3))
    net = net.to(_dev()).eval()
    x = seeded.synthetic_images(2, 3, 128, 128, 3).to(_dev())
    fused = net.predict_probs(x).cpu()
    monkeypatch.setenv("ROBOSAT_TAIL_FUSE", "0")
    plain = net.predict_probs(x).cpu()
    print("fused vs unfused tails: max |dprob|", float((fused - plain).abs().max()))
    assert float((fused - plain).abs().max()) <= 1e-5


@pytest.mark.parametrize("k,s,p", [(3, 2, 1), (2, 2, 0)])
def test_maxpool(k, s, p):
    from robosat_amd import ops

    x = rnd(2, 64, 18, 22, seed=16)
    x[0, :, 3:6, 3:6] = 0.5  # ties: first maximum in window order must win (as torch)
    want, idx = F.max_pool2d(x, k, s, p, return_indices=True)
    got, amax = ops.maxpool2d(nhwc(x), k, s, p, want_argmax=True)
    assert torch.equal(nchw(got), want)
    # decode our tap index to torch's flat input index
    ho, wo = want.shape[2:]
    oy = torch.arange(ho).view(1, 1, ho, 1)
    ox = torch.arange(wo).view(1, 1, 1, wo)
    tap = amax.permute(0, 3, 1, 2).cpu().long()
    flat = (oy * s - p + tap // k) * x.shape[3] + (ox * s - p + tap % k)
    assert torch.equal(flat, idx)


def test_bn_fold_and_final():
    from robosat_amd import ops

    c = 256
    g, b, m, v = rnd(c, seed=17), rnd(c, seed=18), rnd(c, seed=19), torch.rand(c) + 0.5
    sc, sh = ops.bn_fold(g.to(_dev()), b.to(_dev()), m.to(_dev()), v.to(_dev()), 1e-5)
    want_sc = g / torch.sqrt(v + 1e-5)
    close(sc.cpu(), want_sc, 1e-6)
    close(sh.cpu(), b - m * want_sc, 1e-6)

    for ncls in (2, 4):
        x, w, bias = rnd(2, 32, 32, 40, seed=20), rnd(ncls, 32, 1, 1, seed=21) * 0.3, rnd(ncls, seed=22)
        want = F.conv2d(x, w, bias)
        got = ops.final_conv1x1(nhwc(x), w.view(ncls, 32).to(_dev()), bias.to(_dev()))
        close(got.cpu(), want, 1e-5)
        gotp = ops.final_conv1x1(nhwc(x), w.view(ncls, 32).to(_dev()), bias.to(_dev()), softmax=True)
        close(gotp.cpu(), F.softmax(want, 1), 1e-5)


def test_cpu_tensor_is_an_error():
    from robosat_amd import ops

    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 4, 32), torch.zeros(32, 1, 1, 32))
