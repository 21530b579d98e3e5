"""conv_wino_f32<dgrad4x4,..> (rs_conv2d_dgrad_phase_wino) against the generic 4x4 / stride-2 kernel and against torch autograd of the
reference formulation (unet.py:63-73), then its time against the generic kernel on the DecoderBlocks of the fp32 bs-8 train step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from robosat_amd import ops

DEV = "cuda:0"
torch.manual_seed(0)
bad = 0
for (n, c1, c2, cout, hs, ws, masks) in [(2, 128, 0, 32, 16, 16, 1), (1, 256, 64, 128, 16, 32, 1), (2, 64, 64, 64, 24, 20, 0), (1, 128, 64, 64, 17, 15, 1),
                                         (3, 64, 0, 32, 31, 33, 0), (2, 192, 128, 64, 16, 16, 1)]:
    a = torch.randn(n, c1, hs, ws, requires_grad=True)
    b = torch.randn(n, c2, hs, ws, requires_grad=True) if c2 else None
    wt = torch.randn(cout, c1 + c2, 3, 3) * (2.0 / ((c1 + c2) * 9)) ** 0.5
    src = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(src, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = torch.randn_like(y)
    y.backward(gy)
    m1 = torch.randn(n, c1, hs, ws) if masks else None
    m2 = torch.randn(n, c2, hs, ws) if (masks and c2) else None
    want1 = a.grad * (m1 > 0) if masks else a.grad
    want2 = (b.grad * (m2 > 0) if masks else b.grad) if c2 else None
    nhwc = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(DEV)
    w_krsc = wt.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = ops.pack_dgrad_phase_weight(w_krsc)
    u = ops.pack_wino_dgrad_weight(wd)
    dz = nhwc(gy)
    if not ops.wino_dgrad_ok(n, hs, ws, c1, c2, cout):
        print((n, c1, c2, cout, hs, ws), "not eligible")
        continue
    split = c2 > 0 and c1 % 64 == 0
    if split:
        g1, g2 = ops.conv2d_dgrad_phase_wino(dz, u, c1, c2, mask1=nhwc(m1), mask2=nhwc(m2), split=True)
        r1, r2 = ops.conv2d_split(dz, wd, c1, stride=2, pad=1, out_hw=(hs, ws), mask1=nhwc(m1), mask2=nhwc(m2))
    else:
        mm = nhwc(torch.cat([m1, m2], 1) if (masks and c2) else m1)
        g1, _ = ops.conv2d_dgrad_phase_wino(dz, u, c1, c2, mask1=mm)
        r1 = ops.conv2d(dz, wd, stride=2, pad=1, out_hw=(hs, ws), relu_mask=mm)
        g2 = r2 = None
        want1 = torch.cat([want1, want2], 1) if c2 else want1
        want2 = None
    torch.cuda.synchronize()
    for got, ref, want, nm in ((g1, r1, want1, "d1"), (g2, r2, want2, "d2")):
        if got is None:
            continue
        e_ref = float((got - ref).abs().max()) / max(1e-6, float(ref.abs().max()))
        e_aut = float((got.cpu().permute(0, 3, 1, 2) - want).abs().max()) / max(1e-6, float(want.abs().max()))
        ok = e_ref < 2e-5 and e_aut < 2e-5
        bad += 0 if ok else 1
        print((n, c1, c2, cout, hs, ws), nm, "split" if split else "one", "| vs generic %.2e | vs autograd %.2e" % (e_ref, e_aut), "ok" if ok else "MISMATCH", flush=True)
print("PARITY OK" if bad == 0 else "PARITY FAILED (%d)" % bad, flush=True)


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot = [0.0, 0.0]
for name, n, c1, c2, cout, hs in [("dec0", 8, 2048, 256, 256, 16), ("dec1", 8, 1024, 256, 256, 32), ("dec2", 8, 512, 256, 64, 64), ("dec3", 8, 256, 64, 128, 128), ("dec4", 8, 128, 0, 32, 256)]:
    wd = ops.pack_dgrad_phase_weight(torch.randn(cout, 3, 3, c1 + c2, device=DEV) * 0.02)
    u = ops.pack_wino_dgrad_weight(wd)
    dz = torch.randn(n, 2 * hs, 2 * hs, cout, device=DEV)
    m1 = torch.randn(n, hs, hs, c1, device=DEV)
    m2 = torch.randn(n, hs, hs, c2, device=DEV) if c2 else None
    if c2:
        tg = bench(lambda: ops.conv2d_split(dz, wd, c1, stride=2, pad=1, out_hw=(hs, hs), mask1=None, mask2=m2))
        tw = bench(lambda: ops.conv2d_dgrad_phase_wino(dz, u, c1, c2, mask1=None, mask2=m2, split=True))
    else:
        tg = bench(lambda: ops.conv2d(dz, wd, stride=2, pad=1, out_hw=(hs, hs), relu_mask=m1))
        tw = bench(lambda: ops.conv2d_dgrad_phase_wino(dz, u, c1, 0, mask1=m1))
    gf = 2.0 * n * hs * hs * cout * (c1 + c2) * 16 / 1e9
    tot[0] += tg
    tot[1] += tw
    print("%s data gradient bs 8: generic 4x4/s2 %.3f ms (%.0f TF executed) | winograd %.3f ms (%.0f TF executed) | x%.2f" % (name, tg, gf / tg, tw, gf * 9 / 16 / tw, tg / tw), flush=True)
print("sum: generic %.3f ms, winograd %.3f ms" % tuple(tot))
