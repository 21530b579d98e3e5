"""conv1x1_ew_f32 (RS_CONV1X1_EW=1: the fp32 1x1 kernel with its epilogue on its own waves) against the generic kernel and
against PyTorch on the CPU, then timed against the generic kernel on the predict pass's 1x1 layers (measurement tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from robosat_amd import ops

dev = torch.device("cuda:0")


def run(x, w, sc, sh, res, relu, ew):
    if ew:
        ops.set_knob("conv1x1_ew", 1)
    else:
        ops.set_knob("conv1x1_ew", 0)
    y = ops.conv2d(x, w, pad=0, scale=sc, shift=sh, residual=res, relu=relu)
    torch.cuda.synchronize()
    ops.set_knob("conv1x1_ew", -1)
    return y


ok = True
g = torch.Generator().manual_seed(5)
for (n, h, w_, cin, cout) in [(2, 16, 16, 64, 256), (3, 17, 13, 32, 64), (1, 64, 64, 256, 64), (2, 32, 32, 512, 128),
                              (5, 40, 24, 64, 192), (16, 32, 32, 256, 1024), (1, 8, 8, 2048, 512)]:
    for (use_sc, use_res, relu) in [(True, True, True), (False, False, False), (True, False, True)]:
        x = torch.randn(n, cin, h, w_, generator=g)
        wt = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
        sc = (torch.rand(cout, generator=g) + 0.5) if use_sc else None
        sh = torch.randn(cout, generator=g) * 0.1 if use_sc else None
        res = torch.randn(n, cout, h, w_, generator=g) if use_res else None
        ref = F.conv2d(x, wt)
        if use_sc:
            ref = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        if use_res:
            ref = ref + res
        if relu:
            ref = F.relu(ref)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = wt.permute(0, 2, 3, 1).contiguous().to(dev)
        rd = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
        scd, shd = (sc.to(dev), sh.to(dev)) if use_sc else (None, None)
        y0 = run(xd, wd, scd, shd, rd, relu, False)
        y1 = run(xd, wd, scd, shd, rd, relu, True)
        y1b = run(xd, wd, scd, shd, rd, relu, True)
        want = ref.permute(0, 2, 3, 1)
        e_ref = float((y1.cpu() - want).abs().max())
        e_gen = float((y1 - y0).abs().max())
        scale = max(1.0, float(want.abs().max()))
        good = e_ref <= 2e-5 * scale and e_gen <= 2e-6 * scale and torch.equal(y1, y1b) and bool(torch.isfinite(y1).all())
        ok &= good
        print("{:>26s} sc={:d} res={:d} relu={:d}  |ew-torch| {:.2e}  |ew-generic| {:.2e}  |generic-torch| {:.2e}  {}".format(
            str((n, h, w_, cin, cout)), use_sc, use_res, relu, e_ref, e_gen, float((y0.cpu() - want).abs().max()), "ok" if good else "FAIL"))
print("PARITY", "OK" if ok else "FAILED")

# timing: the 1x1 layers of the bs-16 predict pass, inputs rotated past the Infinity Cache
layers = [(64, 128, 256, True), (256, 128, 64, False), (64, 128, 64, False), (256, 128, 128, False), (128, 64, 512, True), (512, 64, 128, False),
          (512, 64, 256, False), (256, 32, 1024, True), (1024, 32, 256, False), (1024, 32, 512, False), (512, 16, 2048, True), (2048, 16, 512, False)]
for cin, hw, cout, use_res in layers:
    nb = 6
    xs = [torch.randn(16, hw, hw, cin, device=dev) for _ in range(nb)]
    rs = [torch.randn(16, hw, hw, cout, device=dev) for _ in range(nb)] if use_res else [None] * nb
    wd = torch.randn(cout, 1, 1, cin, device=dev) * (1.0 / cin) ** 0.5
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    out = torch.empty(16, hw, hw, cout, device=dev)
    t = {}
    for ew in (False, True, False, True):
        if ew:
            ops.set_knob("conv1x1_ew", 1)
        else:
            ops.set_knob("conv1x1_ew", 0)
        for i in range(3):
            ops.conv2d(xs[i % nb], wd, pad=0, scale=sc, shift=sh, residual=rs[i % nb], relu=True, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            ops.conv2d(xs[i % nb], wd, pad=0, scale=sc, shift=sh, residual=rs[i % nb], relu=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        t.setdefault(ew, []).append(e0.elapsed_time(e1) / 24 * 1e3)
    ops.set_knob("conv1x1_ew", -1)
    print("{:>5d} -> {:<5d} at {:>3d}^2 res={:d}   generic {:7.1f} {:7.1f} us   ew {:7.1f} {:7.1f} us   x{:.2f}".format(
        cin, cout, hw, use_res, t[False][0], t[False][1], t[True][0], t[True][1], min(t[False]) / min(t[True])))
