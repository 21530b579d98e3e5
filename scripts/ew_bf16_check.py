"""conv1x1_ew_bf16 (RS_CONV1X1_EW_BF16=1: the train-mode bf16 1x1 forward with its epilogue on its own waves) against the
generic kernel: the raw output must be the same bits, the per-tile BatchNorm partial rows the same sums up to fp32
association; then timed on the 1x1 grids of the bs-32 train step (measurement tool; written in round 4, first run pending)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

dev = torch.device("cuda:0")


def run(x, w, ew):
    if ew:
        ops.set_knob("conv1x1_ew_bf16", 1)  # (needs a `make -C robosat_amd/csrc EXP=1` library)
    else:
        ops.set_knob("conv1x1_ew_bf16", 0)
    y, part = ops.conv2d_bnstats(x, w, pad=0)
    torch.cuda.synchronize()
    ops.set_knob("conv1x1_ew_bf16", 0)
    return y, part


ok = True
g = torch.Generator().manual_seed(11)
for (n, h, w_, cin, cout) in [(2, 16, 16, 64, 256), (3, 17, 13, 64, 128), (1, 64, 64, 256, 128), (2, 32, 32, 512, 128),
                              (4, 32, 32, 256, 1024), (1, 8, 8, 2048, 512), (32, 32, 32, 256, 1024)]:
    x = torch.randn(n, h, w_, cin, generator=g).to(dev).to(torch.bfloat16)
    wt = (torch.randn(cout, 1, 1, cin, generator=g) * (1.0 / cin) ** 0.5).to(dev).to(torch.bfloat16)
    y0, p0 = run(x, wt, False)
    y1, p1 = run(x, wt, True)
    y1b, p1b = run(x, wt, True)
    same_out = torch.equal(y0, y1) and torch.equal(y1, y1b) and torch.equal(p1, p1b)
    s0, s1 = p0.double().sum(0), p1.double().sum(0)  # [2][Cout]: whole-tensor sums from either kernel's rows
    rel = float(((s0 - s1).abs() / (s0.abs() + 1e-3)).max())
    rows_equal = p0.shape == p1.shape and float((p0 - p1).abs().max()) <= 1e-3 * float(p0.abs().max())
    good = same_out and rel <= 1e-5 and rows_equal
    ok &= good
    print("{:>28s}  out identical {}  partial rows {} max|d| {:.2e}  total sums rel {:.2e}  {}".format(
        str((n, h, w_, cin, cout)), same_out, tuple(p1.shape), float((p0 - p1).abs().max()) if p0.shape == p1.shape else -1.0, rel,
        "ok" if good else "FAIL"))
print("PARITY", "OK" if ok else "FAILED")

for cin, hw, cout in [(64, 128, 256), (256, 128, 128), (128, 64, 512), (512, 64, 128), (256, 32, 1024), (1024, 32, 256), (512, 16, 2048), (2048, 16, 512)]:
    nb = 6
    xs = [torch.randn(32, hw, hw, cin, device=dev).to(torch.bfloat16) for _ in range(nb)]
    wd = (torch.randn(cout, 1, 1, cin, device=dev) * (1.0 / cin) ** 0.5).to(torch.bfloat16)
    t = {}
    for ew in (False, True, False, True):
        if ew:
            ops.set_knob("conv1x1_ew_bf16", 1)  # (needs a `make -C robosat_amd/csrc EXP=1` library)
        else:
            ops.set_knob("conv1x1_ew_bf16", 0)
        for i in range(3):
            ops.conv2d_bnstats(xs[i % nb], wd, pad=0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(24):
            ops.conv2d_bnstats(xs[i % nb], wd, pad=0)
        e1.record()
        torch.cuda.synchronize()
        t.setdefault(ew, []).append(e0.elapsed_time(e1) / 24 * 1e3)
    ops.set_knob("conv1x1_ew_bf16", 0)
    print("{:>5d} -> {:<5d} at {:>3d}^2 bs 32   generic {:7.1f} {:7.1f} us   ew {:7.1f} {:7.1f} us   x{:.2f}".format(
        cin, cout, hw, t[False][0], t[False][1], t[True][0], t[True][1], min(t[False]) / min(t[True])))
