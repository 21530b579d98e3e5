"""Winograd vs phase form of the fp32 DecoderBlock layers at the benchmark's shapes (measurement tool).
    python scripts/bench_wino.py [--batch 16] [--size 512]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--wino-only", action="store_true", help="time the Winograd kernels only and print a bit checksum of their outputs (A/B of block shapes)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
s = a.size
layers = [("center", 2048, 0, 256, s // 64), ("dec0", 2048, 256, 256, s // 32), ("dec1", 1024, 256, 256, s // 16),
          ("dec2", 512, 256, 64, s // 8), ("dec3", 256, 64, 128, s // 4), ("dec4", 128, 0, 32, s // 2)]
tot = {"phase": 0.0, "wino": 0.0}
for name, c1, c2, cout, hs in layers:
    x1 = torch.randn(a.batch, hs, hs, c1, device=dev)
    x2 = torch.randn(a.batch, hs, hs, c2, device=dev) if c2 else None
    w = torch.randn(cout, 3, 3, c1 + c2, device=dev) * (2.0 / (9 * (c1 + c2))) ** 0.5
    wp = ops.pack_phase_weight(w)
    u = ops.pack_wino_phase_weight(wp)
    flops = 2.0 * a.batch * cout * (c1 + c2) * 9 * (2 * hs) ** 2
    if a.wino_only:
        fn = lambda: ops.conv2d_phase_wino(x1, u, src2=x2, relu=True)
        if not ops.wino_ok(x1, x2, cout, force=True):
            continue
        ops.PROFILE = []
        out = fn()
        torch.cuda.synchronize()
        kname, ops.PROFILE = ops.PROFILE[0][0], None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        tot["wino"] += ms
        print("{:7s} {:32s} {:7.3f} ms ({:6.1f} TF exec) bits {}".format(name, kname, ms, flops / 4 / ms / 1e9, int(out.view(torch.int32).long().sum())))
        continue
    res = {}
    for kind, fn in (("phase", lambda: ops.conv2d_phase(x1, wp, src2=x2, relu=True)),
                     ("wino", (lambda: ops.conv2d_phase_wino(x1, u, src2=x2, relu=True)) if ops.wino_ok(x1, x2, cout, force=True) else None)):
        if fn is None:
            res[kind] = None
            continue
        for _ in range(2):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        res[kind] = e0.elapsed_time(e1) / a.iters
        tot[kind] += res[kind]
    if res["wino"] is None:
        tot["wino"] += res["phase"]
    err = float((ops.conv2d_phase(x1, wp, src2=x2, relu=True) - ops.conv2d_phase_wino(x1, u, src2=x2, relu=True)).abs().max()) if res["wino"] else float("nan")
    print("{:7s} {:5d}+{:<4d}->{:<4d} @{:<4d} phase {:7.3f} ms ({:6.1f} TF exec)  wino {} | max|diff| {:.2e}".format(
        name, c1, c2, cout, hs, res["phase"], flops * 4 / 9 / res["phase"] / 1e9,
        "{:7.3f} ms ({:6.1f} TF exec, x{:.2f})".format(res["wino"], flops / 4 / res["wino"] / 1e9, res["phase"] / res["wino"]) if res["wino"] else "   n/a", err))
print("sum: phase {:.3f} ms, wino {:.3f} ms".format(tot["phase"], tot["wino"]))

# the eval-mode stride-1 3x3 convolutions: generic implicit GEMM vs Winograd F(2x2, 3x3)
tot = {"gen": 0.0, "wino": 0.0}
for name, c, hs, count in (("layer1.conv2", 64, s // 4, 3), ("layer2.conv2", 128, s // 8, 3), ("layer3.conv2", 256, s // 16, 5),
                           ("layer4.conv2", 512, s // 32, 2), ("dec5", 32, s, 1)):
    x = torch.randn(a.batch, hs, hs, c, device=dev)
    w = torch.randn(c, 3, 3, c, device=dev) * (2.0 / (9 * c)) ** 0.5
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    u = ops.pack_wino33_weight(w)
    flops = 2.0 * a.batch * c * c * 9 * hs * hs
    if a.wino_only:
        fn = lambda: ops.conv2d_wino33(x, u, scale=sc, shift=sh, relu=True)
        if not ops.wino33_ok(x, c):
            continue
        ops.PROFILE = []
        out = fn()
        torch.cuda.synchronize()
        kname, ops.PROFILE = ops.PROFILE[0][0], None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        tot["wino"] += ms * count
        print("{:13s} {:32s} {:7.3f} ms x{} ({:6.1f} TF exec) bits {}".format(name, kname, ms, count, flops * 4 / 9 / ms / 1e9, int(out.view(torch.int32).long().sum())))
        continue
    res = {}
    for kind, fn in (("gen", lambda: ops.conv2d(x, w, pad=1, scale=sc, shift=sh, relu=True)),
                     ("wino", (lambda: ops.conv2d_wino33(x, u, scale=sc, shift=sh, relu=True)) if ops.wino33_ok(x, c) else None)):
        if fn is None:
            res[kind] = None
            continue
        for _ in range(2):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        res[kind] = e0.elapsed_time(e1) / a.iters
        tot[kind] += res[kind] * count
    if res["wino"] is None:
        tot["wino"] += res["gen"] * count
    err = float((ops.conv2d(x, w, pad=1, scale=sc, shift=sh, relu=True) - ops.conv2d_wino33(x, u, scale=sc, shift=sh, relu=True)).abs().max()) if res["wino"] else float("nan")
    print("{:13s} {:4d}->{:<4d} @{:<4d} x{}  generic {:7.3f} ms ({:6.1f} TF)  wino {} | max|diff| {:.2e}".format(
        name, c, c, hs, count, res["gen"], flops / res["gen"] / 1e9,
        "{:7.3f} ms ({:6.1f} TF exec, x{:.2f})".format(res["wino"], flops * 4 / 9 / res["wino"] / 1e9, res["gen"] / res["wino"]) if res["wino"] else "   n/a", err))
print("sum over the pass: generic {:.3f} ms, wino {:.3f} ms".format(tot["gen"], tot["wino"]))
