"""Streaming rate of the two BatchNorm apply passes through the C ABI, per benchmark layer shape (measurement tool).

usage: python scripts/bench_bn.py [--dtype bf16] [--batch 32]
Prints, per (pixels, channels) of the ResNet-50 encoder at 512x512: ms and algorithmic TB/s of bn_apply (read y [+ residual],
write z) and of the backward apply (read g, y, write dy).  Buffers rotate so that consecutive launches do not hit the
256 MB Infinity Cache with the same lines."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
es = 2 if a.dtype == "bf16" else 4
shapes = [(256, 64), (128, 64), (128, 256), (64, 128), (64, 512), (32, 256), (32, 1024), (16, 512), (16, 2048)]


def timed(fn, n):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        fn(i % n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


for hw, c in shapes:
    m = a.batch * hw * hw
    nbuf = max(2, min(6, int(700e6 // (m * c * es)) + 1))
    ys = [torch.randn(m, c, device=dev).to(dt) for _ in range(nbuf)]
    gs = [torch.randn(m, c, device=dev).to(dt) for _ in range(nbuf)]
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    mean, inv, gamma = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) + 0.5
    part = torch.randn(64, 2, c, device=dev)
    t1 = timed(lambda i: ops.bn_apply(ys[i], sc, sh, relu=True), nbuf)
    t2 = timed(lambda i: ops.bn_apply(ys[i], sc, sh, residual=gs[i], relu=True), nbuf)
    t3 = timed(lambda i: ops.bn_bwd_from_partials(gs[i], ys[i], mean, inv, gamma, part), nbuf)
    b = m * c * es
    print("%4d^2 x %4d  (%6.1f MB)  apply %.4f ms %5.2f TB/s | apply+res %.4f ms %5.2f TB/s | bwd (incl. 2 finalize launches) %.4f ms %5.2f TB/s" % (
        hw, c, b / 1e6, t1, 2 * b / t1 / 1e9, t2, 3 * b / t2 / 1e9, t3, 3 * b / t3 / 1e9), flush=True)
