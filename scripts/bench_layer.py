"""Micro-benchmark of single convolution layers through the C ABI (measurement tool; not part of the product path).

usage: python scripts/bench_layer.py [--bf16] [--stats] N,Cin,H,W,Cout,k,stride,pad ...
Prints per layer: ms, effective HBM GB/s (input + output bytes, weights ignored) and executed TFLOP/s.  Inputs rotate over
enough buffers to exceed the 256 MB Infinity Cache, so reads come from HBM as they do in a real step."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robosat_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--stats", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", action="store_true", help="compare with torch conv2d (fp32 math on the same rounded operands)")
ap.add_argument("layers", nargs="+")
a = ap.parse_args()
dt = torch.bfloat16 if a.bf16 else torch.float32
dev = torch.device("cuda:0")
for spec in a.layers:
    n, cin, h, w, cout, k, st, pad = [int(v) for v in spec.split(",")]
    es = 2 if a.bf16 else 4
    inb = n * h * w * cin * es
    nbuf = max(2, int(600e6 // inb) + 1)
    xs = [torch.randn(n, h, w, cin, device=dev).to(dt) for _ in range(min(nbuf, 8))]
    wt = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(dt)
    fn = (lambda x: ops.conv2d_bnstats(x, wt, stride=st, pad=pad)) if a.stats else (lambda x: ops.conv2d(x, wt, stride=st, pad=pad))
    y = fn(xs[0])
    y = y[0] if isinstance(y, tuple) else y
    torch.cuda.synchronize()
    err = ""
    if a.check:
        ref = torch.nn.functional.conv2d(xs[0].float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), stride=st, padding=pad)
        got = y.float().permute(0, 3, 1, 2)
        err = "  maxerr/scale=%.2e" % (float((got - ref).abs().max()) / max(1.0, float(ref.abs().max())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        fn(xs[i % len(xs)])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    outb = y.numel() * es
    fl = 2.0 * y.numel() * k * k * cin
    print("%-28s ms=%.4f  io=%.0f MB  %.0f GB/s  %.0f TF" % (spec, ms, (inb + outb) / 1e6, (inb + outb) / ms / 1e6, fl / ms / 1e9) + err, flush=True)
