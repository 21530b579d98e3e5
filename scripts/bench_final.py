"""ms and algorithmic TB/s of the `final` 1x1 convolution forward / backward at the benchmark shape (measurement tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

dev = torch.device("cuda:0")
for dt, n in ((torch.bfloat16, 32), (torch.float32, 16)):
    es = 2 if dt == torch.bfloat16 else 4
    for c in (2, 4):
        xs = [torch.randn(n, 512, 512, 32, device=dev).to(dt) for _ in range(3)]
        w, b = torch.randn(c, 32, 1, 1, device=dev) * 0.1, torch.randn(c, device=dev)
        dl = torch.randn(n, c, 512, 512, device=dev)

        def timed(fn):
            fn(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                fn(i % 3)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 12

        px = n * 512 * 512
        t1 = timed(lambda i: ops.final_conv1x1(xs[i], w, b))
        t2 = timed(lambda i: ops.final_conv1x1_bwd(xs[i], w, dl))
        print("%s bs%d C=%d  fwd %.4f ms %.2f TB/s | bwd %.4f ms %.2f TB/s" % (
            str(dt).split(".")[1], n, c, t1, px * (32 * es + 4 * c) / t1 / 1e9, t2, px * (64 * es + 4 * c) / t2 / 1e9), flush=True)
