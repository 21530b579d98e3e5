"""us per launch of rs_bottleneck_tail_f32 at the benchmark's shape (16 x 128 x 128 pixels) against the two launches it replaces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robosat_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
n, h, w = 16, 128, 128
nb = 4  # rotate over buffers larger than the Infinity Cache together
xs = [torch.randn(n, h, w, 64, generator=g).to(DEV) for _ in range(nb)]
ids = [torch.randn(n, h, w, 256, generator=g).to(DEV) for _ in range(nb)]
w3 = (torch.randn(256, 1, 1, 64, generator=g) * 0.1).to(DEV)
w1 = (torch.randn(64, 1, 1, 256, generator=g) * 0.05).to(DEV)
s3, t3, s1, t1 = (torch.rand(256, generator=g).to(DEV) + 0.5, torch.randn(256, generator=g).to(DEV), torch.rand(64, generator=g).to(DEV) + 0.5,
                  torch.randn(64, generator=g).to(DEV))


def timed(fn, reps=3, it=24):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / it * 1e3)
    return ["%.1f" % v for v in out]


print(os.environ.get("ROBOSAT_HIP_LIB", "shipped library").split("/")[-1], "fused", timed(lambda i: ops.bottleneck_tail(xs[i % nb], w3, s3, t3, ids[i % nb], w1, s1, t1)))


def two(i):
    o = ops.conv2d(xs[i % nb], w3, scale=s3, shift=t3, residual=ids[i % nb], relu=True)
    return ops.conv2d(o, w1, scale=s1, shift=t1, relu=True)


print("   two launches", timed(two))
