"""Host (CPU) time to ISSUE one train step vs the GPU time to run it (measurement tool).

The bench loop never synchronises inside the timed region, so the host may run ahead of the GPU - if issuing a step takes
less host time than the GPU needs to execute it.  This prints both, per phase of the step, with the GPU idle at the start
of every measured step (nothing blocks the issue):  python scripts/host_overhead.py [--batch 32] [--dtype bf16] [--fused]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from robosat_amd import losses

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--fused", action="store_true", help="torch.optim.Adam(fused=True)")
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
net = bench.build_model(2, dev, True, a.dtype)
g = torch.Generator().manual_seed(100)
x = torch.randn(a.batch, 3, a.size, a.size, generator=g).to(dev)
tgt = torch.randint(0, 2, (a.batch, a.size, a.size), generator=g).to(dev)
crit = losses.LovaszLoss2d().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True) if a.fused else torch.optim.Adam(net.parameters(), lr=1e-4)


def step(marks=None):
    t = [time.perf_counter()]
    opt.zero_grad()
    t.append(time.perf_counter())
    loss = crit(net(x), tgt)
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    if marks is not None:
        marks.append([1e3 * (b - a_) for a_, b in zip(t[:-1], t[1:])])


for _ in range(3):
    step()
torch.cuda.synchronize()
rows, gpu = [], []
for _ in range(a.steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(rows)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    gpu.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
for r, (i, t) in zip(rows, gpu):
    print("host issue: zero_grad %.2f  forward+loss %.2f  backward %.2f  optimizer %.2f  = %.2f ms   |  step until idle %.2f ms" % (*r, i, t))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("free-running: %.2f ms/step" % (1e2 * (time.perf_counter() - t0)))
