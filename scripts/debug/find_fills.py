"""Which torch ops a bf16 train step still launches beside our kernels (measurement tool): runs two steps under
torch.profiler with Python stacks and prints, per aten op that launches a device kernel / copy (fill_, copy_, zero_, ...),
how often it ran per step and the innermost robosat_amd / bench frame it came from.

    python scripts/debug/find_fills.py [--batch 8] [--size 256]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from robosat_amd import losses

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
net = bench.build_model(2, dev, True, "bf16")
g = torch.Generator().manual_seed(100)
x = torch.randn(a.batch, 3, a.size, a.size, generator=g).to(dev)
tgt = torch.randint(0, 2, (a.batch, a.size, a.size), generator=g).to(dev)
crit = losses.LovaszLoss2d().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)


def step():
    opt.zero_grad()
    loss = crit(net(x), tgt)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()

by = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.name not in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::zeros", "aten::ones_like", "aten::to",
                       "aten::_to_copy", "aten::div_", "aten::mul", "aten::add_", "aten::_foreach_add_", "aten::contiguous"):
        continue
    where = "?"
    for fr in ev.stack or []:
        if "robosat_amd" in fr or "bench.py" in fr or "find_fills" in fr or "torch/optim" in fr or "autograd" in fr:
            where = fr
            break
    by[(ev.name, where)] += 1
for (name, where), n in by.most_common(40):
    print("{:6.1f} per step  {:22s} {}".format(n / STEPS, name, where))
print()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
