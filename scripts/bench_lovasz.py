"""LovaszLoss2d kernels alone (robosat_amd/csrc/lovasz.hip): ms per forward + gradient at the benchmark's key counts.
usage: python scripts/bench_lovasz.py  (measurement tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

dev = torch.device("cuda:0")
for n, c, hw in ((32, 2, 512), (32, 4, 512), (8, 2, 512)):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(n, c, hw, hw, generator=g) * 0.7).to(dev)
    t = torch.randint(0, c, (n, hw, hw), generator=g).to(dev)
    for _ in range(3):
        ops.lovasz_fwd(x, t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.lovasz_fwd(x, t)
    e1.record()
    torch.cuda.synchronize()
    print("lovasz fwd + grad  N {} C {} {}x{}  ({} keys per image): {:.3f} ms".format(n, c, hw, hw, c * hw * hw, e0.elapsed_time(e1) / 20))
