"""The other big-LDS kernels of the bf16 step beside the LDS-using neighbour of scripts/flaky_phase4_coresident.py: are THEY bit-reproducible?
  conv_halo_bf16<3x3,512x128> (109 KB)   layer2's conv2 at bs 32
  conv_wgrad_bf16<128x128>, ring of three (99 KB)   a 3x3 / stride-2 weight gradient"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(3)
side = torch.cuda.Stream()
nx = torch.randn(32, 64, 64, 256, device=DEV, generator=g).to(BF)
nw = (torch.randn(64, 1, 1, 256, device=DEV, generator=g) * 0.05).to(BF)


def neighbour(k):
    with torch.cuda.stream(side):
        for _ in range(k):
            ops.conv2d(nx, nw)


def screen(name, make, fn, rounds):
    bad = 0
    for r in range(rounds):
        args = make()
        torch.cuda.synchronize()
        neighbour(6)
        one = fn(*args)
        neighbour(6)
        two = fn(*args)
        bad += int(not torch.equal(one, two))
    print(name, "unequal rounds:", bad, "of", rounds, flush=True)


wt = (torch.randn(128, 3, 3, 128, device=DEV, generator=g) * 0.05).to(BF)
print(ops.conv_tile_name(ops.ConvDesc(32, 64, 64, 128, 0, 0, 3, 3, 1, 1, 64, 64, 128, 0, 0), True, False))
screen("halo 3x3 512x128", lambda: (torch.randn(32, 64, 64, 128, device=DEV, generator=g).to(BF),), lambda x: ops.conv2d(x, wt, pad=1), 150)
screen("wgrad 128x128 ring 3", lambda: (torch.randn(16, 32, 32, 128, device=DEV, generator=g).to(BF), torch.randn(16, 64, 64, 128, device=DEV, generator=g).to(BF)),
       lambda dy, x: ops.conv2d_wgrad(dy, x, 3, 3, stride=2, pad=1), 150)
