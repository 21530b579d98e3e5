"""Does conv_wgrad_phase4_bf16 stay bit-reproducible when blocks of an LDS-using kernel share its CUs?  (scripts/flaky_phase4.py's screen had
only an elementwise neighbour.)  The neighbour: a bf16 1x1 convolution on the 128x64 tile (29 KB of LDS per block: fits beside phase4's 122 KB)."""
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


def screen(n, c1, c2, cout, h, w, rounds, phase4):
    bad = 0
    with ops.knob("wgrad_phase4", phase4):
        for r in range(rounds):
            a = torch.randn(n, h, w, c1, device=DEV, generator=g).to(BF)
            b = torch.randn(n, h, w, c2, device=DEV, generator=g).to(BF) if c2 else None
            dz = torch.randn(n, 2 * h, 2 * w, cout, device=DEV, generator=g).to(BF)
            torch.cuda.synchronize()
            neighbour(6)
            one = ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1)
            neighbour(6)
            two = ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1)
            bad += int(not torch.equal(one, two))
    print((n, c1, c2, cout, h, w), "phase4" if phase4 else "pair kernel", "unequal rounds:", bad, "of", rounds, flush=True)


print(ops.conv_tile_name(ops.ConvDesc(32, 64, 64, 256, 0, 0, 1, 1, 1, 0, 64, 64, 64, 0, 0), True, False))
screen(2, 256, 64, 128, 32, 48, 300, 1)
screen(2, 256, 64, 128, 32, 48, 100, 0)
screen(8, 256, 64, 128, 64, 64, 100, 1)
