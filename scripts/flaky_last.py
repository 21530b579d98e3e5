"""The last screen of round 5, beside the LDS-using neighbour of scripts/flaky_phase4_coresident.py: (a) the shipped weight-gradient
pipeline (two buffers, drained waits) on the shape that failed with the ring, (b) conv_thin_bf16 -- the one other kernel whose counted
wait is followed by a read in the same barrier phase (its ReLU-mask patch) -- on dec5's data gradient and forward."""
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


assert ops.get_knob("wgrad_ring") == 2 and ops.get_knob("wgrad_phase4") == 0 and ops.get_knob("wgrad_blocks") == 192
screen("wgrad 128x128, two buffers (shipped)", lambda: (torch.randn(16, 32, 32, 128, device=DEV, generator=g).to(BF), torch.randn(16, 64, 64, 128, device=DEV, generator=g).to(BF)),
       lambda dy, x: ops.conv2d_wgrad(dy, x, 3, 3, stride=2, pad=1), 120)
w5 = (torch.randn(32, 3, 3, 32, device=DEV, generator=g) * 0.05).to(BF)
print(ops.conv_tile_name(ops.ConvDesc(8, 512, 512, 32, 0, 0, 3, 3, 1, 1, 512, 512, 32, 0, 0), True, False))
screen("conv_thin 3x3 with ReLU mask (dec5 data gradient)",
       lambda: (torch.randn(8, 512, 512, 32, device=DEV, generator=g).to(BF), torch.randn(8, 512, 512, 32, device=DEV, generator=g).to(BF)),
       lambda d, m: ops.conv2d(d, w5, pad=1, relu_mask=m), 60)
screen("conv_thin 3x3 forward", lambda: (torch.randn(8, 512, 512, 32, device=DEV, generator=g).to(BF),), lambda x: ops.conv2d(x, w5, pad=1, relu=True), 40)
