"""Round 6: what exactly is wrong with a counted s_waitcnt vmcnt(N) over a PADDED tail?  conv_wgrad_bf16<128x64, RING = 4> beside the
LDS-using neighbour (the round-5 control, scripts/flaky_ring4.py: 29-138 of 60-150 launches unequal), with the chunks past the end of a
split issued four ways (knob wgrad_ring = 4 + DEAD):
  4  out-of-range pieces into the dead ring slot (the control)          5  in-range loads of a 1 KiB zero line into the dead slot
  6  out-of-range pieces into a scratch KiB nobody ever reads           7  whatever follows the split in memory (ordinary loads)
Per round: the launch twice on the same data (equal bits?) and against the two-buffer result.  scripts/probes/probe_dma_order.hip asks
the hardware the underlying question directly."""
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


def screen(ring, rounds):
    unequal = differs = 0
    for r in range(rounds):
        dy = torch.randn(16, 32, 32, 128, device=DEV, generator=g).to(BF)
        x = torch.randn(16, 64, 64, 64, device=DEV, generator=g).to(BF)
        torch.cuda.synchronize()
        ref = ops.conv2d_wgrad(dy, x, 3, 3, stride=2, pad=1)  # two buffers, alone
        torch.cuda.synchronize()
        with ops.knob("wgrad_ring", ring):
            neighbour(6)
            one = ops.conv2d_wgrad(dy, x, 3, 3, stride=2, pad=1)
            neighbour(6)
            two = ops.conv2d_wgrad(dy, x, 3, 3, stride=2, pad=1)
        unequal += int(not torch.equal(one, two))
        differs += int(not torch.equal(one, ref))
    print("wgrad_ring", ring, "rounds", rounds, "| launch twice, unequal:", unequal, "| differs from the two-buffer result:", differs, flush=True)


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
print(ops.wgrad_kernel_name(ops.ConvDesc(16, 64, 64, 64, 0, 0, 3, 3, 2, 1, 32, 32, 128, 0, 0)))
for ring in (4, 5, 6, 7, 3, 2, 4):
    screen(ring, rounds)
