"""Round 6 forensics on the padded ring (knob wgrad_ring = 4 + DEAD, see scripts/dma_order_bisect.py): WHICH chunk does a wrong launch get wrong?
1x1 weight gradient with dy = 1 and x[pixel][ci] = 2 ** (position of the pixel's 64-pixel chunk inside its split): a block's partial tile is
sum_chunks 64 * 2 ** pos, exact in fp32, so  (expected - got)  decodes into the chunk positions (powers of two) and pixel counts that went missing
(or, for a stale read of the slot's previous tenant, +2 ** (pos - RING) - 2 ** pos)."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(3)
side = torch.cuda.Stream()
nx = torch.randn(32, 64, 64, 256, device=DEV, generator=g).to(BF)
nw = (torch.randn(64, 1, 1, 256, device=DEV, generator=g) * 0.05).to(BF)
N, H, W, CIN, COUT, CPS = 16, 32, 32, 192, 512, 16  # 12 tiles of 128 x 64 -> 16 splits of 16 chunks (plan(): target 192 blocks)


def neighbour(k):
    with torch.cuda.stream(side):
        for _ in range(k):
            ops.conv2d(nx, nw)


def main(ring, rounds):
    m = torch.arange(N * H * W, device=DEV)
    pos = (m // 64) % CPS
    x = (2.0 ** pos.float()).view(N, H, W, 1).expand(N, H, W, CIN).contiguous().to(BF)
    dy = torch.ones(N, H, W, COUT, device=DEV, dtype=BF)
    want = float(64 * (2 ** CPS - 1) * (N * H * W // 64 // CPS))
    ref = ops.conv2d_wgrad(dy, x, 1, 1)
    torch.cuda.synchronize()
    assert float(ref.min()) == want and float(ref.max()) == want, (float(ref.min()), float(ref.max()), want)
    bad = 0
    for r in range(rounds):
        with ops.knob("wgrad_ring", ring):
            neighbour(6)
            got = ops.conv2d_wgrad(dy, x, 1, 1)
        torch.cuda.synchronize()
        d = (ref.double() - got.double()).view(COUT, CIN)
        if bool((d != 0).any()):
            bad += 1
            if bad <= 6:
                vals = Counter(d[d != 0].long().tolist()).most_common(6)
                rows = (d != 0).any(1).nonzero().flatten().tolist()
                cols = (d != 0).any(0).nonzero().flatten().tolist()
                print("  ring", ring, "round", r, "wrong elements", int((d != 0).sum()), "| expected - got (value: count)", vals,
                      "| couts", rows[0], "..", rows[-1], "(", len(rows), ") cins", cols[0], "..", cols[-1], "(", len(cols), ")", flush=True)
    print("ring", ring, "wrong launches", bad, "of", rounds, flush=True)


print(ops.wgrad_kernel_name(ops.ConvDesc(N, H, W, CIN, 0, 0, 1, 1, 1, 0, H, W, COUT, 0, 0)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for ring in (4, 7, 5, 6, 3, 2):
    main(ring, rounds)
