import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robosat_amd import ops
DEV = "cuda:0"; BF = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(29)
n, s, c, classes = 4, 256, 32, 2
w_krsc = torch.randn(c, 3, 3, c, device=DEV, generator=g) * 0.1
u = ops.pack_wino33_weight(w_krsc)
fw, fb = torch.randn(classes, c, device=DEV, generator=g) * 0.2, torch.randn(classes, device=DEV, generator=g)
side = torch.cuda.Stream()
nx = torch.randn(32, 64, 64, 256, device=DEV, generator=g).to(BF)
nw = (torch.randn(64, 1, 1, 256, device=DEV, generator=g) * 0.05).to(BF)
def neighbour(k):
    with torch.cuda.stream(side):
        for _ in range(k):
            ops.conv2d(nx, nw)
for ov in [int(a) for a in sys.argv[1:]] or [0, 77]:
    bad = 0
    for r in range(20):
        x = torch.randn(n, s, s, c, device=DEV, generator=g)
        torch.cuda.synchronize()
        ref = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
        torch.cuda.synchronize()
        neighbour(6); a = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits", overlap=ov)
        torch.cuda.synchronize()
        bad += int(not torch.equal(a, ref))
    print("experiment switch", ov, "| wrong launches", bad, "of 20", flush=True)
