import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from collections import Counter
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
shown = 0
for r in range(12):
    x = torch.randn(n, s, s, c, device=DEV, generator=g)
    torch.cuda.synchronize()
    ref = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    y = ops.conv2d_wino33(x, u, relu=True)
    q = torch.stack([torch.einsum("nhwk,ck->nchw", y[..., 4 * i:4 * i + 4], fw[:, 4 * i:4 * i + 4]) for i in range(8)])  # [8 groups][n][c][h][w]
    torch.cuda.synchronize()
    neighbour(6)
    a = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    torch.cuda.synchronize()
    d = a - ref
    idx = (d != 0).nonzero().tolist()
    hyp = Counter()
    for nn, cc, yy, xx in idx[:400]:
        e = float(d[nn, cc, yy, xx])
        qs = [float(q[i, nn, cc, yy, xx]) for i in range(8)]
        found = None
        for k in range(1, 5):
            for S in itertools.combinations(range(8), k):
                if abs(e + sum(qs[i] for i in S)) < 3e-5: found = "missing groups " + str(S)
                elif abs(e - sum(qs[i] for i in S)) < 3e-5: found = "doubled groups " + str(S)
            if found: break
        hyp[found or "other"] += 1
        if found is None and shown < 6:
            shown += 1
            print("   e.g. (n, c, y, x) =", (nn, cc, yy, xx), "err", round(e, 5), "ref", round(float(ref[nn, cc, yy, xx]), 5), "groups", [round(v, 4) for v in qs], "bias", round(float(fb[cc]), 4))
    print("round", r, "wrong logits", len(idx), dict(hyp), flush=True)
