import os, sys
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
prev = None
for r in range(6):
    x = torch.randn(n, s, s, c, device=DEV, generator=g)
    torch.cuda.synchronize()
    ref = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    y = ops.conv2d_wino33(x, u, relu=True)
    p0 = torch.einsum("nhwk,ck->nchw", y[..., :16], fw[:, :16])
    p1 = torch.einsum("nhwk,ck->nchw", y[..., 16:], fw[:, 16:])
    torch.cuda.synchronize()
    out = torch.full_like(ref, 12345.0)  # a known sentinel where the result goes: a missing store shows
    neighbour(6)
    a = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    torch.cuda.synchronize()
    d = a - ref
    idx = (d != 0).nonzero().tolist()
    hyp = Counter()
    for nn, cc, yy, xx in idx[:3000]:
        e = float(d[nn, cc, yy, xx]); tol = 3e-5
        ty, tx = yy & ~1, xx & ~1
        found = "other"
        for (half, name) in ((p0, "cg0 half"), (p1, "cg1 half")):
            own = float(half[nn, cc, yy, xx])
            if abs(e + own) < tol: found = name + " missing (0)"
            for dy in (0, 1):
                for dx in (0, 1):
                    if (ty + dy, tx + dx) != (yy, xx) and abs(e - (float(half[nn, cc, ty + dy, tx + dx]) - own)) < tol:
                        found = name + " of sibling pixel (%d,%d) instead of (%d,%d)" % (dy, dx, yy & 1, xx & 1)
            oc = 1 - cc
            if abs(e - (float(half[nn, oc, yy, xx]) - own)) < tol: found = name + " of the OTHER class"
        if prev is not None and abs(float(a[nn, cc, yy, xx]) - float(prev[nn, cc, yy, xx])) < 1e-7: found = "previous round's value (store lost)"
        hyp[found] += 1
    print("round", r, "wrong logits", len(idx), dict(hyp), flush=True)
    prev = a
