import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robosat_amd import ops
DEV = "cuda:0"; BF = torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(29)
n, s, c, classes = 4, 256, 32, 8
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
for r in range(6):
    x = torch.randn(n, s, s, c, device=DEV, generator=g)
    torch.cuda.synchronize()
    ref = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    y = ops.conv2d_wino33(x, u, relu=True)                      # [n, h, w, 32]
    p1 = torch.einsum("nhwk,ck->nchw", y[..., 16:], fw[:, 16:])  # the cg = 1 waves' half of every logit
    torch.cuda.synchronize()
    neighbour(6); a = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    torch.cuda.synchronize()
    d = a - ref
    idx = (d != 0).nonzero()
    print("round", r, "wrong logits", idx.shape[0])
    from collections import Counter
    cls = Counter(int(i[1]) for i in idx.tolist()); print("   by class", sorted(cls.items()))
    par = Counter((int(i[2]) % 2, int(i[3]) % 2) for i in idx.tolist()); print("   by pixel of the 2x2 tile (y, x parity)", sorted(par.items()))
    img = Counter(int(i[0]) for i in idx.tolist()); print("   by image", sorted(img.items()))
    hyp = {"read as 0": 0, "read as class c+4's": 0, "read as class c-4's": 0, "same pixel of image n-2": 0, "same pixel of image n-1": 0, "same pixel of image n+1": 0, "other": 0}
    for i in idx.tolist()[:4000]:
        nn, cc, yy, xx = i
        e = float(d[nn, cc, yy, xx]); own = float(p1[nn, cc, yy, xx])
        alt4 = float(p1[nn, cc + 4, yy, xx]) if cc + 4 < classes else None
        altm = float(p1[nn, cc - 4, yy, xx]) if cc >= 4 else None
        tol = 2e-5 * (1 + abs(own))
        if abs(e + own) < tol: hyp["read as 0"] += 1
        elif alt4 is not None and abs(e - (alt4 - own)) < tol: hyp["read as class c+4's"] += 1
        elif altm is not None and abs(e - (altm - own)) < tol: hyp["read as class c-4's"] += 1
        elif nn >= 2 and abs(e - (float(p1[nn - 2, cc, yy, xx]) - own)) < tol: hyp["same pixel of image n-2"] += 1
        elif nn >= 1 and abs(e - (float(p1[nn - 1, cc, yy, xx]) - own)) < tol: hyp["same pixel of image n-1"] += 1
        elif nn + 1 < n and abs(e - (float(p1[nn + 1, cc, yy, xx]) - own)) < tol: hyp["same pixel of image n+1"] += 1
        else: hyp["other"] += 1
    print("   what the wrong logit's cg = 1 half looks like:", hyp)
