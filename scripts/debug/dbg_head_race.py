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
for mode in ("softmax", "logits", "argmax"):
    for k in (0, 6):
        bad = 0; worst = 0.0; cnt = 0
        for r in range(20):
            x = torch.randn(n, s, s, c, device=DEV, generator=g)
            torch.cuda.synchronize()
            neighbour(k); a = ops.conv2d_wino33_head(x, u, fw, fb, mode=mode)
            neighbour(k); b = ops.conv2d_wino33_head(x, u, fw, fb, mode=mode)
            torch.cuda.synchronize()
            if not torch.equal(a, b):
                bad += 1; d = (a.float() - b.float()).abs(); worst = max(worst, float(d.max())); cnt = max(cnt, int((d != 0).sum()))
        print(mode, "neighbour launches", k, "unequal rounds", bad, "of 20 | worst |a-b|", worst, "| most differing elements", cnt, flush=True)
# the two-launch form for comparison
x = torch.randn(n, s, s, c, device=DEV, generator=g)
y = ops.conv2d_wino33(x, u, relu=True)
ref = ops.final_conv1x1(y, fw, fb, softmax=True)
got = ops.conv2d_wino33_head(x, u, fw, fb, mode="softmax")
print("fused vs two launches", float((ref - got).abs().max()))

print("---- plain wino33 (no head) beside the neighbour")
bad = 0
for r in range(20):
    x = torch.randn(n, s, s, c, device=DEV, generator=g)
    torch.cuda.synchronize()
    neighbour(6); a = ops.conv2d_wino33(x, u, relu=True)
    neighbour(6); b = ops.conv2d_wino33(x, u, relu=True)
    torch.cuda.synchronize()
    bad += int(not torch.equal(a, b))
print("plain wino33 unequal rounds", bad, "of 20")
print("---- where are the wrong pixels (logits, vs the launch alone)")
import collections
for r in range(4):
    x = torch.randn(n, s, s, c, device=DEV, generator=g)
    torch.cuda.synchronize()
    ref = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    torch.cuda.synchronize()
    neighbour(6); a = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
    torch.cuda.synchronize()
    d = (a != ref).any(1)  # [n, h, w]
    idx = d.nonzero()
    if idx.numel() == 0:
        print("round", r, "equal"); continue
    patches = collections.Counter((int(i[0]), int(i[1]) // 16, int(i[2]) // 16) for i in idx.tolist())
    inpatch = collections.Counter((int(i[1]) % 16, int(i[2]) % 16) for i in idx.tolist())
    items = sorted(set((p[0] * 16 + p[1]) * 16 + p[2] for p in patches))
    print("round", r, "wrong pixels", idx.shape[0], "in", len(patches), "patches; pixels per patch", sorted(set(patches.values()))[:8],
          "| item index mod 256:", sorted(set(i % 256 for i in items))[:12], "| items", items[:12])
    print("   positions inside a patch (y, x): count", inpatch.most_common(8))
    bothwrong = ((a != ref).all(1) & d).sum()
    print("   pixels with both classes wrong", int(bothwrong), "| max |err|", float((a - ref).abs().max()))
