"""us per launch of the fp32 stem at the benchmark's shape (16 x 3 x 512 x 512), folded-BatchNorm + ReLU epilogue; RS_STEM_STAGGER from the environment."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robosat_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(16, 3, 512, 512, generator=g).to(DEV)
w = (torch.randn(64, 7, 7, 3, generator=g) * 0.05).to(DEV)
sc, sh = torch.rand(64, generator=g).to(DEV) + 0.5, torch.randn(64, generator=g).to(DEV)
x4 = ops.nchw_to_nhwc4(x)
packed = ops.pack_stem_weight(w)
for bands in (3, 4):
    for _ in range(5):
        y = ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=bands, scale=sc, shift=sh, relu=True)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            y = ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=bands, scale=sc, shift=sh, relu=True)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 40 * 1e3)
    print("stagger", os.environ.get("RS_STEM_STAGGER", "0"), "bands", bands, "us per launch", ["%.1f" % r for r in res])
