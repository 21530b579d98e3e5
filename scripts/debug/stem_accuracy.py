"""Forward error of the fp32 stem against a float64 convolution on the host (where, how large).  ROBOSAT_HIP_LIB picks the library."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import robosat_ref as R, seeded  # noqa: E402
from robosat_amd import ops  # noqa: E402

DEV = "cuda:0"
sd = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 2)
wt = sd["resnet.conv1.weight"].float()
for shape in ((2, 3, 128, 128), (2, 3, 512, 512)):
    x = seeded.synthetic_images(shape[0], 3, shape[2], shape[3], 2)
    want = F.conv2d(x.double(), wt.double(), stride=2, padding=3)
    x4 = ops.nchw_to_nhwc4(x.to(DEV))
    packed = ops.pack_stem_weight(wt.permute(0, 2, 3, 1).contiguous().to(DEV))
    for bands in (3, 4):
        got = ops.conv2d(x4, packed, stride=2, pad=3, stem=7, bands=bands).permute(0, 3, 1, 2).cpu().double()
        err = (got - want).abs()
        idx = torch.nonzero(err == err.max())[0].tolist()
        print(shape, "bands", bands, "max|err| %.3e at %s  rms err %.3e  rms want %.3e  max want %.3e  (fp32 host conv: max|err| %.3e)" % (
            float(err.max()), idx, float(err.pow(2).mean().sqrt()), float(want.pow(2).mean().sqrt()), float(want.abs().max()),
            float((F.conv2d(x, wt, stride=2, padding=3).double() - want).abs().max())))
        # error by output column / row band: a border or tile-seam defect shows as a spike
        percol = err.amax(dim=(0, 1, 2))
        perrow = err.amax(dim=(0, 1, 3))
        print("   worst columns", torch.topk(percol, 4).indices.tolist(), ["%.2e" % v for v in torch.topk(percol, 4).values.tolist()],
              "worst rows", torch.topk(perrow, 4).indices.tolist(), ["%.2e" % v for v in torch.topk(perrow, 4).values.tolist()])
