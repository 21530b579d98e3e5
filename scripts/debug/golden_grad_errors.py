"""Per-tensor elementwise errors of one fp32 training step against the reference's golden gradients (tests/golden/train_step_*.npz),
printed worst first -- what tests/test_gpu_train_step.py::test_train_step_matches_reference_golden bounds -- with the stem's output taken
from four equally accurate sources: the stem kernel's RGB form, its four-band form, the host's fp32 convolution (what the golden's own
forward computed), a float64 convolution rounded to fp32.  Everything behind the stem is the same GPU path each time: the spread between
the rows is what a rounding-level change of ONE layer's output does to these gradients."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import robosat_ref as R, seeded  # noqa: E402
from robosat_amd import losses, ops  # noqa: E402
from robosat_amd.unet import UNet  # noqa: E402

DEV = "cuda:0"
real_conv2d = ops.conv2d
for loss_name in ("CrossEntropy", "Lovasz"):
    g = np.load(os.path.join("tests", "golden", "train_step_{}.npz".format(loss_name)))
    for source in ("kernel, RGB form", "kernel, four-band form", "host fp32 conv", "host float64 conv, rounded"):
        net = UNet(2, pretrained=False)
        net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 2))
        net = net.to(DEV).train()
        x = seeded.synthetic_images(2, 3, 128, 128, 2).to(DEV)
        t = seeded.synthetic_targets(2, 2, 128, 128, 2).to(DEV)
        wt = net.resnet.conv1.weight.detach().cpu()

        def patched(src1, weight, *a, **kw):
            if not kw.get("stem"):
                return real_conv2d(src1, weight, *a, **kw)
            if source.startswith("kernel"):
                kw["bands"] = 3 if "RGB" in source else 4
                return real_conv2d(src1, weight, *a, **kw)
            xin = x.cpu()
            y = F.conv2d(xin, wt, stride=2, padding=3) if "fp32" in source else F.conv2d(xin.double(), wt.double(), stride=2, padding=3).float()
            return y.permute(0, 2, 3, 1).contiguous().to(DEV)

        ops.conv2d = patched
        crit = (losses.CrossEntropyLoss2d(weight=torch.tensor([1.6248, 5.762827])) if loss_name == "CrossEntropy" else losses.LovaszLoss2d()).to(DEV)
        logits = net(x)
        loss = crit(logits, t)
        loss.backward()
        ops.conv2d = real_conv2d
        params = dict(net.named_parameters())
        rows = []
        for key in g.files:
            if key.startswith("grad/"):
                want = torch.from_numpy(g[key])
                got = params[key[5:]].grad.cpu()
                rows.append((float((got - want).abs().max()) / max(1e-8, float(want.abs().max())), key[5:].replace("resnet.", "")))
        rows.sort(reverse=True)
        print("{:12s} stem = {:28s} logits err {:.2e} | ".format(loss_name, source, float(np.abs(logits.detach().cpu().numpy() - g["logits"]).max()))
              + "  ".join("{} {:.4f}".format(k, v) for v, k in rows[:5]))
