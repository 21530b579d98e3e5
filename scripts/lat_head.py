"""Single-tile fp32 latency of the eval forward with the fused dec5 + final head and with the two launches (measurement tool).
    python scripts/lat_head.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd.unet import UNet

torch.manual_seed(0)
net = UNet(2, pretrained=False).to("cuda:0").eval()
x = torch.randn(1, 3, 512, 512, device="cuda:0")
with torch.no_grad():
    for fused in ("1", "0", "1", "0"):
        os.environ["ROBOSAT_FUSED_HEAD"] = fused
        for _ in range(10):
            net._forward_eval(x, False, argmax=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            net._forward_eval(x, False, argmax=True)
        e1.record()
        torch.cuda.synchronize()
        print("fused head {}: argmax forward 1x512^2 fp32 {:.3f} ms".format(fused, e0.elapsed_time(e1) / 100))
