"""Single-tile latency of the serve / predict path (uint8 tile in HBM -> class mask / probability bytes), eager launches vs
hipGraph replay, fp32 and bf16.  Prints one JSON line per case (measurement tool)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd.unet import UNet

dev = torch.device("cuda:0")
for dtype in ("fp32", "bf16"):
    torch.manual_seed(0)
    net = UNet(2, pretrained=False, compute_dtype=dtype).to(dev).eval()
    for n, size in ((1, 512), (4, 512), (1, 1024)):
        u8 = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8).to(dev)
        for mode in ("0", "1"):
            os.environ["ROBOSAT_GRAPHS"] = mode
            for _ in range(5):
                net.predict_classes(u8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 50
            for _ in range(iters):
                net.predict_classes(u8)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            print(json.dumps({"case": "predict_classes {}x{}^2 {}".format(n, size, dtype), "launch": "hipGraph replay" if mode == "1" else "eager",
                              "ms": round(ms, 3), "tiles_per_s": round(n / ms * 1e3, 1)}), flush=True)
