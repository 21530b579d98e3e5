import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from oracle import robosat_ref as R, seeded
from robosat_amd import ops
from robosat_amd.unet import UNet
DEV = "cuda:0"
classes = 4
net = UNet(classes, pretrained=False)
net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(classes).state_dict(), 41))
net = net.to(DEV).eval()
g = torch.Generator().manual_seed(8)
u8 = torch.randint(0, 256, (2, 128, 160, 3), generator=g, dtype=torch.uint8)
print("A", flush=True)
got = net.predict_quantized(u8.to(DEV), overlap=16)
torch.cuda.synchronize(); print("quantized ok", got.shape, flush=True)
from robosat_amd.transforms import ImageToTensor, Normalize
norm = Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
x = torch.stack([norm(ImageToTensor()(Image.fromarray(im.numpy(), mode="RGB"))) for im in u8])
print("B", x.shape, x.dtype, x.is_contiguous(), flush=True)
xd = x.to(DEV)
torch.cuda.synchronize(); print("upload ok", flush=True)
_orig = {}
for name in ("nchw_to_nhwc4", "conv2d", "maxpool2d", "conv2d_phase", "final_conv1x1", "bn_fold", "pack_stem_weight", "pack_phase_weight"):
    fn = getattr(ops, name)
    def wrap(*a, _fn=fn, _name=name, **k):
        r = _fn(*a, **k)
        torch.cuda.synchronize()
        shp = tuple(r.shape) if hasattr(r, "shape") else None
        print("  ", _name, shp, flush=True)
        return r
    setattr(ops, name, wrap)
probs = net.predict_probs(xd)
torch.cuda.synchronize(); print("probs ok", probs.shape, flush=True)
m = net.predict_classes(u8.to(DEV))
torch.cuda.synchronize(); print("classes ok", m.shape, flush=True)
