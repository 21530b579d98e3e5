"""Which torch calls in a bf16 training step issue device copies / fills (the __amd_rocclr_copyBuffer / fillBuffer launches of
the kernel trace)?  torch.profiler with stacks over two steps at a small shape; prints copy-like events grouped by Python frame."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

from robosat_amd import losses
from robosat_amd.unet import UNet

dev = torch.device("cuda:0")
net = UNet(2, pretrained=False, compute_dtype="bf16").to(dev).train()
x = torch.randn(8, 3, 256, 256, device=dev)
t = torch.randint(0, 2, (8, 256, 256), device=dev)
crit = losses.LovaszLoss2d().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)


def step():
    opt.zero_grad()
    loss = crit(net(x), t)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ops = collections.Counter()
for ev in prof.events():
    name = ev.name
    if any(k in name.lower() for k in ("copy", "memcpy", "memset", "fill", "zero_", "clone", "contiguous", "aten::to", "_to_copy")):
        stack = [s for s in (ev.stack or []) if "robosat_amd" in s or "bench" in s or "optim" in s or "find_copies" in s]
        ops[(name, tuple(ev.input_shapes or ())[:2].__repr__()[:60], stack[0][-90:] if stack else "")] += 1
for (name, shapes, where), n in sorted(ops.items(), key=lambda kv: -kv[1])[:40]:
    print("{:4d}  {:32s} {:60s} {}".format(n, name[:32], shapes, where))
print("--- device kernels of the step (top by count)")
k = collections.Counter(ev.name[:70] for ev in prof.events() if ev.device_type is not None and "cuda" in str(ev.device_type).lower())
for name, n in k.most_common(25):
    print("{:4d}  {}".format(n, name))
