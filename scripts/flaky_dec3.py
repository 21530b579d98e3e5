"""Which kind of bug makes dec3's weight gradient differ between otherwise identical training steps (scripts/flaky_graph_step.py)?
The test's seven eager steps, four times per mode; per run the step at which dec3.block.block.weight first differs from run 0.
  normal     as shipped
  poison     every scratch buffer of robosat_amd.ops filled with 0x7f bytes (3.4e38 as fp32) before every step: a kernel that reads
             scratch it did not write this launch blows up instead of reading last step's leftovers
  onestream  ROBOSAT_WGRAD_STREAM=0: the weight gradients on the main stream (no cross-stream lifetime question left)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import robosat_ref as R, seeded
from robosat_amd import losses, ops
from robosat_amd.unet import UNet

DEV = "cuda:0"
init = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 9)
batches = [(seeded.synthetic_images(2, 3, 128, 192, 40 + i).to(DEV), seeded.synthetic_targets(2, 2, 128, 192, 40 + i).to(DEV)) for i in range(6)]
odd = (seeded.synthetic_images(1, 3, 64, 64, 77).to(DEV), seeded.synthetic_targets(1, 2, 64, 64, 77).to(DEV))
order = batches[:4] + [odd] + batches[4:]


def run(poison):
    net = UNet(2, pretrained=False, compute_dtype="bf16")
    net.load_state_dict(init)
    net = net.to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
    crit = losses.LovaszLoss2d().to(DEV)
    snaps = []
    for x, t in order:
        if poison:
            torch.cuda.synchronize()
            for ws in ops._WORKSPACE.values():
                ws.fill_(0x7F)
            torch.cuda.synchronize()
        opt.zero_grad()
        loss = crit(net(x), t)
        loss.backward()
        g = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}
        opt.step()
        snaps.append(g)
    torch.cuda.synchronize()
    return snaps


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        bad = [k for k in x if x[k] is not None and not torch.equal(x[k], y[k])]
        if bad:
            worst = max(float((x[k] - y[k]).abs().max()) for k in bad)
            finite = all(bool(torch.isfinite(y[k]).all()) for k in bad)
            return i, bad[:3], len(bad), worst, finite
    return None


for knobs in ({"wgrad_phase4": 1, "wgrad_ring": 2}, {"wgrad_phase4": 1}):  # (the kernel was the default when this was written)
    saved = {k: ops.get_knob(k) for k in knobs}
    for k, v in knobs.items():
        ops.set_knob(k, v)
    for mode in ("normal", "poison", "onestream"):
        os.environ["ROBOSAT_WGRAD_STREAM"] = "0" if mode == "onestream" else "1"
        runs = [run(mode == "poison") for _ in range(4)]
        print(knobs or "default", mode, [first_diff(runs[0], r) for r in runs[1:]], flush=True)
    for k, v in saved.items():
        ops.set_knob(k, v)
os.environ["ROBOSAT_WGRAD_STREAM"] = "1"
