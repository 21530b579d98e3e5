"""How far the REFERENCE's own gradients move when its stem output is rounded differently (CPU, fp32, no GPU).

tests/test_gpu_train_step.py::test_train_step_matches_reference_golden bounds the elementwise error of eleven gradient tensors of one
training step against goldens written by the unmodified reference on the CPU.  A training step through 53 train-mode BatchNorms and a
sort-based loss amplifies rounding: this script runs the oracle (oracle/robosat_ref.py, pinned bit for bit to the reference) on the golden's
inputs twice -- as is, and with resnet.conv1's output perturbed by Gaussian noise of relative size `eps` (the fp32 rounding of a
147-term dot product measured against float64 is 2.2e-7 rms relative: scripts/debug/stem_accuracy.py) -- and prints, for the golden's eleven tensors,
max|g' - g| / max|g|.  The bound of the test must sit above these numbers: a kernel that is exactly as accurate as the host's convolution but
sums in another order cannot do better than the reference does against itself.

    python tests/golden/grad_noise_floor.py        (about two minutes on 8 cores)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import robosat_ref as R, seeded  # noqa: E402


def step(loss_name, eps, seed):
    net = R.UNetRef(2)
    net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 2))
    net.train()
    x = seeded.synthetic_images(2, 3, 128, 128, 2)
    t = seeded.synthetic_targets(2, 2, 128, 128, 2)
    if eps:
        gen = torch.Generator().manual_seed(seed)
        net.resnet.conv1.register_forward_hook(lambda m, i, o: o + eps * o.pow(2).mean().sqrt() * torch.randn(o.shape, generator=gen))
    logits = net(x)
    if loss_name == "Lovasz":
        loss = R.lovasz2d(logits, t)
    else:
        loss = R.cross_entropy2d(logits, t, weight=torch.tensor([1.6248, 5.762827]))
    loss.backward()
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for loss_name in ("CrossEntropy", "Lovasz"):
        keys = [k[5:] for k in np.load(os.path.join(here, "train_step_{}.npz".format(loss_name))).files if k.startswith("grad/")]
        base = step(loss_name, 0.0, 0)
        for eps in (2e-7, 1e-6):
            worst = {k: 0.0 for k in keys}
            for seed in (1, 2, 3):
                g = step(loss_name, eps, seed)
                for k in keys:
                    worst[k] = max(worst[k], float((g[k] - base[k]).abs().max()) / max(1e-8, float(base[k].abs().max())))
            top = sorted(worst.items(), key=lambda kv: -kv[1])
            print("{:12s} stem output noise {:.0e} rms relative, 3 draws: ".format(loss_name, eps)
                  + "  ".join("{} {:.4f}".format(k.replace("resnet.", ""), v) for k, v in top[:5]))


if __name__ == "__main__":
    main()
