"""Generate tests/golden/*.npz by running the UNMODIFIED reference (dev container only).

    python tests/golden/make_golden.py

The reference's own tests hold no vectors for unet/losses/metrics, so the pins are
outputs of the reference itself: ``robosat.unet.UNet``, ``robosat.losses.*`` and ``robosat.metrics.Metrics``
imported from /root/reference through ``oracle/refshim.py`` and executed on CPU (fp32) with the seeded
parameters/inputs of ``oracle/seeded.py``.  The fixtures are small and committed; this script is their provenance.
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import refshim, seeded  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# parameters whose full gradients are stored (the rest are pinned by their L2 norms)
FULL_GRADS = [
    "final.weight", "final.bias", "dec5.block.weight", "dec4.block.block.weight", "resnet.conv1.weight",
    "resnet.bn1.weight", "resnet.bn1.bias", "resnet.layer1.0.conv1.weight", "resnet.layer1.0.bn3.weight",
    "resnet.layer4.2.bn3.bias", "resnet.layer2.0.downsample.0.weight",
]
TRACKED_BN = ["resnet.bn1", "resnet.layer1.0.bn2", "resnet.layer3.0.downsample.1", "resnet.layer4.2.bn3"]


def ref_unet(ref, num_classes, seed):
    net = ref.unet.UNet(num_classes, pretrained=False)
    net.load_state_dict(seeded.seeded_state_dict(net.state_dict(), seed))
    return net


def golden_forward(ref):
    for tag, (n, c, h, w, k, seed) in {"c2_64": (1, 3, 64, 64, 2, 0), "c3_64x128": (2, 3, 64, 128, 3, 1)}.items():
        net = ref_unet(ref, k, seed).eval()
        x = seeded.synthetic_images(n, c, h, w, seed)
        with torch.no_grad():
            logits = net(x)
            probs = torch.nn.functional.softmax(logits, dim=1)
        np.savez_compressed(
            os.path.join(OUT, "unet_fwd_{}.npz".format(tag)),
            shape=np.array([n, c, h, w, k, seed]), logits=logits.numpy(), probs=probs.numpy(),
        )
        print("forward", tag, float(logits.abs().mean()))


def golden_losses(ref):
    out = {}
    for tag, (n, c, h, w, seed) in {"c2": (2, 2, 32, 32, 3), "c4": (2, 4, 16, 16, 4)}.items():
        g = torch.Generator().manual_seed(77 + seed)
        logits = (torch.randn(n, c, h, w, generator=g) * 2).requires_grad_(True)
        targets = seeded.synthetic_targets(n, c, h, w, seed)
        weight = torch.tensor([1.6248, 5.762827, 2.5, 0.75][:c])
        crits = {
            "CrossEntropy": ref.losses.CrossEntropyLoss2d(weight=weight),
            "Focal": ref.losses.FocalLoss2d(weight=weight),
            "mIoU": ref.losses.mIoULoss2d(weight=weight),
            "Lovasz": ref.losses.LovaszLoss2d(),
        }
        out[tag + "_logits"] = logits.detach().numpy()
        out[tag + "_targets"] = targets.numpy()
        out[tag + "_weight"] = weight.numpy()
        for name, crit in crits.items():
            logits.grad = None
            loss = crit(logits, targets)
            loss.backward()
            out["{}_{}_loss".format(tag, name)] = np.float32(loss.item())
            out["{}_{}_grad".format(tag, name)] = logits.grad.numpy().copy()
            print("loss", tag, name, loss.item())
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)


def golden_metrics(ref):
    g = torch.Generator().manual_seed(5)
    out = {}
    m = ref.metrics.Metrics(range(2))
    scores = torch.randn(3, 2, 32, 32, generator=g)
    actual = seeded.synthetic_targets(3, 2, 32, 32, 9)
    for a, s in zip(actual, scores):
        m.add(a, s)
    out["scores"], out["actual"] = scores.numpy(), actual.numpy()
    out["counts"] = np.array([m.tn, m.fn, m.fp, m.tp], dtype=np.int64)
    out["scores3"] = np.array([m.get_miou(), m.get_fg_iou(), m.get_mcc()], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics", out["counts"], out["scores3"])


def golden_train_step(ref):
    """One fwd + loss + bwd in train mode (BatchNorm batch statistics), 2x3x128x128, 2 classes."""

    for loss_name in ("CrossEntropy", "Lovasz"):
        torch.manual_seed(0)
        net = ref_unet(ref, 2, 2).train()
        x = seeded.synthetic_images(2, 3, 128, 128, 2)
        t = seeded.synthetic_targets(2, 2, 128, 128, 2)
        weight = torch.tensor([1.6248, 5.762827])
        crit = ref.losses.CrossEntropyLoss2d(weight=weight) if loss_name == "CrossEntropy" else ref.losses.LovaszLoss2d()
        logits = net(x)
        loss = crit(logits, t)
        loss.backward()
        out = {"loss": np.float32(loss.item()), "logits": logits.detach().numpy()}
        names, norms = [], []
        for name, p in net.named_parameters():
            if p.grad is None:
                continue
            names.append(name)
            norms.append(float(p.grad.double().norm()))
            if name in FULL_GRADS:
                out["grad/" + name] = p.grad.numpy().copy()
        out["grad_names"] = np.array(names)
        out["grad_norms"] = np.array(norms, dtype=np.float64)
        sd = net.state_dict()
        for bn in TRACKED_BN:
            out["bn/" + bn + ".running_mean"] = sd[bn + ".running_mean"].numpy().copy()
            out["bn/" + bn + ".running_var"] = sd[bn + ".running_var"].numpy().copy()
            out["bn/" + bn + ".num_batches_tracked"] = sd[bn + ".num_batches_tracked"].numpy().copy()
        np.savez_compressed(os.path.join(OUT, "train_step_{}.npz".format(loss_name)), **out)
        print("train", loss_name, loss.item(), len(names))


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    ref = refshim.load_reference()
    golden_forward(ref)
    golden_losses(ref)
    golden_metrics(ref)
    golden_train_step(ref)
