"""Pins the CPU oracle (oracle/robosat_ref.py): (1) against the committed golden vectors, which were produced by
the UNMODIFIED reference (tests/golden/make_golden.py); (2) against the live reference wherever /root/reference
exists (dev container).  CPU only."""

import os

import numpy as np
import pytest
import torch

from oracle import refshim, robosat_ref as R, seeded


def _oracle_net(num_classes, seed):
    net = R.UNetRef(num_classes)
    net.load_state_dict(seeded.seeded_state_dict(net.state_dict(), seed))
    return net


@pytest.mark.parametrize("tag", ["c2_64", "c3_64x128"])
def test_forward_matches_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "unet_fwd_{}.npz".format(tag)))
    n, c, h, w, k, seed = [int(v) for v in g["shape"]]
    net = _oracle_net(k, seed).eval()
    probs = R.predict_probs(net, seeded.synthetic_images(n, c, h, w, seed)).numpy()
    # same torch build + same ops => the restatement reproduces the reference to rounding
    assert np.abs(probs - g["probs"]).max() <= 1e-6


@pytest.mark.parametrize("tag", ["c2", "c4"])
@pytest.mark.parametrize("name", ["CrossEntropy", "Focal", "mIoU", "Lovasz"])
def test_losses_match_golden(golden_dir, tag, name):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    logits = torch.from_numpy(g[tag + "_logits"]).requires_grad_(True)
    targets = torch.from_numpy(g[tag + "_targets"])
    weight = torch.from_numpy(g[tag + "_weight"])
    fn = R.LOSSES[name]
    loss = fn(logits, targets) if name == "Lovasz" else fn(logits, targets, weight=weight)
    loss.backward()
    assert abs(loss.item() - float(g["{}_{}_loss".format(tag, name)])) <= 1e-6 * max(1.0, abs(loss.item()))
    assert np.abs(logits.grad.numpy() - g["{}_{}_grad".format(tag, name)]).max() <= 1e-7


def test_metrics_match_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    counts = np.zeros(4, dtype=np.int64)
    for a, s in zip(torch.from_numpy(g["actual"]), torch.from_numpy(g["scores"])):
        counts += np.array(R.confusion_counts(a, s))
    assert (counts == g["counts"]).all()
    assert np.allclose(R.metric_scores(*counts), g["scores3"], rtol=0, atol=1e-12)


def test_state_dict_layout():
    """329 keys / 39 390 314 parameters -- the checkpoint contract."""
    net = R.UNetRef(2)
    sd = net.state_dict()
    assert len(sd) == 329
    assert sum(p.numel() for p in net.parameters()) == 39390314
    assert tuple(sd["dec0.block.block.weight"].shape) == (256, 2304, 3, 3)
    assert tuple(sd["final.weight"].shape) == (2, 32, 1, 1)


def test_quantize_quirk():
    q = R.quantize_probs(np.array([0.0, 0.5, 1.0 - 1e-9, 1.0]))
    assert q.tolist() == [1, 128, 255, 0]  # p == 1.0 wraps to bin 0 (reference quirk, predict.py:102-103)


@pytest.mark.skipif(not refshim.available(), reason="reference tree only exists in the dev container")
def test_restatement_equals_live_reference():
    ref = refshim.load_reference()
    theirs = ref.unet.UNet(2, pretrained=False)
    ours = R.UNetRef(2)
    assert list(theirs.state_dict().keys()) == list(ours.state_dict().keys())
    sd = seeded.seeded_state_dict(theirs.state_dict(), 7)
    theirs.load_state_dict(sd)
    ours.load_state_dict(sd)
    x = seeded.synthetic_images(1, 3, 64, 64, 7)
    t = seeded.synthetic_targets(1, 2, 64, 64, 7)
    for mode in ("eval", "train"):
        getattr(theirs, mode)()
        getattr(ours, mode)()
        a, b = theirs(x), ours(x)
        assert torch.equal(a, b)
    la = ref.losses.LovaszLoss2d()(a, t)
    lb = R.lovasz2d(b, t)
    assert abs(la.item() - lb.item()) <= 1e-6
