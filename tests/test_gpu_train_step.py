"""One full training step on the MI355X (train-mode forward with batch statistics, loss, hand-scheduled backward)
against the golden vectors produced by the unmodified reference (tests/golden/make_golden.py: 2x3x128x128, 2 classes).
"""

import os

import numpy as np
import pytest
import torch

from oracle import robosat_ref as R, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("loss_name", ["CrossEntropy", "Lovasz"])
def test_train_step_matches_reference_golden(golden_dir, loss_name):
    from robosat_amd import losses
    from robosat_amd.unet import UNet

    g = np.load(os.path.join(golden_dir, "train_step_{}.npz".format(loss_name)))
    net = UNet(2, pretrained=False)
    net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 2))
    net = net.to(DEV).train()
    x = seeded.synthetic_images(2, 3, 128, 128, 2).to(DEV)
    t = seeded.synthetic_targets(2, 2, 128, 128, 2).to(DEV)
    crit = (losses.CrossEntropyLoss2d(weight=torch.tensor([1.6248, 5.762827])) if loss_name == "CrossEntropy" else losses.LovaszLoss2d()).to(DEV)

    logits = net(x)
    loss = crit(logits, t)
    loss.backward()

    err_logits = float(np.abs(logits.detach().cpu().numpy() - g["logits"]).max())
    print(loss_name, "loss", loss.item(), float(g["loss"]), "max|dlogits|", err_logits)
    assert err_logits <= 2e-3 * max(1.0, float(np.abs(g["logits"]).max()))
    assert abs(loss.item() - float(g["loss"])) <= 1e-3 * max(1.0, abs(float(g["loss"])))

    params = dict(net.named_parameters())
    assert params["resnet.fc.weight"].grad is None  # unused in forward, as in the reference
    worst = 0.0
    for name, want_norm in zip(g["grad_names"], g["grad_norms"]):
        grad = params[str(name)].grad
        assert grad is not None, name
        got_norm = float(grad.double().norm())
        rel = abs(got_norm - want_norm) / max(want_norm, 1e-12)
        worst = max(worst, rel)
        assert rel <= 2e-2, (str(name), got_norm, float(want_norm))
    print("worst grad-norm relative error", worst)
    # Elementwise bound on the golden's eleven gradient tensors.  The step amplifies rounding (53 train-mode BatchNorms; the Lovasz
    # gradient is piecewise constant in the SORT ORDER of the errors), and how far is measured, not assumed: the reference itself,
    # with resnet.conv1's output perturbed at fp32-rounding level, moves these tensors by up to 0.6 % (CrossEntropy) / 2.2 % (Lovasz)
    # of their largest element (tests/golden/grad_noise_floor.py), and this path with the stem's output taken from four equally
    # accurate sources -- the kernel, the GPU box's own host convolution, a float64 convolution rounded -- lands anywhere between
    # 0.3 % and 2.1 % on resnet.bn1.bias under Lovasz (profiles/r06/golden_noise_floor.txt; rounds 1-5 sat at 0.2 % because the old
    # stem kernel happened to add in the golden machine's order).  The norms above are not chaotic and keep their 2 % bound.
    elementwise = 2e-2 if loss_name == "CrossEntropy" else 5e-2
    worst_el, worst_cos = (0.0, None), (1.0, None)
    for key in g.files:
        if key.startswith("grad/"):
            want = torch.from_numpy(g[key])
            got = params[key[5:]].grad.cpu()
            err = float((got - want).abs().max())
            worst_el = max(worst_el, (err / max(1e-8, float(want.abs().max())), key))
            worst_cos = min(worst_cos, (float(torch.nn.functional.cosine_similarity(got.flatten().double(), want.flatten().double(), dim=0)), key))
            assert err <= elementwise * max(1e-8, float(want.abs().max())), (key, err, float(want.abs().max()))
    print("worst elementwise gradient error relative to the tensor's largest element", worst_el, "worst cosine", worst_cos)
    assert worst_cos[0] >= 0.9995, worst_cos  # (measured 0.99993-0.99998: direction is not chaotic, single elements are)
    sd = net.state_dict()
    for key in g.files:
        if key.startswith("bn/"):
            want = torch.from_numpy(g[key])
            got = sd[key[3:]].cpu()
            if want.dtype == torch.int64:
                assert int(got) == int(want), key
            else:
                assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), key


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_optimizer_updates_that_bypass_version_counters_are_seen(dtype):
    """``torch.optim.Adam(fused=True)`` (what ``rs train`` uses on the GPU) rewrites the parameters without bumping their
    version counters; the derived copies the kernels read (bf16 casts, packed phase filters, folded BatchNorm, captured
    graphs) must follow all the same.  Three steps with the fused and with the default optimizer from one initial state
    give the same losses, and the eval-mode output afterwards is the oracle's on the trained state dict."""
    from robosat_amd import losses
    from robosat_amd.unet import UNet

    x = seeded.synthetic_images(2, 3, 64, 64, 5).to(DEV)
    t = seeded.synthetic_targets(2, 2, 64, 64, 5).to(DEV)
    init = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 5)
    runs = {}
    for fused in (False, True):
        net = UNet(2, pretrained=False, compute_dtype=dtype)
        net.load_state_dict(init)
        net = net.to(DEV)
        net.eval()
        before = net.predict_probs(x).cpu()  # (fills the eval-mode caches that must not survive the steps)
        net.train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=fused)
        crit = losses.CrossEntropyLoss2d(weight=torch.tensor([1.0, 2.0])).to(DEV)
        ls = []
        for _ in range(3):
            opt.zero_grad()
            loss = crit(net(x), t)
            loss.backward()
            opt.step()
            ls.append(float(loss))
        net.eval()
        after = net.predict_probs(x).cpu()
        ref = R.UNetRef(2)
        ref.load_state_dict({k: v.detach().float().cpu() for k, v in net.state_dict().items()})
        ref.eval()
        want = R.predict_probs(ref, x.cpu())
        tol = 1e-3 if dtype == "fp32" else 3e-2
        assert float((after - want).abs().max()) <= tol, (fused, float((after - want).abs().max()))
        assert float((after - before).abs().max()) > 1e-3  # the steps did move the output
        runs[fused] = ls
    print(runs)
    assert runs[True][0] == pytest.approx(runs[False][0], rel=1e-5)
    for a, b in zip(runs[True], runs[False]):  # later losses see the updates: a stale weight copy shows here
        assert a == pytest.approx(b, rel=2e-2 if dtype == "bf16" else 2e-3)


@pytest.mark.parametrize("dtype,loss_name", [("bf16", "Lovasz"), ("fp32", "CrossEntropy")])
def test_graphed_train_step_is_the_eager_step(dtype, loss_name):
    """The hipGraph replay of the training step (robosat_amd.graph.TrainStepGraph: what `rs train` and the bench's train legs
    run) launches the very kernels of the eager step in the same order on the same two streams: six steps over changing
    batches give bit-identical losses, logits, parameters, BatchNorm buffers and Adam state either way -- and a batch of
    another shape in the middle falls back to the eager step without disturbing the graph."""
    from robosat_amd import losses
    from robosat_amd.graph import TrainStepGraph
    from robosat_amd.unet import UNet

    init = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 9)
    batches = [(seeded.synthetic_images(2, 3, 128, 192, 40 + i).to(DEV), seeded.synthetic_targets(2, 2, 128, 192, 40 + i).to(DEV))
               for i in range(6)]
    odd = (seeded.synthetic_images(1, 3, 64, 64, 77).to(DEV), seeded.synthetic_targets(1, 2, 64, 64, 77).to(DEV))
    order = batches[:4] + [odd] + batches[4:]
    runs = {}
    for graphed in (False, True):
        net = UNet(2, pretrained=False, compute_dtype=dtype)
        net.load_state_dict(init)
        net = net.to(DEV).train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
        crit = (losses.LovaszLoss2d() if loss_name == "Lovasz" else losses.CrossEntropyLoss2d(weight=torch.tensor([1.0, 3.0]))).to(DEV)
        stepper = TrainStepGraph(net, crit, opt, warmup=2, enabled=graphed)
        ls, outs = [], []
        for i, (x, t) in enumerate(order):
            loss, logits = stepper(x, t)
            ls.append(loss.clone())  # (the graph's static outputs are overwritten by the next replay)
            outs.append(logits.clone())
            assert stepper.captured == (graphed and i >= 2), (graphed, i)
        torch.cuda.synchronize()
        net.eval()
        runs[graphed] = {"loss": torch.stack(ls).cpu(), "outs": [o.cpu() for o in outs], "probs": net.predict_probs(batches[0][0]).cpu(),
                         "state": {k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
                         "adam": [st["exp_avg_sq"].detach().cpu().clone() for st in opt.state.values()],
                         "steps": {float(st["step"]) for st in opt.state.values()}}
    a, b = runs[False], runs[True]
    print(dtype, "losses eager", a["loss"].tolist(), "graphed", b["loss"].tolist())
    assert torch.equal(a["loss"], b["loss"])
    for x, y in zip(a["outs"], b["outs"]):
        assert torch.equal(x, y)
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k
    for x, y in zip(a["adam"], b["adam"]):
        assert torch.equal(x, y)
    assert a["steps"] == b["steps"] == {7.0}
    assert torch.equal(a["probs"], b["probs"])  # eval after graph replays reads the CURRENT weights (derived caches dropped)
    assert int(b["state"]["resnet.bn1.num_batches_tracked"]) == 7
