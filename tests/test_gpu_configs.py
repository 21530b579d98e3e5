"""The BASELINE.json configurations on the MI355X at their FULL sizes, through the public operator
surface -- tile and row-size dispatch depends on M and the grid, so "same kernels as the small tests" is checked, not
inferred:

  configs[1]  rs predict bs 16, 3x512x512 fp32 (the headline benchmark shape) -> oracle parity on four of the 16 tiles
              (<= 1e-3) and bit-for-bit batch independence of all 16;
  configs[2]  rs train bf16 bs 32, 3x512x512 (the train leg's shape) -> one full training step against the fp32 CPU
              oracle on the same seeded weights / batch: loss within 2 %, decoder + head gradients cosine >= 0.98, every
              gradient finite, mean cosine over all 168 tensors reported and held to >= 0.90;

  configs[3]  rs predict 1024x1024 3-band tiles, bs 8, fp32      -> oracle parity on three of the eight 1024^2 tiles (the
              CPU oracle needs ~15 s per such tile) + the size-independent property that a tile's probabilities do not
              depend on its batch neighbours (bs 8 vs bs 1, bit-for-bit, all eight tiles: every output pixel's reduction
              order is fixed by the kernel);
  configs[4]  4-band (RGB+IR) multi-class (4 classes) train with the Lovasz loss -> one full fp32 training step against
              the CPU oracle on the same seeded weights (loss, logits, every parameter gradient) at 2 x 4 x 128^2, and the
              configuration AS BASELINE WORDS IT -- bs 32, 4 x 512^2, 4 classes, bf16 -- as one training step against the
              fp32 oracle (loss 2 %, decoder cosine >= 0.98, mean >= 0.90, worst tensor >= 0.85), with the Lovasz kernel
              alone held to 2e-5 / 2e-3 on 1 048 576 keys per image.
"""

import pytest
import torch

from oracle import robosat_ref as R, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(num_classes, seed, in_channels=3, compute_dtype=torch.float32):
    from robosat_amd.unet import UNet

    ref = R.UNetRef(num_classes, in_channels=in_channels)
    sd = seeded.seeded_state_dict(ref.state_dict(), seed)
    ref.load_state_dict(sd)
    net = UNet(num_classes, pretrained=False, in_channels=in_channels, compute_dtype=compute_dtype)
    net.load_state_dict(sd)
    return ref, net.to(DEV)


def test_cfg4_predict_1024_bs8():
    ref, net = _pair(2, 21)
    ref.eval()
    net.eval()
    x = seeded.synthetic_images(8, 3, 1024, 1024, seed=9)
    got8 = net.predict_probs(x.to(DEV))
    assert got8.shape == (8, 2, 1024, 1024)
    pick = [0, 3, 7]  # three of the eight tiles through the CPU oracle (~15 s each), the first, a middle one and the last
    want = R.predict_probs(ref, x[pick])
    err = float((got8[pick].cpu() - want).abs().max())
    print("cfg4 1024^2: max|dprob| vs oracle over tiles", pick, err)
    assert err <= 1e-3  # north_star tolerance (fp32)
    for i in range(8):  # batch independence, EVERY tile: same numbers whether it travels alone or in the batch of 8
        got1 = net.predict_probs(x[i:i + 1].to(DEV))
        assert torch.equal(got1[0], got8[i]), i
    s = got8.sum(1)
    assert float((s - 1).abs().max()) <= 1e-5  # softmax rows sum to one


def test_cfg5_four_band_four_class_lovasz_train_step():
    from robosat_amd import losses

    n, c, k, size = 2, 4, 4, 128
    x = seeded.synthetic_images(n, c, size, size, 3)
    t = seeded.synthetic_targets(n, k, size, size, 3)
    ref, net = _pair(k, 7, in_channels=c)
    ref.train()
    out = ref(x)
    rl = R.lovasz2d(out, t)
    rl.backward()

    net.train()
    crit = losses.LovaszLoss2d().to(DEV)
    logits = net(x.to(DEV))
    loss = crit(logits, t.to(DEV))
    loss.backward()
    print("cfg5 loss", loss.item(), "oracle", rl.item())
    assert float((logits.detach().cpu() - out.detach()).abs().max()) <= 2e-3 * max(1.0, float(out.abs().max()))
    assert abs(loss.item() - rl.item()) <= 1e-3 * max(1.0, abs(rl.item()))
    rp = dict(ref.named_parameters())
    worst = 0.0
    for name, p in net.named_parameters():
        want = rp[name].grad
        if want is None:
            assert p.grad is None, name
            continue
        wn = float(want.norm())
        if wn < 1e-9:
            continue
        rel = float((p.grad.cpu() - want).norm()) / wn
        worst = max(worst, rel)
        assert rel <= 3e-2, (name, rel)
    print("cfg5 worst relative gradient error", worst)
    assert tuple(net.resnet.conv1.weight.grad.shape) == (64, 4, 7, 7)

    # bf16 variant of the same step (BASELINE configs[2] precision on configs[4] shapes)
    _, nb = _pair(k, 7, in_channels=c, compute_dtype=torch.bfloat16)
    nb.train()
    lb = crit(nb(x.to(DEV)), t.to(DEV))
    lb.backward()
    assert abs(lb.item() - rl.item()) <= 2e-2 * max(1.0, abs(rl.item()))
    for name, p in nb.named_parameters():
        if rp[name].grad is not None:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name


def test_cfg2_predict_bs16_512_fp32_headline_shape():
    ref, net = _pair(2, 31)
    ref.eval()
    net.eval()
    x = seeded.synthetic_images(16, 3, 512, 512, seed=12)
    got = net.predict_probs(x.to(DEV))
    assert got.shape == (16, 2, 512, 512) and got.dtype == torch.float32
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for i in (0, 5, 10, 15):  # a quarter of the batch through the CPU oracle (~4 s per tile)
        want = R.predict_probs(ref, x[i:i + 1])
        err = float((got[i:i + 1].cpu() - want).abs().max())
        print("cfg2 bs16 512^2 tile", i, "max|dprob| vs oracle", err)
        assert err <= 1e-3  # north_star tolerance (fp32)
    # ... and ALL 16 tiles ride on those four: a tile's probabilities are bit-for-bit the same whether it travels alone, in a
    # pair or in the batch of 16 (every output pixel's reduction order is fixed by the layer geometry, never by the batch)
    for i in range(16):
        alone = net.predict_probs(x[i:i + 1].to(DEV))
        assert torch.equal(alone[0], got[i]), i
    pair = net.predict_probs(x[[5, 9]].to(DEV))
    assert torch.equal(pair[0], got[5]) and torch.equal(pair[1], got[9])
    assert float((got.sum(1) - 1).abs().max()) <= 1e-5


def test_cfg3_train_bs32_512_bf16_step_vs_oracle():
    from robosat_amd import losses

    import psutil

    if psutil.virtual_memory().available < 64e9:  # the fp32 CPU oracle keeps ~1.2 GB of autograd state per 512^2 tile
        # NOT a skip: a full-size training test that silently vanishes on a smaller box would read as green
        pytest.xfail("host has < 64 GB free for the bs-32 CPU oracle step: configs[2] at full size NOT checked on this box")
    n, size = 32, 512
    x = seeded.synthetic_images(n, 3, size, size, 13)
    t = seeded.synthetic_targets(n, 2, size, size, 13)
    ref, net = _pair(2, 33, compute_dtype=torch.bfloat16)
    torch.set_num_threads(min(32, torch.get_num_threads()))  # (torch's intra-op pool degrades badly on a 256-thread host)
    ref.train()
    out = ref(x)
    rl = R.lovasz2d(out, t)
    rl.backward()
    rgrads = {k: p.grad for k, p in ref.named_parameters()}
    del out

    net.train()
    crit = losses.LovaszLoss2d().to(DEV)
    logits = net(x.to(DEV))
    loss = crit(logits, t.to(DEV))
    loss.backward()
    print("cfg3 bf16 bs32 512^2 loss", loss.item(), "oracle fp32", rl.item())
    assert abs(loss.item() - rl.item()) <= 2e-2 * max(1.0, abs(rl.item()))
    cos = {}
    for name, p in net.named_parameters():
        want = rgrads[name]
        if want is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        if float(want.norm()) < 1e-9:
            continue
        g = p.grad.float().cpu()
        cos[name] = float((g * want).sum() / (g.norm() * want.norm() + 1e-30))
    mean, worst = sum(cos.values()) / len(cos), min(cos.items(), key=lambda kv: kv[1])
    print("cfg3 gradient cosine vs fp32 oracle: mean {:.4f}, worst {} {:.4f} ({} tensors)".format(mean, worst[0], worst[1], len(cos)))
    assert len(cos) >= 160
    for name, c in cos.items():
        if name.startswith(("dec", "center", "final")):
            assert c >= 0.98, (name, c)
    # 16 384 samples per channel in the deepest BatchNorm (vs 32 in the bs-2 128^2 calibration test): bf16 storage noise
    # averages out and the encoder's gradients line up with fp32 far better than at toy sizes
    assert mean >= 0.90, mean
    # ... and no single tensor may point somewhere else: the WORST encoder gradient is bounded too, at what is measured
    # (round 3: resnet.bn1.bias 0.871; the bar names the tensor when it fails)
    assert worst[1] >= 0.85, "worst gradient cosine: {} {:.4f}".format(*worst)


def test_cfg5_train_bs32_512_bf16_four_band_four_class_step_vs_oracle():
    """BASELINE configs[4] at its full size: bs 32, 4 bands (RGB + IR), 4 classes, 512^2, Lovasz (1 048 576 keys per image),
    bf16 -- one full training step against the fp32 CPU oracle on the same seeded weights and batch (the oracle needs all 32
    tiles too: train-mode BatchNorm statistics are over the batch).  The fp32 Lovasz of the bf16 path's OWN logits is also
    held to the oracle's loss function tightly: that isolates the 1 M-key sort / scan from bf16 noise in the network."""
    from robosat_amd import losses

    import psutil

    if psutil.virtual_memory().available < 64e9:
        pytest.xfail("host has < 64 GB free for the bs-32 CPU oracle step: configs[4] at full size NOT checked on this box")
    n, bands, k, size = 32, 4, 4, 512
    x = seeded.synthetic_images(n, bands, size, size, 15)
    t = seeded.synthetic_targets(n, k, size, size, 15)
    ref, net = _pair(k, 35, in_channels=bands, compute_dtype=torch.bfloat16)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref.train()
    out = ref(x)
    rl = R.lovasz2d(out, t)
    rl.backward()
    rgrads = {name: p.grad for name, p in ref.named_parameters()}
    del out

    net.train()
    crit = losses.LovaszLoss2d().to(DEV)
    logits = net(x.to(DEV))
    assert tuple(logits.shape) == (n, k, size, size) and logits.dtype == torch.float32
    loss = crit(logits, t.to(DEV))
    loss.backward()
    print("cfg5 bf16 bs32 4x512^2 C=4 loss", loss.item(), "oracle fp32", rl.item())
    assert abs(loss.item() - rl.item()) <= 2e-2 * max(1.0, abs(rl.item()))
    # the loss kernel alone, at this size, on these very logits (two images through the CPU oracle's loss: 2 x 1 M keys)
    sub = logits.detach()[:2].cpu().requires_grad_(True)
    want = R.lovasz2d(sub, t[:2])
    want.backward()
    got_in = logits.detach()[:2].clone().requires_grad_(True)
    got = crit(got_in, t[:2].to(DEV))
    got.backward()
    assert abs(got.item() - want.item()) <= 2e-5 * max(1.0, abs(want.item()))
    from lovasz_check import assert_lovasz_grad_close  # (per group of equal errors: the reference's sort is not stable)

    rel = assert_lovasz_grad_close(sub.detach(), t[:2], got_in.grad.cpu(), sub.grad, 2e-3)
    print("cfg5 Lovasz kernel on 2 x 1M keys: loss", got.item(), "oracle", want.item(), "gradient max err over tie groups",
          "{:.2e} of the largest entry".format(rel))

    cos = {}
    for name, p in net.named_parameters():
        want_g = rgrads[name]
        if want_g is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        if float(want_g.norm()) < 1e-9:
            continue
        g = p.grad.float().cpu()
        cos[name] = float((g * want_g).sum() / (g.norm() * want_g.norm() + 1e-30))
    assert tuple(net.resnet.conv1.weight.grad.shape) == (64, 4, 7, 7)
    mean, worst = sum(cos.values()) / len(cos), min(cos.items(), key=lambda kv: kv[1])
    print("cfg5 gradient cosine vs fp32 oracle: mean {:.4f}, worst {} {:.4f} ({} tensors)".format(mean, worst[0], worst[1], len(cos)))
    for name, c in cos.items():
        if name.startswith(("dec", "center", "final")):
            assert c >= 0.98, (name, c)
    assert mean >= 0.90, mean
    assert worst[1] >= 0.85, "worst gradient cosine: {} {:.4f}".format(*worst)  # (round 3 measured 0.895)
