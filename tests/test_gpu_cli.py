"""End to end on the MI355X: ``rs train`` then ``rs predict`` on a synthetic slippy-map dataset (BASELINE configs[0]
shape: 256x256 2-class tiles, 1 epoch), checking the reference's artifacts (log lines, history plot, checkpoint layout,
probability PNGs) and parity with the CPU oracle fed the very same batches / checkpoint."""

import argparse
import os
import random
import re

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import robosat_ref as R, seeded
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rs_train_then_predict(tmp_path):
    from robosat_amd.tools import predict as predict_tool
    from robosat_amd.tools import train as train_tool

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=8, n_val=4, size=256)
    ckdir = os.path.join(tmp, "pth")
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, ckdir, loss="Lovasz", batch_size=2, image_size=256, epochs=1)

    random.seed(0)
    torch.manual_seed(0)
    train_tool.main(argparse.Namespace(model=model_toml, dataset=ds_toml, checkpoint=None, resume=False, workers=0))

    # --- artifacts, exactly as the reference names them (tools/train.py:114-160) ---
    log = open(os.path.join(ckdir, "log")).read().splitlines()
    assert log[0].startswith("--- Hyper Parameters on Dataset: ") and "Epoch: 1/1" in log
    pat = r"^(Train   |Validate) loss: \d+\.\d{4}, mIoU: (\d\.\d{3}|nan), parking IoU: (\d\.\d{3}|nan), MCC: (-?\d\.\d{3}|nan)$"
    assert sum(bool(re.match(pat, l)) for l in log) == 2, log
    assert os.path.exists(os.path.join(ckdir, "history-00001-of-00001.png"))
    ck_path = os.path.join(ckdir, "checkpoint-00001-of-00001.pth")
    ck = torch.load(ck_path, map_location="cpu")
    assert set(ck) == {"epoch", "state_dict", "optimizer"} and ck["epoch"] == 1
    ref = R.UNetRef(2)
    assert list(ck["state_dict"]) == ["module." + k for k in ref.state_dict()]
    assert len(ck["optimizer"]["state"]) == 168 and len(ck["optimizer"]["param_groups"][0]["params"]) == 170
    # the checkpoint loads into the reference architecture as-is
    ref.load_state_dict({k[len("module."):]: v for k, v in ck["state_dict"].items()})
    assert int(ck["state_dict"]["module.resnet.bn1.num_batches_tracked"]) == 4  # 8 tiles / batch 2

    # --- rs predict on the validation tiles; compare with the oracle run on the same checkpoint ---
    tiles_dir, probs_dir = os.path.join(ds_root, "validation", "images"), os.path.join(tmp, "probs")
    predict_tool.main(argparse.Namespace(batch_size=2, checkpoint=ck_path, overlap=32, tile_size=256, workers=0,
                                         tiles=tiles_dir, probs=probs_dir, model=model_toml, dataset=ds_toml))
    from robosat_amd.datasets import BufferedSlippyMapDirectory
    from robosat_amd.transforms import Compose, ConvertImageMode, ImageToTensor, Normalize

    tf = Compose([ConvertImageMode("RGB"), ImageToTensor(), Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    directory = BufferedSlippyMapDirectory(tiles_dir, transform=tf, size=256, overlap=32)
    assert len(directory) == 4
    ref.eval()
    worst = 0
    for i in range(len(directory)):
        image, tile = directory[i]
        x, y, z = (int(v) for v in tile)
        png = Image.open(os.path.join(probs_dir, str(z), str(x), "{}.png".format(y)))
        assert png.mode == "P" and png.size == (256, 256)
        got = np.array(png).astype(np.int32)
        probs = directory.unbuffer(R.predict_probs(ref, image.unsqueeze(0))[0].numpy())
        want = R.quantize_probs(probs[1]).astype(np.int32)
        diff = np.abs(got - want)
        diff = np.minimum(diff, 256 - diff)  # bin 0 <-> 255 wrap (p == 1.0 quirk)
        worst = max(worst, int(diff.max()))
        assert (diff > 0).mean() < 0.02  # only pixels sitting on a bin edge may move, and by one bin
    assert worst <= 1


def test_rs_train_several_epochs_then_resume(tmp_path):
    """Checkpoints are written every epoch while training goes on (the saved optimizer state must be a copy: the fused Adam's
    step counters live on the device, the file carries them on the host like a stock Adam's), and `--resume` continues from
    one (reference tools/train.py:85-92,143-147)."""
    from robosat_amd.config import load_config, save_config
    from robosat_amd.tools import train as train_tool

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=8, n_val=4, size=256)
    ckdir = os.path.join(tmp, "pth")
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, ckdir, loss="CrossEntropy", batch_size=2, image_size=256, epochs=2)
    train_tool.main(argparse.Namespace(model=model_toml, dataset=ds_toml, checkpoint=None, resume=False, workers=0))
    ck2 = os.path.join(ckdir, "checkpoint-00002-of-00002.pth")
    ck = torch.load(ck2, map_location="cpu")
    steps = {float(st["step"]) for st in ck["optimizer"]["state"].values()}
    assert steps == {8.0} and ck["epoch"] == 2  # 2 epochs x (8 tiles / batch 2)
    assert all(st["step"].device.type == "cpu" for st in ck["optimizer"]["state"].values())
    assert int(ck["state_dict"]["module.resnet.bn1.num_batches_tracked"]) == 8

    cfg = load_config(model_toml)
    cfg["opt"]["epochs"] = 3
    save_config(cfg, model_toml)
    train_tool.main(argparse.Namespace(model=model_toml, dataset=ds_toml, checkpoint=ck2, resume=True, workers=0))
    ck3 = torch.load(os.path.join(ckdir, "checkpoint-00003-of-00003.pth"), map_location="cpu")
    assert ck3["epoch"] == 3 and {float(st["step"]) for st in ck3["optimizer"]["state"].values()} == {12.0}
    assert int(ck3["state_dict"]["module.resnet.bn1.num_batches_tracked"]) == 12
    log = open(os.path.join(ckdir, "log")).read()
    assert "Epoch: 3/3" in log and "Epoch: 1/3" not in log


def test_device_side_predict_pipeline_is_bit_identical_to_host_steps(tmp_path):
    """N1: uint8 tiles in, quantised bytes out, all on the device, must reproduce the bytes of the
    reference's host-side steps -- ToTensor + Normalize (fp32), softmax, unbuffer, np.digitize -- exactly."""
    from robosat_amd.tools import predict as predict_tool
    from robosat_amd.tools.predict import quantize
    from robosat_amd.transforms import ImageToTensor, Normalize
    from robosat_amd.unet import UNet

    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    net = UNet(2, pretrained=False)
    net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 6))
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (3, 192, 256, 3), generator=g, dtype=torch.uint8)
    u8[0, :, :, :] = 0  # extremes of the input range
    u8[1, :40] = 255
    overlap = 32
    got = net.predict_quantized(u8.to(DEV), overlap=overlap, mean=mean, std=std).cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == (3, 192 - 64, 256 - 64)
    # host steps as the reference does them (PIL -> ToTensor -> Normalize; softmax output -> crop -> digitize)
    norm = Normalize(mean, std)
    x = torch.stack([norm(ImageToTensor()(Image.fromarray(im.numpy(), mode="RGB"))) for im in u8])
    probs = net.predict_probs(x.to(DEV)).cpu().numpy()
    want = np.stack([quantize(p[1:, overlap:-overlap, overlap:-overlap]).squeeze() for p in probs])
    assert np.array_equal(got, want), int((got != want).sum())

    # and through the tool: both pipelines write identical PNG pixels
    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=2, n_val=4, size=256, seed=7)
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, os.path.join(tmp, "pth"), batch_size=2, image_size=256)
    ck_path = os.path.join(tmp, "ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in net.state_dict().items()}}, ck_path)
    tiles_dir = os.path.join(ds_root, "validation", "images")
    outs = {}
    for mode in ("0", "1"):
        os.environ["ROBOSAT_PREDICT_HOST_PIPELINE"] = mode
        try:
            probs_dir = os.path.join(tmp, "probs" + mode)
            predict_tool.main(argparse.Namespace(batch_size=2, checkpoint=ck_path, overlap=32, tile_size=256, workers=0,
                                                 tiles=tiles_dir, probs=probs_dir, model=model_toml, dataset=ds_toml))
        finally:
            os.environ.pop("ROBOSAT_PREDICT_HOST_PIPELINE", None)
        files = sorted(os.path.join(d, f) for d, _, fs in os.walk(probs_dir) for f in fs)
        assert len(files) == 4
        outs[mode] = [np.array(Image.open(f)) for f in files]
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a, b)


def test_training_trajectory_matches_cpu_oracle(tmp_path):
    """Same initial weights, same batches, same Adam: per-step losses and epoch metrics of the GPU path track the CPU
    oracle (fp32 tolerance grows with the step count, Adam normalises even tiny gradient differences)."""
    from robosat_amd import losses
    from robosat_amd.metrics import Metrics
    from robosat_amd.tools.train import get_dataset_loaders
    from robosat_amd.unet import UNet

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=8, n_val=4, size=128, seed=3)
    model = {"common": {"image_size": 128, "batch_size": 2}}
    dataset = {"common": {"dataset": ds_root}}
    random.seed(1)
    loader, _ = get_dataset_loaders(model, dataset, 0)
    batches = [(im.clone(), mk.clone()) for im, mk, _ in loader]
    assert len(batches) == 4

    sd = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 4)
    ref = R.UNetRef(2)
    ref.load_state_dict(sd)
    ref.train()
    net = UNet(2, pretrained=False)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    crit = losses.LovaszLoss2d().to(DEV)
    m, counts = Metrics(range(2)), np.zeros(4, dtype=np.int64)
    for step, (images, masks) in enumerate(batches):
        opt_ref.zero_grad()
        out_ref = ref(images)
        loss_ref = R.lovasz2d(out_ref, masks)
        loss_ref.backward()
        opt_ref.step()
        for a, s in zip(masks, out_ref.detach()):
            counts += np.array(R.confusion_counts(a, s))

        opt.zero_grad()
        out = net(images.to(DEV))
        loss = crit(out, masks.to(DEV))
        loss.backward()
        opt.step()
        m.add_batch(masks.to(DEV), out.detach())
        rel = abs(loss.item() - loss_ref.item()) / max(1.0, abs(loss_ref.item()))
        print("step", step, "loss", loss.item(), "oracle", loss_ref.item())
        assert rel <= 5e-3 * (step + 1)
    want = R.metric_scores(*counts)
    got = (m.get_miou(), m.get_fg_iou(), m.get_mcc())
    print("metrics", got, want)
    assert abs(got[0] - want[0]) <= 0.01 and abs(got[1] - want[1]) <= 0.01


def test_rs_train_predict_masks_four_band_four_class(tmp_path):
    """BASELINE configs[4] through the TOOLS: a 4-band (RGB directory + single-band IR directory), 4-class dataset goes
    through ``rs train`` (Lovasz) -> ``rs predict --extra_tiles`` -> ``rs masks``.  The band layout comes from the dataset
    config's ``image_dirs`` / ``image_modes`` (the reference's dataset layer concatenates image directories on the channel
    axis, datasets.py:44-78); the checkpoint holds a [64,4,7,7] stem; probabilities are one byte per non-background class
    (mode RGB for 4 classes) and match the CPU oracle run on the same checkpoint and the same 4-band composites."""
    from robosat_amd.tools import masks as masks_tool
    from robosat_amd.tools import predict as predict_tool
    from robosat_amd.tools import train as train_tool

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=8, n_val=4, size=256, seed=21, classes=4, ir=True)
    ckdir = os.path.join(tmp, "pth")
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, ckdir, loss="Lovasz", batch_size=2, image_size=256, epochs=1, classes=4,
                                              ir=True)
    random.seed(0)
    torch.manual_seed(0)
    train_tool.main(argparse.Namespace(model=model_toml, dataset=ds_toml, checkpoint=None, resume=False, workers=0))
    log = open(os.path.join(ckdir, "log")).read().splitlines()
    pat = r"^(Train   |Validate) loss: \d+\.\d{4}, mIoU: (\d\.\d{3}|nan), parking IoU: (\d\.\d{3}|nan), MCC: (-?\d\.\d{3}|nan)$"
    assert sum(bool(re.match(pat, l)) for l in log) == 2, log
    ck_path = os.path.join(ckdir, "checkpoint-00001-of-00001.pth")
    ck = torch.load(ck_path, map_location="cpu")
    assert tuple(ck["state_dict"]["module.resnet.conv1.weight"].shape) == (64, 4, 7, 7)
    assert tuple(ck["state_dict"]["module.final.weight"].shape) == (4, 32, 1, 1)
    ref = R.UNetRef(4, in_channels=4)
    assert list(ck["state_dict"]) == ["module." + k for k in ref.state_dict()]
    ref.load_state_dict({k[len("module."):]: v for k, v in ck["state_dict"].items()})

    rgb_dir, ir_dir = os.path.join(ds_root, "validation", "images"), os.path.join(ds_root, "validation", "ir")
    probs_dir = os.path.join(tmp, "probs")
    predict_tool.main(argparse.Namespace(batch_size=2, checkpoint=ck_path, overlap=32, tile_size=256, workers=0, tiles=rgb_dir,
                                         probs=probs_dir, model=model_toml, dataset=ds_toml, extra_tiles=[ir_dir]))
    # without the IR directory the tool must refuse (a 4-band model cannot be fed 3 bands)
    with pytest.raises(SystemExit):
        predict_tool.main(argparse.Namespace(batch_size=2, checkpoint=ck_path, overlap=32, tile_size=256, workers=0, tiles=rgb_dir,
                                             probs=os.path.join(tmp, "nope"), model=model_toml, dataset=ds_toml, extra_tiles=[]))

    from robosat_amd.bands import bands_from_config, split_per_source
    from robosat_amd.config import load_config
    from robosat_amd.datasets import BufferedSlippyMapConcatenation
    from robosat_amd.transforms import Compose, ConvertImageMode, ImageToTensor, Normalize

    bands = bands_from_config(load_config(ds_toml), load_config(model_toml))
    assert bands.channels == 4 and bands.modes == ["RGB", "L"]
    tfs = [Compose([ConvertImageMode(md), ImageToTensor(), Normalize(m, s)])
           for md, m, s in zip(bands.modes, split_per_source(bands, bands.mean), split_per_source(bands, bands.std))]
    directory = BufferedSlippyMapConcatenation([rgb_dir, ir_dir], tfs, bands.modes, size=256, overlap=32, cat_dim=0)
    assert len(directory) == 4
    ref.eval()
    worst = 0
    for i in range(len(directory)):
        image, tile = directory[i]
        assert tuple(image.shape) == (4, 320, 320)
        x, y, z = (int(v) for v in tile)
        png = Image.open(os.path.join(probs_dir, str(z), str(x), "{}.png".format(y)))
        assert png.mode == "RGB" and png.size == (256, 256)  # three non-background classes, one byte each
        got = np.array(png).astype(np.int32)
        probs = directory.unbuffer(R.predict_probs(ref, image.unsqueeze(0))[0].numpy())
        want = np.stack([R.quantize_probs(probs[c]) for c in (1, 2, 3)], axis=-1).astype(np.int32)
        diff = np.abs(got - want)
        diff = np.minimum(diff, 256 - diff)
        worst = max(worst, int(diff.max()))
        assert (diff > 0).mean() < 0.02
    assert worst <= 1

    masks_dir = os.path.join(tmp, "masks")
    masks_tool.main(argparse.Namespace(masks=masks_dir, probs=[probs_dir], weights=None, dataset=ds_toml, batch_size=4))
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(masks_dir) for f in fs)
    assert len(files) == 4
    for f in files:
        m = Image.open(f)
        assert m.mode == "P" and int(np.array(m).max()) <= 3


def test_rs_train_rejects_band_mismatch(tmp_path):
    """`[model] in_channels` that disagrees with the dataset's bands is an error at start-up, not an assertion on
    the first batch."""
    from robosat_amd.config import load_config, save_config
    from robosat_amd.tools import train as train_tool

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=2, n_val=2, size=256)
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, os.path.join(tmp, "pth"), batch_size=2, image_size=256)
    cfg = load_config(model_toml)
    cfg["model"]["in_channels"] = 4
    save_config(cfg, model_toml)
    with pytest.raises(SystemExit, match="in_channels"):
        train_tool.main(argparse.Namespace(model=model_toml, dataset=ds_toml, checkpoint=None, resume=False, workers=0))


def test_bf16_training_reaches_the_fp32_oracles_miou(tmp_path):
    """The metric's second half -- "mIoU vs CPU ref" -- at the precision the train leg runs at (missing 3):
    the same learnable synthetic set, the same initial weights, the same batches in the same order, Adam with the same
    learning rate; the fp32 CPU oracle (reference arithmetic) and the bf16 MI355X path train for 2 epochs and are then
    validated on the same held-out tiles.  Validation mIoU must agree within 0.02 (the confusion counts of
    metrics.py:27-84 on each side) and both must have learned something."""
    from robosat_amd import losses
    from robosat_amd.metrics import Metrics
    from robosat_amd.tools.train import get_dataset_loaders
    from robosat_amd.unet import UNet

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=24, n_val=8, size=128, seed=31)
    model = {"common": {"image_size": 128, "batch_size": 4}}
    dataset = {"common": {"dataset": ds_root}}
    random.seed(2)
    train_loader, val_loader = get_dataset_loaders(model, dataset, 0)
    epochs = []
    for e in range(2):
        train_loader.batch_sampler.set_epoch(e)
        epochs.append([(im.clone(), mk.clone()) for im, mk, _ in train_loader])
    val = [(im.clone(), mk.clone()) for im, mk, _ in val_loader]
    assert len(epochs[0]) == 6 and len(val) == 2

    sd = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 8)
    ref = R.UNetRef(2)
    ref.load_state_dict(sd)
    net = UNet(2, pretrained=False, compute_dtype=torch.bfloat16)
    net.load_state_dict(sd)
    net = net.to(DEV)
    lr = 3e-4
    opt_ref = torch.optim.Adam(ref.parameters(), lr=lr)
    opt = torch.optim.Adam(net.parameters(), lr=lr, fused=True)
    crit = losses.LovaszLoss2d().to(DEV)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref.train()
    net.train()
    for batches in epochs:
        for images, masks in batches:
            opt_ref.zero_grad()
            R.lovasz2d(ref(images), masks).backward()
            opt_ref.step()
            opt.zero_grad()
            crit(net(images.to(DEV)), masks.to(DEV)).backward()
            opt.step()
    ref.eval()
    net.eval()
    counts = np.zeros(4, dtype=np.int64)
    m = Metrics(range(2))
    with torch.no_grad():
        for images, masks in val:
            for a, s in zip(masks, ref(images)):
                counts += np.array(R.confusion_counts(a, s))
            m.add_batch(masks.to(DEV), net(images.to(DEV)))
    want, got = R.metric_scores(*counts)[0], m.get_miou()
    print("validation mIoU after 2 epochs: bf16 MI355X {:.4f}, fp32 CPU oracle {:.4f}".format(got, want))
    assert abs(got - want) <= 0.02, (got, want)
    assert want > 0.6 and got > 0.6  # (an untrained model sits near 0.3-0.45 on this set)
