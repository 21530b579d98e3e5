"""Feeders against the UNMODIFIED reference, on the CPU: the slippy-map datasets, the joint
transform chain of ``rs train`` and the buffered tiles of ``rs predict`` hand the model the same tensors the reference's
own classes do on the same directory.  Needs /root/reference (dev container); skipped where it is absent."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import refshim

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference tree not present")

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]  # train.py:246, predict.py:71


def _reference_modules():
    """The reference's own datasets / transforms modules.  Resize, CenterCrop, Normalize and ToTensor are torchvision's in
    the reference (train.py:14, transforms.py:14): here the refshim stand-ins, so for those four the comparison only says
    both sides follow torchvision's semantics -- the datasets, the joint/random transforms, ConvertImageMode, MaskToTensor
    and the tile buffering are the reference's own code."""
    import types

    refshim.load_reference()  # installs the stand-ins and puts the reference on sys.path
    import robosat.datasets as rd  # noqa: E402  (the reference's own files)
    import robosat.transforms as rt  # noqa: E402
    import torchvision.transforms as tvt  # noqa: E402  (stand-in)

    ns = types.SimpleNamespace(**{k: getattr(rt, k) for k in dir(rt) if not k.startswith("_")})
    ns.Resize, ns.CenterCrop, ns.Normalize, ns.Compose = tvt.Resize, tvt.CenterCrop, tvt.Normalize, tvt.Compose
    return rd, ns


def _chain(t):
    """The training transform of train.py:248-260 (the random flips/rotations draw from `random`)."""
    return t.JointCompose([
        t.JointTransform(t.ConvertImageMode("RGB"), t.ConvertImageMode("P")),
        t.JointTransform(t.Resize((96, 96), 2), t.Resize((96, 96), 0)),  # bilinear / nearest (PIL codes)
        t.JointTransform(t.CenterCrop((64, 64)), t.CenterCrop((64, 64))),
        t.JointRandomHorizontalFlip(0.5),
        t.JointRandomRotation(0.5, 90),
        t.JointRandomRotation(0.5, 90),
        t.JointRandomRotation(0.5, 90),
        t.JointTransform(t.ImageToTensor(), t.MaskToTensor()),
        t.JointTransform(t.Normalize(mean=MEAN, std=STD), None),
    ])


def test_training_dataset_items_match_reference(tmp_path):
    rd, rt = _reference_modules()
    from robosat_amd import datasets as md, transforms as mt

    root = synth.make_dataset(str(tmp_path / "ds"), n_train=6, n_val=2, size=128, seed=11)
    img, lab = os.path.join(root, "training", "images"), os.path.join(root, "training", "labels")
    ours = md.SlippyMapTilesConcatenation([img], lab, _chain(mt))
    ref = rd.SlippyMapTilesConcatenation([img], lab, _chain(rt))
    assert len(ours) == len(ref) == 6
    for i in range(len(ref)):
        random.seed(100 + i)
        a_img, a_mask, a_tiles = ours[i]
        random.seed(100 + i)
        b_img, b_mask, b_tiles = ref[i]
        assert [tuple(t) for t in a_tiles] == [tuple(t) for t in b_tiles]
        assert a_img.dtype == b_img.dtype == torch.float32 and a_mask.dtype == b_mask.dtype == torch.int64
        assert torch.equal(a_img, b_img) and torch.equal(a_mask, b_mask), i


def test_buffered_predict_tiles_match_reference(tmp_path):
    rd, rt = _reference_modules()
    from robosat_amd import datasets as md, transforms as mt

    root = synth.make_dataset(str(tmp_path / "ds"), n_train=8, n_val=0, size=256, seed=12)
    img = os.path.join(root, "training", "images")

    def chain(t):  # predict.py:73
        return t.Compose([t.ConvertImageMode("RGB"), t.ImageToTensor(), t.Normalize(mean=MEAN, std=STD)])

    ours = md.BufferedSlippyMapDirectory(img, transform=chain(mt), size=256, overlap=32)
    ref = rd.BufferedSlippyMapDirectory(img, transform=chain(rt), size=256, overlap=32)
    assert len(ours) == len(ref) == 8
    got = {tuple(t.tolist()): im for im, t in (ours[i] for i in range(len(ours)))}
    for i in range(len(ref)):
        b_img, b_tile = ref[i]
        a_img = got[tuple(b_tile.tolist())]
        assert a_img.shape == (3, 320, 320)
        assert torch.equal(a_img, b_img), b_tile
    probs = np.random.default_rng(0).random((2, 320, 320)).astype(np.float32)
    assert np.array_equal(ours.unbuffer(probs), ref.unbuffer(probs)) and ours.unbuffer(probs).shape == (2, 256, 256)


def test_split_chain_items_match_reference(tmp_path):
    """The default `rs train` loader splits the reference's chain (train.py:248-260) where it stops being deterministic: the
    worker does mode conversion / resize / crop and DRAWS (robosat_amd.datasets.UnaugmentedTiles), the device does the
    transposes + ToTensor + Normalize (rs_augment_tiles; here its CPU statement oracle/tools_ref.augment).  Same seed ->
    bit-equal items to the reference's own dataset + transform classes, and the same number of draws consumed."""
    rd, rt = _reference_modules()
    from oracle import tools_ref as T
    from robosat_amd import datasets as md

    size = 96
    root = synth.make_dataset(str(tmp_path / "ds"), n_train=8, n_val=2, size=128, seed=12)
    img, lab = os.path.join(root, "training", "images"), os.path.join(root, "training", "labels")
    chain = rt.JointCompose([
        rt.JointTransform(rt.ConvertImageMode("RGB"), rt.ConvertImageMode("P")),
        rt.JointTransform(rt.Resize((size, size), 2), rt.Resize((size, size), 0)),
        rt.JointTransform(rt.CenterCrop((size, size)), rt.CenterCrop((size, size))),
        rt.JointRandomHorizontalFlip(0.5),
        rt.JointRandomRotation(0.5, 90),
        rt.JointRandomRotation(0.5, 90),
        rt.JointRandomRotation(0.5, 90),
        rt.JointTransform(rt.ImageToTensor(), rt.MaskToTensor()),
        rt.JointTransform(rt.Normalize(mean=MEAN, std=STD), None),
    ])
    ref = rd.SlippyMapTilesConcatenation([img], lab, chain)
    ours = md.UnaugmentedTiles([img], lab, size, draw=True)
    assert len(ours) == len(ref) == 8
    seen = set()
    for i in range(len(ref)):
        random.seed(300 + i)
        b_img, b_mask, b_tiles = ref[i]
        after_ref = random.random()
        random.seed(300 + i)
        u8, m8, code, tiles = ours[i]
        after_ours = random.random()
        assert after_ref == after_ours  # four draws each: the streams stay aligned for the next item
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (size, size, 3) and m8.dtype == torch.uint8
        assert [tuple(t) for t in tiles] == [tuple(t) for t in b_tiles]
        draws = [0.0 if code & 1 else 1.0] + [0.0] * (code >> 1) + [1.0] * (3 - (code >> 1))
        a_img, a_mask = T.augment(u8.numpy(), m8.numpy(), draws, MEAN, STD)
        assert np.array_equal(a_img, b_img.numpy()) and np.array_equal(a_mask, b_mask.numpy()), (i, code)
        seen.add(code)
    assert len(seen) >= 4  # (the seeds exercise several elements of the flip / rotation group)


def test_four_band_items_are_the_reference_chain_per_source(tmp_path):
    """BASELINE configs[4] feeds RGB + IR: two image directories concatenated on the channel axis by the dataset layer
    (datasets.py:44-78), each converted to ITS mode (RGB / L) and normalised with ITS statistics (robosat_amd.bands).  The
    reference's own classes, run once per directory with that directory's mode and statistics under the same seed (the joint
    random transforms draw once per item, whatever the number of images), must give the same bands; and the split chain the
    default ``rs train`` loader uses (worker: decode / convert / resize / crop / draw; device: transposes + normalise, here
    its CPU statement) must give the same item again."""
    rd, rt = _reference_modules()
    from oracle import tools_ref as T
    from robosat_amd import datasets as md, transforms as mt
    from robosat_amd.bands import bands_from_config, split_per_source

    size = 96
    root = synth.make_dataset(str(tmp_path / "ds"), n_train=6, n_val=2, size=128, seed=13, classes=4, ir=True)
    bands = bands_from_config({"common": {"image_dirs": ["images", "ir"], "image_modes": ["RGB", "L"]}}, {"model": {"in_channels": 4}})
    assert bands.channels == 4 and bands.mean[:3] == MEAN and len(bands.mean) == 4
    dirs = [os.path.join(root, "training", d) for d in bands.dirs]
    lab = os.path.join(root, "training", "labels")
    means, stds = split_per_source(bands, bands.mean), split_per_source(bands, bands.std)

    def ref_chain(mode, mean, std):
        return rt.JointCompose([
            rt.JointTransform(rt.ConvertImageMode(mode), rt.ConvertImageMode("P")),
            rt.JointTransform(rt.Resize((size, size), 2), rt.Resize((size, size), 0)),
            rt.JointTransform(rt.CenterCrop((size, size)), rt.CenterCrop((size, size))),
            rt.JointRandomHorizontalFlip(0.5),
            rt.JointRandomRotation(0.5, 90),
            rt.JointRandomRotation(0.5, 90),
            rt.JointRandomRotation(0.5, 90),
            rt.JointTransform(rt.ImageToTensor(), rt.MaskToTensor()),
            rt.JointTransform(rt.Normalize(mean=mean, std=std), None),
        ])

    refs = [rd.SlippyMapTilesConcatenation([d], lab, ref_chain(m, mu, sd)) for d, m, mu, sd in zip(dirs, bands.modes, means, stds)]
    ours = md.SlippyMapTilesConcatenation(dirs, lab, mt.JointCompose([
        mt.JointPerSource([mt.ConvertImageMode(m) for m in bands.modes], mt.ConvertImageMode("P")),
        mt.JointTransform(mt.Resize((size, size), 2), mt.Resize((size, size), 0)),
        mt.JointTransform(mt.CenterCrop((size, size)), mt.CenterCrop((size, size))),
        mt.JointRandomHorizontalFlip(0.5),
        mt.JointRandomRotation(0.5, 90),
        mt.JointRandomRotation(0.5, 90),
        mt.JointRandomRotation(0.5, 90),
        mt.JointTransform(mt.ImageToTensor(), mt.MaskToTensor()),
        mt.JointPerSource([mt.Normalize(mean=mu, std=sd) for mu, sd in zip(means, stds)], None),
    ]))
    split = md.UnaugmentedTiles(dirs, lab, size, draw=True, modes=bands.modes)
    assert len(ours) == len(split) == 6
    for i in range(len(ours)):
        parts = []
        for ref in refs:
            random.seed(500 + i)
            b_img, b_mask, _ = ref[i]
            parts.append(b_img)
        want = torch.cat(parts, dim=0)
        random.seed(500 + i)
        a_img, a_mask, a_tiles = ours[i]
        assert tuple(a_img.shape) == (4, size, size) and len(a_tiles) == 2 and a_tiles[0] == a_tiles[1]
        assert torch.equal(a_img, want) and torch.equal(a_mask, b_mask), i
        assert int(a_mask.max()) <= 3
        random.seed(500 + i)
        u8, m8, code, _ = split[i]
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (size, size, 4)
        draws = [0.0 if code & 1 else 1.0] + [0.0] * (code >> 1) + [1.0] * (3 - (code >> 1))
        s_img, s_mask = T.augment(u8.numpy(), m8.numpy(), draws, bands.mean, bands.std)
        assert np.array_equal(s_img, want.numpy()) and np.array_equal(s_mask, b_mask.numpy()), (i, code)


def test_four_band_buffered_predict_tiles(tmp_path):
    """``rs predict --extra_tiles``: every source composited with its own neighbours in its own mode, then concatenated --
    band for band what the reference's BufferedSlippyMapDirectory gives for each directory on its own."""
    rd, rt = _reference_modules()
    from robosat_amd import datasets as md, transforms as mt

    root = synth.make_dataset(str(tmp_path / "ds"), n_train=8, n_val=0, size=256, seed=14, classes=4, ir=True)
    rgb, ir = os.path.join(root, "training", "images"), os.path.join(root, "training", "ir")
    both = md.BufferedSlippyMapConcatenation([rgb, ir], [mt.Compose([mt.ConvertImageMode("RGB"), mt.ImageToUint8()]),
                                                         mt.Compose([mt.ConvertImageMode("L"), mt.ImageToUint8()])],
                                             ["RGB", "L"], size=256, overlap=32, cat_dim=2)
    ref_rgb = rd.BufferedSlippyMapDirectory(rgb, transform=rt.Compose([rt.ConvertImageMode("RGB"), rt.ImageToTensor()]), size=256, overlap=32)
    # (the reference composites in RGB only, tiles.py:186: an L-mode file comes back as three equal planes)
    ref_ir = rd.BufferedSlippyMapDirectory(ir, transform=rt.Compose([rt.ConvertImageMode("RGB"), rt.ImageToTensor()]), size=256, overlap=32)
    want = {}
    for i in range(len(ref_rgb)):
        a, tile = ref_rgb[i]
        want.setdefault(tuple(tile.tolist()), {})["rgb"] = a
    for i in range(len(ref_ir)):
        a, tile = ref_ir[i]
        want[tuple(tile.tolist())]["ir"] = a
    assert len(both) == 8
    for i in range(len(both)):
        u8, tile = both[i]
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (320, 320, 4)
        w = want[tuple(tile.tolist())]
        got = u8.permute(2, 0, 1).float().div(255)
        assert torch.equal(got[:3], w["rgb"]) and torch.equal(got[3], w["ir"][0]), tile


def test_resize_treats_an_alpha_plane_as_a_data_band():
    """``image_modes = ["RGBA"]`` carries IR in the alpha plane.  PIL's own resize premultiplies by alpha (RGB 200
    under alpha 10 comes back as 204, under alpha 0 as 0); the transform must resample the four bands independently, i.e. give
    what resizing each band as an 'L' image gives -- also where the alpha plane holds zeros."""
    from PIL import Image

    from robosat_amd.transforms import Resize

    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(48, 48, 3), dtype=np.uint8)
    ir = rng.integers(0, 256, size=(48, 48), dtype=np.uint8)
    ir[:16] = 0        # a block of zero IR: premultiplied resampling would wipe RGB there
    ir[16:24] = 10     # and a block of small IR: it would quantise RGB to multiples of ~25
    img = Image.fromarray(np.dstack([rgb, ir]), mode="RGBA")
    for size in ((32, 32), (64, 40)):
        got = np.asarray(Resize(size, Image.BILINEAR)(img))
        want = np.dstack([np.asarray(Image.fromarray(b).resize((size[1], size[0]), Image.BILINEAR)) for b in list(rgb.transpose(2, 0, 1)) + [ir]])
        assert got.shape == want.shape and np.array_equal(got, want)
        naive = np.asarray(img.resize((size[1], size[0]), Image.BILINEAR))
        assert not np.array_equal(naive[..., :3], want[..., :3])  # (what PIL alone does with the alpha: the bug this guards)
    # the reference's modes are untouched: same object semantics, same pixels as PIL
    opaque = Image.fromarray(rgb, mode="RGB")
    assert np.array_equal(np.asarray(Resize((32, 32), Image.BILINEAR)(opaque)), np.asarray(opaque.resize((32, 32), Image.BILINEAR)))
    # same size: a copy, whatever the mode
    assert np.array_equal(np.asarray(Resize((48, 48), Image.BILINEAR)(img)), np.asarray(img))


def test_history_plot_is_the_references_picture(tmp_path):
    """``history-*.png`` (robosat/utils.py:8-25, written by tools/train.py:146-148): drawn here without pyplot state, the
    same pixels as the reference's function where the reference is present (this container), a valid PNG anywhere."""

    import collections
    import importlib.util

    import numpy as np
    from PIL import Image

    from robosat_amd.utils import plot

    history = collections.OrderedDict([("train loss", [0.9, 0.7, 0.5]), ("train miou", [0.1, 0.3, 0.5]),
                                       ("val loss", [1.0, 0.8, 0.75]), ("val miou", [0.2, 0.4, 0.45])])
    mine = str(tmp_path / "mine.png")
    plot(mine, history)
    got = np.asarray(Image.open(mine).convert("RGB"))
    assert got.shape == (480, 640, 3) and got.min() < 128  # something was drawn
    plot(str(tmp_path / "empty.png"), {})  # first epoch of a run that has no entries yet: no exception
    ref_src = "/root/reference/robosat/utils.py"
    if os.path.exists(ref_src):
        spec = importlib.util.spec_from_file_location("reference_utils", ref_src)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        theirs = str(tmp_path / "theirs.png")
        ref.plot(theirs, history)
        assert np.array_equal(got, np.asarray(Image.open(theirs).convert("RGB")))
