"""Synthetic slippy-map datasets for the end-to-end tests: random-texture RGB tiles with
rectangular foreground objects whose colour statistics differ from the background, mode-P label PNGs.  With ``ir=True`` a
second, single-band directory ``ir/`` holds an infrared plane (BASELINE configs[4]: RGB + IR), and with ``classes > 2`` the
rectangles carry class labels 1..classes-1 that the bands tell apart (class k is bright in band (k - 1) % bands)."""

import os

import numpy as np
from PIL import Image

CLASS_NAMES = ["background", "parking", "road", "building", "water", "forest", "field", "rail"]
COLOR_NAMES = ["denim", "orange", "green", "red", "blue", "teal", "yellow", "purple"]


def make_tile(rng, size, classes=2, bands=3):
    """(image uint8 [size,size,bands], mask uint8 [size,size]); with 2 classes and 3 bands exactly the round-1 tiles."""

    img = rng.integers(0, 120, size=(size, size, bands), dtype=np.uint8)
    mask = np.zeros((size, size), dtype=np.uint8)
    for _ in range(int(rng.integers(1, 4))):
        w, h = rng.integers(size // 8, size // 2, size=2)
        x0, y0 = rng.integers(0, size - w), rng.integers(0, size - h)
        if classes == 2:
            mask[y0:y0 + h, x0:x0 + w] = 1
            img[y0:y0 + h, x0:x0 + w] = rng.integers(130, 256, size=(h, w, bands), dtype=np.uint8)
        else:
            k = int(rng.integers(1, classes))
            mask[y0:y0 + h, x0:x0 + w] = k
            img[y0:y0 + h, x0:x0 + w] = rng.integers(0, 120, size=(h, w, bands), dtype=np.uint8)
            img[y0:y0 + h, x0:x0 + w, (k - 1) % bands] = rng.integers(150, 256, size=(h, w), dtype=np.uint8)
    return img, mask


def write_split(root, split, count, size, rng, zoom=18, x0=1000, y0=2000, classes=2, ir=False):
    for i in range(count):
        img, mask = make_tile(rng, size, classes, 4 if ir else 3)
        x, y = x0 + i // 4, y0 + i % 4
        planes = [("images", img[:, :, :3]), ("labels", mask)] + ([("ir", img[:, :, 3])] if ir else [])
        for kind, arr in planes:
            d = os.path.join(root, split, kind, str(zoom), str(x))
            os.makedirs(d, exist_ok=True)
            if kind == "images":
                Image.fromarray(np.ascontiguousarray(arr), mode="RGB").save(os.path.join(d, "{}.png".format(y)))
            elif kind == "ir":
                Image.fromarray(np.ascontiguousarray(arr), mode="L").save(os.path.join(d, "{}.png".format(y)))
            else:
                im = Image.fromarray(arr, mode="P")
                im.putpalette([0, 0, 0, 250, 0, 0] + [0] * (254 * 3))
                im.save(os.path.join(d, "{}.png".format(y)))


def make_dataset(root, n_train=8, n_val=4, size=256, seed=0, classes=2, ir=False):
    rng = np.random.default_rng(seed)
    write_split(root, "training", n_train, size, rng, classes=classes, ir=ir)
    write_split(root, "validation", n_val, size, rng, x0=3000, classes=classes, ir=ir)
    return root


def write_configs(tmp, dataset_root, checkpoint_dir, loss="Lovasz", batch_size=2, image_size=256, epochs=1, lr=1e-4, classes=2,
                  ir=False, compute_dtype=None):
    from robosat_amd.config import save_config

    model, ds = os.path.join(tmp, "model.toml"), os.path.join(tmp, "dataset.toml")
    extra = {"pretrained": False}  # (no ImageNet file in the sandbox: random encoder on request)
    if ir:
        extra["in_channels"] = 4
    if compute_dtype:
        extra["compute_dtype"] = compute_dtype
    save_config({"common": {"cuda": True, "batch_size": batch_size, "image_size": image_size, "checkpoint": checkpoint_dir},
                 "opt": {"epochs": epochs, "lr": lr, "loss": loss},
                 "model": extra}, model)
    common = {"dataset": dataset_root, "classes": CLASS_NAMES[:classes], "colors": COLOR_NAMES[:classes]}
    if ir:
        common.update({"image_dirs": ["images", "ir"], "image_modes": ["RGB", "L"]})
    weights = [1.6248, 5.762827] + [4.0] * (classes - 2)
    save_config({"common": common, "weights": {"values": weights}}, ds)
    return model, ds
