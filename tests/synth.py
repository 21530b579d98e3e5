"""Synthetic slippy-map datasets for the end-to-end tests: random-texture RGB tiles with
rectangular foreground objects whose colour statistics differ from the background, mode-P label PNGs."""

import os

import numpy as np
from PIL import Image


def make_tile(rng, size):
    img = rng.integers(0, 120, size=(size, size, 3), dtype=np.uint8)
    mask = np.zeros((size, size), dtype=np.uint8)
    for _ in range(int(rng.integers(1, 4))):
        w, h = rng.integers(size // 8, size // 2, size=2)
        x0, y0 = rng.integers(0, size - w), rng.integers(0, size - h)
        mask[y0:y0 + h, x0:x0 + w] = 1
        img[y0:y0 + h, x0:x0 + w] = rng.integers(130, 256, size=(h, w, 3), dtype=np.uint8)
    return img, mask


def write_split(root, split, count, size, rng, zoom=18, x0=1000, y0=2000):
    for i in range(count):
        img, mask = make_tile(rng, size)
        x, y = x0 + i // 4, y0 + i % 4
        for kind, arr in (("images", img), ("labels", mask)):
            d = os.path.join(root, split, kind, str(zoom), str(x))
            os.makedirs(d, exist_ok=True)
            if kind == "images":
                Image.fromarray(arr, mode="RGB").save(os.path.join(d, "{}.png".format(y)))
            else:
                im = Image.fromarray(arr, mode="P")
                im.putpalette([0, 0, 0, 250, 0, 0] + [0] * (254 * 3))
                im.save(os.path.join(d, "{}.png".format(y)))


def make_dataset(root, n_train=8, n_val=4, size=256, seed=0):
    rng = np.random.default_rng(seed)
    write_split(root, "training", n_train, size, rng)
    write_split(root, "validation", n_val, size, rng, x0=3000)
    return root


def write_configs(tmp, dataset_root, checkpoint_dir, loss="Lovasz", batch_size=2, image_size=256, epochs=1, lr=1e-4):
    from robosat_amd.config import save_config

    model, ds = os.path.join(tmp, "model.toml"), os.path.join(tmp, "dataset.toml")
    save_config({"common": {"cuda": True, "batch_size": batch_size, "image_size": image_size, "checkpoint": checkpoint_dir},
                 "opt": {"epochs": epochs, "lr": lr, "loss": loss},
                 "model": {"pretrained": False}}, model)  # (no ImageNet file in the sandbox: random encoder on request)
    save_config({"common": {"dataset": dataset_root, "classes": ["background", "parking"], "colors": ["denim", "orange"]},
                 "weights": {"values": [1.6248, 5.762827]}}, ds)
    return model, ds
