"""Generate tests/golden/tools.npz by running the UNMODIFIED reference tools (dev container only):

    python tests/golden/make_golden_tools.py

  * ``rs masks``   -- ``robosat.tools.masks.main`` on two synthetic probability tilesets (the bytes ``rs predict`` writes),
                      unweighted and weighted: inputs + the mask PNG pixels it produced;
  * ``rs weights`` -- ``robosat.tools.weights.main`` on a synthetic label set: labels + the list it printed.

Third-party modules the reference imports but this image lacks are stood in by ``oracle/refshim.py`` (mercantile.Tile,
toml, torchvision.transforms); the reference's own files run untouched."""

import argparse
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import refshim  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def write_tiles(root, arrays, palette):
    """Probability tiles exactly as the reference's predict tool writes them (tools/predict.py:105-113): mode P + palette."""
    for i, arr in enumerate(arrays):
        d = os.path.join(root, "18", str(100 + i // 2))
        os.makedirs(d, exist_ok=True)
        out = Image.fromarray(arr, mode="P")
        out.putpalette(palette)
        out.save(os.path.join(d, "{}.png".format(200 + i % 2)), optimize=True)


def read_tiles(root, n):
    return np.stack([np.array(Image.open(os.path.join(root, "18", str(100 + i // 2), "{}.png".format(200 + i % 2)))) for i in range(n)])


def main():
    refshim.load_reference()  # installs the stand-ins and puts /root/reference on sys.path
    import robosat.tools.masks as ref_masks
    import robosat.tools.weights as ref_weights
    from robosat.colors import continuous_palette_for_color

    palette = continuous_palette_for_color("pink", 256)

    rng = np.random.default_rng(7)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- rs masks: 3 models x 4 tiles of 64x64 quantised foreground probabilities; ties and extremes included
        k, t, s = 3, 4, 64
        q = rng.integers(0, 256, size=(k, t, s, s), dtype=np.uint8)
        q[1] = np.where(rng.random((t, s, s)) < 0.3, 255 - q[0], q[1])  # models that disagree symmetrically: exact 0.5 averages
        q[:, 0, :4, :] = 0
        q[:, 0, 4:8, :] = 255
        q[:, 1, :8, :8] = 128
        dirs = []
        for m in range(k):
            d = os.path.join(tmp, "probs{}".format(m))
            write_tiles(d, list(q[m]), palette)
            dirs.append(d)
        out["masks_q"] = q
        for name, w in (("masks_unweighted", None), ("masks_weighted", [0.5, 1.5, 2.25]), ("masks_two_models", None)):
            use = dirs[:2] if name == "masks_two_models" else dirs
            dst = os.path.join(tmp, name)
            ref_masks.main(argparse.Namespace(masks=dst, probs=use, weights=w))
            out[name] = read_tiles(dst, t)
        out["masks_weights"] = np.array([0.5, 1.5, 2.25])

        # ---- rs weights: 6 label tiles of 96x96 over 3 classes, very unbalanced
        labels = (rng.random((6, 96, 96)) < 0.04).astype(np.uint8) + (rng.random((6, 96, 96)) < 0.01).astype(np.uint8)
        ds = os.path.join(tmp, "ds")
        for i, lab in enumerate(labels):
            d = os.path.join(ds, "training", "labels", "18", str(300 + i))
            os.makedirs(d)
            im = Image.fromarray(lab, mode="P")
            im.putpalette([0, 0, 0, 250, 0, 0, 0, 250, 0] + [0] * (253 * 3))
            im.save(os.path.join(d, "7.png"))
        cfg = os.path.join(tmp, "dataset.toml")
        with open(cfg, "w") as fp:
            fp.write("[common]\n  dataset = '{}'\n  classes = ['background', 'parking', 'road']\n  colors = ['denim', 'orange', 'green']\n".format(ds))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_weights.main(argparse.Namespace(dataset=cfg))
        printed = buf.getvalue().strip().splitlines()[-1]
        out["weights_labels"] = labels
        out["weights_printed"] = np.array(printed)
        out["weights_values"] = np.array(eval(printed), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "tools.npz"), **out)
    print("masks", {k: v.shape for k, v in out.items() if k.startswith("masks")}, "weights", printed)


if __name__ == "__main__":
    main()
