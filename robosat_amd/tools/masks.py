"""``rs masks``: probability PNGs of one or several models -> class-index mask PNGs by weighted soft voting -- same
arguments and files as the reference (``robosat/tools/masks.py:28-84``).  Un-quantisation, the weighted average over the
models and the argmax run on the MI355X in one kernel (``rs_softvote_masks``), a batch of tiles per launch, in float64
and in the reference's summation order: the masks are byte-identical to numpy's.

Probability encoding (what ``rs predict`` writes): binary models -> the reference's single-channel mode-P PNG holding the
quantised foreground probability.  Models with 3 / 4 / 5 classes (which the reference's predict asserts away,
predict.py:98) -> mode LA / RGB / RGBA PNGs holding the quantised probability of every non-background class; the
background is 1 - their sum.  ``--dataset`` (optional, extension) takes the palette from the dataset's ``colors``."""

import argparse
import os
import sys

import numpy as np
import torch
from PIL import Image
from tqdm import tqdm

from robosat_amd import ops, png
from robosat_amd.colors import make_palette
from robosat_amd.tiles import tiles_from_slippy_map

# one colour per class index for masks of models with more than two classes (the reference hard-codes denim / orange)
DEFAULT_COLORS = ["denim", "orange", "green", "purple", "yellow", "cyan", "red", "mustard"]
MODE_CHANNELS = {"P": 1, "L": 1, "LA": 2, "RGB": 3, "RGBA": 4}
CHANNEL_MODES = {1: "P", 2: "LA", 3: "RGB", 4: "RGBA"}


def add_parser(subparser):
    parser = subparser.add_parser(
        "masks",
        help="compute masks from prediction probabilities",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter,
    )
    parser.add_argument("masks", type=str, help="slippy map directory to save masks to")
    parser.add_argument("probs", type=str, nargs="+", help="slippy map directories with class probabilities")
    parser.add_argument("--weights", type=float, nargs="+", help="weights for weighted average soft-voting")
    parser.add_argument("--dataset", type=str, required=False, help="dataset configuration file to take the mask colors from")
    parser.add_argument("--batch_size", type=int, default=16, help="tiles per device launch")
    parser.set_defaults(func=main)


def load_quantized(path):
    """PNG -> uint8 [H, W, C-1]: the stored bytes, one channel per non-background class."""

    image = Image.open(path)
    if image.mode not in MODE_CHANNELS:
        image = image.convert("P")  # (the reference converts whatever it finds to "P")
    q = np.array(image)
    return q[:, :, None] if q.ndim == 2 else q


def main(args):
    if args.weights and len(args.probs) != len(args.weights):
        sys.exit("Error: number of slippy map directories and weights must be the same")
    if not torch.cuda.is_available():
        sys.exit("Error: this build computes on the MI355X only")
    device = torch.device("cuda", 0)

    colors = None
    if args.dataset:
        from robosat_amd.config import load_config

        colors = load_config(args.dataset)["common"]["colors"]

    tilesets = list(zip(*map(tiles_from_slippy_map, args.probs)))
    batch = max(1, args.batch_size)
    for start in tqdm(range(0, len(tilesets), batch), desc="Masks", unit="batch", ascii=True):
        group = tilesets[start:start + batch]
        for tileset in group:
            # (the reference's `assert len(set(...))` can never fail -- masks.py:60; a second probability directory listing
            # other tiles must not be voted against this one silently)
            assert len(set(tile for tile, _ in tileset)) == 1, "tilesets in sync"
        # [K models][tiles of the batch][H][W][C-1]
        stacks = [[load_quantized(path) for _, path in tileset] for tileset in group]
        shape = stacks[0][0].shape
        assert all(q.shape == shape for tileset in stacks for q in tileset), "tiles of one batch share a shape"
        k, (h, w, cq) = len(stacks[0]), shape
        q = np.stack([np.stack([tileset[m] for tileset in stacks]) for m in range(k)])  # [K, B, H, W, Cq]
        dq = torch.from_numpy(q).to(device).view(k, -1, cq)
        masks = ops.softvote_masks(dq, args.weights).view(len(group), h, w).cpu().numpy()

        names = colors or (["denim", "orange"] if cq == 1 else DEFAULT_COLORS[:cq + 1])
        if len(names) < cq + 1:
            sys.exit("Error: the probabilities encode {} classes but the dataset config lists {} colors".format(cq + 1, len(names)))
        palette = make_palette(*names)
        for tileset, mask in zip(group, masks):
            x, y, z = tileset[0][0]
            os.makedirs(os.path.join(args.masks, str(z), str(x)), exist_ok=True)
            png.write_png(os.path.join(args.masks, str(z), str(x), str(y) + ".png"), mask.astype(np.uint8), "P", palette)


def softvote(probs, axis=0, weights=None):
    """The reference's host expression (masks.py:73-84), kept as the definition the kernel is tested against."""

    return np.argmax(np.average(probs, axis=axis, weights=weights), axis=axis)
