"""``rs weights``: class weights ``1 / ln(1.02 + p_c)`` from the label histogram of the training set -- same flag,
arithmetic and printed list as the reference (``robosat/tools/weights.py:26-59``).  The histogram (the reference's
per-tile ``np.bincount`` loop) runs on the MI355X: label tiles are uploaded as bytes in batches and counted by
``rs_label_histogram_u8``; the final weights are the reference's float64 numpy expression on those counts."""

import argparse
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader
from tqdm import tqdm

from robosat_amd import ops
from robosat_amd.config import load_config
from robosat_amd.datasets import SlippyMapTiles
from robosat_amd.transforms import Compose, ConvertImageMode


def add_parser(subparser):
    parser = subparser.add_parser(
        "weights", help="computes class weights on dataset", formatter_class=argparse.ArgumentDefaultsHelpFormatter
    )
    parser.add_argument("--dataset", type=str, required=True, help="path to dataset configuration file")
    parser.set_defaults(func=main)


class _MaskToBytes:
    """Mode-P label tile -> flat uint8 tensor (the bytes np.bincount sees in the reference)."""

    def __call__(self, image):
        return torch.from_numpy(np.array(image, dtype=np.uint8).reshape(-1))


def _collate(items):
    return torch.cat([labels for labels, _ in items])


def class_counts(label_dir, num_classes, device, batch_size=64, workers=0):
    """(pixels, int64 counts per class) of every label tile under ``label_dir``, counted on the device."""

    dataset = SlippyMapTiles(label_dir, transform=Compose([ConvertImageMode(mode="P"), _MaskToBytes()]))
    counts = torch.zeros(256, device=device, dtype=torch.int64)
    n = 0
    loader = DataLoader(dataset, batch_size=batch_size, num_workers=workers, collate_fn=_collate)
    for labels in tqdm(loader, desc="Loading", unit="batch", ascii=True):
        n += labels.numel()
        ops.label_histogram_u8(labels.to(device, non_blocking=True).contiguous(), counts)
    counts = counts.cpu().numpy()
    if counts[num_classes:].any():  # np.bincount would have grown past `num_classes` and the reference's `+=` would raise
        raise ValueError("labels outside [0, {}) in {}".format(num_classes, label_dir))
    return n, counts[:num_classes].astype(np.int64)


def weights_from_counts(n, counts):
    """The reference's arithmetic, verbatim (weights.py:55-58): float64 numpy, rounded to 6 digits."""

    probs = counts / n
    weights = 1 / np.log(1.02 + probs)
    weights.round(6, out=weights)
    return weights.tolist()


def main(args):
    dataset = load_config(args.dataset)
    if not torch.cuda.is_available():
        sys.exit("Error: this build computes on the MI355X only")
    device = torch.device("cuda", 0)
    path = dataset["common"]["dataset"]
    num_classes = len(dataset["common"]["classes"])
    n, counts = class_counts(os.path.join(path, "training", "labels"), num_classes, device)
    assert n > 0, "dataset with masks must not be empty"
    print(weights_from_counts(n, counts))
