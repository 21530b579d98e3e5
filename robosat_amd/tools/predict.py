"""``rs predict``: same flags and output files as the reference (``robosat/tools/predict.py``) -- one mode-P PNG per
tile holding the 8-bit quantised foreground probability with the continuous pink palette -- computed by the
MI355X-native model with the softmax fused into the last kernel.  Models with 3 / 4 / 5 classes, which the reference
refuses (predict.py:98 asserts a binary model), get the same encoding per non-background class in a mode LA / RGB / RGBA
PNG (``rs masks`` reads both; C = 2 stays byte-identical to the reference).  A plain ``rs predict`` uses every visible GPU like
the reference's ``DataParallel`` (tools/predict.py:63): it re-executes itself once per GPU (``robosat_amd.launch``);
the batches are dealt round-robin to the ranks by the batch sampler (each rank decodes only its own tiles) and there is
no collective."""

import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from torch.utils.data import DataLoader
from tqdm import tqdm

from robosat_amd import launch, png
from robosat_amd.bands import bands_from_config, split_per_source
from robosat_amd.colors import continuous_palette_for_color
from robosat_amd.config import check_num_classes, load_config
from robosat_amd.datasets import BufferedSlippyMapConcatenation, BufferedSlippyMapDirectory
from robosat_amd.transforms import Compose, ConvertImageMode, ImageToTensor, ImageToUint8, Normalize
from robosat_amd.unet import UNet


def add_parser(subparser):
    parser = subparser.add_parser(
        "predict",
        help="predicts probability masks for slippy map tiles",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter,
    )
    parser.add_argument("--batch_size", type=int, default=1, help="images per batch")
    parser.add_argument("--checkpoint", type=str, required=True, help="model checkpoint to load")
    parser.add_argument("--overlap", type=int, default=32, help="tile pixel overlap to predict on")
    parser.add_argument("--tile_size", type=int, required=True, help="tile size for slippy map tiles")
    parser.add_argument("--workers", type=int, default=0, help="number of workers pre-processing images")
    parser.add_argument("tiles", type=str, help="directory to read slippy map image tiles from")
    parser.add_argument("probs", type=str, help="directory to save slippy map probability masks to")
    parser.add_argument("--model", type=str, required=True, help="path to model configuration file")
    parser.add_argument("--dataset", type=str, required=True, help="path to dataset configuration file")
    # extension (multi-band models, BASELINE configs[4]): one more slippy-map directory per further entry of the dataset's
    # `[common] image_dirs` (e.g. the infrared tiles), same z/x/y as `tiles`
    parser.add_argument("--extra_tiles", type=str, nargs="*", default=[], help="directories with the further image bands")
    parser.set_defaults(func=main)


def argv_from_args(args):
    """The ``rs predict`` command line equivalent to the namespace ``main()`` received (see tools/train.py)."""

    argv = ["predict", "--batch_size", str(args.batch_size), "--checkpoint", args.checkpoint, "--overlap", str(args.overlap),
            "--tile_size", str(args.tile_size), "--workers", str(args.workers), "--model", args.model, "--dataset", args.dataset]
    extra = list(getattr(args, "extra_tiles", None) or [])
    # (positionals first: `--extra_tiles` takes any number of values and would swallow them)
    return argv + [args.tiles, args.probs] + (["--extra_tiles"] + extra if extra else [])


def strip_module_prefix(state_dict):
    """Checkpoints carry the ``module.`` prefix of the reference's DataParallel wrapper (tools/train.py:69,158)."""

    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def quantize(foreground):
    """Probabilities in [0,1] -> uint8 bins exactly as the reference (predict.py:102-103): 1-based ``np.digitize``
    over 256 anchors; p == 1.0 lands in bin 256 and wraps to 0."""

    return np.digitize(foreground, np.linspace(0, 1, 256)).astype(np.uint8)


class RankBatchSampler:
    """The reference's sequential batches (DataLoader(batch_size=B), predict.py:78), of which rank r takes every
    ``world``-th: each rank loads and composites only its own tiles."""

    def __init__(self, num_items, batch_size, rank=0, world=1):
        self.n, self.bs, self.rank, self.world = num_items, batch_size, rank, world

    def __iter__(self):
        for i, start in enumerate(range(0, self.n, self.bs)):
            if i % self.world == self.rank:
                yield list(range(start, min(start + self.bs, self.n)))

    def __len__(self):
        nb = (self.n + self.bs - 1) // self.bs
        return (nb - self.rank + self.world - 1) // self.world


def main(args):
    model = load_config(args.model)
    dataset = load_config(args.dataset)

    if not model["common"]["cuda"]:
        # (a deliberate deviation from the reference, which falls back to the CPU here -- tools/train.py:60-63 -- and from BASELINE.json
        # configs[0], "rs train ... PyTorch CPU, 1 epoch (plumbing, no GPU)": there is no CPU compute path in this package; the same plumbing
        # run is covered on the GPU by tests/test_gpu_cli.py::test_rs_train_then_predict, the CPU side of it by the oracle)
        sys.exit("Error: this build computes on the MI355X only (no CPU path: BASELINE configs[0]'s `cuda = false` run is not supported); "
                 "set [common] cuda = true")
    if not torch.cuda.is_available():
        sys.exit("Error: CUDA requested but not available")

    if not launch.under_launcher():
        gpus = int(os.environ.get("ROBOSAT_GPUS", torch.cuda.device_count()))
        if gpus > 1 and getattr(args, "spawn", True):  # (`spawn=False`: the library caller's opt-out)
            status = launch.run_per_gpu(gpus, argv_from_args(args), module="robosat_amd.tools")
            if status != 0:
                sys.exit(status)
            return
    world, rank, local = launch.dist_env()
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    num_classes = len(dataset["common"]["classes"])
    try:
        check_num_classes(num_classes, "predict")
        bands = bands_from_config(dataset, model)  # (default: the reference's one RGB directory, ImageNet statistics)
    except ValueError as err:
        sys.exit("Error: {}".format(err))
    roots = [args.tiles] + list(getattr(args, "extra_tiles", None) or [])
    if len(roots) != len(bands.dirs):
        sys.exit("Error: the dataset config names {} image source(s) {}; give the further ones with --extra_tiles".format(
            len(bands.dirs), bands.dirs))

    chkpt = torch.load(args.checkpoint, map_location=device)
    net = UNet(num_classes, pretrained=False, compute_dtype=model.get("model", {}).get("compute_dtype", "fp32"),
               in_channels=bands.channels).to(device)
    net.load_state_dict(strip_module_prefix(chkpt["state_dict"]))
    net.eval()

    mean, std = bands.mean, bands.std
    # Default: the device-side pipeline -- tiles go up as uint8, ToTensor + Normalize, the
    # network, softmax, the un-buffer crop and the 8-bit quantisation all run on the GPU and one byte per pixel comes
    # back.  ROBOSAT_PREDICT_HOST_PIPELINE=1 keeps the reference's host-side steps (same bytes; the parity tests compare).
    host_pipeline = os.environ.get("ROBOSAT_PREDICT_HOST_PIPELINE", "0") == "1"
    if host_pipeline:
        transforms = [Compose([ConvertImageMode(mode=md), ImageToTensor(), Normalize(mean=m, std=s)])
                      for md, m, s in zip(bands.modes, split_per_source(bands, mean), split_per_source(bands, std))]
    else:
        transforms = [Compose([ConvertImageMode(mode=md), ImageToUint8()]) for md in bands.modes]

    if len(roots) == 1:
        directory = BufferedSlippyMapDirectory(args.tiles, transform=transforms[0], size=args.tile_size, overlap=args.overlap,
                                               mode=bands.modes[0])
    else:
        directory = BufferedSlippyMapConcatenation(roots, transforms, bands.modes, size=args.tile_size, overlap=args.overlap,
                                                   cat_dim=0 if host_pipeline else 2)
    assert len(directory) > 0, "at least one tile in dataset"

    loader = DataLoader(directory, num_workers=args.workers, pin_memory=True,
                        batch_sampler=RankBatchSampler(len(directory), args.batch_size, rank, world))
    palette = continuous_palette_for_color("pink", 256)

    # PNG encoding costs 3-10 ms per 512^2 tile: a pool encodes and writes while the next batches compute.  Pillow's encoder
    # holds the GIL while it deflates (a thread pool then encodes ONE tile at a time: ~50 tiles/s); robosat_amd.png goes
    # through zlib.compress, which does not.  Same mode, pixels and palette as the reference's files (predict.py:105-113).
    def write_png(q, x, y, z):
        os.makedirs(os.path.join(args.probs, str(z), str(x)), exist_ok=True)
        path = os.path.join(args.probs, str(z), str(x), str(y) + ".png")
        if num_classes == 2:
            png.write_png(path, q, "P", palette)
        else:
            png.write_png(path, q, {2: "LA", 3: "RGB", 4: "RGBA"}[num_classes - 1])

    writers = ThreadPoolExecutor(max_workers=int(os.environ.get("ROBOSAT_PNG_THREADS", "8")))
    pending = []

    # ROBOSAT_TIMING=1: where the wall time of the loop went (waiting for the loader / device work incl. the copy back /
    # waiting for the PNG writers), one line on stderr at the end
    timing = os.environ.get("ROBOSAT_TIMING", "0") == "1"
    spent = {"loader": 0.0, "device": 0.0, "writers": 0.0}
    mark = time.perf_counter()

    def lap(key):
        nonlocal mark
        now = time.perf_counter()
        spent[key] += now - mark
        mark = now

    def submit(tiles, quantized):
        for tile, q in zip(tiles, quantized):
            x, y, z = list(map(int, tile))
            pending.append(writers.submit(write_png, np.array(q, copy=True), x, y, z))  # (a copy: the staging buffer is reused)
        while len(pending) > 256:  # bounded backlog
            pending.pop(0).result()

    # The device-side pipeline runs one batch AHEAD of the host: batch i's bytes come back (asynchronously, into one of two
    # pinned buffers) while batch i+1 is already uploading and computing, so the GPU does not wait for the loader hand-over,
    # the PNG submissions or the copy itself.
    staging, inflight = [None, None], None
    ahead = os.environ.get("ROBOSAT_PREDICT_AHEAD", "1") == "1"  # (measurement knob: 0 = collect every batch at once)

    def collect(entry):
        host, event, tiles = entry
        event.synchronize()
        lap("device")
        submit(tiles, host.numpy())
        lap("writers")

    for step, (images, tiles) in enumerate(tqdm(loader, desc="Eval", unit="batch", ascii=True, disable=rank != 0)):
        lap("loader")
        if host_pipeline:
            probs = net.predict_probs(images.to(device, non_blocking=True)).cpu().numpy()
            quantized = []
            for prob in probs:
                prob = directory.unbuffer(prob)
                assert np.allclose(np.sum(prob, axis=0), 1.0, atol=1e-6), "single channel requires probabilities to sum up to one"
                q = quantize(prob[1:, :, :])
                quantized.append(q.squeeze() if num_classes == 2 else np.ascontiguousarray(q.transpose(1, 2, 0)))
            lap("device")
            submit(tiles, quantized)
            lap("writers")
            continue
        q_dev = net.predict_quantized(images.to(device, non_blocking=True), overlap=args.overlap, mean=mean, std=std)
        slot = step & 1
        if staging[slot] is None or staging[slot].shape[1:] != q_dev.shape[1:] or staging[slot].shape[0] < q_dev.shape[0]:
            staging[slot] = torch.empty(q_dev.shape, dtype=q_dev.dtype, pin_memory=True)
        host = staging[slot][:q_dev.shape[0]]
        host.copy_(q_dev, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        if inflight is not None:
            collect(inflight)  # (the previous batch: its slot is free again before the next batch reuses it)
        inflight = (host, event, tiles)
        lap("device")
        if not ahead:
            collect(inflight)
            inflight = None
    if inflight is not None:
        collect(inflight)

    for job in pending:
        job.result()  # (re-raises a failed write)
    writers.shutdown()
    lap("writers")
    if timing:
        print("rs predict rank {}: {} tiles; seconds waiting for the loader {:.2f}, on the device path {:.2f}, for the PNG writers {:.2f}".format(
            rank, len(directory), spent["loader"], spent["device"], spent["writers"]), file=sys.stderr)
