"""Loader-inclusive throughput of the tools (measurement; SURVEY.md section 7 "report both"): `rs predict` over a real
slippy-map directory of decoded-from-disk tiles, and `rs train` for one epoch with the host loader (PIL transforms in
DataLoader workers) and with the device-side augmentation (decoded-tile cache in HBM).  Every case runs the whole tool
twice, on a quarter of the tiles and on all of them: `steady_tiles_per_s` = the extra tiles over the extra seconds, i.e.
without interpreter / model start-up and the checkpoint write.  Prints one JSON line per case.

usage: python scripts/loader_bench.py [--tiles 512] [--size 512] [--workers 16] [--batch 16]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
from PIL import Image

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", type=int, default=1024)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--workers", type=int, default=16)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--train-tiles", type=int, default=1024)
ap.add_argument("--only", choices=["predict", "train"], default=None)
a = ap.parse_args()


def rs(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    e.setdefault("ROBOSAT_GPUS", "1")
    t0 = time.perf_counter()
    e["ROBOSAT_TIMING"] = "1"
    r = subprocess.run([sys.executable, "-m", "robosat_amd.tools"] + args, env=e, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-2000:])
    for line in r.stderr.splitlines():
        if line.startswith("rs predict rank"):
            print("#", line, flush=True)
    return dt


with tempfile.TemporaryDirectory() as tmp:
    import synth
    from robosat_amd.config import load_config, save_config
    from robosat_amd.unet import UNet

    rng = np.random.default_rng(0)
    # aerial-imagery-like tiles: smooth texture + objects, stored as JPEG (what tile servers deliver); labels PNG
    def write(split, count, x0):
        for i in range(count):
            img, mask = synth.make_tile(rng, a.size)
            base = np.kron(rng.integers(40, 200, size=(a.size // 16, a.size // 16, 3)), np.ones((16, 16, 1))).astype(np.uint8)
            img = (0.5 * img + 0.5 * base).astype(np.uint8)
            x, y = x0 + i // 32, 5000 + i % 32
            for kind, arr in (("images", img), ("labels", mask)):
                d = os.path.join(tmp, "ds", split, kind, "18", str(x))
                os.makedirs(d, exist_ok=True)
                if kind == "images":
                    Image.fromarray(arr, mode="RGB").save(os.path.join(d, "{}.jpg".format(y)), quality=90)
                else:
                    im = Image.fromarray(arr, mode="P")
                    im.putpalette([0, 0, 0, 250, 0, 0] + [0] * (254 * 3))
                    im.save(os.path.join(d, "{}.png".format(y)))

    write("validation", a.tiles, 1000)
    write("training", a.train_tiles, 3000)
    ds_root = os.path.join(tmp, "ds")
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, os.path.join(tmp, "pth"), batch_size=a.batch, image_size=a.size, epochs=1)
    net = UNet(2, pretrained=False)
    ck = os.path.join(tmp, "ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in net.state_dict().items()}}, ck)

    def subset(split, count, name):
        """A dataset root holding the first `count` tiles of `split` (symlinks) -- the quarter-size run."""
        root = os.path.join(tmp, name)
        for kind in ("images", "labels"):
            src = os.path.join(ds_root, split, kind)
            files = sorted(os.path.join(d, f) for d, _, fs in os.walk(src) for f in fs)[:count]
            for f in files:
                dst = os.path.join(root, split, kind, os.path.relpath(f, src))
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                os.symlink(f, dst)
        return root

    small = subset("validation", a.tiles // 4, "ds_small")
    subset("training", a.train_tiles // 4, "ds_small")

    for workers in (() if a.only == "train" else (0, a.workers)):
        times = []
        for root, n in ((small, a.tiles // 4), (ds_root, a.tiles)):
            times.append(rs(["predict", "--batch_size", str(a.batch), "--checkpoint", ck, "--overlap", "32", "--tile_size", str(a.size), "--workers",
                             str(workers), "--model", model_toml, "--dataset", ds_toml, os.path.join(root, "validation", "images"),
                             os.path.join(tmp, "probs{}_{}".format(workers, n))]))
        steady = (a.tiles - a.tiles // 4) / max(1e-9, times[1] - times[0])
        print(json.dumps({"case": "rs predict, {} JPEG tiles of {}^2 from disk, overlap 32 (9-tile composites), PNG out, bs {}, {} loader workers".format(
            a.tiles, a.size, a.batch, workers), "steady_tiles_per_s": round(steady, 1), "wall_s": [round(t, 2) for t in times],
            "tiles": [a.tiles // 4, a.tiles]}), flush=True)

    cases = [("fp32", "host", 1), ("bf16", "host", 1), ("fp32", "split", 1), ("bf16", "split", 1), ("bf16", "cache", 1), ("bf16", "cache", 3)]
    for dtype, feed, epochs in (() if a.only == "predict" else cases):
        times = []
        for root, n in ((small, (a.train_tiles + a.tiles) // 4), (ds_root, a.train_tiles + a.tiles)):
            cfg = load_config(model_toml)
            cfg["model"]["compute_dtype"] = dtype
            cfg["model"]["device_augment"] = feed == "cache"
            cfg["opt"]["epochs"] = epochs
            cfg["common"]["checkpoint"] = os.path.join(tmp, "pth_{}_{}_{}_{}".format(dtype, feed, epochs, n))
            save_config(cfg, model_toml)
            dcfg = load_config(ds_toml)
            dcfg["common"]["dataset"] = root
            save_config(dcfg, ds_toml)
            times.append(rs(["train", "--model", model_toml, "--dataset", ds_toml, "--workers", str(a.workers)],
                            env={"ROBOSAT_TRAIN_HOST_PIPELINE": "1" if feed == "host" else "0"}))
        n_all, n_small = (a.train_tiles + a.tiles) * epochs, (a.train_tiles + a.tiles) // 4 * epochs
        steady = (n_all - n_small) / max(1e-9, times[1] - times[0])
        what = {"host": "the reference's whole transform chain in {} DataLoader workers (PIL, fp32 tensors)".format(a.workers),
                "split": "default: {} workers decode/resize/crop + draw, the device flips/rotates/normalises".format(a.workers),
                "cache": "device_augment: tiles decoded once ({} workers) into HBM, augmented on the device".format(a.workers)}[feed]
        print(json.dumps({"case": "rs train {} epoch(s) (train + validation pass each), {} + {} tiles of {}^2, bs {}, {}, {}".format(
            epochs, a.train_tiles, a.tiles, a.size, a.batch, dtype, what),
            "steady_tiles_per_s": round(steady, 1), "wall_s": [round(t, 2) for t in times], "tile_passes": [n_small, n_all]}), flush=True)
