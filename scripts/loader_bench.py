"""Loader-inclusive throughput of the tools (measurement; SURVEY.md section 7 "report both"): `rs predict` over a real
slippy-map directory of decoded-from-disk tiles, and `rs train` with the reference-style host loader (the whole PIL transform
chain in DataLoader workers), the default split chain (workers decode + draw, the device augments) and the decoded-tile cache
in HBM.  Rates come from the tools' own clocks (`ROBOSAT_TIMING=1`: seconds inside the batch loop, i.e. without interpreter
/ model start-up, with the first batch's warm-up and the DataLoader workers' start), on enough tiles to amortise those: the
1 024 distinct JPEG tiles are linked four times under different tile columns.  Prints one JSON line per case.

usage: python scripts/loader_bench.py [--tiles 1024] [--repeat 4] [--size 512] [--workers 16] [--batch 16] [--only predict|train]"""
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
ap.add_argument("--repeat", type=int, default=4, help="links per distinct tile (different tile columns)")
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
    lines = [line for line in r.stderr.splitlines() if line.startswith("rs predict rank") or line.startswith("rs train rank")]
    for line in lines:
        print("#", line, flush=True)
    return dt, lines


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

    def repeated(split, count, times, name):
        """A dataset root holding the first `count` tiles of `split` `times` times over (symlinks under shifted tile columns)."""
        root = os.path.join(tmp, name)
        for kind in ("images", "labels"):
            src = os.path.join(ds_root, split, kind)
            files = sorted(os.path.join(d, f) for d, _, fs in os.walk(src) for f in fs)[:count]
            for rep in range(times):
                for f in files:
                    z, x, y = os.path.relpath(f, src).split(os.sep)
                    dst = os.path.join(root, split, kind, z, str(int(x) + 100 * rep), y)
                    os.makedirs(os.path.dirname(dst), exist_ok=True)
                    os.symlink(f, dst)
        return root

    import re

    big = repeated("validation", a.tiles, a.repeat, "ds_big")
    repeated("training", a.train_tiles, a.repeat, "ds_big")
    n_val, n_train = a.tiles * a.repeat, a.train_tiles * a.repeat

    for workers in (() if a.only == "train" else (0, a.workers)):
        root, n = (ds_root, a.tiles) if workers == 0 else (big, n_val)
        wall, lines = rs(["predict", "--batch_size", str(a.batch), "--checkpoint", ck, "--overlap", "32", "--tile_size", str(a.size), "--workers",
                          str(workers), "--model", model_toml, "--dataset", ds_toml, os.path.join(root, "validation", "images"),
                          os.path.join(tmp, "probs{}".format(workers))])
        secs = [float(v) for v in re.findall(r"(\d+\.\d+)", lines[-1].split(";", 1)[1])]
        print(json.dumps({"case": "rs predict, {} JPEG tiles of {}^2 from disk, overlap 32 (9-tile composites), PNG out, bs {}, {} loader workers".format(
            n, a.size, a.batch, workers), "tiles_per_s": round(n / sum(secs), 1), "loop_s": round(sum(secs), 2),
            "waiting_for_loader_device_writers_s": secs, "wall_s_incl_startup": round(wall, 2)}), flush=True)

    cases = [("bf16", "host", 1), ("fp32", "split", 1), ("bf16", "split", 1), ("bf16", "cache", 2)]
    for dtype, feed, epochs in (() if a.only == "predict" else cases):
        cfg = load_config(model_toml)
        cfg["model"]["compute_dtype"] = dtype
        cfg["model"]["device_augment"] = feed == "cache"
        cfg["opt"]["epochs"] = epochs
        cfg["common"]["checkpoint"] = os.path.join(tmp, "pth_{}_{}_{}".format(dtype, feed, epochs))
        save_config(cfg, model_toml)
        dcfg = load_config(ds_toml)
        dcfg["common"]["dataset"] = big
        save_config(dcfg, ds_toml)
        wall, lines = rs(["train", "--model", model_toml, "--dataset", ds_toml, "--workers", str(a.workers)],
                         env={"ROBOSAT_TRAIN_HOST_PIPELINE": "1" if feed == "host" else "0"})
        passes = [(m.group(1), int(m.group(2)), float(m.group(3))) for m in (re.search(r": (\w+) pass of (\d+) tiles in (\d+\.\d+) s", l) for l in lines) if m]
        last_train = [p for p in passes if p[0] == "Train"][-1]
        last_val = [p for p in passes if p[0] == "Validate"][-1]
        what = {"host": "the reference's whole transform chain in {} DataLoader workers (PIL, fp32 tensors)".format(a.workers),
                "split": "default: {} workers decode/resize/crop + draw, the device flips/rotates/normalises".format(a.workers),
                "cache": "device_augment: tiles decoded once ({} workers) into HBM, augmented on the device".format(a.workers)}[feed]
        print(json.dumps({"case": "rs train epoch {} of {}, {} training + {} validation tiles of {}^2, bs {}, {}, {}".format(
            epochs, epochs, n_train, n_val, a.size, a.batch, dtype, what),
            "train_tiles_per_s": round(last_train[1] / last_train[2], 1), "validate_tiles_per_s": round(last_val[1] / last_val[2], 1),
            "passes": passes, "wall_s_incl_startup": round(wall, 2)}), flush=True)
