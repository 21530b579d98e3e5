"""``rs serve``: on-demand tile server running the segmentation model per request -- the ``Predictor`` and the
``/<z>/<x>/<y>.png`` endpoint of the reference (``robosat/tools/serve.py:135-192``, ``:48-70``) on the MI355X-native model.

``Predictor.segment(image)`` is the single-image latency path: uint8 tile up, ToTensor + Normalize + U-Net + ``self.final``
+ argmax on the device (``UNet.predict_classes``), one byte per pixel back, palette PNG out.  The reference's Mapbox-GL
demo page (templates/map.html) and its access token are not part of the hot path and are not reproduced; ``/`` answers
with a one-line description instead.  ``--url`` may be an ``http(s)://.../{z}/{x}/{y}`` endpoint (needs `requests`) or a
local slippy-map directory template such as ``/data/tiles/{z}/{x}/{y}.png``."""

import argparse
import io
import os
import sys

import numpy as np
import torch
from PIL import Image

from robosat_amd.bands import bands_from_config
from robosat_amd.colors import make_palette
from robosat_amd.config import load_config
from robosat_amd.unet import UNet

predictor = None
tiles = None
size = None


def add_parser(subparser):
    parser = subparser.add_parser(
        "serve",
        help="serves predicted masks with on-demand tileserver",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter,
    )
    parser.add_argument("--model", type=str, required=True, help="path to model configuration file")
    parser.add_argument("--dataset", type=str, required=True, help="path to dataset configuration file")
    parser.add_argument("--url", type=str, help="endpoint with {z}/{x}/{y} variables to fetch image tiles from")
    parser.add_argument("--checkpoint", type=str, required=True, help="model checkpoint to load")
    parser.add_argument("--tile_size", type=int, default=512, help="tile size for slippy map tiles")
    parser.add_argument("--host", type=str, default="127.0.0.1", help="host to serve on")
    parser.add_argument("--port", type=int, default=5000, help="port to serve on")
    parser.set_defaults(func=main)


class Predictor:
    """Reference ``serve.py:135-192``: loads a (``module.``-prefixed) checkpoint once, segments one image per call."""

    def __init__(self, checkpoint, model, dataset):
        cuda = model["common"]["cuda"]
        assert torch.cuda.is_available() or not cuda, "cuda is available when requested"
        if not cuda:
            raise RuntimeError("robosat_amd computes on the MI355X only; set [common] cuda = true")
        self.cuda = cuda
        self.device = torch.device("cuda")
        self.checkpoint = checkpoint
        self.model = model
        self.dataset = dataset
        # the band layout of the dataset config (default: one RGB image, ImageNet statistics -- serve.py:152-153); a served
        # tile is ONE image, so a multi-band model needs a single 4-band source (`image_modes = ["RGBA"]`)
        self.bands = bands_from_config(dataset, model)
        if len(self.bands.dirs) != 1:
            raise ValueError("rs serve fetches one image per tile: the dataset config must name a single image source")
        self.net = self.net_from_chkpt_()
        self.palette = make_palette(*self.dataset["common"]["colors"])

    def segment(self, image):
        """PIL image -> mode-P mask image: ``argmax`` of the logits, palette from the dataset's colours."""

        mean, std = self.bands.mean, self.bands.std
        pixels = np.array(image.convert(self.bands.modes[0]), dtype=np.uint8)
        u8 = torch.from_numpy(pixels if pixels.ndim == 3 else pixels[:, :, None]).unsqueeze(0)
        mask = self.net.predict_classes(u8.to(self.device, non_blocking=True), mean=mean, std=std)[0].cpu().numpy()
        mask = Image.fromarray(mask, mode="P")
        mask.putpalette(self.palette)
        return mask

    def net_from_chkpt_(self):
        chkpt = torch.load(self.checkpoint, map_location=self.device)
        num_classes = len(self.dataset["common"]["classes"])
        net = UNet(num_classes, pretrained=False, compute_dtype=self.model.get("model", {}).get("compute_dtype", "fp32"),
                   in_channels=self.bands.channels).to(self.device)
        state = chkpt["state_dict"]
        net.load_state_dict({(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()})
        net.eval()
        return net


def fetch_tile(url):
    """The tile image behind ``url`` (http(s) through `requests`, anything else as a local path); None if unavailable."""

    if url.startswith("http://") or url.startswith("https://"):
        import requests

        try:
            resp = requests.get(url, timeout=10)
            resp.raise_for_status()
            return io.BytesIO(resp.content)
        except Exception:
            return None
    return url if os.path.isfile(url) else None


def make_app():
    from flask import Flask, abort, send_file

    app = Flask(__name__)

    @app.route("/")
    def index():
        return "robosat_amd tile server: GET /<z>/<x>/<y>.png returns the segmentation mask of that tile ({} px)\n".format(size)

    @app.route("/<int:z>/<int:x>/<int:y>.png")
    def tile(z, x, y):
        if z != 18:  # (the reference's restriction, serve.py:52-54)
            abort(404)
        res = fetch_tile(tiles.format(x=x, y=y, z=z))
        if not res:
            abort(500)
        mask = predictor.segment(Image.open(res))
        output = io.BytesIO()
        mask.save(output, format="png", optimize=True)
        output.seek(0)
        return send_file(output, mimetype="image/png")

    @app.after_request
    def after_request(response):
        response.headers["Access-Control-Allow-Origin"] = "*"
        return response

    return app


def main(args):
    model = load_config(args.model)
    dataset = load_config(args.dataset)
    if model["common"]["cuda"] and not torch.cuda.is_available():
        sys.exit("Error: CUDA requested but not available")
    if not args.url:
        sys.exit("Error: --url needed: endpoint or directory template with {z}/{x}/{y} to fetch image tiles from")

    global size, tiles, predictor
    size = args.tile_size
    tiles = args.url
    predictor = Predictor(args.checkpoint, model, dataset)
    make_app().run(host=args.host, port=args.port, threaded=False)  # one request at a time on the GPU, as the reference
