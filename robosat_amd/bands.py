"""Which image directories feed the network, and as what: the band layout of a dataset.

The reference's dataset layer already concatenates several slippy-map image directories on the channel axis
(``SlippyMapTilesConcatenation``, robosat/datasets.py:44-78) but its tools always pass ONE directory converted to RGB
(tools/train.py:250,262-268; tools/predict.py:71) with a 3-entry mean / std (train.py:246).  BASELINE configs[4] is a
4-band (RGB + IR) model, so the tools here read the band layout from the DATASET config -- all keys optional, the
defaults are exactly the reference's behaviour:

    [common]
      image_dirs  = ["images", "ir"]     # sub-directories of <dataset>/{training,validation}/   (default ["images"])
      image_modes = ["RGB", "L"]         # PIL mode each source is converted to                   (default RGB, then L)
      mean        = [0.485, 0.456, 0.406, 0.449]   # per band, after ToTensor                     (defaults below)
      std         = [0.229, 0.224, 0.225, 0.226]

A single 4-band source is ``image_dirs = ["images"], image_modes = ["RGBA"]`` (IR in the alpha plane).  The bands of all
sources, in order, are the network's input channels: their count must equal ``[model] in_channels`` when that key is
given, and is what ``in_channels`` defaults to when it is not.
"""

import collections

MODE_CHANNELS = {"RGB": 3, "L": 1, "RGBA": 4}
# ImageNet statistics for RGB (tools/train.py:246); for a single band their luminance-weighted equivalents
# (torchvision's grayscale convention); an alpha-plane band gets the single-band numbers
_MODE_MEAN = {"RGB": [0.485, 0.456, 0.406], "L": [0.449], "RGBA": [0.485, 0.456, 0.406, 0.449]}
_MODE_STD = {"RGB": [0.229, 0.224, 0.225], "L": [0.226], "RGBA": [0.229, 0.224, 0.225, 0.226]}

Bands = collections.namedtuple("Bands", ["dirs", "modes", "mean", "std", "channels"])


def bands_from_config(dataset, model=None):
    """The ``Bands`` of a dataset config (+ the model config's ``in_channels`` to check against)."""

    common = dataset.get("common", {})
    dirs = list(common.get("image_dirs", ["images"]))
    if not dirs or not all(isinstance(d, str) and d for d in dirs):
        raise ValueError("[common] image_dirs must be a non-empty list of directory names")
    modes = list(common.get("image_modes", ["RGB"] + ["L"] * (len(dirs) - 1)))
    if len(modes) != len(dirs):
        raise ValueError("[common] image_modes needs one PIL mode per entry of image_dirs ({} vs {})".format(len(modes), len(dirs)))
    for m in modes:
        if m not in MODE_CHANNELS:
            raise ValueError("[common] image_modes: unsupported mode {!r} (one of {})".format(m, sorted(MODE_CHANNELS)))
    channels = sum(MODE_CHANNELS[m] for m in modes)
    if not 1 <= channels <= 4:
        raise ValueError("the network's stem takes 1..4 input bands; image_dirs / image_modes give {}".format(channels))
    mean = list(common.get("mean", [v for m in modes for v in _MODE_MEAN[m]]))
    std = list(common.get("std", [v for m in modes for v in _MODE_STD[m]]))
    if len(mean) != channels or len(std) != channels:
        raise ValueError("[common] mean / std need one entry per band ({} bands)".format(channels))
    if model is not None:
        want = model.get("model", {}).get("in_channels")
        if want is not None and int(want) != channels:
            raise ValueError("[model] in_channels = {} but the dataset's image_dirs / image_modes give {} band(s)".format(want, channels))
    return Bands(dirs, modes, [float(v) for v in mean], [float(v) for v in std], channels)


def split_per_source(bands, values):
    """``values`` (one per band) cut into one list per source."""

    out, at = [], 0
    for m in bands.modes:
        out.append(list(values[at:at + MODE_CHANNELS[m]]))
        at += MODE_CHANNELS[m]
    return out
