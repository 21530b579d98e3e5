"""Import the UNMODIFIED reference (``/root/reference``) in the dev container.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  ``/root/reference`` does
not exist on the GPU box, so nothing that runs there may import this module.

The reference needs four third-party modules that are absent from this image
(SURVEY.md section 8c).  We pre-populate ``sys.modules`` with minimal stand-ins
and then import the reference's own files untouched:

* ``torchvision.models.resnet50`` -- torchvision 0.3.0 (pinned ``torchvision~=0.3``
  in reference ``setup.py:38``; not in the reference tree).  The stand-in is the
  plain ``torch.nn`` restatement in ``oracle/robosat_ref.py`` (``ResNet50``).
* ``torchvision.transforms`` -- ToTensor / Normalize / Compose / Resize / CenterCrop.
* ``mercantile.Tile`` -- ``namedtuple("Tile", "x y z")``.
* ``toml.load`` -- via ``tomli``.
"""

import collections
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("ROBOSAT_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "robosat"))


def _install_standins():
    from oracle import robosat_ref

    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        models = types.ModuleType("torchvision.models")

        def resnet50(pretrained=False, **kwargs):
            # no network here: ``pretrained`` is neutralised (random init)
            return robosat_ref.ResNet50()

        models.resnet50 = resnet50

        tr = types.ModuleType("torchvision.transforms")

        class ToTensor:
            def __call__(self, pic):
                arr = np.asarray(pic, dtype=np.uint8)
                if arr.ndim == 2:
                    arr = arr[:, :, None]
                return torch.from_numpy(arr.transpose(2, 0, 1).copy()).float().div(255)

        class Normalize:
            def __init__(self, mean, std):
                self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
                self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

            def __call__(self, t):
                return (t - self.mean) / self.std

        class Compose:
            def __init__(self, transforms):
                self.transforms = transforms

            def __call__(self, x):
                for t in self.transforms:
                    x = t(x)
                return x

        class Resize:
            def __init__(self, size, interpolation=2):
                self.size, self.interpolation = size, interpolation

            def __call__(self, img):
                h, w = self.size
                return img.resize((w, h), self.interpolation)

        class CenterCrop:
            def __init__(self, size):
                self.size = size

            def __call__(self, img):
                h, w = self.size
                W, H = img.size
                left, top = int(round((W - w) / 2.0)), int(round((H - h) / 2.0))
                return img.crop((left, top, left + w, top + h))

        tr.ToTensor, tr.Normalize, tr.Compose, tr.Resize, tr.CenterCrop = ToTensor, Normalize, Compose, Resize, CenterCrop
        tv.models, tv.transforms = models, tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = models
        sys.modules["torchvision.transforms"] = tr

    if "mercantile" not in sys.modules:
        merc = types.ModuleType("mercantile")
        merc.Tile = collections.namedtuple("Tile", "x y z")
        sys.modules["mercantile"] = merc

    if "toml" not in sys.modules:
        import tomli

        toml = types.ModuleType("toml")

        def load(path):
            with open(path, "rb") as fp:
                return tomli.load(fp)

        toml.load = load
        sys.modules["toml"] = toml


def load_reference():
    """Returns a namespace with the reference's own ``unet``, ``losses``, ``metrics`` modules."""

    if not available():
        raise RuntimeError("reference tree not present at {}".format(REFERENCE_ROOT))

    _install_standins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import robosat.unet as unet  # noqa: E402  (the reference's own file)
    import robosat.losses as losses  # noqa: E402
    import robosat.metrics as metrics  # noqa: E402

    return types.SimpleNamespace(unet=unet, losses=losses, metrics=metrics)
