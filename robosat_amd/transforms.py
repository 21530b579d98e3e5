"""Image / mask transformations feeding the hot path (reference ``robosat/transforms.py`` + the torchvision
transforms it imports: ToTensor, Normalize, Resize, CenterCrop -- torchvision is not a dependency here).

Augmentation randomness comes from Python's ``random`` module, one draw per transform, as in the reference."""

import random

import numpy as np
import torch
from PIL import Image


class ImageToTensor:
    """PIL image (H,W[,C] uint8) -> float32 [C,H,W] in [0,1] (torchvision ``ToTensor``)."""

    def __call__(self, image):
        arr = np.asarray(image, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).float().div_(255)


class ImageToUint8:
    """PIL image -> uint8 [H,W,C] tensor, untouched: ToTensor + Normalize then run on the device
    (``UNet.predict_quantized``), so a tile crosses PCIe as 1 byte per sample."""

    def __call__(self, image):
        arr = np.asarray(image, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(arr))


class MaskToTensor:
    """PIL label image -> int64 [H,W]."""

    def __call__(self, image):
        return torch.from_numpy(np.array(image, dtype=np.uint8)).long()


class ConvertImageMode:
    def __init__(self, mode):
        self.mode = mode

    def __call__(self, image):
        return image.convert(self.mode)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, tensor):
        return (tensor - self.mean) / self.std


class Resize:
    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, image):
        h, w = self.size
        if image.mode in ("RGBA", "LA") and image.size != (w, h):
            # An alpha plane that carries DATA (robosat_amd.bands: IR in the A of an RGBA source).  PIL's filters treat A as
            # coverage: Image.resize converts to premultiplied alpha, resamples and divides back, which rescales RGB by IR / 255
            # (lossy) and zeroes it wherever IR == 0.  Resample every band on its own -- for the opaque modes the reference uses
            # (RGB, L, P) this is exactly what PIL does anyway.
            return Image.merge(image.mode, [band.resize((w, h), self.interpolation) for band in image.split()])
        return image.resize((w, h), self.interpolation)


class CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, image):
        h, w = self.size
        iw, ih = image.size
        left, top = int(round((iw - w) / 2.0)), int(round((ih - h) / 2.0))
        return image.crop((left, top, left + w, top + h))


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for fn in self.transforms:
            x = fn(x)
        return x


class JointCompose:
    """Chain of joint ``(images, mask) -> (images, mask)`` transformations."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, images, mask):
        for fn in self.transforms:
            images, mask = fn(images, mask)
        return images, mask


class JointTransform:
    """Lifts independent (stateless) image / mask transformations to a joint one; ``None`` = leave untouched."""

    def __init__(self, image_transform, mask_transform):
        self.image_transform, self.mask_transform = image_transform, mask_transform

    def __call__(self, images, mask):
        if self.image_transform is not None:
            images = [self.image_transform(v) for v in images]
        if self.mask_transform is not None:
            mask = self.mask_transform(mask)
        return images, mask


class JointPerSource:
    """One (stateless) transformation PER image source -- e.g. ``ConvertImageMode("RGB")`` for the first directory and
    ``ConvertImageMode("L")`` for an infrared one, or each source's own ``Normalize`` -- where ``JointTransform`` applies the
    same one to all of them (the band layout of ``robosat_amd.bands``); ``None`` entries leave a source / the mask untouched."""

    def __init__(self, image_transforms, mask_transform=None):
        self.image_transforms, self.mask_transform = list(image_transforms), mask_transform

    def __call__(self, images, mask):
        assert len(images) == len(self.image_transforms), "one transformation per image source"
        images = [v if fn is None else fn(v) for fn, v in zip(self.image_transforms, images)]
        if self.mask_transform is not None:
            mask = self.mask_transform(mask)
        return images, mask


class _JointRandomTranspose:
    """With probability ``p`` applies one PIL transpose ``method`` to all images and the mask alike."""

    def __init__(self, p, method):
        self.p, self.method = p, method

    def __call__(self, images, mask):
        if random.random() < self.p:
            return [v.transpose(self.method) for v in images], mask.transpose(self.method)
        return images, mask


class JointRandomVerticalFlip(_JointRandomTranspose):
    def __init__(self, p):
        super().__init__(p, Image.FLIP_TOP_BOTTOM)


class JointRandomHorizontalFlip(_JointRandomTranspose):
    def __init__(self, p):
        super().__init__(p, Image.FLIP_LEFT_RIGHT)


class JointRandomRotation(_JointRandomTranspose):
    def __init__(self, p, degree):
        methods = {90: Image.ROTATE_90, 180: Image.ROTATE_180, 270: Image.ROTATE_270}
        if degree not in methods:
            raise NotImplementedError("We only support multiple of 90 degree rotations for now")
        super().__init__(p, methods[degree])
