"""Datasets over slippy-map tile directories: the class surface of the reference's ``robosat/datasets.py``
(``SlippyMapTiles``, ``SlippyMapTilesConcatenation``, ``BufferedSlippyMapDirectory``), feeding batches to the GPU path.
Parity with the reference's own classes on the same directory: ``tests/test_feeders.py``."""

import collections

import torch
import torch.utils.data
from PIL import Image

from .tiles import buffer_tile_image, tiles_from_slippy_map


class _TileDirectory(torch.utils.data.Dataset):
    """A ``z/x/y.*`` directory listed once into ``self.tiles = [(tile, path), ...]``."""

    def __init__(self, root, ordered):
        super().__init__()
        listing = tiles_from_slippy_map(root)
        self.tiles = sorted(listing, key=lambda entry: entry[0]) if ordered else list(listing)

    def __len__(self):
        return len(self.tiles)


class SlippyMapTiles(_TileDirectory):
    """Items are ``(image, tile)`` in sorted tile order (``datasets.py:16-39``)."""

    def __init__(self, root, transform=None):
        super().__init__(root, ordered=True)
        self.transform = transform

    def __getitem__(self, i):
        tile, path = self.tiles[i]
        image = Image.open(path)
        return (image if self.transform is None else self.transform(image)), tile


class SlippyMapTilesConcatenation(torch.utils.data.Dataset):
    """Several image directories (concatenated on the channel axis) plus one label directory; items are
    ``(images [sum C,H,W], mask [H,W], tiles)`` after the joint transform (``datasets.py:44-78``)."""

    def __init__(self, inputs, target, joint_transform=None):
        super().__init__()
        self.joint_transform = joint_transform
        self.inputs = [SlippyMapTiles(directory) for directory in inputs]  # transforms are joint: applied in __getitem__
        self.target = SlippyMapTiles(target)
        counts = {len(ds) for ds in self.inputs}
        assert len(counts) == 1, "same number of tiles in all images"
        assert counts == {len(self.target)}, "same number of tiles in images and label"

    def __len__(self):
        return len(self.target)

    def __getitem__(self, i):
        images, tiles = zip(*(ds[i] for ds in self.inputs))
        mask, mask_tile = self.target[i]
        assert len(set(tiles)) == 1, "all images are for the same tile"
        assert tiles[0] == mask_tile, "image tile is the same as label tile"
        images = list(images)
        if self.joint_transform is not None:
            images, mask = self.joint_transform(images, mask)
        return torch.cat(images, dim=0), mask, list(tiles)


class BufferedSlippyMapDirectory(_TileDirectory):
    """Tiles composited with an ``overlap``-pixel border from their neighbours (``datasets.py:83-136``); ``unbuffer``
    crops a prediction back to the tile.  The tile -> path map is built once (the reference rebuilds it per item).

    Every composite needs nine decoded files, six of which the next tile needs again: tiles are therefore visited in
    sorted (x, y) order -- the reference walks ``os.listdir`` order; each tile's output is independent of the order -- and
    the last ``cache_tiles`` decoded neighbours are kept (per DataLoader worker), which turns ~9 decodes per tile into ~1-3."""

    def __init__(self, root, transform=None, size=512, overlap=32, cache_tiles=192):
        assert overlap >= 0
        assert size >= 256
        super().__init__(root, ordered=True)
        self.transform, self.size, self.overlap = transform, size, overlap
        self._store = dict(self.tiles)
        self._cache, self._cache_tiles = collections.OrderedDict(), cache_tiles

    def _open(self, path):
        image = self._cache.get(path)
        if image is None:
            image = Image.open(path).convert("RGB")
            image.load()
            self._cache[path] = image
            if len(self._cache) > self._cache_tiles:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(path)
        return image

    def __getitem__(self, i):
        tile = self.tiles[i][0]
        image = buffer_tile_image(tile, self._store, overlap=self.overlap, tile_size=self.size, opener=self._open)
        if self.transform is not None:
            image = self.transform(image)
        return image, torch.IntTensor([tile.x, tile.y, tile.z])

    def unbuffer(self, probs):
        o = self.overlap
        return probs[:, o:probs.shape[1] - o, o:probs.shape[2] - o]


class DecodedTileCache:
    """Every (image, mask) tile of a training / validation split decoded ONCE into uint8 tensors in HBM (SURVEY.md section
    8f, N4).  What is cached is the output of the DETERMINISTIC head of the reference's transform chain -- mode
    conversion, resize, centre crop (tools/train.py:250-253) -- i.e. exactly the pixels the random flips / rotations and
    ``ToTensor`` + ``Normalize`` start from.  A 512x512 RGB tile + its mask is 1 MiB: 100 000 tiles fit the MI355X's 288 GB
    with room to spare, and an epoch then costs no PNG/JPEG decode at all."""

    def __init__(self, image_dirs, label_dir, size, device, head_transform=None):
        import numpy as np
        from PIL import Image

        from .transforms import CenterCrop, ConvertImageMode, Resize

        self.source = SlippyMapTilesConcatenation(image_dirs, label_dir, joint_transform=None)
        target = (size, size)
        to_image = [ConvertImageMode("RGB"), Resize(target, Image.BILINEAR), CenterCrop(target)]
        to_mask = [ConvertImageMode("P"), Resize(target, Image.NEAREST), CenterCrop(target)]
        images, masks, self.tiles = [], [], []
        for i in range(len(self.source.target)):
            tiles_and_images = [ds[i] for ds in self.source.inputs]
            mask, mask_tile = self.source.target[i]
            assert all(tile == mask_tile for _, tile in tiles_and_images), "image tile is the same as label tile"
            planes = []
            for image, _ in tiles_and_images:
                for fn in to_image:
                    image = fn(image)
                planes.append(np.asarray(image, dtype=np.uint8))
            for fn in to_mask:
                mask = fn(mask)
            images.append(np.concatenate(planes, axis=2))
            masks.append(np.asarray(mask, dtype=np.uint8))
            self.tiles.append(mask_tile)
        self.size = size
        self.images = torch.from_numpy(np.stack(images)).to(device)  # [T, S, S, C] uint8
        self.masks = torch.from_numpy(np.stack(masks)).to(device)    # [T, S, S] uint8
        self.device = device

    def __len__(self):
        return len(self.tiles)


class DeviceAugmentLoader:
    """Iterates a ``DecodedTileCache`` like the reference's training ``DataLoader`` (tools/train.py:248-274): batches of
    ``(images [N,C,S,S] fp32 normalised, masks [N,S,S] int64, tiles)`` -- already on the device.  The random part of the
    transform chain -- horizontal flip with p = 0.5, then three independent 90-degree rotations with p = 0.5 each -- is drawn
    on the host from Python's ``random`` in the reference's order (four numbers per sample), so a seeded run augments every
    sample exactly as the host chain would; flip, rotations, ``ToTensor`` and ``Normalize`` then run in ONE kernel
    (``rs_augment_tiles``) straight from the cache."""

    def __init__(self, cache, batch_sampler, mean, std):
        self.cache, self.batch_sampler, self.mean, self.std = cache, batch_sampler, list(mean), list(std)

    def __len__(self):
        return len(self.batch_sampler)

    @staticmethod
    def draw_op():
        import random

        flip = random.random() < 0.5
        turns = sum(random.random() < 0.5 for _ in range(3))
        return int(flip) + 2 * turns

    def __iter__(self):
        from . import ops

        c = self.cache
        channels = c.images.shape[3]
        mean, std = (self.mean * channels)[:channels], (self.std * channels)[:channels]  # (several image dirs: per-band repeat)
        for batch in self.batch_sampler:
            codes = [self.draw_op() for _ in batch]
            index = torch.tensor(batch, dtype=torch.int32).to(c.device, non_blocking=True)
            op = torch.tensor(codes, dtype=torch.int32).to(c.device, non_blocking=True)
            images, masks = ops.augment_tiles(c.images, c.masks, index, op, mean, std)
            yield images, masks, [[c.tiles[i]] for i in batch]
