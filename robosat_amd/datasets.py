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

    def __init__(self, root, transform=None, size=512, overlap=32, cache_tiles=192, mode="RGB"):
        assert overlap >= 0
        assert size >= 256
        super().__init__(root, ordered=True)
        self.transform, self.size, self.overlap, self.mode = transform, size, overlap, mode
        self._store = dict(self.tiles)
        self._cache, self._cache_tiles = collections.OrderedDict(), cache_tiles

    def _open(self, path):
        image = self._cache.get(path)
        if image is None:
            image = Image.open(path).convert(self.mode)
            image.load()
            self._cache[path] = image
            if len(self._cache) > self._cache_tiles:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(path)
        return image

    def __getitem__(self, i):
        tile = self.tiles[i][0]
        image = buffer_tile_image(tile, self._store, overlap=self.overlap, tile_size=self.size, opener=self._open, mode=self.mode)
        if self.transform is not None:
            image = self.transform(image)
        return image, torch.IntTensor([tile.x, tile.y, tile.z])

    def unbuffer(self, probs):
        o = self.overlap
        return probs[:, o:probs.shape[1] - o, o:probs.shape[2] - o]


class BufferedSlippyMapConcatenation(torch.utils.data.Dataset):
    """Several ``BufferedSlippyMapDirectory`` sources over the same tiles, concatenated on the channel axis: what
    ``SlippyMapTilesConcatenation`` (``datasets.py:44-78``) is to ``SlippyMapTiles``, for ``rs predict`` on multi-band models
    (BASELINE configs[4]: RGB + IR).  Every source is composited with its own neighbours and converted to its own mode; the
    per-source transforms must return tensors that concatenate on ``cat_dim`` (0 for ``[C,H,W]`` floats, 2 for ``[H,W,C]``
    bytes)."""

    def __init__(self, roots, transforms, modes, size=512, overlap=32, cat_dim=2):
        super().__init__()
        assert len(roots) == len(transforms) == len(modes) and roots
        self.sources = [BufferedSlippyMapDirectory(r, transform=t, size=size, overlap=overlap, mode=m)
                        for r, t, m in zip(roots, transforms, modes)]
        first = [tile for tile, _ in self.sources[0].tiles]
        for src in self.sources[1:]:
            assert [tile for tile, _ in src.tiles] == first, "same tiles in all image directories"
        self.cat_dim, self.overlap, self.size = cat_dim, overlap, size

    def __len__(self):
        return len(self.sources[0])

    def __getitem__(self, i):
        items = [src[i] for src in self.sources]
        return torch.cat([image for image, _ in items], dim=self.cat_dim), items[0][1]

    def unbuffer(self, probs):
        return self.sources[0].unbuffer(probs)


def draw_flip_rotations():
    """The four draws of the random part of the reference's training transform (tools/train.py:254-257:
    ``JointRandomHorizontalFlip(0.5)`` then three ``JointRandomRotation(0.5, 90)``), in its order, from Python's ``random``
    as it does -- as ONE op code (flip + 2 * quarter turns) for ``rs_augment_tiles`` instead of four PIL transposes."""
    import random

    flip = random.random() < 0.5
    turns = sum(random.random() < 0.5 for _ in range(3))
    return int(flip) + 2 * turns


class UnaugmentedTiles(torch.utils.data.Dataset):
    """Items ``(image uint8 [S,S,C], mask uint8 [S,S], op code, tiles)``: the DETERMINISTIC head of the reference's transform
    chain -- mode conversion, resize, centre crop (tools/train.py:250-253) -- done on the host (in a DataLoader worker), i.e.
    exactly the pixels its random flips / rotations and ``ToTensor`` + ``Normalize`` start from.  With ``draw=True`` the four
    random numbers of the random part are drawn right here, where the reference's chain draws them (same worker, same order),
    and travel as an op code; the transposes, ``ToTensor`` and ``Normalize`` then run on the device (``rs_augment_tiles``).
    A tile crosses the DataLoader's queues and PCIe as 1 MiB of bytes instead of 5 MiB of floats + int64 labels."""

    def __init__(self, image_dirs, label_dir, size, draw, modes=None):
        super().__init__()
        from .transforms import CenterCrop, ConvertImageMode, Resize

        self.source = SlippyMapTilesConcatenation(image_dirs, label_dir, joint_transform=None)
        target = (size, size)
        modes = list(modes) if modes is not None else ["RGB"] * len(image_dirs)  # (the reference converts every source to RGB)
        assert len(modes) == len(image_dirs), "one mode per image directory"
        self.to_image = [[ConvertImageMode(m), Resize(target, Image.BILINEAR), CenterCrop(target)] for m in modes]
        self.to_mask = [ConvertImageMode("P"), Resize(target, Image.NEAREST), CenterCrop(target)]
        self.draw = draw

    def __len__(self):
        return len(self.source)

    def __getitem__(self, i):
        import numpy as np

        tiles_and_images = [ds[i] for ds in self.source.inputs]
        mask, mask_tile = self.source.target[i]
        assert all(tile == mask_tile for _, tile in tiles_and_images), "image tile is the same as label tile"
        planes = []
        for (image, _), chain in zip(tiles_and_images, self.to_image):
            for fn in chain:
                image = fn(image)
            plane = np.asarray(image, dtype=np.uint8)
            planes.append(plane if plane.ndim == 3 else plane[:, :, None])  # (a mode-L source is one band)
        for fn in self.to_mask:
            mask = fn(mask)
        image = torch.from_numpy(np.ascontiguousarray(np.concatenate(planes, axis=2)))
        mask = torch.from_numpy(np.array(mask, dtype=np.uint8))
        code = draw_flip_rotations() if self.draw else 0
        return image, mask, code, [tile for _, tile in tiles_and_images]


class DecodedTileCache:
    """Every (image, mask) tile of a training / validation split decoded ONCE into uint8 tensors in HBM (SURVEY.md section
    8f, N4): the items of ``UnaugmentedTiles`` (decoded by ``workers`` DataLoader processes).  A 512x512 RGB tile + its
    mask is 1 MiB: 100 000 tiles fit the MI355X's 288 GB with room to spare, and an epoch then costs no PNG/JPEG decode at all."""

    def __init__(self, image_dirs, label_dir, size, device, workers=0, modes=None):
        source = UnaugmentedTiles(image_dirs, label_dir, size, draw=False, modes=modes)
        images, masks = [], []
        for image, mask, _, _ in torch.utils.data.DataLoader(source, batch_size=64, num_workers=workers):
            images.append(image.to(device, non_blocking=True))
            masks.append(mask.to(device, non_blocking=True))
        self.tiles = [tile for tile, _ in source.source.target.tiles]
        self.size = size
        self.images = torch.cat(images)  # [T, S, S, C] uint8
        self.masks = torch.cat(masks)    # [T, S, S] uint8
        self.device = device

    def __len__(self):
        return len(self.tiles)


def _per_band(values, channels):
    """``values`` as one entry per band: as given when it has that many (``robosat_amd.bands``), else the reference's 3-entry
    RGB statistics repeated per RGB source (tools/train.py:246 applies the same ``Normalize`` to every image)."""

    values = list(values)
    return values if len(values) == channels else (values * channels)[:channels]


class HostDecodeLoader:
    """The reference's training / validation ``DataLoader`` (tools/train.py:262-274) with its transform chain split where
    it stops being deterministic: ``workers`` processes decode, convert, resize, crop and DRAW (``UnaugmentedTiles``); flip /
    rot90 / ``ToTensor`` / ``Normalize`` run in one kernel on the device.  Yields what the reference's loader yields --
    ``(images [N,C,S,S] fp32 normalised, masks [N,S,S] int64, tiles)`` -- already on the device; same augmentation
    distribution, same seeded draws per worker as the host chain, bit-equal tensors (tests/test_gpu_tools.py)."""

    def __init__(self, image_dirs, label_dir, size, batch_sampler, workers, device, mean, std, modes=None):
        self.dataset = UnaugmentedTiles(image_dirs, label_dir, size, draw=True, modes=modes)
        self.loader = torch.utils.data.DataLoader(self.dataset, num_workers=workers, pin_memory=True, batch_sampler=batch_sampler)
        self.device, self.mean, self.std = device, list(mean), list(std)
        self.batch_sampler = batch_sampler  # (``set_epoch`` of the sharded sampler is reached through it, as on a DataLoader)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        from . import ops

        for images, masks, codes, tiles in self.loader:
            images = images.to(self.device, non_blocking=True)
            masks = masks.to(self.device, non_blocking=True)
            n, channels = images.shape[0], images.shape[3]
            mean, std = _per_band(self.mean, channels), _per_band(self.std, channels)
            index = torch.arange(n, dtype=torch.int32, device=self.device)
            op = codes.to(torch.int32).to(self.device, non_blocking=True)
            out, om = ops.augment_tiles(images, masks, index, op, mean, std)
            yield out, om, tiles


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

    draw_op = staticmethod(draw_flip_rotations)

    def __iter__(self):
        from . import ops

        c = self.cache
        channels = c.images.shape[3]
        mean, std = _per_band(self.mean, channels), _per_band(self.std, channels)
        for batch in self.batch_sampler:
            codes = [self.draw_op() for _ in batch]
            index = torch.tensor(batch, dtype=torch.int32).to(c.device, non_blocking=True)
            op = torch.tensor(codes, dtype=torch.int32).to(c.device, non_blocking=True)
            images, masks = ops.augment_tiles(c.images, c.masks, index, op, mean, std)
            yield images, masks, [[c.tiles[i]] for i in batch]
