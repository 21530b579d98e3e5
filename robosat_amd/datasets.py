"""Datasets over slippy-map tile directories: the class surface of the reference's ``robosat/datasets.py``
(``SlippyMapTiles``, ``SlippyMapTilesConcatenation``, ``BufferedSlippyMapDirectory``), feeding batches to the GPU path.
Parity with the reference's own classes on the same directory: ``tests/test_feeders.py``."""

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
    crops a prediction back to the tile.  The tile -> path map is built once (the reference rebuilds it per item)."""

    def __init__(self, root, transform=None, size=512, overlap=32):
        assert overlap >= 0
        assert size >= 256
        super().__init__(root, ordered=False)
        self.transform, self.size, self.overlap = transform, size, overlap
        self._store = dict(self.tiles)

    def __getitem__(self, i):
        tile = self.tiles[i][0]
        image = buffer_tile_image(tile, self._store, overlap=self.overlap, tile_size=self.size)
        if self.transform is not None:
            image = self.transform(image)
        return image, torch.IntTensor([tile.x, tile.y, tile.z])

    def unbuffer(self, probs):
        o = self.overlap
        return probs[:, o:probs.shape[1] - o, o:probs.shape[2] - o]
