"""Datasets over slippy-map tile directories (reference ``robosat/datasets.py``), feeding batches to the GPU path."""

import torch
import torch.utils.data
from PIL import Image

from .tiles import buffer_tile_image, tiles_from_slippy_map


class SlippyMapTiles(torch.utils.data.Dataset):
    """One ``z/x/y.*`` directory; items are ``(image, tile)`` in sorted tile order."""

    def __init__(self, root, transform=None):
        super().__init__()
        self.transform = transform
        self.tiles = sorted(tiles_from_slippy_map(root), key=lambda entry: entry[0])

    def __len__(self):
        return len(self.tiles)

    def __getitem__(self, i):
        tile, path = self.tiles[i]
        image = Image.open(path)
        if self.transform is not None:
            image = self.transform(image)
        return image, tile


class SlippyMapTilesConcatenation(torch.utils.data.Dataset):
    """Several image directories (concatenated on the channel axis) plus one label directory; items are
    ``(images [sum C,H,W], mask [H,W], tiles)`` after the joint transform."""

    def __init__(self, inputs, target, joint_transform=None):
        super().__init__()
        self.joint_transform = joint_transform
        self.inputs = [SlippyMapTiles(inp) for inp in inputs]
        self.target = SlippyMapTiles(target)
        assert len({len(ds) for ds in self.inputs}) == 1, "same number of tiles in all images"
        assert len(self.target) == len(self.inputs[0]), "same number of tiles in images and label"

    def __len__(self):
        return len(self.target)

    def __getitem__(self, i):
        pairs = [ds[i] for ds in self.inputs]
        images, tiles = [p[0] for p in pairs], [p[1] for p in pairs]
        mask, mask_tile = self.target[i]
        assert len(set(tiles)) == 1, "all images are for the same tile"
        assert tiles[0] == mask_tile, "image tile is the same as label tile"
        if self.joint_transform is not None:
            images, mask = self.joint_transform(images, mask)
        return torch.cat(images, dim=0), mask, tiles


class BufferedSlippyMapDirectory(torch.utils.data.Dataset):
    """Tiles composited with an ``overlap`` border from their neighbours; ``unbuffer`` crops predictions back."""

    def __init__(self, root, transform=None, size=512, overlap=32):
        super().__init__()
        assert overlap >= 0
        assert size >= 256
        self.transform, self.size, self.overlap = transform, size, overlap
        self.tiles = list(tiles_from_slippy_map(root))
        self._store = dict(self.tiles)  # built once (the reference rebuilds this mapping for every item)

    def __len__(self):
        return len(self.tiles)

    def __getitem__(self, i):
        tile, _ = self.tiles[i]
        image = buffer_tile_image(tile, self._store, overlap=self.overlap, tile_size=self.size)
        if self.transform is not None:
            image = self.transform(image)
        return image, torch.IntTensor([tile.x, tile.y, tile.z])

    def unbuffer(self, probs):
        o = self.overlap
        _, h, w = probs.shape
        return probs[:, o:h - o, o:w - o]
