"""Slippy-map tile directories (``z/x/y.ext``) -- the feeder side of the hot path (reference ``robosat/tiles.py``).

Only what ``rs train`` / ``rs predict`` need: tile discovery and the overlap-buffered composite of a tile with its
eight neighbours.  ``Tile`` stands in for ``mercantile.Tile`` (same field order x, y, z)."""

import collections
import csv
import os

from PIL import Image

Tile = collections.namedtuple("Tile", ["x", "y", "z"])


def _as_int(text):
    try:
        return int(text)
    except ValueError:
        return None


def tiles_from_slippy_map(root):
    """Yields ``(Tile, path)`` for every ``root/z/x/y.*`` whose three components are integers."""

    for zname in os.listdir(root):
        z = _as_int(zname)
        if z is None:
            continue
        zdir = os.path.join(root, zname)
        for xname in os.listdir(zdir):
            x = _as_int(xname)
            if x is None:
                continue
            xdir = os.path.join(zdir, xname)
            for fname in os.listdir(xdir):
                y = _as_int(os.path.splitext(fname)[0])
                if y is not None:
                    yield Tile(x=x, y=y, z=z), os.path.join(xdir, fname)


def tiles_from_csv(path):
    """Yields a ``Tile`` per non-empty ``x,y,z`` row."""

    with open(path) as fp:
        for row in csv.reader(fp):
            if row:
                yield Tile(*(int(v) for v in row))


def buffer_tile_image(tile, tiles, overlap, tile_size, nodata=0, opener=None, mode="RGB"):
    """The tile's RGB image with an ``overlap``-pixel border taken from its 8 neighbours (``nodata`` where a
    neighbour is missing): size ``tile_size + 2*overlap`` squared (reference tiles.py:162-227).

    ``mode`` (extension; default the reference's "RGB") is the PIL mode of the composite: "L" for a single-band source.
    ``tiles`` is a mapping ``Tile -> path`` (or an iterable of pairs).  ``opener(path)`` returns the decoded image in that mode
    (default: ``Image.open(path).convert("RGB")``); ``BufferedSlippyMapDirectory`` passes a small LRU cache here, because
    the composite of every tile decodes nine files and neighbouring tiles share six of them."""

    store = tiles if isinstance(tiles, dict) else dict(tiles)
    size = tile_size + 2 * overlap
    composite = Image.new(mode=mode, size=(size, size), color=nodata)

    # per axis and neighbour offset: (destination start, source start, length)
    span = {-1: (0, tile_size - overlap, overlap), 0: (overlap, 0, tile_size), 1: (overlap + tile_size, 0, overlap)}

    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            path = store.get(Tile(x=int(tile.x) + dx, y=int(tile.y) + dy, z=int(tile.z)))
            if path is None:
                if dx == 0 and dy == 0:
                    raise KeyError(tile)
                continue
            (tx, sx, w), (ty, sy, h) = span[dx], span[dy]
            if w == 0 or h == 0:
                continue
            piece = Image.open(path).convert(mode) if opener is None else opener(path)
            if dx == 0 and dy == 0:
                composite.paste(piece, box=(tx, ty))
            else:
                composite.paste(piece.crop(box=(sx, sy, sx + w, sy + h)), box=(tx, ty, tx + w, ty + h))
    return composite
