"""8-bit PNG files written without the interpreter lock held.

``rs predict`` / ``rs masks`` store one PNG per tile (reference ``tools/predict.py:105-113``, ``tools/masks.py:66-70``:
``Image.fromarray(...).putpalette(...).save(path, optimize=True)``).  Pillow's PNG encoder keeps the GIL while it deflates
(~3-10 ms per 512x512 tile), so a pool of writer *threads* encodes one tile at a time and the tool tops out near 50 tiles/s
however fast the GPU is (``profiles/r02/loader_bench.txt``, round 2).  zlib's ``compressobj`` releases the GIL: the same pixels
and palette written as signature + IHDR [+ PLTE] + IDAT + IEND here scale with the writer threads.  The files decode to
exactly what the reference's decode to (same mode, pixels and palette; tests/test_tools_oracle.py); they are not the same
bytes on disk (deflate parameters differ between Pillow versions as well).
"""

import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_COLOR_TYPE = {"P": (3, 1), "L": (0, 1), "LA": (4, 2), "RGB": (2, 3), "RGBA": (6, 4)}


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def encode_png(array, mode, palette=None, level=6):
    """PNG bytes of a uint8 array [H,W] (modes P, L) or [H,W,channels] (LA, RGB, RGBA); ``palette``: flat RGB list for P."""

    color_type, channels = _COLOR_TYPE[mode]
    a = np.ascontiguousarray(array, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] != channels:
        raise ValueError("mode {} needs {} channel(s), got an array of shape {}".format(mode, channels, array.shape))
    h, w = a.shape[:2]
    rows = np.zeros((h, 1 + w * channels), dtype=np.uint8)  # filter type 0 ("None") in front of every scanline
    rows[:, 1:] = a.reshape(h, w * channels)
    out = [_SIGNATURE, _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))]
    if mode == "P":
        if palette is None:
            raise ValueError("mode P needs a palette")
        pal = bytes(bytearray(palette))
        if len(pal) % 3 or not 3 <= len(pal) <= 768:
            raise ValueError("palette must hold 1..256 RGB triples")
        out.append(_chunk(b"PLTE", pal))
    # compressobj, not zlib.compress: both release the GIL inside deflate, but the one-shot call grows its output buffer under
    # the GIL in many small steps and does not scale across threads on poorly compressible tiles (measured: 126 vs 902
    # tiles/s with 8 threads on noise-like 512^2 tiles; smooth tiles ~3000/s either way)
    deflate = zlib.compressobj(level, zlib.DEFLATED, 15, 9)
    out.append(_chunk(b"IDAT", deflate.compress(rows.tobytes()) + deflate.flush()))
    out.append(_chunk(b"IEND", b""))
    return b"".join(out)


def write_png(path, array, mode, palette=None, level=6):
    data = encode_png(array, mode, palette, level)
    with open(path, "wb") as f:
        f.write(data)
