"""Where a tile's time goes inside stem_conv_f32: shader-clock stamps of wave 0 of every block at eight points of the tile loop
(the -DRS_STEM_TRACE build of stem_f32.hip as gpurun_in/libstemtrace.so:
`cd robosat_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRS_STEM_TRACE -shared stem_f32.hip -o ../../gpurun_in/libstemtrace.so`).  Prints the median cycles of each phase over all blocks and tiles."""
import ctypes
import os
import sys

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_in", os.environ.get("STEM_TRACE_LIB", "libstemtrace.so")))
DEV = "cuda:0"
N, H, W = 16, 512, 512
g = torch.Generator().manual_seed(0)
x4 = torch.zeros(N, H, W, 4, device=DEV)
x4[..., :3] = torch.randn(N, H, W, 3, generator=g).to(DEV)
w = (torch.randn(64, 7, 8, 4, generator=g) * 0.05).to(DEV)
w[:, :, 7, :] = 0
w[..., 3] = 0
out = torch.empty(N, H // 2, W // 2, 64, device=DEV)
for grid in (512, 256):
    trace = torch.zeros(grid * 64 * 8, dtype=torch.int64, device=DEV)
    for _ in range(3):
        rc = lib.rs_stem_f32_trace(N, H, W, ctypes.c_void_p(x4.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                   ctypes.c_void_p(trace.data_ptr()), grid)
        assert rc == 0
    torch.cuda.synchronize()
    tiles = N * (H // 2) * 2 // grid
    t = trace.view(grid, 64, 8)[:, :tiles].cpu().double()
    names = ["barrier (strip in place)", "fetch issue", "MFMA loop", "barrier", "stage write + barrier + read + barrier", "put (next strip)", "scale / relu / stores"]
    print("grid", grid, "tiles per block", tiles, "| whole tile (stamp 0 -> next stamp 0), median cycles:", float((t[:, 1:, 0] - t[:, :-1, 0]).median()),
          "| block life / tiles:", float(((t[:, -1, 7] - t[:, 0, 0]) / tiles).median()))
    for k in range(7):
        d = t[:, 1:-1, k + 1] - t[:, 1:-1, k]
        print("   %-42s median %8.0f   p10 %8.0f   p90 %8.0f cycles" % (names[k], float(d.median()), float(d.quantile(0.1)), float(d.quantile(0.9))))
    d = t[:, 2:, 0] - t[:, 1:-1, 7]
    print("   %-42s median %8.0f" % ("loop back edge", float(d.median())))
    full = trace.view(grid, 64, 8).cpu().double()
    entry, ready, done = full[:, 63, 0], full[:, 63, 1], full[:, 63, 2]
    print("   per block: entry -> filter staged %.0f | -> first tile's top %.0f | last tile's end -> exit %.0f | entry -> exit %.0f (medians)" % (
        float((ready - entry).median()), float((t[:, 0, 0] - ready).median()), float((done - t[:, -1, 7]).median()), float((done - entry).median())))
    for x in range(8):  # (blocks b = x mod 8 share an XCD and its counter)
        sel = torch.arange(grid) % 8 == x
        print("   XCD %d: first entry -> last exit %.0f cycles; entries spread over %.0f, exits over %.0f" % (
            x, float(done[sel].max() - entry[sel].min()), float(entry[sel].max() - entry[sel].min()), float(done[sel].max() - done[sel].min())))
