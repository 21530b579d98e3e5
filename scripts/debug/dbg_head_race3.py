import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robosat_amd import ops
DEV = "cuda:0"; BF = torch.bfloat16
dl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdummy_neighbour.so"))
dl.launch_dummy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
srcbuf = torch.arange(16 << 20, device='cuda:0', dtype=torch.int32)
g = torch.Generator(device=DEV).manual_seed(29)
n, s, c, classes = 4, 256, 32, 2
w_krsc = torch.randn(c, 3, 3, c, device=DEV, generator=g) * 0.1
u = ops.pack_wino33_weight(w_krsc)
fw, fb = torch.randn(classes, c, device=DEV, generator=g) * 0.2, torch.randn(classes, device=DEV, generator=g)
side = torch.cuda.Stream()
sink = torch.zeros(16, device=DEV, dtype=torch.int32)
def run(label, blocks, lds, iters, mode, rounds=20):
    bad = 0
    for r in range(rounds):
        x = torch.randn(n, s, s, c, device=DEV, generator=g)
        torch.cuda.synchronize()
        ref = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
        torch.cuda.synchronize()
        if blocks:
            assert dl.launch_dummy(blocks, lds, iters, mode, side.cuda_stream, sink.data_ptr(), srcbuf.data_ptr(), 64 << 20) == 0
        a = ops.conv2d_wino33_head(x, u, fw, fb, mode="logits")
        torch.cuda.synchronize()
        bad += int(not torch.equal(a, ref))
    print(label, "| wrong launches", bad, "of", rounds, flush=True)
run("alone", 0, 0, 0, 0)
run("neighbour: bf16 MFMA loop, no LDS", 1024, 0, 6000, 3)
run("neighbour: packed-VALU loop, no LDS", 1024, 0, 20000, 5)
run("neighbour: LDS-DMA stream (32 KB of LDS, L2-resident source)", 1024, 32768, 300, 4)
run("neighbour: LDS-DMA stream, 512 blocks", 512, 32768, 600, 4)
run("neighbour: ds traffic, 29 KB", 1024, 29184, 20000, 1)
