import sys, os
sys.path.insert(0, "/root/repo")
import torch
from robosat_amd import ops
DEV="cuda:0"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
bs=8
for name, (c1, c2, cout, hs) in {"dec1": (1024, 256, 256, 32), "dec3": (256, 64, 128, 128)}.items():
    g = torch.Generator(device=DEV).manual_seed(5)
    dz = torch.randn(bs, 2 * hs, 2 * hs, cout, device=DEV, generator=g)
    s1 = torch.randn(bs, hs, hs, c1, device=DEV, generator=g)
    s2 = torch.randn(bs, hs, hs, c2, device=DEV, generator=g)
    out = torch.empty(cout, 3, 3, c1 + c2, device=DEV)
    for blocks in (1024, 2048, 4096):
        with ops.knob("wgrad_f32_wino_blocks", blocks):
            t = timeit(lambda: ops.conv2d_wgrad(dz, s1, 3, 3, src2=s2, ups=1, pad=1, out=out))
        gf = 2.0 * bs * hs * hs * 9 * (c1 + c2) * cout / 1e9
        print(name, "ko", os.environ.get("ROBOSAT_HIP_LIB", "0")[-6:], "blocks", blocks, "%.3f ms  %.0f TF executed" % (t, gf / t))
