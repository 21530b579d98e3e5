"""fp32 DecoderBlock weight gradient in the Winograd domain (conv_wgrad_wino_f32.hip) against the phase form and torch autograd, then
the five DecoderBlock layers of the benchmark's U-Net at bs 8 (fp32 train leg) timed in both forms.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from robosat_amd import ops

DEV = "cuda:0"
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(DEV)

ok = True
for (n, c1, c2, cout, h, w) in [(2, 64, 0, 64, 8, 8), (1, 128, 64, 32, 9, 13), (3, 64, 64, 64, 16, 12), (2, 192, 128, 128, 7, 5), (1, 64, 0, 96, 33, 20), (5, 64, 64, 32, 16, 16)]:
    a = rnd(n, c1, h, w, seed=1).requires_grad_(True)
    b = rnd(n, c2, h, w, seed=2).requires_grad_(True) if c2 else None
    wt = (rnd(cout, c1 + c2, 3, 3, seed=3) * 0.05).requires_grad_(True)
    cat = torch.cat([a, b], 1) if c2 else a
    y = F.conv2d(F.interpolate(cat, scale_factor=2, mode="nearest"), wt, padding=1)
    gy = rnd(*y.shape, seed=4)
    y.backward(gy)
    args = (nhwc(gy), nhwc(a.detach()), 3, 3)
    kw = dict(src2=nhwc(b.detach()) if c2 else None, ups=1, pad=1)
    d = ops.ConvDesc(n, h, w, c1, c2, 1, 3, 3, 1, 1, 2 * h, 2 * w, cout, 0, 0)
    import ctypes
    from robosat_amd import _lib
    form = _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d))
    dw = ops.conv2d_wgrad(*args, **kw)
    with ops.knob("wgrad_f32_wino", 0):
        ph = ops.conv2d_wgrad(*args, **kw)
    ref = wt.grad.permute(0, 2, 3, 1).to(DEV)
    e1 = float((dw - ph).abs().max() / ph.abs().max()); e2 = float((dw - ref).abs().max() / ref.abs().max()); e3 = float((ph - ref).abs().max() / ref.abs().max())
    good = form == 3 and e1 < 2e-5 and e2 < 2e-5
    ok &= good
    print((n, c1, c2, cout, h, w), "form", form, "| vs phase %.2e | vs autograd %.2e (phase form vs autograd %.2e)" % (e1, e2, e3), "ok" if good else "BAD")
print("PARITY OK" if ok else "PARITY FAILED")

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tot = [0.0, 0.0]
for name, (c1, c2, cout, hs) in {"center": (2048, 0, 256, 8), "dec0": (2048, 256, 256, 16), "dec1": (1024, 256, 256, 32), "dec2": (512, 256, 64, 64), "dec3": (256, 64, 128, 128), "dec4": (64, 64, 32, 256)}.items():
    g = torch.Generator(device=DEV).manual_seed(5)
    dz = torch.randn(bs, 2 * hs, 2 * hs, cout, device=DEV, generator=g)
    s1 = torch.randn(bs, hs, hs, c1, device=DEV, generator=g)
    s2 = torch.randn(bs, hs, hs, c2, device=DEV, generator=g) if c2 else None
    import ctypes
    from robosat_amd import _lib
    d = ops.ConvDesc(bs, hs, hs, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * hs, cout, 0, 0)
    form = _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d))
    out = torch.empty(cout, 3, 3, c1 + c2, device=DEV)
    res = {}
    for blocks in (512, 1024, 2048):
        with ops.knob("wgrad_f32_wino_blocks", blocks):
            res[blocks] = timeit(lambda: ops.conv2d_wgrad(dz, s1, 3, 3, src2=s2, ups=1, pad=1, out=out))
    with ops.knob("wgrad_f32_wino", 0):
        tp = timeit(lambda: ops.conv2d_wgrad(dz, s1, 3, 3, src2=s2, ups=1, pad=1, out=out))
    gf = 2.0 * bs * hs * hs * 16 * (c1 + c2) * cout / 1e9
    tw = res[1024]
    tot[0] += tp; tot[1] += tw if form == 3 else tp
    print("%s weight gradient bs %d (form %d): phase %.3f ms (%.0f TF executed) | winograd blocks 512/1024/2048: %.3f / %.3f / %.3f ms (%.0f TF executed at 1024) | x%.2f" %
          (name, bs, form, tp, gf / tp, res[512], res[1024], res[2048], gf * 9 / 16 / tw, tp / tw))
print("sum: phase %.3f ms, winograd %.3f ms" % tuple(tot))
