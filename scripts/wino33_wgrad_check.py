"""fp32 weight gradient of the stride-1 3x3 convolutions in the Winograd domain of F(2x2, 3x3) (conv_wgrad_wino33_f32.hip) against the
nine-tap kernel and torch autograd, then the benchmark's U-Net layers at bs 8 (fp32 train leg) timed in both forms.  GPU box only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from robosat_amd import _lib, ops

DEV = "cuda:0"
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(DEV)

ok = True
for (n, cin, cout, h, w) in [(2, 64, 64, 8, 8), (1, 32, 32, 9, 13), (3, 64, 128, 16, 12), (2, 128, 64, 7, 5), (1, 64, 96, 33, 20), (5, 32, 64, 16, 16), (2, 64, 64, 34, 50)]:
    a = rnd(n, cin, h, w, seed=1)
    wt = (rnd(cout, cin, 3, 3, seed=3) * 0.05).requires_grad_(True)
    y = F.conv2d(a, wt, padding=1)
    gy = rnd(*y.shape, seed=4)
    y.backward(gy)
    args = (nhwc(gy), nhwc(a), 3, 3)
    d = ops.ConvDesc(n, h, w, cin, 0, 0, 3, 3, 1, 1, h, w, cout, 0, 0)
    form = _lib.lib().rs_conv2d_wgrad_form(ctypes.byref(d))
    dw = ops.conv2d_wgrad(*args, pad=1)
    with ops.knob("wgrad_f32_wino33", 0):
        nine = ops.conv2d_wgrad(*args, pad=1)
    ref = wt.grad.permute(0, 2, 3, 1).to(DEV)
    e1 = float((dw - nine).abs().max() / nine.abs().max()); e2 = float((dw - ref).abs().max() / ref.abs().max()); e3 = float((nine - ref).abs().max() / ref.abs().max())
    good = form == 4 and e1 < 2e-5 and e2 < 2e-5
    ok &= good
    print((n, cin, cout, h, w), "form", form, "| vs nine taps %.2e | vs autograd %.2e (nine taps vs autograd %.2e)" % (e1, e2, e3), "ok" if good else "BAD")
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
for name, (c, hs, cnt) in {"layer1.conv2": (64, 128, 3), "layer2.conv2": (128, 64, 3), "layer3.conv2": (256, 32, 5), "layer4.conv2": (512, 16, 2), "dec5": (32, 512, 1)}.items():
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn(bs, hs, hs, c, device=DEV, generator=g)
    x = torch.randn(bs, hs, hs, c, device=DEV, generator=g)
    out = torch.empty(c, 3, 3, c, device=DEV)
    res = {}
    for blocks in (128, 256, 512):
        with ops.knob("wgrad_f32_wino33_blocks", blocks):
            res[blocks] = timeit(lambda: ops.conv2d_wgrad(dy, x, 3, 3, pad=1, out=out))
    with ops.knob("wgrad_f32_wino33", 0):
        t9 = timeit(lambda: ops.conv2d_wgrad(dy, x, 3, 3, pad=1, out=out))
    gf = 2.0 * bs * hs * hs * 9 * c * c / 1e9
    tw = res[256]
    tot[0] += cnt * t9; tot[1] += cnt * tw
    print("%s x%d weight gradient bs %d: nine taps %.3f ms (%.0f TF) | winograd blocks 128/256/512: %.3f / %.3f / %.3f ms (%.0f TF executed at 256) | x%.2f" %
          (name, cnt, bs, t9, gf / t9, res[128], res[256], res[512], gf * 16 / 36 / tw, t9 / tw))
print("sum over the step's launches: nine taps %.3f ms, winograd %.3f ms" % tuple(tot))
