"""conv1x1_np_f32 (knob conv1x1_np = 1) against the generic fp32 1x1 kernel: bit-identical results on ragged / small / benchmark shapes, then
the time of every stride-1 1x1 launch of the bs-16 predict pass, both kernels, alternating (inputs rotated past the Infinity Cache)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import ops

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(1)
bad = 0
for (n, h, w, cin, cout) in [(2, 16, 16, 64, 256), (3, 17, 13, 32, 64), (1, 64, 64, 256, 64), (2, 32, 32, 512, 128), (5, 40, 24, 64, 192), (16, 32, 32, 256, 1024),
                             (1, 8, 8, 2048, 512), (2, 9, 7, 128, 512), (1, 5, 5, 96, 64), (3, 31, 29, 160, 320), (16, 64, 64, 128, 512)]:
    for sc, res, relu in ((1, 1, 1), (0, 0, 0), (1, 0, 1), (0, 1, 0)):
        x = torch.randn(n, h, w, cin, device=DEV, generator=g)
        wt = torch.randn(cout, 1, 1, cin, device=DEV, generator=g) * 0.05
        s = torch.rand(cout, device=DEV, generator=g) + 0.5 if sc else None
        b = torch.randn(cout, device=DEV, generator=g) if sc else None
        r = torch.randn(n, h, w, cout, device=DEV, generator=g) if res else None
        with ops.knob("conv1x1_np", 0), ops.knob("conv1x1_ew", 0):
            ref = ops.conv2d(x, wt, scale=s, shift=b, residual=r, relu=bool(relu))
        with ops.knob("conv1x1_np", 1), ops.knob("conv1x1_ew", 0):
            got = ops.conv2d(x, wt, scale=s, shift=b, residual=r, relu=bool(relu))
        torch.cuda.synchronize()
        ok = torch.equal(ref, got)
        bad += 0 if ok else 1
        if not ok:
            print("  MISMATCH", (n, h, w, cin, cout), sc, res, relu, float((ref - got).abs().max()), int((ref != got).sum()))
print("PARITY OK (bit-identical, 44 cases)" if bad == 0 else "PARITY FAILED: %d cases" % bad, flush=True)


def bench(n, hw, cin, cout, res, knob, relu=True):
    nbuf = max(2, int(600e6 // (n * hw * hw * max(cin, cout) * 4)) + 1)
    xs = [torch.randn(n, hw, hw, cin, device=DEV) for _ in range(nbuf)]
    rs = [torch.randn(n, hw, hw, cout, device=DEV) for _ in range(nbuf)] if res else [None] * nbuf
    wt = torch.randn(cout, 1, 1, cin, device=DEV) * 0.05
    s, b = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV)
    out = torch.empty(n, hw, hw, cout, device=DEV)
    with ops.knob("conv1x1_np", knob):
        for i in range(5):
            ops.conv2d(xs[i % nbuf], wt, scale=s, shift=b, residual=rs[i % nbuf], relu=relu, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            ops.conv2d(xs[i % nbuf], wt, scale=s, shift=b, residual=rs[i % nbuf], relu=relu, out=out)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


layers = [(64, 128, 256, 1, 4, "layer1 conv3 (EW by rule)"), (256, 128, 64, 0, 2, "layer1 conv1"), (256, 128, 128, 0, 1, "layer2.0 conv1"), (128, 64, 512, 1, 4, "layer2 conv3"),
          (512, 64, 128, 0, 3, "layer2 conv1"), (512, 64, 256, 0, 1, "layer3.0 conv1"), (256, 32, 1024, 1, 6, "layer3 conv3"), (1024, 32, 256, 0, 5, "layer3 conv1"),
          (1024, 32, 512, 0, 1, "layer4.0 conv1"), (512, 16, 2048, 1, 3, "layer4 conv3"), (2048, 16, 512, 0, 2, "layer4 conv1")]
tot = [0.0, 0.0]
for cin, hw, cout, res, cnt, name in layers:
    a1, b1 = bench(16, hw, cin, cout, res, 0), bench(16, hw, cin, cout, res, 1)
    a2, b2 = bench(16, hw, cin, cout, res, 0), bench(16, hw, cin, cout, res, 1)
    a, b = min(a1, a2), min(b1, b2)
    tot[0] += cnt * a
    tot[1] += cnt * b
    print("%5d -> %-5d at %3d^2 res=%d x%d  generic %7.1f %7.1f us   np %7.1f %7.1f us   x%.2f   %s" % (cin, cout, hw, res, cnt, a1, a2, b1, b2, a / b, name), flush=True)
print("sum over the pass's launches: generic %.1f us, np %.1f us" % (tot[0], tot[1]))

print("knock-outs on 256 -> 1024 at 32^2 and 128 -> 512 at 64^2 with residual (np kernel): relu bits 1 = as shipped, 3 = no stores, 5 = no residual loads, 7 = neither")
for cin, hw, cout in ((256, 32, 1024), (128, 64, 512)):
    print(cin, cout, [round(bench(16, hw, cin, cout, 1, 1, relu=r), 1) for r in (1, 3, 5, 7)], "generic", round(bench(16, hw, cin, cout, 1, 0), 1))
