#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v22; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3"
python - <<'PY' 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $OUT/check256.txt
import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from robosat_amd import ops
dev = "cuda:0"
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
q = lambda t: t.to(torch.bfloat16).float()
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
for (n, cin, h, w, cout, k, stride, pad) in [(2, 256, 20, 20, 256, 3, 1, 1), (1, 256, 15, 23, 512, 1, 1, 0), (3, 512, 9, 7, 256, 1, 1, 0), (2, 512, 16, 16, 512, 3, 2, 1)]:
    x = q(rnd(n, cin, h, w, seed=1)).requires_grad_(True)
    wt = (rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    gy = q(rnd(*y.shape, seed=3))
    y.backward(gy)
    res = {}
    for v in (0, 1):
        with ops.knob("wgrad_256", v):
            res[v] = ops.conv2d_wgrad(nhwc(gy), nhwc(x.detach()), k, k, stride=stride, pad=pad)
    want = wt.grad
    e0 = float((res[0].permute(0, 3, 1, 2).cpu() - want).abs().max() / want.abs().max())
    e1 = float((res[1].permute(0, 3, 1, 2).cpu() - want).abs().max() / want.abs().max())
    d = float((res[0] - res[1]).abs().max() / res[0].abs().max())
    print((n, cin, h, w, cout, k, stride, pad), "rel err 256x128 %.2e  256x256 %.2e  between %.2e" % (e0, e1, d), "OK" if max(e0, e1) < 3e-4 and d < 2e-5 else "FAIL")
PY
run() { env "$1" timeout 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warm
for i in 1 2 3; do run RS_WGRAD_256=0; run RS_WGRAD_256=1; done
} | tee $OUT/wgrad256_ab.txt
