"""Micro-benchmark of single convolution launches through the C ABI (measurement tool; not part of the product path).

usage: python scripts/bench_layer.py [--variants "auto 128x128/64 256x128/128 ..."] [--iters N] SPEC [SPEC ...]

SPEC = dtype:form:N,C1[+C2],H,W,Cout[,k,stride,pad]
  dtype  f32 | bf16
  form   conv   plain convolution, eval epilogue (scale/shift + ReLU; "+res" adds the residual: Bottleneck conv3)
         relu   plain convolution with only a ReLU ("+mask": only a ReLU mask): ConvRelu / its data gradient
         stats  train-mode forward: raw output + fused BatchNorm partial sums
         bwd    data gradient into a BatchNorm: ReLU mask + bn_y statistics (+res)
         phase  DecoderBlock in phase form on cat[C1, C2] at source size H x W (k/stride/pad ignored)
         dg4    the phase form's 4x4 / stride-2 data gradient: input dz [N, 2H, 2W, C1], output [N, H, W, Cout]
  e.g.   f32:conv+res:16,64,128,128,256,1,1,0     bf16:phase:32,256+64,128,128,128

Each variant forces the dispatcher (`rs_conv2d_set_tuning`; "auto" = the measured heuristics) and prints ms, the launched
symbol, algorithmic HBM GB/s and executed TFLOP/s.  Inputs rotate over enough buffers to exceed the 256 MB Infinity
Cache, so reads come from HBM as they do in a real step."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robosat_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--variants", type=str, default="auto")
ap.add_argument("specs", nargs="+")
a = ap.parse_args()
dev = torch.device("cuda:0")


def variants():
    for v in a.variants.split():
        if v == "auto":
            yield v, None, 0
        else:
            t, r = v.split("/")
            yield v, t, int(r)


def rot(shape, dt, nbytes):
    n = max(2, min(8, int(600e6 // max(1, nbytes)) + 1))
    return [torch.randn(*shape, device=dev).to(dt) for _ in range(n)]


for spec in a.specs:
    dts, form, dims = spec.split(":")
    dt = torch.bfloat16 if dts == "bf16" else torch.float32
    es = 2 if dts == "bf16" else 4
    form, _, opt = form.partition("+")
    parts = dims.split(",")
    n = int(parts[0])
    c1, _, c2 = parts[1].partition("+")
    c1, c2 = int(c1), int(c2 or 0)
    h, w, cout = int(parts[2]), int(parts[3]), int(parts[4])
    k, st, pad = (int(parts[5]), int(parts[6]), int(parts[7])) if len(parts) > 5 else (3, 1, 1)
    res = opt == "res"
    if form == "phase":
        xs = rot((n, h, w, c1), dt, n * h * w * c1 * es)
        x2 = torch.randn(n, h, w, c2, device=dev).to(dt) if c2 else None
        wk = torch.randn(cout, 3, 3, c1 + c2, device=dev) * 0.02
        wt = ops.pack_phase_weight(wk, dt)
        fn = lambda x: ops.conv2d_phase(x, wt, src2=x2, relu=True)
        d = _lib.ConvDesc(n, h, w, c1, c2, 1, 3, 3, 1, 1, 2 * h, 2 * w, cout, 1, 0)
        name = lambda: ops.conv_tile_name(d, es == 2, phase=True)
        flops = 2.0 * n * 4 * h * w * cout * (c1 + c2) * 4
        nbytes = es * (n * h * w * (c1 + c2) + n * 4 * h * w * cout)
    elif form == "dg4":
        xs = rot((n, 2 * h, 2 * w, c1), dt, n * 4 * h * w * c1 * es)
        wt = (torch.randn(cout, 4, 4, c1, device=dev) * 0.02).to(dt)
        zm = torch.randn(n, h, w, cout, device=dev).to(dt) if opt == "mask" else None
        fn = lambda x: ops.conv2d(x, wt, stride=2, pad=1, out_hw=(h, w), relu_mask=zm)
        def name():
            nm = ops.conv_tile_name(ops.conv_desc(xs[0], wt, stride=2, pad=1, out_hw=(h, w)), es == 2)
            return nm if nm.startswith("conv_thin") else nm.replace("<", "<dgrad4x4,")
        flops = 2.0 * n * h * w * cout * c1 * 16
        nbytes = es * (n * 4 * h * w * c1 + (2 if opt == "mask" else 1) * n * h * w * cout)
    else:
        ho, wo = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
        xs = rot((n, h, w, c1), dt, n * h * w * c1 * es)
        wt = (torch.randn(cout, k, k, c1, device=dev) * 0.05).to(dt)
        outshape = (n, ho, wo, cout)
        extra = 0
        if form == "conv":
            sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
            r = torch.randn(*outshape, device=dev).to(dt) if res else None
            extra = 1 if res else 0
            fn = lambda x: ops.conv2d(x, wt, stride=st, pad=pad, scale=sc, shift=sh, residual=r, relu=True)
        elif form == "relu":
            z = torch.randn(*outshape, device=dev).to(dt) if opt == "mask" else None
            extra = 1 if z is not None else 0
            fn = lambda x: ops.conv2d(x, wt, stride=st, pad=pad, relu=z is None, relu_mask=z)
        elif form == "stats":
            fn = lambda x: ops.conv2d_bnstats(x, wt, stride=st, pad=pad)
        elif form == "bwd":
            y, z = torch.randn(*outshape, device=dev).to(dt), torch.randn(*outshape, device=dev).to(dt)
            mean, inv = torch.randn(cout, device=dev), torch.rand(cout, device=dev) + 0.5
            r = torch.randn(*outshape, device=dev).to(dt) if res else None
            extra = 2 + (1 if res else 0)
            fn = lambda x: ops.conv2d_dgrad_bnstats(x, wt, (ho, wo), y, mean, inv, pad=pad, residual=r, relu_mask=z)
        else:
            raise SystemExit("unknown form " + form)
        name = lambda: ops.conv_tile_name(ops.conv_desc(xs[0], wt, stride=st, pad=pad), es == 2)
        flops = 2.0 * n * ho * wo * cout * c1 * k * k
        nbytes = es * (n * h * w * c1 + (1 + extra) * n * ho * wo * cout)
    for label, tile, rowb in variants():
        try:
            with ops.tuning(tile, rowb):
                sym = name()
                fn(xs[0])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.iters):
                    fn(xs[i % len(xs)])
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print("%-44s %-14s %-40s ms=%.4f  %6.0f GB/s  %7.1f TF" % (spec, label, sym, ms, nbytes / ms / 1e6, flops / ms / 1e9), flush=True)
        except Exception as exc:  # a variant the launch cannot run (e.g. Cout not a multiple of the tile)
            print("%-44s %-14s skipped: %s" % (spec, label, exc), flush=True)
