"""Does the predict pass gain from running consecutive batches on alternating HIP streams?

The bs-16 fp32 pass is a chain of launches that are either matrix-core bound (the Winograd DecoderBlocks, 3x3 layers: ~6 of
its 10 ms) or bound by HBM / the epilogue's stores (layer1/2's 1x1 group, stem, head: ~3 ms), one after the other on one
stream: each kind leaves the other resource idle.  Two batches in flight on two streams are at different depths of the
network most of the time, so an HBM-bound launch of one can share the chip with an MFMA-bound launch of the other.
`rs predict` streams batches anyway (tools/predict.py keeps one batch in flight on the host side already).

Measurement tool: same model, same resident inputs, K passes on 1 stream vs round-robin over S streams; outputs compared
bit for bit with the single-stream results.  usage: python scripts/two_stream_predict.py [--batch 16] [--size 512] [--steps 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--streams", type=int, nargs="*", default=[1, 2, 3, 2, 1])
ap.add_argument("--split", type=int, default=0, help="also: ONE batch split into this many sub-batches on as many streams, joined per step")
args = ap.parse_args()

dev = torch.device("cuda:0")
net = bench.build_model(2, dev, False, args.dtype, 3)
g = torch.Generator().manual_seed(100)
xs = [torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev) for _ in range(3)]
with torch.no_grad():
    want = [net.predict_probs(x).clone() for x in xs]
torch.cuda.synchronize()


def run(nstreams, steps):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream(dev)]
    outs = [None] * steps
    for s in streams:
        s.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]):
            outs[i] = net.predict_probs(xs[i % 3])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return el, outs


def run_split(parts, steps):
    """one batch per step, split into `parts` sub-batches on `parts` streams, all joined at the end of every step"""
    main = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    n = args.batch // parts
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        x = xs[i % 3]
        outs = []
        for k, s in enumerate(streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(net.predict_probs(x[k * n:(k + 1) * n]))
        for s in streams:
            main.wait_stream(s)
        last = outs
    torch.cuda.synchronize()
    return time.perf_counter() - t0, last


for ns in args.streams:
    run(ns, 6)  # warm-up: workspaces and allocator pools of the new streams
    el, outs = run(ns, args.steps)
    same = all(torch.equal(outs[i], want[i % 3]) for i in range(args.steps))
    print("streams {}: {:8.3f} ms per batch  {:8.1f} tiles/s   outputs identical to the single-stream pass: {}".format(
        ns, el / args.steps * 1e3, args.batch * args.steps / el, same), flush=True)
if args.split > 1:
    run_split(args.split, 6)
    el, outs = run_split(args.split, args.steps)
    got = torch.cat(outs)
    print("one batch as {} sub-batches on {} streams, joined every step: {:8.3f} ms per batch  {:8.1f} tiles/s   identical: {}".format(
        args.split, args.split, el / args.steps * 1e3, args.batch * args.steps / el, torch.equal(got, want[(args.steps - 1) % 3])))
