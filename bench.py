"""bench.py -- 512x512 tiles/sec of the U-Net hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S]

Workload (BASELINE.json configs[1]): ``rs predict`` of the ResNet-50 U-Net, batch 16 x 3x512x512 fp32 per GPU, 2
classes: layout conversion -> 61 convolutions -> fused final 1x1 + softmax, i.e. everything ``rs predict`` runs on the
device per batch (reference tools/predict.py:83-87).  Inputs are synthetic and already resident in HBM when the timed
region starts.  Tiles shard across ranks with no data-path collective ("weak" scaling: B tiles per GPU per step).

One JSON line on stdout (rank 0): metric/value/unit... (the predict leg), plus
  "train"        -- the train leg of the metric (configs[2]: bf16, bs 32 per GPU, fwd + Lovasz + bwd + gradient
                    all-reduce + Adam): value (tiles/s over all ranks), ms_per_step, its own roofline object;
  "roofline"     -- dominant kernel, algorithmic FLOPs / HIP-event time over the launches of one pass, vs the fp32
                    MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s);
  "cpu_baseline" -- the CPU oracle (oracle/robosat_ref.py, kind "port") timed on this box's host cores on a bounded
                    sample of the same workload (N=1, rank 0 only).
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same table: dense bf16 (v_mfma_f32_32x32x16_bf16); never the 2:1-sparsity figure
HBM_PEAK_GBS = 8000.0  # same guide: HBM3E ~8 TB/s


def kernel_peak(name):
    return BF16_MFMA_PEAK_TFLOPS if "bf16" in name else FP32_MFMA_PEAK_TFLOPS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--phase", choices=["predict", "train"], default="predict",
                    help="predict = BASELINE configs[1] (default); train = fwd + loss + bwd + grad all-reduce + Adam")
    ap.add_argument("--loss", choices=["CrossEntropy", "Lovasz", "Focal"], default="Lovasz", help="train phase criterion")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                    help="compute dtype: fp32 (exact-fp32 MFMA, the parity path; BASELINE configs[1]) or bf16 (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-line comparison of one output tile with the CPU oracle")
    ap.add_argument("--no-train-leg", action="store_true", help="predict phase only: skip the bf16 train leg reported under \"train\"")
    ap.add_argument("--train-batch", type=int, default=32, help="tiles per GPU per step of the train leg (configs[2]: 32)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU-oracle sample")
    ap.add_argument("--layers-json", type=str, default="", help="also dump the per-layer roofline table here")
    return ap.parse_args()


def build_model(classes, device, train=False, dtype="fp32"):
    from robosat_amd.unet import UNet

    torch.manual_seed(0)
    net = UNet(classes, pretrained=False, compute_dtype=dtype)  # random init of the reference architecture
    g = torch.Generator().manual_seed(1)
    for name, buf in net.named_buffers():  # non-trivial BatchNorm statistics (fresh init would make BN an identity)
        if name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    net = net.to(device)
    return net.train() if train else net.eval()


def roofline(step):
    """HIP events around every convolution launch of one predict pass (same stream as the launches)."""

    from robosat_amd import ops

    # per-launch timings must not overlap each other: the train step's weight-gradient side stream is switched off for the
    # profiled pass (the TIMED steps of run_phase keep it)
    keep = os.environ.get("ROBOSAT_WGRAD_STREAM")
    os.environ["ROBOSAT_WGRAD_STREAM"] = "0"
    try:
        step()
        torch.cuda.synchronize()
        ops.PROFILE = []
        step()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
    finally:
        if keep is None:
            del os.environ["ROBOSAT_WGRAD_STREAM"]
        else:
            os.environ["ROBOSAT_WGRAD_STREAM"] = keep
    per_kernel, layers = {}, []
    for name, flops, shape, e0, e1, nbytes, executed in recs:
        ms = e0.elapsed_time(e1)
        k = per_kernel.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0])
        k[0] += flops
        k[1] += ms
        k[2] += 1
        k[3] += nbytes
        k[4] += executed
        layers.append({"kernel": name, "cin_cout_k_stride_ups_ho_wo": list(shape), "gflop": flops / 1e9, "ms": ms,
                       "tflops": flops / ms / 1e9 if ms > 0 else 0.0})
    dom = max(per_kernel, key=lambda n: per_kernel[n][1])
    flops, ms, launches, alg_bytes, exe_flops = per_kernel[dom]
    total_ms = sum(v[1] for v in per_kernel.values())
    total_fl = sum(v[0] for v in per_kernel.values())
    # time-weighted peak of the launches (fp32 and bf16 kernels coexist in the bf16 path: the stem stays fp32)
    ideal_ms = sum(v[4] / kernel_peak(n) / 1e9 for n, v in per_kernel.items())  # on EXECUTED flops
    total_exe = sum(v[4] for v in per_kernel.values())
    peak = kernel_peak(dom)
    # which roof binds the dominant kernel: its EXECUTED flops at the dense MFMA peak of its dtype, or its algorithmic bytes
    # (every tensor once) at HBM_PEAK -- whichever takes longer
    t_mfma, t_hbm = exe_flops / peak / 1e9, alg_bytes / HBM_PEAK_GBS / 1e6
    out = {"bound": "mfma" if t_mfma >= t_hbm else "hbm", "kernel": dom}
    if t_mfma >= t_hbm:
        # what the matrix cores EXECUTE per second against their peak.  (The decoder's phase form computes conv3x3 over a
        # nearest-x2 upsample as four 2x2 convolutions with pre-summed taps: 4/9 of the reference-shape multiply-adds.  Its
        # rate in those algorithmic FLOPs -- `algorithmic` below -- can exceed the MFMA peak and is not a roofline fraction.)
        achieved = exe_flops / ms / 1e9
        out.update({"achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4)})
    else:
        achieved = alg_bytes / ms / 1e6
        out.update({"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4)})
    out.update({
        "traffic": pmc_traffic(dom), "traffic_source": traffic_source(), "launches": launches,
        "executed": {"tflops": round(exe_flops / ms / 1e9, 2), "frac": round(exe_flops / ms / 1e9 / peak, 4)},
        "algorithmic": {"tflops": round(flops / ms / 1e9, 2), "gflop_per_launch": round(flops / launches / 1e9, 3),
                        "what": "reference-shape FLOPs, 2*N*Cout*Cin*k*k*Ho*Wo (SURVEY.md section 8d)"},
        "hbm": {"algorithmic_gbs": round(alg_bytes / ms / 1e6, 1), "frac": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4)},
        "avg_launch_ms": round(ms / launches, 4), "gflop_per_launch": round(exe_flops / launches / 1e9, 3),
        "algorithmic_bytes_per_launch": round(alg_bytes / launches),
        "all_convs": {"tflops": round(total_fl / total_ms / 1e9, 2), "executed_tflops": round(total_exe / total_ms / 1e9, 2),
                      "executed_frac": round(ideal_ms / total_ms, 4), "ms": round(total_ms, 3), "gflop": round(total_fl / 1e9, 2),
                      "executed_gflop": round(total_exe / 1e9, 2),
                      # the launches' roofline times (max of the MFMA and the HBM time of each) over their measured times
                      "roofline_frac": round(sum(max(v[4] / kernel_peak(n) / 1e9, v[3] / HBM_PEAK_GBS / 1e6)
                                                 for n, v in per_kernel.items()) / total_ms, 4)},
        "per_kernel": {n: {"tflops": round(v[0] / v[1] / 1e9, 2), "executed_tflops": round(v[4] / v[1] / 1e9, 2),
                           "gbs": round(v[3] / v[1] / 1e6, 1), "ms": round(v[1], 3), "launches": v[2]}
                       for n, v in per_kernel.items()},
    })
    return out, layers


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
    passes, corrected as MI355X_MICROARCH.md prescribes) -- profiles/pmc_traffic.json, written by scripts/pmc_summary.py
    from a run of this same command.  None when no counter run has been committed for this kernel."""

    return _pmc_table().get(kernel)


def _pmc_table():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fp:
            return json.load(fp)
    except (OSError, ValueError):
        return {}


def traffic_source():
    """Where `roofline.traffic` comes from: the counter passes are separate rocprofv3 runs (a --pmc run cannot share a
    process with this one), so the number is read from the committed summary; `_meta` there names the commit, the command
    and the date of the passes (scripts/gpu_round.sh writes it), and is passed through so a reader can tell whether the
    kernel has changed since."""

    meta = _pmc_table().get("_meta")
    return {"file": "profiles/pmc_traffic.json", "measured": meta} if meta else {"file": "profiles/pmc_traffic.json", "measured": None}


def cpu_baseline(classes, size, budget_s, phase="predict", loss_name="Lovasz"):
    """The CPU oracle on this box's host cores: bounded sample of the same workload (tiles of the same size).

    torch's intra-op pool does not scale to every core of a 2-socket host for these convolutions, so the thread count
    is chosen by a short sweep on a quarter-size tile and reported as ``cores``."""

    from oracle import robosat_ref as R

    torch.manual_seed(0)
    net = R.UNetRef(classes)
    net = net.train() if phase == "train" else net.eval()
    crit = R.LOSSES[loss_name]
    opt = torch.optim.Adam(net.parameters(), lr=1e-4) if phase == "train" else None

    def run(x, t):
        if phase == "predict":
            R.predict_probs(net, x)
        else:
            opt.zero_grad()
            out = net(x)
            l = crit(out, t) if loss_name == "Lovasz" else crit(out, t, weight=torch.ones(classes))
            l.backward()
            opt.step()

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({c for c in (8, 16, 32, 64, 128) if c <= avail} | {min(avail, 8)})
    bs = 2 if phase == "train" else 1  # BatchNorm needs > 1 sample per channel at the 16x16 bottleneck
    xs, ts = torch.randn(bs, 3, size // 2, size // 2), torch.randint(0, classes, (bs, size // 2, size // 2))
    best, best_t = cands[0], float("inf")
    for c in cands:
        torch.set_num_threads(c)
        run(xs, ts)
        t0 = time.perf_counter()
        run(xs, ts)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
    torch.set_num_threads(best)
    x, t = torch.randn(bs, 3, size, size), torch.randint(0, classes, (bs, size, size))
    run(x, t)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        run(x, t)
        n += bs
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    what = "UNetRef+softmax" if phase == "predict" else "UNetRef train step (fwd+{}+bwd+Adam)".format(loss_name)
    return {"value": round(n / el, 3), "unit": "tiles/s", "cores": best, "kind": "port",
            "sample": "{} tiles of 3x{}x{} (batch {}), fp32, oracle/robosat_ref.py {} with {} of {} host cores (best of {})".format(
                n, size, size, bs, what, best, avail, cands)}


def parity_vs_oracle(net, x, classes, got=None):
    """The in-line output check: probabilities of ONE tile of the benchmark batch from the model the timed loop just ran
    (``got``: that loop's own last output row when given) against the CPU oracle (oracle/robosat_ref.py) holding the same
    state dict.  A kernel that wrote zeros -- or anything else -- at the benchmarked shapes shows up here, in the same JSON
    line as the throughput it would have posted."""

    from oracle import robosat_ref as R

    ref = R.UNetRef(classes)
    ref.load_state_dict({k: v.detach().float().cpu() for k, v in net.state_dict().items()})
    ref.eval()
    was_training = net.training
    net.eval()
    try:
        if got is None:
            got = net.predict_probs(x[:1])[0]
    finally:
        net.train(was_training)
    want = R.predict_probs(ref, x[:1].float().cpu())[0]
    got = got.float().cpu()
    return {"max_abs_vs_oracle": float((got - want).abs().max()), "argmax_agreement": float((got.argmax(0) == want.argmax(0)).float().mean()),
            "what": "softmax probabilities of tile 0 of the benchmark batch vs oracle/robosat_ref.py (fp32, CPU) on the same weights"}


def run_phase(args, phase, dtype, batch, steps, warmup, device, dist, rank):
    """Builds the model for `phase`, runs `warmup` untimed + `steps` timed steps bracketed by barrier + synchronize, and
    returns (max-over-ranks seconds, step function, a callable producing rank 0's parity record)."""

    import torch.distributed as td

    train = phase == "train"
    net = build_model(args.classes, device, train, dtype)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(batch, 3, args.size, args.size, generator=g).to(device)  # resident in HBM

    if train:
        from robosat_amd import losses
        from robosat_amd.parallel import GradReducer

        tgt = torch.randint(0, args.classes, (batch, args.size, args.size), generator=g).to(device)
        crit = {"CrossEntropy": lambda: losses.CrossEntropyLoss2d(weight=torch.ones(args.classes)),
                "Focal": lambda: losses.FocalLoss2d(weight=torch.ones(args.classes)),
                "Lovasz": lambda: losses.LovaszLoss2d()}[args.loss]().to(device)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)  # (as `rs train` builds it on a GPU)
        if dist:
            from robosat_amd import parallel

            parallel.broadcast_module(net)  # (the replicas are seeded identically; this is what `rs train` does)
            net.grad_reducer = GradReducer()  # bucketed RCCL all-reduce overlapped with the backward kernels

        def step():
            opt.zero_grad()
            loss = crit(net(x), tgt)
            loss.backward()
            opt.step()
            return loss
    else:
        def step():
            return net.predict_probs(x)

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    barrier()
    el = time.perf_counter() - t0
    if dist:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        el = float(t.item())
    def parity():  # (outside the timed region, and after the roofline passes: the CPU oracle lets the GPU go idle)
        if rank != 0 or args.no_parity:
            return None
        rec = parity_vs_oracle(net, x, args.classes, got=None if train else last[0])
        if train:
            rec["train_loss_last_step"] = float(last)
        return rec

    return el, step, parity


def workload(args, phase, dtype, batch, world):
    train = phase == "train"
    return {"workload": ("rs predict ResNet50-UNet, bs={} 3x{}x{} {} per GPU, {} classes (BASELINE configs[1])" if not train else
                         "rs train ResNet50-UNet, bs={} 3x{}x{} {} per GPU, {} classes, " + args.loss + " loss + Adam (BASELINE configs[2])").format(
        batch, args.size, args.size, dtype, args.classes), "phase": phase, "tiles_per_gpu_per_step": batch,
        "tile": args.size, "parallelism": ("tiles sharded over {} rank(s), no collective" if not train else
                                           "dp{}: replica per GPU, flat-arena RCCL all-reduce of 37.3M gradients per step").format(world)}


def main():
    args = parse()
    from robosat_amd import launch, parallel

    if not launch.under_launcher() and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU), exactly what the driver's
        # `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` provides from outside
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    world, rank, local = launch.dist_env()
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus {} but the launcher started {} rank(s); reporting n_gpus = {}".format(args.gpus, world, world),
              file=sys.stderr)
    dist = world > 1
    if dist:
        import torch.distributed as td

        parallel.init_process_group(world, rank)
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist:
        # Create the communicator now and push whatever RCCL wrote through C stdio (it prints a version banner on first
        # use) out of the buffer, so that the JSON line below stays the LAST line of stdout.
        import ctypes

        if td.get_backend() == "nccl":
            td.barrier(device_ids=[local])
        else:
            td.barrier()
        ctypes.CDLL(None).fflush(None)

    el, step, parity = run_phase(args, args.phase, args.dtype, args.batch, args.steps, args.warmup, device, dist, rank)
    line = None
    # every rank runs the two untimed roofline passes: a train step contains the gradient all-reduce, so rank 0 alone
    # would wait for its peers forever
    roof, layers = roofline(step)
    if rank == 0:
        if args.layers_json:
            with open(args.layers_json, "w") as fp:
                json.dump(layers, fp, indent=1)
        line = {
            "metric": "512x512 tiles/sec train+predict, 1/2/4/8 MI355X; mIoU vs CPU ref",
            "value": round(world * args.batch * args.steps / el, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": workload(args, args.phase, args.dtype, args.batch, world),
            "roofline": roof, "parity": parity(),
        }
    del step
    torch.cuda.empty_cache()

    # The metric is "train+predict": the headline `value` above is the predict leg (BASELINE configs[1]); the train leg
    # (configs[2]: bf16, bs 32 per GPU, fwd + Lovasz + bwd + RCCL gradient all-reduce + Adam) rides in the same line.
    if args.phase == "predict" and not args.no_train_leg:
        tb, ts, tw = args.train_batch, max(1, min(args.steps, 10)), max(3, min(args.warmup, 5))
        tel, tstep, tparity = run_phase(args, "train", "bf16", tb, ts, tw, device, dist, rank)
        troof, _ = roofline(tstep)
        if rank == 0:
            line["train"] = {"value": round(world * tb * ts / tel, 2), "unit": "tiles/s", "steps": ts, "warmup": tw,
                             "ms_per_step": round(tel / ts * 1e3, 3), "dtype": "bf16", "scaling": "weak",
                             "config": workload(args, "train", "bf16", tb, world), "roofline": troof, "parity": tparity()}
        del tstep
        torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.classes, args.size, args.cpu_seconds, args.phase, args.loss)
            if "train" in line:  # the train leg's own CPU number (the oracle's fwd + Lovasz + bwd + Adam), shorter sample
                line["train"]["cpu_baseline"] = cpu_baseline(args.classes, args.size, args.cpu_seconds * 0.7, "train", args.loss)
    if dist:
        import ctypes

        td.barrier()
        td.destroy_process_group()
        ctypes.CDLL(None).fflush(None)  # anything RCCL still holds in C stdio goes out BEFORE the JSON line
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
