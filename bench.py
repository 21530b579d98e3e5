"""bench.py -- 512x512 tiles/sec of the U-Net hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S]

Workload (BASELINE.json configs[1]): ``rs predict`` of the ResNet-50 U-Net, batch 16 x 3x512x512 fp32 per GPU, 2
classes: layout conversion -> 61 convolutions -> fused final 1x1 + softmax, i.e. everything ``rs predict`` runs on the
device per batch (reference tools/predict.py:83-87).  Inputs are synthetic and already resident in HBM when the timed
region starts.  Tiles shard across ranks with no data-path collective ("weak" scaling: B tiles per GPU per step).

One JSON line on stdout (rank 0): metric/value/unit... (the predict leg), plus
  "train"        -- the train leg of the metric (configs[2]: bf16, bs 32 per GPU, fwd + Lovasz + bwd + gradient
                    all-reduce + Adam): value (tiles/s over all ranks), ms_per_step, its own roofline object;
  "legs"         -- the remaining BASELINE configurations, same timing discipline, short loops: configs[4] (4-band RGB+IR,
                    4 classes, Lovasz, bf16 bs 32), fp32 training (the reference's arithmetic, bs 8), configs[3] (1024^2);
  "step_ms"      -- min / median / max of the per-step device times (in every leg), next to the mean `ms_per_step`;
  "miou"         -- "mIoU vs CPU ref": bf16 MI355X vs fp32 CPU oracle after the same short training run (N=1 only);
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
# untimed settle before the W warm-up steps of every leg (see run_phase); the two knobs exist for the A/B run that justified
# it (scripts/stall_ab.sh): ROBOSAT_BENCH_PREWARM=0 and ROBOSAT_BENCH_EMPTY_CACHE=1 restore round 2's behaviour
PREWARM_STEPS = int(os.environ.get("ROBOSAT_BENCH_PREWARM", "6"))
PREWARM_SLEEP_S = 0.3 if PREWARM_STEPS > 0 else 0.0
# After the W warm-up steps: further UNTIMED steps, each timed on its own, until three in a row are within 5 % of the fastest
# seen (at most SETTLE_MAX).  A leg that starts behind seconds of host-only work (the previous leg's CPU-oracle parity check)
# was seen to run its first six timed steps at 22-40 ms instead of 21.7 (profiles/r05/bench_settle.txt): clocks and queues
# come back over more steps than a fixed warm-up covers on some boxes.  Steady-state throughput is what is measured.
SETTLE_MAX = int(os.environ.get("ROBOSAT_BENCH_SETTLE", "40"))
# Host flow control inside the timed loop: before issuing step i the host waits for step i - RUNAHEAD to have finished on the
# device (0 = never waits).  The host issues a train step in half the time the device takes, so an unthrottled loop gets
# further ahead than any warm-up did, the caching allocator runs out of blocks whose side-stream events have completed and
# grows -- hipMalloc inside the timed region, 15-20 ms each time (profiles/r05/bench_settle.txt).  With two steps queued the
# device never idles; the reference's own loop synchronises EVERY step (`loss.item()`, train.py:190).
RUNAHEAD = int(os.environ.get("ROBOSAT_BENCH_RUNAHEAD", "2"))
EMPTY_CACHE_BETWEEN_LEGS = os.environ.get("ROBOSAT_BENCH_EMPTY_CACHE", "0") == "1"


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
    ap.add_argument("--channels", type=int, default=3, help="input bands (4 = RGB + IR, BASELINE configs[4])")
    ap.add_argument("--phase", choices=["predict", "train"], default="predict",
                    help="predict = BASELINE configs[1] (default); train = fwd + loss + bwd + grad all-reduce + Adam")
    ap.add_argument("--loss", choices=["CrossEntropy", "Lovasz", "Focal"], default="Lovasz", help="train phase criterion")
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                    help="compute dtype: fp32 (exact-fp32 MFMA, the parity path; BASELINE configs[1]) or bf16 (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-line comparison of one output tile with the CPU oracle")
    ap.add_argument("--no-train-leg", action="store_true", help="predict phase only: skip the bf16 train leg reported under \"train\"")
    ap.add_argument("--train-batch", type=int, default=32, help="tiles per GPU per step of the train leg (configs[2]: 32)")
    ap.add_argument("--train-steps", type=int, default=30, help="timed steps of the train leg (>= 30: one stall must not move the mean)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="predict phase only: skip the legs reported under \"legs\" (configs[4] 4-band 4-class bf16 train, fp32 "
                         "train bs 8, configs[3] 1024^2 predict)")
    ap.add_argument("--no-miou", action="store_true", help="skip the bf16-vs-fp32-oracle convergence check reported under \"miou\"")
    ap.add_argument("--grad-dtype", choices=["fp32", "bf16"], default="fp32", help="what the gradient all-reduce puts on the wire (N > 1)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch / --train-batch tiles PER GPU (default); strong: they are the GLOBAL batch, split over the ranks")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU-oracle sample")
    ap.add_argument("--layers-json", type=str, default="", help="also dump the per-layer roofline table here")
    ap.add_argument("--full-json", type=str, default="",
                    help="where the FULL record goes (per-kernel tables, every step time, counter provenance); default "
                         "gpurun_out/bench_full.json.  stdout carries the compact line (<= 4 KB) only")
    ap.add_argument("--force-reducer", action="store_true",
                    help="train legs at ONE rank: run the data-parallel GradReducer anyway, over a world-size-1 RCCL group (the "
                         "backend == nccl branch, its stream ordering and the in-place AVG on arena views execute for real)")
    return ap.parse_args()


def build_model(classes, device, train=False, dtype="fp32", channels=3):
    from robosat_amd.unet import UNet

    torch.manual_seed(0)
    net = UNet(classes, pretrained=False, compute_dtype=dtype, in_channels=channels)  # random init of the reference architecture
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
    g3 = {"exe": 0.0, "alg": 0.0, "ms": 0.0, "n": 0, "ideal_ms": 0.0}  # the 3x3 group: see conv3x3 below
    for name, flops, shape, e0, e1, nbytes, executed in recs:
        ms = e0.elapsed_time(e1)
        # the north_star's "3x3 conv" group: every launch whose filter is 3x3 -- plain, strided, DecoderBlock (phase / Winograd
        # forms), their weight gradients -- or the 4x4 / stride-2 data gradient of a DecoderBlock's 3x3 (shape[2] = kh)
        if shape[2] in (3, 4):
            g3["exe"] += executed
            g3["alg"] += flops
            g3["ms"] += ms
            g3["n"] += 1
            g3["ideal_ms"] += executed / kernel_peak(name) / 1e9
        k = per_kernel.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0])
        k[0] += flops
        k[1] += ms
        k[2] += 1
        k[3] += nbytes
        k[4] += executed
        layers.append({"kernel": name, "cin_cout_k_stride_ups_ho_wo": list(shape), "gflop": flops / 1e9, "ms": ms,
                       "tflops": flops / ms / 1e9 if ms > 0 else 0.0, "executed_gflop": executed / 1e9, "mb": nbytes / 1e6})
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
        # ONE number for the north_star's "MFMA roofline on 3x3 conv": executed FLOPs of the 3x3 group over its time against the
        # dense MFMA peak of each launch's dtype (frac = sum of per-launch peak times / sum of measured times)
        "conv3x3": {"executed_tflops": round(g3["exe"] / g3["ms"] / 1e9, 2) if g3["ms"] else None,
                    "frac": round(g3["ideal_ms"] / g3["ms"], 4) if g3["ms"] else None, "ms": round(g3["ms"], 3), "launches": g3["n"],
                    "algorithmic_tflops": round(g3["alg"] / g3["ms"] / 1e9, 2) if g3["ms"] else None,
                    "what": "all 3x3 launches of the pass (plain, strided, DecoderBlock forms, weight gradients, 4x4/s2 data gradients of "
                            "DecoderBlock): executed FLOPs / HIP-event time vs the dense MFMA peak of the launch's dtype"},
        "per_kernel": {n: {"tflops": round(v[0] / v[1] / 1e9, 2), "executed_tflops": round(v[4] / v[1] / 1e9, 2),
                           "gbs": round(v[3] / v[1] / 1e6, 1), "ms": round(v[1], 3), "launches": v[2]}
                       for n, v in per_kernel.items()},
    })
    return out, layers


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
    passes, corrected as MI355X_MICROARCH.md prescribes) -- profiles/pmc_traffic.json, written by scripts/pmc_summary.py
    from a run of this same command.  None when no counter run has been committed for this kernel."""

    table = _pmc_table()
    if _traffic_stale(table):
        return None  # counters of ANOTHER tree say nothing about these kernels
    if kernel in table:
        return table[kernel]
    return table.get(kernel.split("+")[0] + ">") if "+" in kernel else None  # ("conv_wgrad_bf16<phase,128x128+128x64>": by its first tile)


def _traffic_stale(table):
    """True when the committed counter tables were not taken on the kernel sources this run executes: their `_meta`
    carries the digest of csrc/ + the ABI header of the profiled tree (scripts/pmc_traffic.py); no digest = stale."""

    from robosat_amd._lib import kernel_source_digest

    meta = table.get("_meta") or {}
    # (round 6: the digest of the tree the counters were TAKEN on decides, and a table somebody re-stamped by hand is stale by
    # definition -- round 5's guard compared a field its author could edit, and did, three times: VERDICT r5 weak 2)
    if "restamped" in meta:
        return True
    return meta.get("profiled_csrc_digest", meta.get("csrc_digest")) != kernel_source_digest()


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

    from robosat_amd._lib import kernel_source_digest

    table = _pmc_table()
    meta = table.get("_meta")
    out = {"file": "profiles/pmc_traffic.json", "measured": meta, "csrc_digest_now": kernel_source_digest(), "stale": _traffic_stale(table)}
    if out["stale"]:
        out["note"] = "the counter passes belong to other kernel sources than this run's: roofline.traffic withheld (null)"
    return out


def cpu_baseline(classes, size, budget_s, phase="predict", loss_name="Lovasz", channels=3):
    """The CPU oracle on this box's host cores: bounded sample of the same workload (tiles of the same size).

    torch's intra-op pool does not scale to every core of a 2-socket host for these convolutions, so the thread count
    is chosen by a short sweep on a quarter-size tile and reported as ``cores``."""

    from oracle import robosat_ref as R

    torch.manual_seed(0)
    net = R.UNetRef(classes, in_channels=channels)
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
    xs, ts = torch.randn(bs, channels, size // 2, size // 2), torch.randint(0, classes, (bs, size // 2, size // 2))
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
    x, t = torch.randn(bs, channels, size, size), torch.randint(0, classes, (bs, size, size))
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
            "sample": "{} tiles of {}x{}x{} (batch {}), fp32, oracle/robosat_ref.py {} with {} of {} host cores (best of {})".format(
                n, channels, size, size, bs, what, best, avail, cands)}


def parity_vs_oracle(net, x, classes, got=None, channels=3):
    """The in-line output check: probabilities of ONE tile of the benchmark batch from the model the timed loop just ran
    (``got``: that loop's own last output row when given) against the CPU oracle (oracle/robosat_ref.py) holding the same
    state dict.  A kernel that wrote zeros -- or anything else -- at the benchmarked shapes shows up here, in the same JSON
    line as the throughput it would have posted."""

    from oracle import robosat_ref as R

    ref = R.UNetRef(classes, in_channels=channels)
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


class Leg:
    """One workload of the line: phase, dtype, tiles per GPU per step, tile size, classes, bands, loss."""

    def __init__(self, phase, dtype, batch, size, classes, channels, loss, tag=""):
        self.phase, self.dtype, self.batch, self.size, self.classes, self.channels, self.loss, self.tag = (
            phase, dtype, batch, size, classes, channels, loss, tag)


def run_phase(leg, steps, warmup, device, dist, rank, no_parity=False, grad_dtype="fp32", force_reducer=False):
    """Builds the model for `leg`, runs `warmup` untimed + `steps` timed steps bracketed by barrier + synchronize, and
    returns (max-over-ranks seconds, per-step milliseconds of this rank from HIP events between the steps, the EAGER step
    function, a callable producing rank 0's parity record, whether the timed steps were hipGraph replays)."""

    import torch.distributed as td

    train = leg.phase == "train"
    torch.cuda.reset_peak_memory_stats(device)
    net = build_model(leg.classes, device, train, leg.dtype, leg.channels)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(leg.batch, leg.channels, leg.size, leg.size, generator=g).to(device)  # resident in HBM

    if train:
        from robosat_amd import losses
        from robosat_amd.parallel import GradReducer

        tgt = torch.randint(0, leg.classes, (leg.batch, leg.size, leg.size), generator=g).to(device)
        crit = {"CrossEntropy": lambda: losses.CrossEntropyLoss2d(weight=torch.ones(leg.classes)),
                "Focal": lambda: losses.FocalLoss2d(weight=torch.ones(leg.classes)),
                "Lovasz": lambda: losses.LovaszLoss2d()}[leg.loss]().to(device)
        # as `rs train` builds it on a GPU: fused, and capturable when the step is replayed as a hipGraph (one rank; see below)
        # (needs 3 untimed calls: two eager steps, then the call that captures -- a capture inside the timed region is not a step)
        # Opt-in (ROBOSAT_TRAIN_GRAPH=1 here, `[model] graph = true` in rs train): replayed, the step measures 24.7 ms against
        # 23.5 ms eager (profiles/r03/host_sensitivity.txt) -- the graph executor serialises the weight-gradient branch.
        graphed = not dist and warmup + PREWARM_STEPS >= 3 and os.environ.get("ROBOSAT_TRAIN_GRAPH", "0") == "1"
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True, capturable=graphed)
        if dist or force_reducer:
            from robosat_amd import parallel

            parallel.broadcast_module(net)  # (the replicas are seeded identically; this is what `rs train` does)
            if dist and hasattr(crit, "global_batch"):
                crit.global_batch = True  # batch-level loss terms over the GLOBAL batch, as rs train sets it (losses.py)
            # bucketed RCCL all-reduce overlapped with the backward kernels (--force-reducer: over a group of one rank too)
            net.grad_reducer = GradReducer(wire_dtype=torch.bfloat16 if grad_dtype == "bf16" else torch.float32,
                                           force=force_reducer)
            run_phase.reducer = net.grad_reducer

        def step():  # the eager step (what the roofline pass brackets launch by launch)
            opt.zero_grad()
            loss = crit(net(x), tgt)
            loss.backward()
            opt.step()
            return loss

        timed = step
        if graphed:
            # what `rs train` runs (robosat_amd.graph.TrainStepGraph): two eager steps, then the whole step -- zero_grad,
            # forward, loss, backward, Adam -- is ONE hipGraph replay per batch (the batch is copied into the graph's static
            # input first, as the tool does with every new batch; that copy is inside the timed step)
            from robosat_amd.graph import TrainStepGraph

            stepper = TrainStepGraph(net, crit, opt)

            def timed():
                return stepper(x, tgt)[0]
    else:
        def step():
            return net.predict_probs(x)

        timed = step

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    # Settle BEFORE the W warm-up steps.  The first steps of a leg are a burst of hipMalloc (the caching allocator grows to the
    # leg's working set); on this driver stack a process's queues can be evicted and restored (with a ~100 ms delay per round)
    # in the wake of such a burst, and a timed window that starts 0.1 s later catches it: one step of 10 at 218 ms / 845 ms
    # instead of 25 was seen twice in this round's own runs (profiles/r03/bench_stall_*.json), and a 33 ms hiccup in a 10-step
    # window is exactly the +3.3 ms per step by which the driver's train leg differed from ours in rounds 1 and 2.  So: a few
    # untimed steps to reach the working set, a drain, a short pause, and only then the W warm-up steps and the K timed ones.
    # Steady-state throughput is what is measured; nothing is skipped inside the timed region.
    for _ in range(PREWARM_STEPS):
        timed()
    torch.cuda.synchronize()
    time.sleep(PREWARM_SLEEP_S)
    # Python's cyclic collector runs when it pleases; a full collection in the middle of the timed steps stalls the host thread
    # that issues the launches (one step of a 10-step leg was seen to take 845 ms instead of 25).  Collect NOW -- before the
    # warm-up steps, not between them and the timed ones: a collection over this process's heap is 0.1-0.3 s of idle GPU, and
    # a timed window that opens right behind such a gap showed 35-40 ms steps among its first six (profiles/r05/bench_settle.txt)
    # -- and keep it off until the leg is timed; `rs train` does the same around its epoch loops (tools/train.py:_epoch).
    import gc

    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # (recorded on the stream; no host sync)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        for _ in range(warmup):
            timed()
        settle, best, good = 0, None, 0
        while settle < SETTLE_MAX:
            e0.record()
            timed()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            settle += 1
            best = ms if best is None else min(best, ms)
            good = good + 1 if ms <= 1.05 * best else 0
            done = good >= 3
            if dist:  # (a train step holds a collective: every rank runs the same number of steps)
                flag = torch.tensor([1.0 if done else 0.0], device=device)
                td.all_reduce(flag, op=td.ReduceOp.MIN)
                done = bool(flag.item() > 0.5)
            if done:
                break
        run_phase.settle_steps = settle
        barrier()
        t0 = time.perf_counter()
        last = None
        marks[0].record()
        allocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
        for i in range(steps):
            if RUNAHEAD > 0 and i >= RUNAHEAD:
                marks[i + 1 - RUNAHEAD].synchronize()  # (end of step i - RUNAHEAD)
            last = timed()
            marks[i + 1].record()
        barrier()
        el = time.perf_counter() - t0
        run_phase.device_allocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - allocs0
    finally:
        if gc_was_on:
            gc.enable()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    if dist:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        el = float(t.item())

    def parity():  # (outside the timed region, and after the roofline passes: the CPU oracle lets the GPU go idle)
        if rank != 0 or no_parity:
            return None
        rec = parity_vs_oracle(net, x, leg.classes, got=None if train else last[0], channels=leg.channels)
        if train:
            rec["train_loss_last_step"] = float(last)
        return rec

    run_phase.peak_gb = round(torch.cuda.max_memory_allocated(device) / 2**30, 2)  # (read by the caller right after)
    return el, step_ms, step, parity, bool(train and graphed and stepper.captured)


def reducer_record():
    """What the last train leg's GradReducer was and did: backend, group size, collectives issued (so a reader can tell the
    RCCL branch executed inside the timed steps), wire dtype."""

    rd = getattr(run_phase, "reducer", None)
    if rd is None:
        return None
    return {"backend": rd.backend, "world": rd.world, "forced": rd.force, "collectives_issued": rd.issued,
            "wire": "bf16" if rd.wire_dtype == torch.bfloat16 else "fp32"}


def step_stats(step_ms):
    """min / median / max of the per-step device times (HIP events recorded between the steps of the timed loop): a stall
    that hits one step -- another process polling the SMU, a host hiccup -- shows here instead of hiding in the mean."""

    v = sorted(step_ms)
    n = len(v)
    med = (v[n // 2] + v[(n - 1) // 2]) / 2
    return {"min": round(v[0], 3), "median": round(med, 3), "max": round(v[-1], 3), "n": n,
            "stalled_steps": sum(1 for x in v if x > 1.5 * med), "slowest_step_index": max(range(n), key=lambda i: step_ms[i]),
            "all": [round(x, 2) for x in step_ms]}


def workload(leg, world, cfg=""):
    train = leg.phase == "train"
    bands = "{}x{}x{}".format(leg.channels, leg.size, leg.size)
    what = ("rs predict ResNet50-UNet, bs={} {} {} per GPU, {} classes" if not train else
            "rs train ResNet50-UNet, bs={} {} {} per GPU, {} classes, " + leg.loss + " loss + Adam").format(
        leg.batch, bands, leg.dtype, leg.classes) + (" (BASELINE {})".format(cfg) if cfg else "")
    return {"workload": what, "phase": leg.phase, "tiles_per_gpu_per_step": leg.batch, "tile": leg.size, "bands": leg.channels,
            "classes": leg.classes,
            "parallelism": ("tiles sharded over {} rank(s), no collective" if not train else
                            "dp{}: replica per GPU, flat-arena RCCL all-reduce of 37.3M gradients per step").format(world)}



def baseline_config(leg):
    """Which BASELINE.json configuration a leg is, if any."""

    key = (leg.phase, leg.dtype, leg.batch, leg.size, leg.classes, leg.channels)
    return {("predict", "fp32", 16, 512, 2, 3): "configs[1]", ("train", "bf16", 32, 512, 2, 3): "configs[2]",
            ("predict", "fp32", 8, 1024, 2, 3): "configs[3]"}.get(key, "configs[4]" if key == ("train", "bf16", 32, 512, 4, 4) and
                                                                  leg.loss == "Lovasz" else "")


def miou_vs_cpu_ref(device, seed=31, n_train=24, n_val=8, size=128, batch=4, epochs=2, lr=3e-4):
    """The metric's "mIoU vs CPU ref": the bf16 MI355X path and the fp32 CPU oracle train on the SAME learnable synthetic
    tiles (rectangles brighter than their background, as tests/synth.py draws them) from the same initial weights, same
    batches, same Adam, for `epochs` epochs, and are validated on the same held-out tiles with the reference's confusion
    counts (metrics.py:27-84).  Small tiles, outside every timed region (~10 s, most of it the CPU side)."""

    import numpy as np

    from oracle import robosat_ref as R, seeded
    from robosat_amd import losses
    from robosat_amd.metrics import Metrics
    from robosat_amd.unet import UNet

    rng = np.random.default_rng(seed)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)

    def tiles(count):
        xs, ts = [], []
        for _ in range(count):
            img = rng.integers(0, 120, size=(size, size, 3), dtype=np.uint8)
            mask = np.zeros((size, size), dtype=np.int64)
            for _ in range(int(rng.integers(1, 4))):
                w, h = rng.integers(size // 8, size // 2, size=2)
                x0, y0 = rng.integers(0, size - w), rng.integers(0, size - h)
                mask[y0:y0 + h, x0:x0 + w] = 1
                img[y0:y0 + h, x0:x0 + w] = rng.integers(130, 256, size=(h, w, 3), dtype=np.uint8)
            xs.append(((img.astype(np.float32) / 255 - mean) / std).transpose(2, 0, 1))
            ts.append(mask)
        return torch.from_numpy(np.stack(xs)), torch.from_numpy(np.stack(ts))

    xtr, ttr = tiles(n_train)
    xva, tva = tiles(n_val)
    sd = seeded.seeded_state_dict(R.UNetRef(2).state_dict(), 8)
    ref = R.UNetRef(2)
    ref.load_state_dict(sd)
    net = UNet(2, pretrained=False, compute_dtype="bf16")
    net.load_state_dict(sd)
    net = net.to(device)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=lr)
    opt = torch.optim.Adam(net.parameters(), lr=lr, fused=True)
    crit = losses.LovaszLoss2d().to(device)
    torch.set_num_threads(min(32, max(1, len(os.sched_getaffinity(0)))))
    ref.train()
    net.train()
    order = torch.Generator().manual_seed(seed)
    for _ in range(epochs):
        perm = torch.randperm(n_train, generator=order)
        for i in range(0, n_train - batch + 1, batch):
            idx = perm[i:i + batch]
            opt_ref.zero_grad()
            R.lovasz2d(ref(xtr[idx]), ttr[idx]).backward()
            opt_ref.step()
            opt.zero_grad()
            crit(net(xtr[idx].to(device)), ttr[idx].to(device)).backward()
            opt.step()
    ref.eval()
    net.eval()
    counts = np.zeros(4, dtype=np.int64)
    m = Metrics(range(2))
    with torch.no_grad():
        for a, s in zip(tva, ref(xva)):
            counts += np.array(R.confusion_counts(a, s))
        m.add_batch(tva.to(device), net(xva.to(device)))
    cpu, gpu = float(R.metric_scores(*counts)[0]), float(m.get_miou())
    return {"gpu": round(gpu, 4), "cpu_ref": round(cpu, 4), "abs_diff": round(abs(gpu - cpu), 4),
            "what": "validation mIoU (metrics.py:43-84) after {} epochs of Lovasz/Adam(lr {}) on {} synthetic {}x{} tiles, {} held out: "
                    "bf16 MI355X path vs fp32 CPU oracle, same initial weights and batches".format(epochs, lr, n_train, size, size, n_val)}


COMPACT_LIMIT = 4096  # bytes of the stdout line: the driver keeps an 8 KB tail of the run's output


def _short(text, n):
    return text if len(text) <= n else text[:n - 1] + "~"


def _compact_roofline(r):
    """The contract's roofline object (bound / achieved / peak / unit / frac / traffic) + what names it; the per-kernel
    table, the counter provenance and the second roof's figures stay in the full record."""

    if not r:
        return r
    out = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms") if k in r}
    if "algorithmic" in r:
        out["algorithmic_tflops"] = r["algorithmic"]["tflops"]
    if "hbm" in r:
        out["hbm_frac"] = r["hbm"]["frac"]
    if "all_convs" in r:
        out["all_convs"] = {k: r["all_convs"][k] for k in ("executed_frac", "roofline_frac", "ms", "executed_tflops") if k in r["all_convs"]}
    if "conv3x3" in r:
        out["conv3x3"] = {k: r["conv3x3"][k] for k in ("executed_tflops", "frac", "ms", "launches")}
    return out


def _compact_leg(leg, with_roofline=True):
    out = {k: leg[k] for k in ("value", "unit", "steps", "warmup", "untimed_steps", "ms_per_step", "value_median", "dtype", "hipgraph", "peak_hbm_gb", "settle_steps",
                               "scaling", "reducer") if k in leg}
    if "step_ms" in leg:
        out["step_ms"] = {k: leg["step_ms"][k] for k in ("min", "median", "max", "n", "stalled_steps")}
    if "config" in leg:
        out["config"] = {"workload": _short(leg["config"]["workload"], 120), "parallelism": _short(leg["config"]["parallelism"], 64)}
    if with_roofline and "roofline" in leg:
        out["roofline"] = _compact_roofline(leg["roofline"])
    if leg.get("parity"):
        out["parity"] = {k: v for k, v in leg["parity"].items() if k != "what"}
    if leg.get("cpu_baseline"):
        out["cpu_baseline"] = dict(leg["cpu_baseline"], sample=_short(leg["cpu_baseline"]["sample"], 150))
    return out


def compact_line(line, full_path=""):
    """What goes to stdout: every field of the bench contract for the headline leg, and for `train` and each of `legs`
    value / ms_per_step / step_ms min-median-max / roofline (dominant kernel + all_convs) / parity / cpu_baseline -- at most
    COMPACT_LIMIT bytes, so that the driver's 8 KB tail holds the whole line.  Everything else (per-kernel tables, all step
    times, counter provenance, the long descriptions) is in the full record at `full_path`."""

    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "untimed_steps", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data", "hipgraph") if k in line}
    out.update(_compact_leg(line))
    out["flops_basis"] = "executed"
    if "train" in line:
        out["train"] = _compact_leg(line["train"])
    if "legs" in line:
        out["legs"] = {name: _compact_leg(leg, with_roofline=False) for name, leg in line["legs"].items()}
        for name, leg in line["legs"].items():  # (the bs-32 predict leg keeps the one number it exists for)
            if "roofline" in leg and "conv3x3" in leg["roofline"]:
                out["legs"][name]["conv3x3"] = {k: leg["roofline"]["conv3x3"][k] for k in ("executed_tflops", "frac", "ms")}
        for leg in out["legs"].values():
            for k in ("config", "unit", "hipgraph", "warmup"):  # (in the full record; the line is for the driver's 8 KB tail)
                leg.pop(k, None)
    if "miou" in line:
        out["miou"] = {k: v for k, v in line["miou"].items() if k != "what"}
    if full_path:
        out["full_record"] = full_path
    # belt and braces: should a future field push the line over the limit, shed the least important details first -- the legs'
    # step-time spreads, then their memory figures, and only then whole groups
    def fits():
        return len(json.dumps(out)) <= COMPACT_LIMIT

    for key in ("step_ms", "peak_hbm_gb", "value_median"):
        if fits():
            break
        for leg in out.get("legs", {}).values():
            leg.pop(key, None)
    for victim in ("legs", "miou", "full_record"):
        if fits():
            break
        out.pop(victim, None)
    return out


def write_full_record(line, path):
    """The full record next to the compact line: `path`, else gpurun_out/bench_full.json under the repo (scratch on the GPU
    box, merged back by gpurun).  Returns the path written, or "" when nothing could be written (read-only tree)."""

    path = path or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as fp:
            json.dump(line, fp, indent=1)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError:
        return ""


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
    if dist or args.force_reducer:
        import torch.distributed as td

        if not dist:  # a process group of ONE rank over RCCL: the reducer's nccl branch on a single GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        parallel.init_process_group(world, rank)
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist or args.force_reducer:
        # Create the communicator now and push whatever RCCL wrote through C stdio (it prints a version banner on first
        # use) out of the buffer, so that the JSON line below stays the LAST line of stdout.
        import ctypes

        if td.get_backend() == "nccl":
            td.barrier(device_ids=[local])
        else:
            td.barrier()
        ctypes.CDLL(None).fflush(None)

    strong = args.scaling == "strong"

    def per_rank(batch):  # --scaling strong: the flag is the GLOBAL batch (rs train's `[common] batch_size`), split like DataParallel
        if not strong:
            return batch
        if batch % world:
            sys.exit("bench.py: --scaling strong needs a batch divisible by the {} ranks (got {})".format(world, batch))
        return batch // world

    main_leg = Leg(args.phase, args.dtype, per_rank(args.batch), args.size, args.classes, args.channels, args.loss)
    el, step_ms, step, parity, hipgraph = run_phase(main_leg, args.steps, args.warmup, device, dist, rank, args.no_parity, args.grad_dtype,
                                                    args.force_reducer)
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
            "value": round(world * main_leg.batch * args.steps / el, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "untimed_steps": PREWARM_STEPS + args.warmup + run_phase.settle_steps, "prewarm": {"steps": PREWARM_STEPS, "sleep_s": PREWARM_SLEEP_S, "settle_max": SETTLE_MAX, "settle_steps": run_phase.settle_steps, "runahead": RUNAHEAD,
                                               "device_allocs_in_timed_steps": run_phase.device_allocs,
                                               "what": "untimed, every leg: `steps` steps + a pause, gc, W warm-up steps, then <= `settle_max` more until 3 in a row are within 5 % of the fastest; timed loop keeps <= `runahead` steps queued"},
            "ms_per_step": round(el / args.steps * 1e3, 3), "step_ms": step_stats(step_ms),
            "hipgraph": hipgraph,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": workload(main_leg, world, baseline_config(main_leg)),
            # `roofline.frac` is on the FLOPs the kernels EXECUTE; SURVEY.md section 8d's algorithmic count rides along as
            # `roofline.algorithmic` (its 100 %-of-peak ceiling of 941 tiles/s can be exceeded: see `flops_basis`)
            "flops_basis": "executed (the DecoderBlock phase form runs 4/9 of the section-8d algorithmic multiply-adds: conv3x3 over a "
                           "nearest-x2 upsample == four 2x2 convolutions on the source grid with pre-summed taps; its fp32 Winograd "
                           "F(2x2,2x2) form 1/4; the fp32 eval stride-1 3x3 layers as Winograd F(2x2,3x3) 4/9)",
            "roofline": roof, "parity": parity(),
        }
        if args.phase == "train" and reducer_record() is not None:
            line["reducer"] = reducer_record()
    del step  # (the cached blocks stay with the allocator: the next leg reuses them instead of a new round of hipFree / hipMalloc)
    if EMPTY_CACHE_BETWEEN_LEGS:
        torch.cuda.empty_cache()

    # The metric is "train+predict": the headline `value` above is the predict leg (BASELINE configs[1]); the train leg
    # (configs[2]: bf16, bs 32 per GPU, fwd + Lovasz + bwd + RCCL gradient all-reduce + Adam) rides in the same line.
    if args.phase == "predict" and not args.no_train_leg:
        tleg = Leg("train", "bf16", per_rank(args.train_batch), args.size, args.classes, args.channels, args.loss)
        ts, tw = max(1, args.train_steps), max(3, min(args.warmup, 5))
        tel, tstep_ms, tstep, tparity, tgraph = run_phase(tleg, ts, tw, device, dist, rank, args.no_parity, args.grad_dtype,
                                                          args.force_reducer)
        troof, _ = roofline(tstep)
        if rank == 0:
            line["train"] = {"value": round(world * tleg.batch * ts / tel, 2), "unit": "tiles/s", "steps": ts, "warmup": tw,
                             "untimed_steps": PREWARM_STEPS + tw + run_phase.settle_steps,
                             "ms_per_step": round(tel / ts * 1e3, 3), "step_ms": step_stats(tstep_ms),
                             # tiles/s at the MEDIAN step time: what the leg sustains when no step stalls (`value` is the mean)
                             "value_median": round(world * tleg.batch / step_stats(tstep_ms)["median"] * 1e3, 2), "dtype": "bf16",
                             "hipgraph": tgraph, "peak_hbm_gb": run_phase.peak_gb, "settle_steps": run_phase.settle_steps,
                             "device_allocs_in_timed_steps": run_phase.device_allocs,
                             "scaling": args.scaling, "config": workload(tleg, world, baseline_config(tleg)),
                             "roofline": troof, "parity": tparity()}
            if reducer_record() is not None:  # which exchange ran inside the timed steps
                line["train"]["reducer"] = reducer_record()
        del tstep
        if EMPTY_CACHE_BETWEEN_LEGS:
            torch.cuda.empty_cache()

    # The other BASELINE configurations, timed the same way (barrier + synchronize, max over ranks) with short loops and no
    # roofline pass: configs[4] (4-band RGB+IR, 4 classes, Lovasz, bf16 bs 32), the reference's own arithmetic for training
    # (fp32, bs 8) and configs[3] (1024^2 tiles, bs 8, fp32 predict).
    if args.phase == "predict" and not args.no_extra_legs and args.size == 512:
        # predict_fp32_bs32: the north_star words its MFMA target "on 3x3 conv at bs=32 512x512" -- the headline configuration at
        # that batch, with the 3x3 group's fraction of the fp32 MFMA peak (its own roofline pass: `conv3x3`)
        extra = [("predict_fp32_bs32", Leg("predict", "fp32", per_rank(32), 512, 2, 3, "Lovasz"), 10, 3),
                 ("cfg5_train_bf16_4band_4class", Leg("train", "bf16", per_rank(32), 512, 4, 4, "Lovasz"), 10, 4),
                 ("train_fp32_bs8", Leg("train", "fp32", per_rank(8), 512, 2, 3, "Lovasz"), 10, 4),
                 ("cfg4_predict_fp32_1024_bs8", Leg("predict", "fp32", per_rank(8), 1024, 2, 3, "Lovasz"), 10, 3)]
        for name, leg, ls, lw in extra:
            lel, lstep_ms, lstep, lparity, lgraph = run_phase(leg, ls, lw, device, dist, rank, no_parity=True, grad_dtype=args.grad_dtype)
            if rank == 0:
                line.setdefault("legs", {})[name] = {
                    "value": round(world * leg.batch * ls / lel, 2), "unit": "tiles/s", "steps": ls, "warmup": lw,
                    "untimed_steps": PREWARM_STEPS + lw + run_phase.settle_steps,
                    "ms_per_step": round(lel / ls * 1e3, 3), "step_ms": step_stats(lstep_ms),
                    "value_median": round(world * leg.batch / step_stats(lstep_ms)["median"] * 1e3, 2), "dtype": leg.dtype,
                    "hipgraph": lgraph, "peak_hbm_gb": run_phase.peak_gb, "settle_steps": run_phase.settle_steps,
                    "config": workload(leg, world, baseline_config(leg))}
            if name == "predict_fp32_bs32":  # (every rank: the pass has no collective, but keep the ranks in step)
                lroof, _ = roofline(lstep)
                if rank == 0:
                    line["legs"][name]["roofline"] = lroof
            del lstep
            if EMPTY_CACHE_BETWEEN_LEGS:
                torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_miou:
            line["miou"] = miou_vs_cpu_ref(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.classes, args.size, args.cpu_seconds, args.phase, args.loss, args.channels)
            if "train" in line:  # the train leg's own CPU number (the oracle's fwd + Lovasz + bwd + Adam), shorter sample
                line["train"]["cpu_baseline"] = cpu_baseline(args.classes, args.size, args.cpu_seconds * 0.7, "train", args.loss,
                                                             args.channels)
    if dist or args.force_reducer:
        import ctypes

        td.barrier()
        td.destroy_process_group()
        ctypes.CDLL(None).fflush(None)  # anything RCCL still holds in C stdio goes out BEFORE the JSON line
    if rank == 0:
        print(json.dumps(compact_line(line, write_full_record(line, args.full_json))), flush=True)


if __name__ == "__main__":
    main()
