"""Rank body of the world-size-2 data-parallel tests (started by robosat_amd.launch.spawn_ranks; see
tests/test_parallel_gloo.py and tests/test_gpu_parallel.py).  ``python dp_worker.py <mode> <outdir>``: every rank writes
``<outdir>/rank<r>.json``."""

import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cpu_logic(rank, world):
    """gloo on CPU tensors: flat-arena reducer, replica broadcast, tile sharding."""
    from robosat_amd import parallel
    from robosat_amd.tools.predict import RankBatchSampler

    torch.manual_seed(100 + rank)  # every rank draws different weights, as independent processes do
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
    net[1].running_mean.add_(rank + 1.0)
    before = float(sum(p.abs().sum() for p in net.parameters()))
    parallel.broadcast_module(net)
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([float(sum(t.double().abs().sum() for t in net.state_dict().values()))], dtype=torch.float64))
    net[1].running_var.mul_(rank + 2.0)
    parallel.broadcast_bn_buffers(net)
    red = parallel.GradReducer()
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red.reduce_async(flat[0:400])
    red.reduce_async(flat[400:1000])
    red.wait()
    # the global-batch branch of mIoULoss2d (robosat_amd.losses._global_batch_miou) on hand-made per-rank statistics: rank 0 has
    # the larger NLL, rank 1 the larger soft-IoU term; the global batch takes the NLL branch with the global denominator
    from robosat_amd import losses

    nc = 2 * 3  # (N * C gradient coefficients per half: unused here)
    miou_r, num_r, sw_r = (0.8, 14.0, 10.0) if rank == 0 else (0.07, 0.4, 11.0)
    stats = torch.zeros(5 + 2 * nc)
    stats[0], stats[1], stats[2] = max(miou_r, num_r / sw_r), sw_r, float(num_r / sw_r > miou_r)
    stats[-2], stats[-1] = miou_r, num_r
    local_branch = float(stats[2])
    got = losses._global_batch_miou(stats[0].clone(), stats, True)
    off = losses._global_batch_miou(stats[0].clone(), stats.clone(), False)
    return {"before": before, "sums": [float(s) for s in sums], "running_var": net[1].running_var.tolist(),
            "miou_dp": {"loss": float(got), "branch": float(stats[2]), "den": float(stats[1]), "local_branch": local_branch,
                        "untouched_without_opt_in": float(off)},
            "flat_ok": bool(torch.allclose(flat, torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1)) / world)),
            "loss": parallel.average_scalars([float(rank)], torch.device("cpu"))[0],
            "counts": parallel.sum_counts(torch.tensor([1, 2, 3, 4 + rank])).tolist(),
            "shards": parallel.shard_indices(10, 2, rank, world),
            "predict_batches": list(RankBatchSampler(10, 3, rank, world))}


def backward_order(net):
    """The order in which robosat_amd.autograd._backward carves the gradient arena, as (parameter, KRSC view?) items with a
    None where it flushes a bucket to the reducer: head + decoder | layer4 | layer3 | layer2 | layer1 | (finish:) stem."""

    out = [(net.final.weight, False), (net.final.bias, False), (net.dec5.block.weight, True)]
    for blk in (net.dec4, net.dec3, net.dec2, net.dec1, net.dec0, net.center):
        out.append((blk.block.block.weight, True))
    out.append(None)
    r = net.resnet
    for layer in (r.layer4, r.layer3, r.layer2, r.layer1):
        for blk in reversed(list(layer)):
            for bn, conv in ((blk.bn3, blk.conv3), (blk.bn2, blk.conv2), (blk.bn1, blk.conv1)):
                out += [(bn.weight, False), (bn.bias, False), (conv.weight, True)]
            if blk.downsample is not None:
                out += [(blk.downsample[1].weight, False), (blk.downsample[1].bias, False), (blk.downsample[0].weight, True)]
        out.append(None)
    out += [(r.bn1.weight, False), (r.bn1.bias, False), (r.conv1.weight, True)]  # (the stem: flushed by arena.finish)
    return out


def arena8(rank, world, wire):
    """The data-parallel bookkeeping of a training step at the job's real size, on host tensors over gloo: the REAL U-Net's
    parameter list carved into the flat arena in the backward's order (robosat_amd.autograd.GradArena), the five bucket
    flushes of their real sizes (53 / 57 / 27 / 5 / 1 MiB + the stem's 38 KB) through GradReducer (fp32 or bf16 wire), the replica broadcast of
    the whole state dict -- with as many ranks as a node has GPUs.  What RCCL replaces in production is the transport."""
    import time

    from robosat_amd import parallel
    from robosat_amd.autograd import GradArena
    from robosat_amd.unet import UNet

    torch.manual_seed(100 + rank)
    net = UNet(2, pretrained=False)  # (parameters only: nothing here computes)
    digest_before = float(sum(t.double().abs().sum() for t in net.state_dict().values()))
    t0 = time.perf_counter()
    parallel.broadcast_module(net)
    t_bcast = time.perf_counter() - t0
    digest = float(sum(t.double().abs().sum() for t in net.state_dict().values()))
    peers = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(peers, torch.tensor([digest], dtype=torch.float64))

    unused = {id(p) for p in net.resnet.fc.parameters()}
    params = [p for p in net.parameters() if id(p) not in unused]
    red = parallel.GradReducer(wire_dtype=torch.bfloat16 if wire == "bf16" else torch.float32)
    arena = GradArena(params, torch.device("cpu"), red)
    buckets, mark = [], 0
    g = torch.Generator().manual_seed(5)
    base = torch.randn(1 << 16, generator=g)  # the same pattern on every rank, scaled per rank: mean = pattern * mean(scale)
    t0 = time.perf_counter()
    for item in backward_order(net):
        if item is None:
            buckets.append((arena.off - mark) * 4 / 2**20)
            mark = arena.off
            arena.flush()
            continue
        p, krsc = item
        v = arena.conv(_Holder(p)) if krsc else arena.take(p, tuple(p.shape))
        flat = v.reshape(-1)
        reps = (flat.numel() + base.numel() - 1) // base.numel()
        flat.copy_(base.repeat(reps)[:flat.numel()] * (rank + 1.0))
    buckets.append((arena.off - mark) * 4 / 2**20)
    arena.finish()
    t_step = time.perf_counter() - t0
    want_scale = sum(range(1, world + 1)) / world
    # every parameter that takes part got a gradient view of its own shape, averaged over the ranks
    worst, covered = 0.0, 0
    for p in params:
        gv = arena.grads[p]
        assert tuple(gv.shape) == tuple(p.shape)
        flat = gv.permute(0, 2, 3, 1).reshape(-1) if gv.dim() == 4 else gv.reshape(-1)  # (conv gradients live in KRSC)
        reps = (flat.numel() + base.numel() - 1) // base.numel()
        want = base.repeat(reps)[:flat.numel()] * want_scale
        worst = max(worst, float((flat - want).abs().max() / want.abs().max()))
        covered += flat.numel()
    return {"world": world, "wire": wire, "buckets_mb": [round(b, 1) for b in buckets], "issued": red.issued, "elements": covered,
            "arena_full": arena.off == arena.flat.numel(), "worst_rel_err": worst, "joins": arena.joins,
            "replicas_equal": len({round(float(x), 3) for x in peers}) == 1, "started_different": True if rank == 0 else digest_before != digest,
            "seconds": {"broadcast": round(t_bcast, 2), "flushes": round(t_step, 2)}}


class _Holder:
    """What GradArena.conv wants: an object with a .weight (robosat_amd.unet._Conv has more; the arena only reads this)."""

    def __init__(self, weight):
        self.weight = weight


def gpu_unet(rank, world, dtype):
    """Two replicas on ONE MI355X (gloo reduces the device tensors through the host): a full U-Net training step through
    GradArena + GradReducer must give every rank the MEAN of the ranks' local gradients, and the replicas must stay
    identical through optimizer steps."""
    from robosat_amd import losses, parallel
    from robosat_amd.unet import UNet

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)
    net = UNet(2, pretrained=False, compute_dtype=dtype).to(dev).train()
    parallel.broadcast_module(net)
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(2, 3, 64, 64, generator=g).to(dev)
    t = torch.randint(0, 2, (2, 64, 64), generator=g).to(dev)
    crit = losses.LovaszLoss2d().to(dev)
    params = [p for n, p in net.named_parameters() if not n.startswith("resnet.fc.")]

    def grads(reducer):
        net.grad_reducer = reducer
        for p in net.parameters():
            p.grad = None
        loss = crit(net(x), t)
        loss.backward()
        torch.cuda.synchronize()
        return torch.cat([p.grad.reshape(-1).float() for p in params]), float(loss)

    local, loss_local = grads(None)
    want = local.clone()
    dist.all_reduce(want, op=dist.ReduceOp.SUM)
    want /= world
    got, loss_dp = grads(parallel.GradReducer())
    wire, _ = grads(parallel.GradReducer(wire_dtype=torch.bfloat16))  # `[model] grad_dtype = "bf16"`: half the bytes on the wire
    wire_err = float((wire - want).norm() / want.norm())
    peers = [torch.zeros_like(got) for _ in range(world)]
    dist.all_gather(peers, got)
    err = float((got - want).abs().max() / want.abs().max())
    differs = float((local - want).abs().max() / want.abs().max())  # the shards are different: local != mean

    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for _ in range(2):
        opt.zero_grad()
        crit(net(x), t).backward()
        opt.step()
    torch.cuda.synchronize()
    drift = 0.0
    for name, v in net.state_dict().items():
        if "running_" in name or "num_batches" in name:  # BatchNorm statistics stay per replica (as DataParallel's)
            continue
        ref = v.detach().clone()
        dist.broadcast(ref, src=0)
        drift = max(drift, float((v.float() - ref.float()).abs().max()))
    fc_grad = net.resnet.fc.weight.grad is None

    # the exchange must not put the wire (or the weight-gradient stream) on the backward's critical path: time from the start
    # of the backward to the main stream's last kernel before its ONE join with the side stream, with and without a reducer
    from robosat_amd.autograd import GradArena

    xb = torch.randn(4, 3, 256, 256, generator=g).to(dev)
    tb = torch.randint(0, 2, (4, 256, 256), generator=g).to(dev)

    def main_stream(reducer):
        net.grad_reducer = reducer
        best, rec = float("inf"), None
        for i in range(4):
            for p in net.parameters():
                p.grad = None
            GradArena.TRACE = []
            crit(net(xb), tb).backward()
            torch.cuda.synchronize()
            rec, = GradArena.TRACE
            if i:
                best = min(best, rec["events"][0].elapsed_time(rec["events"][1]))
        GradArena.TRACE = None
        dist.barrier()
        return best, rec["joins"], rec["flushes"]

    (ms_plain, joins_plain, _), (ms_dp, joins_dp, flushes) = main_stream(None), main_stream(parallel.GradReducer())
    return {"wire_bf16_rel_err": wire_err, "main_stream_ms": [ms_plain, ms_dp], "joins": [joins_plain, joins_dp],
            "flushes_on_side_stream": flushes, "grad_rel_err": err, "local_vs_mean": differs, "peer_equal": bool(all(torch.equal(p, got) for p in peers)),
            "replica_drift": drift, "fc_has_no_grad": bool(fc_grad), "loss_local": loss_local, "loss_dp": loss_dp}


def gpu_weighted_ce(rank, world):
    """Weighted cross-entropy / focal loss on shards with different class mixes: the average of the ranks' losses and
    gradients must equal the loss / gradient of ONE evaluation over the global batch, which is what the reference's
    DataParallel computes (tools/train.py:180-186) -- not the mean of per-shard normalised losses."""
    import torch.nn.functional as F

    from robosat_amd import losses

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    c, n, h, w = 3, 2, 32, 48
    weight = torch.tensor([0.5, 2.0, 5.0])
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(world * n, c, h, w, generator=g)
    # rank 0's shard is almost all class 0, rank 1's mostly class 2: very different weight sums
    probs = [torch.tensor([0.9, 0.08, 0.02]), torch.tensor([0.1, 0.2, 0.7])]
    targets = torch.cat([torch.multinomial(probs[r % 2], n * h * w, True, generator=g).view(n, h, w) for r in range(world)])
    out = {}
    def opted_in(crit):
        crit.global_batch = True  # (what rs train / bench.py set on the ranks of a data-parallel job)
        return crit

    for name, crit, ref in (("ce", opted_in(losses.CrossEntropyLoss2d(weight=weight)), lambda x: F.nll_loss(F.log_softmax(x, 1), targets, weight=weight)),
                            ("focal", opted_in(losses.FocalLoss2d(weight=weight)),
                             lambda x: F.nll_loss((1 - F.softmax(x, 1)) ** 2 * F.log_softmax(x, 1), targets, weight=weight))):
        x_ref = logits.clone().requires_grad_(True)
        want = ref(x_ref)
        want.backward()
        mine = logits[rank * n:(rank + 1) * n].to(dev).requires_grad_(True)
        loss = crit.to(dev)(mine, targets[rank * n:(rank + 1) * n].to(dev))
        loss.backward()
        mean_loss = loss.detach().clone()
        dist.all_reduce(mean_loss)
        mean_loss /= world
        got = mine.grad.cpu() / world  # (what the gradient all-reduce's average contributes for this shard)
        want_g = x_ref.grad[rank * n:(rank + 1) * n]
        out[name] = {"loss_err": abs(float(mean_loss) - float(want)) / abs(float(want)),
                     "grad_err": float((got - want_g).abs().max() / want_g.abs().max())}
    return out


def gpu_miou(rank, world):
    """mIoULoss2d on shards that would pick DIFFERENT branches of the reference's ``max(miou, nll)`` (losses.py:83): rank 0's
    logits are noise (NLL larger), rank 1's are confident and mostly right (soft-IoU term larger).  The reference evaluates
    the max ONCE over the gathered batch; the ranks' averaged loss and gradients must equal that single evaluation (the
    oracle's ``miou2d`` on the concatenated batch), and without the exchange they must not (or the case tests nothing)."""
    from oracle import robosat_ref as R
    from robosat_amd import losses

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    c, n, h, w = 3, 2, 32, 40
    weight = torch.tensor([1.0, 2.0, 0.5])
    g = torch.Generator().manual_seed(11)
    targets = torch.randint(0, c, (world * n, h, w), generator=g)
    logits = torch.randn(world * n, c, h, w, generator=g)
    onehot = torch.nn.functional.one_hot(targets, c).permute(0, 3, 1, 2).float()
    for r in range(1, world, 2):  # odd ranks: confident and right (p ~ 0.96): NLL ~ 0.04 < soft-IoU term ~ 0.07 -> the IoU branch alone
        sl = slice(r * n, (r + 1) * n)
        logits[sl] = 4.0 * onehot[sl] + 0.1 * logits[sl]
    x_ref = logits.clone().requires_grad_(True)
    want = R.miou2d(x_ref, targets, weight=weight)
    want.backward()
    per_shard = [float(R.miou2d(logits[r * n:(r + 1) * n], targets[r * n:(r + 1) * n], weight=weight)) for r in range(world)]
    out = {"want": float(want), "per_shard_oracle": per_shard}
    for name, opt_in in (("global", True), ("per_shard", False)):
        crit = losses.mIoULoss2d(weight=weight).to(dev)
        crit.global_batch = opt_in
        mine = logits[rank * n:(rank + 1) * n].to(dev).requires_grad_(True)
        loss = crit(mine, targets[rank * n:(rank + 1) * n].to(dev))
        loss.backward()
        mean_loss = loss.detach().clone()
        dist.all_reduce(mean_loss)
        mean_loss /= world
        got = mine.grad.cpu() / world
        want_g = x_ref.grad[rank * n:(rank + 1) * n]
        out[name] = {"loss": float(loss), "loss_err": abs(float(mean_loss) - float(want)) / abs(float(want)),
                     "grad_err": float((got - want_g).abs().max() / want_g.abs().max())}
    return out


def gpu_rccl1(dtype):
    """The RCCL branch of GradReducer on ONE MI355X: a process group of one rank over the nccl backend, the reducer forced
    (``force=True``) so that every bucket really goes through ``ProcessGroupNCCL`` -- the communicator stream ordered after
    the weight-gradient stream, five overlapping in-place AVG collectives on views of the one gradient arena, the bf16 wire's
    cast / SUM / cast back, the single end-of-backward join.  With one rank the average is the identity, so the step's
    gradients must be BIT-identical to the reducer-less step (bf16 wire: to the bf16 rounding of it)."""
    import warnings

    from robosat_amd import losses, parallel
    from robosat_amd.autograd import GradArena
    from robosat_amd.unet import UNet

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(3)
    net = UNet(2, pretrained=False, compute_dtype=dtype).to(dev).train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, 128, 128, generator=g).to(dev)
    t = torch.randint(0, 2, (4, 128, 128), generator=g).to(dev)
    crit = losses.LovaszLoss2d().to(dev)
    params = [p for n, p in net.named_parameters() if not n.startswith("resnet.fc.")]

    def grads(reducer, check_sync=False):
        net.grad_reducer = reducer
        for p in net.parameters():
            p.grad = None
        GradArena.TRACE = []
        loss = crit(net(x), t)
        torch.cuda.synchronize()
        synced = []
        if check_sync:  # torch warns on every call that makes the HOST wait for the device while this mode is on
            torch.cuda.set_sync_debug_mode("warn")
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                loss.backward()
            synced = [str(wn.message) for wn in caught if "synchroniz" in str(wn.message).lower()]
        finally:
            if check_sync:
                torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        rec, = GradArena.TRACE
        GradArena.TRACE = None
        return torch.cat([p.grad.reshape(-1).float() for p in params]), rec, synced

    local, rec0, _ = grads(None)
    red = parallel.GradReducer(force=True)
    grads(red)  # (the communicator is created lazily by the first collective: RCCL's own set-up may synchronise)
    got, rec1, synced = grads(red, check_sync=True)
    red16 = parallel.GradReducer(wire_dtype=torch.bfloat16, force=True)
    wire, rec2, _ = grads(red16)
    # a few optimizer steps with the reducer in the loop: training goes on, weights stay finite
    net.grad_reducer = red
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    for _ in range(3):
        opt.zero_grad()
        crit(net(x), t).backward()
        opt.step()
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(p).all()) for p in net.parameters())
    return {"backend": red.backend, "world": red.world, "issued_fp32": red.issued, "issued_bf16": red16.issued,
            "bit_identical_fp32": bool(torch.equal(got, local)), "max_abs_diff_fp32": float((got - local).abs().max()),
            "bit_identical_bf16_rounding": bool(torch.equal(wire, local.bfloat16().float())),
            "bf16_rel_err": float((wire - local).norm() / local.norm()),
            "joins": [rec0["joins"], rec1["joins"], rec2["joins"]], "flushes_on_side_stream": [rec1["flushes"], rec2["flushes"]],
            "host_syncs_in_backward": synced, "finite_after_steps": finite, "grad_norm": float(local.norm())}


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    from robosat_amd import parallel

    parallel.init_process_group(world, rank, backend="nccl" if mode.startswith("gpu_rccl1") else "gloo")
    if mode == "cpu":
        res = cpu_logic(rank, world)
    elif mode == "gpu_wce":
        res = gpu_weighted_ce(rank, world)
    elif mode == "gpu_miou":
        res = gpu_miou(rank, world)
    elif mode.startswith("gpu_rccl1"):
        res = gpu_rccl1(torch.bfloat16 if mode.endswith("bf16") else torch.float32)
    elif mode.startswith("arena8"):
        res = arena8(rank, world, "bf16" if mode.endswith("bf16") else "fp32")
    elif mode == "fail":
        if rank == 1:
            sys.exit(3)  # a lost rank: the launcher must stop the survivor (who would wait forever below)
        dist.barrier()
        res = {}
    else:
        res = gpu_unet(rank, world, torch.bfloat16 if mode == "gpu_bf16" else torch.float32)
    with open(os.path.join(outdir, "rank{}.json".format(rank)), "w") as fp:
        json.dump(res, fp)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
