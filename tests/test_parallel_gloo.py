"""N>1 data-parallel path on CPU: two gloo ranks started by the product launcher (``robosat_amd.launch.spawn_ranks``,
the code behind ``rs train`` / ``bench.py --gpus N``) exercise the flat-arena gradient reducer, the replica broadcast
and the tile sharding (one process per GPU in production, RCCL instead of gloo).  The same two-rank job with REAL
kernels runs on the GPU box: tests/test_gpu_parallel.py."""

import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dp_worker.py")


def run_world2(mode, outdir, timeout=300, world=2):
    from robosat_amd import launch

    env = dict(os.environ)
    env["ROBOSAT_DIST_BACKEND"] = "gloo"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    rc = launch.spawn_ranks([sys.executable, WORKER, mode, str(outdir)], world, env=env, timeout=timeout)
    res = []
    for r in range(world):
        path = os.path.join(str(outdir), "rank{}.json".format(r))
        res.append(json.load(open(path)) if os.path.exists(path) else None)
    return rc, res


def test_launcher_reducer_broadcast_and_sharding_world2(tmp_path):
    rc, res = run_world2("cpu", tmp_path)
    assert rc == 0
    for rank, r in enumerate(res):
        assert r["flat_ok"]
        assert abs(r["loss"] - 0.5) < 1e-12
        assert r["counts"] == [2, 4, 6, 9]
        assert r["sums"][0] == r["sums"][1]  # after broadcast_module both replicas hold rank 0's parameters + buffers
        assert r["running_var"] == [2.0] * 4  # rank 0's running_var (x2), also on rank 1 (which had x3)
    # mIoULoss2d under data parallelism: ONE branch decision over the global batch (losses.py:72-83 under DataParallel).
    # rank 0: miou 0.8 < nll 1.4; rank 1: miou 0.07 > nll 0.036 -> alone they differ; global: nll (14.4 / 21) > miou 0.435
    for rank, r in enumerate(res):
        m = r["miou_dp"]
        assert m["branch"] == 1.0 and abs(m["den"] - 10.5) < 1e-6, m            # NLL branch everywhere, D = mean_r sum w
        assert abs(m["loss"] - ((14.0, 0.4)[rank] / 10.5)) < 1e-6, m            # num_r / D: the ranks' mean is 14.4 / 21
        assert m["local_branch"] == (1.0, 0.0)[rank]                              # (the shards alone choose differently)
        assert abs(m["untouched_without_opt_in"] - (1.4, 0.07)[rank]) < 1e-6, m  # no collective, no change without the opt-in
    assert abs((res[0]["miou_dp"]["loss"] + res[1]["miou_dp"]["loss"]) / 2 - 14.4 / 21.0) < 1e-6
    assert res[0]["before"] != res[1]["before"]  # ... and they really started different
    # 10 samples, 2 per rank, 2 ranks -> 2 global batches of 4 (drop_last), split contiguously like DataParallel
    assert res[0]["shards"] == [[0, 1], [4, 5]] and res[1]["shards"] == [[2, 3], [6, 7]]
    # rs predict: sequential batches of 3 dealt round-robin, every tile exactly once
    assert res[0]["predict_batches"] == [[0, 1, 2], [6, 7, 8]] and res[1]["predict_batches"] == [[3, 4, 5], [9]]


def test_launcher_stops_the_survivors_of_a_failed_rank(tmp_path):
    rc, res = run_world2("fail", tmp_path, timeout=120)
    assert rc == 3
    assert res == [None, None]


def test_a_failed_rank_takes_an_eight_rank_job_down(tmp_path):
    """The node-sized job: rank 1 of EIGHT exits, the launcher stops the seven that would wait in the collective forever."""
    rc, res = run_world2("fail", tmp_path, timeout=180, world=8)
    assert rc == 3
    assert res == [None] * 8


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_arena_and_reducer_at_the_real_size_with_eight_ranks(tmp_path, wire):
    """nothing in the N > 1 path should be first-run code when a node appears.  Eight gloo ranks carve the
    REAL U-Net's 168 gradient tensors into the flat arena in the backward's order, flush the six buckets of their real sizes
    through GradReducer (fp32 wire: exact mean; bf16 wire: pre-scaled contributions, one rounding each way) and broadcast the
    whole 329-entry state dict from rank 0.  (The kernels and RCCL's transport are what this leaves out: tests/test_gpu_parallel.py
    runs the same classes on device tensors, with the nccl backend at world size 1.)"""
    rc, res = run_world2("arena8_" + wire, tmp_path, timeout=900, world=8)
    assert rc == 0, res
    for rank, r in enumerate(res):
        assert r["world"] == 8 and r["issued"] == 6 and r["arena_full"] and r["joins"] == 1, r
        assert r["elements"] == 37341314  # every parameter with a gradient (resnet.fc has none)
        # head + decoder 55 MB (52.8 MiB), layer4 60, layer3 28, layer2 5, layer1 1, the stem's 38 KB at finish
        assert [round(b) for b in r["buckets_mb"]] == [53, 57, 27, 5, 1, 0], r["buckets_mb"]
        assert r["replicas_equal"] and r["started_different"]
        assert r["worst_rel_err"] <= (1e-6 if wire == "fp32" else 2.0 ** -7), r["worst_rel_err"]


def test_shard_indices_single_rank_is_reference_order():
    from robosat_amd.parallel import shard_indices

    assert shard_indices(7, 2, 0, 1) == [[0, 1], [2, 3], [4, 5]]  # drop_last=True as the reference loaders
    assert shard_indices(5, 2, 0, 1, epoch_order=[4, 3, 2, 1, 0]) == [[4, 3], [2, 1]]


def test_ranks_for_batch_follows_dataparallel_scatter():
    from robosat_amd.launch import ranks_for_batch

    assert ranks_for_batch(2, 8) == 2    # the reference's default batch_size = 2 keeps two of eight GPUs busy
    assert ranks_for_batch(32, 8) == 8
    assert ranks_for_batch(12, 8) == 6
    assert ranks_for_batch(7, 4) == 1
    assert ranks_for_batch(16, 1) == 1


def test_sharding_properties_for_every_world_size():
    """For 1..8 ranks: the ranks' shares of a global batch, concatenated in rank order, ARE the reference's batch (what
    DataParallel's scatter along dim 0 gives, tools/train.py:69), with the reference's drop_last; ``rs predict``'s
    sequential batches are dealt so that every tile is predicted exactly once and ``len()`` tells each rank's count."""

    import random

    from robosat_amd.parallel import shard_indices
    from robosat_amd.tools.predict import RankBatchSampler

    rng = random.Random(7)
    for world in range(1, 9):
        for n, per_rank in ((0, 2), (5, 1), (64, 4), (67, 2), (100, 3)):
            order = list(range(n))
            rng.shuffle(order)
            shards = [shard_indices(n, per_rank, r, world, epoch_order=order) for r in range(world)]
            gb = per_rank * world
            assert all(len(s) == n // gb for s in shards)  # drop_last on the GLOBAL batch, the same step count on every rank
            for b in range(n // gb):
                assert sum((shards[r][b] for r in range(world)), []) == order[b * gb:(b + 1) * gb]
            # validation order: no permutation given = range(n)
            assert shard_indices(n, per_rank, 0, world)[:1] == ([list(range(per_rank))] if n >= gb else [])
        for n, bs in ((0, 3), (1, 3), (10, 3), (4096, 16), (4097, 16)):
            per_rank = [list(RankBatchSampler(n, bs, r, world)) for r in range(world)]
            assert [len(RankBatchSampler(n, bs, r, world)) for r in range(world)] == [len(p) for p in per_rank]
            seen = sorted(i for p in per_rank for batch in p for i in batch)
            assert seen == list(range(n))  # every tile once
            # batch b of the reference's sequential loader goes to rank b % world, whole
            for r, p in enumerate(per_rank):
                for k, batch in enumerate(p):
                    b = k * world + r
                    assert batch == list(range(b * bs, min((b + 1) * bs, n)))
