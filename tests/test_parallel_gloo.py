"""N>1 data-parallel path on CPU: world_size-2 gloo processes exercise the flat-arena gradient reducer and the
tile sharding (one process per GPU in production, RCCL instead of gloo)."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robosat_amd.parallel import GradReducer, average_scalars, shard_indices, sum_counts

    red = GradReducer()
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)  # the arena
    red.reduce_async(flat[0:400])  # bucket 1 while "backward" continues
    red.reduce_async(flat[400:1000])
    red.wait()
    want = torch.arange(1000, dtype=torch.float32) * (1 + 2) / 2.0
    ok = torch.allclose(flat, want)
    loss = average_scalars([float(rank)], torch.device("cpu"))[0]
    counts = sum_counts(torch.tensor([1, 2, 3, 4 + rank], dtype=torch.int64))
    shards = shard_indices(10, 2, rank, world)
    out.put((rank, ok, loss, counts.tolist(), shards))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, loss, counts, shards in res:
        assert ok
        assert abs(loss - 0.5) < 1e-12
        assert counts == [2, 4, 6, 9]
    # 10 samples, batch 2 per rank, 2 ranks -> 2 global batches of 4 (drop_last), split contiguously like DataParallel
    assert res[0][4] == [[0, 1], [4, 5]] and res[1][4] == [[2, 3], [6, 7]]


def test_shard_indices_single_rank_is_reference_order():
    from robosat_amd.parallel import shard_indices

    assert shard_indices(7, 2, 0, 1) == [[0, 1], [2, 3], [4, 5]]  # drop_last=True as the reference loaders
    assert shard_indices(5, 2, 0, 1, epoch_order=[4, 3, 2, 1, 0]) == [[4, 3], [2, 1]]
