"""Data parallelism for ``rs train``: one process per MI355X, gradients summed with RCCL over xGMI.

The reference wraps the model in single-process ``torch.nn.DataParallel`` (tools/train.py:69): each step it broadcasts
157.8 MB of parameters, scatters the batch, gathers the logits and reduce-adds 149.4 MB of gradients onto GPU 0
(SURVEY.md section 2.3).  Here every rank owns a replica and a shard of the tiles, so the only exchange left is ONE
average of the gradients per step -- and because the backward pass writes all gradients into one flat arena in
production order (``robosat_amd.autograd.GradArena``), that exchange is a handful of large in-place all-reduces on
contiguous ranges (head+decoder 55 MB, layer4 60 MB, layer3 28 MB, layer2 5 MB, layer1+stem 1 MB) launched as each
range completes and overlapped with the remaining backward kernels.  xGMI is point-to-point, so few large messages
are what keeps all 7 links busy; per-tensor collectives (168 of them) would be latency-bound.

BatchNorm statistics stay per-rank (unsynchronised), exactly like the per-replica statistics of ``DataParallel``.
"""

import torch
import torch.distributed as dist


class GradReducer:
    """Averages ranges of the flat gradient arena across ranks, asynchronously."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self._pending = []

    def reduce_async(self, flat):
        """Enqueue an in-place average of ``flat`` (a contiguous 1-D view).  With the RCCL backend the collective runs
        on the communicator's own stream after the kernels enqueued so far and overlaps with later compute."""

        if self.world == 1:
            return
        if self.backend == "nccl":
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self._pending.append((work, None))
        else:  # gloo (CPU tests): no AVG
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, flat))

    def wait(self):
        for work, flat in self._pending:
            work.wait()
            if flat is not None:
                flat.div_(self.world)
        self._pending = []


def shard_indices(num_items, batch_size, rank, world, epoch_order=None, drop_last=True):
    """The slice of an epoch's sample order that ``rank`` processes.

    The global order (``epoch_order``: a permutation for the shuffled training loader, ``range`` for validation --
    reference tools/train.py:273-274) is cut into global batches of ``batch_size * world`` samples with the reference's
    ``drop_last=True`` semantics; each rank takes its contiguous ``batch_size`` share of every global batch, which is
    how ``DataParallel`` splits a batch along dim 0 (tools/train.py:69)."""

    order = list(range(num_items)) if epoch_order is None else list(epoch_order)
    gb = batch_size * world
    nb = len(order) // gb if drop_last else (len(order) + gb - 1) // gb
    out = []
    for b in range(nb):
        chunk = order[b * gb:(b + 1) * gb]
        out.append(chunk[rank * batch_size:(rank + 1) * batch_size])
    return out


def average_scalars(values, device):
    """All-reduce (mean) a few Python floats -- the per-step loss for logging."""

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).tolist()


def sum_counts(counts):
    """All-reduce (sum) the int64[4] confusion counters at the end of an epoch."""

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
