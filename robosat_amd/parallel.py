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

import os

import torch
import torch.distributed as dist


def init_process_group(world, rank, backend=None):
    """Join the job's process group: RCCL ("nccl") on the MI355X; ``ROBOSAT_DIST_BACKEND=gloo`` selects gloo (the
    world-size-2 tests that run two ranks on ONE GPU -- RCCL refuses two ranks per device; gloo reduces device tensors
    through host memory)."""

    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or os.environ.get("ROBOSAT_DIST_BACKEND", "nccl")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)


def broadcast_module(module, src=0):
    """Every parameter and buffer of ``module`` := rank ``src``'s.  ``DataParallel`` re-broadcasts the parameters of
    device 0 on every forward (reference tools/train.py:69,180); with one replica per process the replicas only have
    to START identical -- the averaged gradients then keep them identical -- so this runs once after construction /
    checkpoint load.  Tensors travel in one flat buffer per dtype (a handful of collectives, not 329)."""

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers())]
    _broadcast_flat(tensors, src)


def broadcast_bn_buffers(module, src=0):
    """BatchNorm running statistics := rank ``src``'s.  The reference's replicas update private copies that are thrown
    away; only device 0's survive.  Called before validation / checkpointing so that the logged
    validation numbers are those of the model rank 0 saves."""

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    _broadcast_flat([b.data for b in module.buffers()], src)


def _broadcast_flat(tensors, src):
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))  # (.copy_ keeps each tensor's memory format, e.g. KRSC weights)
            off += n


class GradReducer:
    """Averages ranges of the flat gradient arena across ranks, asynchronously.

    ``reduce_async`` is called with the stream the collective should be ordered after as torch's CURRENT stream
    (``ProcessGroupNCCL`` makes its communicator stream wait for the current stream, runs the collective there and returns):
    ``GradArena.flush`` calls it under the weight-gradient side stream, so a bucket's exchange starts when that stream has
    finished the bucket's last weight gradient -- the backward's main stream (data gradients, BatchNorm) is never made to
    wait for the side stream or for the wire.  ``wait`` -- called once, at the end of the backward --
    makes the current stream wait for every pending collective.

    ``wire_dtype=torch.bfloat16`` (``[model] grad_dtype = "bf16"``): the bucket travels as bf16 (74.7 MB per step instead of
    149.4 MB): cast on the device (``rs_cast_f32_to_bf16``), summed by RCCL, and converted back into the fp32 arena with the
    1/world scale (``rs_cast_bf16_to_f32_scaled``) -- the optimizer still reads fp32 gradients, local accumulation (the
    weight-gradient kernels) stays fp32; what is rounded is each rank's contribution to the sum and the partial sums on the
    ring.  Default: fp32 on the wire, bit-identical replicas, the reference's arithmetic."""

    def __init__(self, group=None, wire_dtype=torch.float32, force=False):
        """``force``: issue the collectives even in a group of ONE rank (a world-size-1 RCCL communicator is legal: the
        all-reduce is then an in-place copy on RCCL's stream).  That is how the ``backend == "nccl"`` branch -- the
        communicator stream's ordering after the side stream, the in-place AVG on views of one arena, the bf16 wire -- runs
        on a one-GPU box (tests/test_gpu_parallel.py, ``bench.py --force-reducer``); a real job never sets it."""

        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        if wire_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("GradReducer: gradients travel as fp32 or bf16")
        self.wire_dtype = wire_dtype
        self.force = bool(force)
        self.issued = 0  # collectives issued so far (tests: the branch really ran)
        self._pending = []

    def reduce_async(self, flat):
        """Enqueue an in-place average of ``flat`` (a contiguous 1-D fp32 view), ordered after torch's current stream."""

        if self.world == 1 and not self.force:
            return
        self.issued += 1
        if self.wire_dtype == torch.bfloat16:
            # every rank puts ALREADY AVERAGED contributions on the wire (x 1/world inside the cast): the ring's partial sums
            # then stay at the magnitude of one gradient however many ranks there are -- summing WORLD unscaled bf16 values and
            # dividing afterwards loses mantissa to the larger partial sums and overflows earlier
            if flat.is_cuda:
                from . import ops

                wire = ops.cast_bf16_scaled(flat, 1.0 / self.world)
            else:  # host tensors (the gloo tests of the wire's bookkeeping): the same scale -> round-to-nearest-even cast
                wire = (flat * (1.0 / self.world)).to(torch.bfloat16)
            work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            # (`wire` is consumed on RCCL's communicator stream; the reference held in _pending until wait() has cast it back
            # is what keeps its memory from being recycled)
            self._pending.append((work, flat, wire))
        elif self.backend == "nccl":
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self._pending.append((work, None, None))
        else:  # gloo (CPU tests, two ranks on one GPU): no AVG
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((work, flat, None))

    def wait(self):
        for work, flat, wire in self._pending:
            work.wait()  # (device tensors: the CURRENT stream waits for the collective; the host does not block on RCCL)
            if wire is not None and not flat.is_cuda:
                flat.copy_(wire.float())
            elif wire is not None:
                from . import ops

                ops.cast_f32_scaled(wire, flat, 1.0)  # (the contributions were scaled by 1/world on the way out)
            elif flat is not None:
                flat.div_(self.world)
        self._pending = []


def shard_indices(num_items, batch_size, rank, world, epoch_order=None, drop_last=True):
    """The slice of an epoch's sample order that ``rank`` processes; ``batch_size`` = samples per rank per step.

    The global order (``epoch_order``: a permutation for the shuffled training loader, ``range`` for validation --
    reference tools/train.py:273-274) is cut into global batches of ``batch_size * world`` samples with the reference's
    ``drop_last=True`` semantics; each rank takes its contiguous ``batch_size`` share of every global batch, which is
    how ``DataParallel`` splits a batch along dim 0 (tools/train.py:69).  ``rs train`` passes
    ``[common] batch_size / world`` here: the TOML batch size stays the GLOBAL batch, as in the reference."""

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
