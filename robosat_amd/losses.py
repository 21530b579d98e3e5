"""Losses of the reference (``robosat/losses.py``) on the MI355X: same class names, constructor arguments and
``forward(inputs[N,C,H,W] float32, targets[N,H,W] int64) -> 0-dim tensor`` contract (tools/train.py:97-106,185), with
the arithmetic in ``librobosat_hip.so`` (``rs_nll_loss_*``, ``rs_lovasz_fwd``).  No CPU path."""

import torch
import torch.nn as nn

from . import ops


def _check(inputs, targets):
    if not inputs.is_cuda:
        raise RuntimeError("robosat_amd losses run on the MI355X only (got a {} tensor); there is no CPU fallback".format(inputs.device))
    assert inputs.dim() == 4 and targets.dim() == 3 and inputs.shape[0] == targets.shape[0] and inputs.shape[2:] == targets.shape[1:]
    assert targets.dtype == torch.int64
    return inputs.contiguous(), targets.contiguous()


def _dp_world(global_batch):
    """World size when ``global_batch`` asks for the data-parallel exchange and there is a group of > 1 ranks, else 1."""

    import torch.distributed as dist

    if not global_batch or not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size()


def _global_batch_normaliser(loss, stats, global_batch):
    """Data-parallel ranks: make the weighted-NLL normaliser the GLOBAL batch's.

    The reference's ``DataParallel`` gathers the logits of all replicas and evaluates ONE loss over the global batch
    (tools/train.py:180-186): ``sum_i w_i l_i / sum_i w_i`` with both sums over every pixel of every shard.  Each rank here
    sees only its shard, and the mean of per-shard ratios is a different number (and gradient) whenever the shards' class
    mixes -- hence their weight sums -- differ.  The fix costs one scalar all-reduce: with ``D = mean_r sum_i w_i`` (the
    global denominator / world) replacing the local denominator, the rank's loss becomes ``num_r / D`` and the average of
    the ranks' losses / gradients that ``rs train`` forms anyway is exactly the global-batch loss / gradient.
    ``stats[1]`` (what ``rs_nll_loss_bwd`` divides by) is rewritten in place; returns the rescaled loss.

    OPT-IN (``criterion.global_batch = True``; ``rs train`` and ``bench.py`` set it on the ranks of a data-parallel job):
    it is a blocking collective inside ``forward``, so EVERY rank must call the loss the same number of times -- a
    rank-0-only evaluation or an unevenly sharded validation pass would hang in it."""

    import torch.distributed as dist

    world = _dp_world(global_batch)
    if world == 1:
        return loss
    local = stats[1].clone()
    den = stats[1:2]
    dist.all_reduce(den)  # (a 4-byte SUM on the device; no host sync)
    den.div_(world).clamp_(min=1e-30)  # (a global batch whose every pixel has weight 0: 0 / tiny = 0, not NaN)
    return loss * (local / den[0])


def _global_batch_miou(loss, stats, global_batch):
    """Data-parallel ranks: ``mIoULoss2d``'s ``max(miou, nll)`` decided ONCE over the global batch (losses.py:72-83 under
    tools/train.py:69), not per shard.

    The soft-IoU term is a mean over (class, image), so the global value is the mean of the ranks' values; the NLL term is
    ``sum_i w_i l_i / sum_i w_i`` over every shard.  The ranks exchange three scalars (their miou, NLL numerator, weight sum:
    one 12-byte all-reduce on the device, no host sync), every rank takes the same branch, and in the NLL branch the
    denominator becomes ``D = mean_r sum_i w_i`` as for the plain weighted losses: the average of the ranks' losses and
    gradients then equals the reference's single evaluation.  ``stats[1]`` / ``stats[2]`` (what ``rs_miou_loss_bwd`` divides
    by / branches on) are rewritten in place.  Opt-in like ``_global_batch_normaliser``."""

    import torch.distributed as dist

    world = _dp_world(global_batch)
    if world == 1:
        return loss
    n = stats.numel()
    miou_r, num_r = stats[n - 2], stats[n - 1]
    v = torch.stack([miou_r, num_r, stats[1]])
    dist.all_reduce(v)
    miou_g = v[0] / world
    den = (v[2] / world).clamp(min=1e-30)
    nll_g = v[1] / v[2].clamp(min=1e-30)
    use_nll = nll_g > miou_g
    stats[1] = den
    stats[2] = use_nll.to(stats.dtype)
    return torch.where(use_nll, num_r / den, miou_r)


class _NLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, weight, mode, gamma, global_batch=False):
        x = inputs.detach().float()
        loss, stats = ops.nll_loss_fwd(x, targets, weight, mode, gamma)
        loss = _global_batch_normaliser(loss, stats, global_batch)
        ctx.save_for_backward(x, targets, stats)
        ctx.weight, ctx.mode, ctx.gamma = weight, mode, gamma
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        x, targets, stats = ctx.saved_tensors
        g = grad_out.detach().float().contiguous()
        return ops.nll_loss_bwd(x, targets, ctx.weight, stats, g, ctx.mode, ctx.gamma), None, None, None, None, None


class _MIoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, weight, global_batch=False):
        x = inputs.detach().float()
        loss, stats = ops.miou_loss_fwd(x, targets, weight)
        loss = _global_batch_miou(loss, stats, global_batch)
        ctx.save_for_backward(x, targets, stats)
        ctx.weight = weight
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        x, targets, stats = ctx.saved_tensors
        return ops.miou_loss_bwd(x, targets, ctx.weight, stats, grad_out.detach().float().contiguous()), None, None, None


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets):
        loss, grad_unit = ops.lovasz_fwd(inputs.detach().float(), targets, want_grad=True)
        ctx.save_for_backward(grad_unit)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (grad_unit,) = ctx.saved_tensors
        return ops.scale_by_scalar(grad_unit, grad_out.detach().float().contiguous()), None


class _WeightedLoss(nn.Module):
    # Set to True on the ranks of a data-parallel job (rs train, bench.py): the batch-level terms of the loss -- the weighted
    # NLL's normaliser, mIoU's branch choice -- are then the GLOBAL batch's, as in the reference's single evaluation over the
    # gathered logits.  It adds a small blocking collective to forward(): every rank must call the loss equally often.
    global_batch = False

    def __init__(self, weight=None):
        super().__init__()
        # a buffer (not a parameter) so that ``criterion.to(device)`` moves it, as nn.NLLLoss(weight) does
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32).clone())


class CrossEntropyLoss2d(_WeightedLoss):
    """Cross-entropy: weighted NLL of log_softmax over the class axis (reference losses.py:8-25)."""

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _NLLFn.apply(inputs, targets, self.weight, ops.NLL_CROSS_ENTROPY, 0.0, self.global_batch)


class FocalLoss2d(_WeightedLoss):
    """Focal loss, gamma = 2 by default (reference losses.py:28-50)."""

    def __init__(self, gamma=2, weight=None):
        super().__init__(weight)
        self.gamma = gamma

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _NLLFn.apply(inputs, targets, self.weight, ops.NLL_FOCAL, float(self.gamma), self.global_batch)


class mIoULoss2d(_WeightedLoss):
    """Soft mean-IoU loss; like the reference it returns ``max(miou, weighted NLL)`` and back-propagates through
    whichever of the two is larger (reference losses.py:53-83); with ``global_batch`` the choice is made once over the
    data-parallel job's global batch (``_global_batch_miou``)."""

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _MIoUFn.apply(inputs, targets, self.weight, self.global_batch)


class LovaszLoss2d(nn.Module):
    """The reference's Lovasz hinge variant over the flattened C*H*W vector per image (losses.py:86-119)."""

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _LovaszFn.apply(inputs, targets)
