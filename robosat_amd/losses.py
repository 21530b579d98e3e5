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


def _global_batch_normaliser(loss, stats):
    """Data-parallel ranks: make the weighted-NLL normaliser the GLOBAL batch's.

    The reference's ``DataParallel`` gathers the logits of all replicas and evaluates ONE loss over the global batch
    (tools/train.py:180-186): ``sum_i w_i l_i / sum_i w_i`` with both sums over every pixel of every shard.  Each rank here
    sees only its shard, and the mean of per-shard ratios is a different number (and gradient) whenever the shards' class
    mixes -- hence their weight sums -- differ.  The fix costs one scalar all-reduce: with ``D = mean_r sum_i w_i`` (the
    global denominator / world) replacing the local denominator, the rank's loss becomes ``num_r / D`` and the average of
    the ranks' losses / gradients that ``rs train`` forms anyway is exactly the global-batch loss / gradient.
    ``stats[1]`` (what ``rs_nll_loss_bwd`` divides by) is rewritten in place; returns the rescaled loss."""

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return loss
    local = stats[1].clone()
    den = stats[1:2]
    dist.all_reduce(den)  # (a 4-byte SUM on the device; no host sync)
    den.div_(dist.get_world_size())
    return loss * (local / den[0])


class _NLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, weight, mode, gamma):
        x = inputs.detach().float()
        loss, stats = ops.nll_loss_fwd(x, targets, weight, mode, gamma)
        loss = _global_batch_normaliser(loss, stats)
        ctx.save_for_backward(x, targets, stats)
        ctx.weight, ctx.mode, ctx.gamma = weight, mode, gamma
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        x, targets, stats = ctx.saved_tensors
        g = grad_out.detach().float().contiguous()
        return ops.nll_loss_bwd(x, targets, ctx.weight, stats, g, ctx.mode, ctx.gamma), None, None, None, None


class _MIoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, weight):
        x = inputs.detach().float()
        loss, stats = ops.miou_loss_fwd(x, targets, weight)
        ctx.save_for_backward(x, targets, stats)
        ctx.weight = weight
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        x, targets, stats = ctx.saved_tensors
        return ops.miou_loss_bwd(x, targets, ctx.weight, stats, grad_out.detach().float().contiguous()), None, None


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
    def __init__(self, weight=None):
        super().__init__()
        # a buffer (not a parameter) so that ``criterion.to(device)`` moves it, as nn.NLLLoss(weight) does
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32).clone())


class CrossEntropyLoss2d(_WeightedLoss):
    """Cross-entropy: weighted NLL of log_softmax over the class axis (reference losses.py:8-25)."""

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _NLLFn.apply(inputs, targets, self.weight, ops.NLL_CROSS_ENTROPY, 0.0)


class FocalLoss2d(_WeightedLoss):
    """Focal loss, gamma = 2 by default (reference losses.py:28-50)."""

    def __init__(self, gamma=2, weight=None):
        super().__init__(weight)
        self.gamma = gamma

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _NLLFn.apply(inputs, targets, self.weight, ops.NLL_FOCAL, float(self.gamma))


class mIoULoss2d(_WeightedLoss):
    """Soft mean-IoU loss; like the reference it returns ``max(miou, weighted NLL)`` and back-propagates through
    whichever of the two is larger (reference losses.py:53-83)."""

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _MIoUFn.apply(inputs, targets, self.weight)


class LovaszLoss2d(nn.Module):
    """The reference's Lovasz hinge variant over the flattened C*H*W vector per image (losses.py:86-119)."""

    def forward(self, inputs, targets):
        inputs, targets = _check(inputs, targets)
        return _LovaszFn.apply(inputs, targets)
