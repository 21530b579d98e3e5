"""Comparing Lovasz gradients (reference losses.py:96-119) when the sort has ties.

The loss sorts the C*H*W margin errors of an image and weights them with the increments of the Jaccard curve.  Elements
with EQUAL errors may be visited in any order: the loss does not depend on it (a tie group contributes error x the
Jaccard increment over the whole group), but how that increment is split among the group's members -- their individual
gradients -- does, and ``torch.sort`` on the CPU is not stable at these sizes (on 2^20 normal samples ~1 % of its indices
differ from ``stable=True``), so the reference's own gradient inside a tie group is the sort implementation's choice.
What IS defined: d loss / d error summed over each group of equal errors.  That is what is compared -- element by element
wherever an error is unique (~99 % of a random image), group sum by group sum elsewhere."""

import torch


def lovasz_grad_groups(logits, targets, grad):
    """Per image, d loss / d error (= -sign * d loss / d logit, error = 1 - sign * logit) summed over groups of equal errors,
    laid out in ascending error order: float64 tensors, one per image."""

    n, c = logits.shape[:2]
    sign = torch.zeros_like(logits).scatter_(1, targets.view(n, 1, *targets.shape[1:]), 1.0) * 2 - 1
    errors = (1.0 - sign * logits).reshape(n, -1)  # (fp32: the very values the reference sorts)
    g_err = (-sign * grad).reshape(n, -1).double()
    out = []
    for e, g in zip(errors, g_err):
        uniq, inverse = torch.unique(e, return_inverse=True)
        out.append(torch.zeros(uniq.numel(), dtype=torch.float64).index_add_(0, inverse, g))
    return out


def assert_lovasz_grad_close(logits, targets, got, want, tol, what="lovasz grad"):
    """``got`` / ``want``: d loss / d logits (CPU tensors) for the same fp32 ``logits``; relative to the largest reference entry."""

    scale = max(float(want.abs().max()), 1e-30)
    worst = 0.0
    for a, b in zip(lovasz_grad_groups(logits, targets, got), lovasz_grad_groups(logits, targets, want)):
        assert a.shape == b.shape
        worst = max(worst, float((a - b).abs().max()))
    assert worst <= tol * scale, "{}: max abs err over tie groups {} (scale {})".format(what, worst, scale)
    # outside the tie groups the comparison above IS elementwise; in addition nothing may be wildly off anywhere: a tie
    # group's members share a handful of neighbouring Jaccard increments
    assert float((got - want).abs().max()) <= 0.2 * scale, "{}: elementwise {} (scale {})".format(what, float((got - want).abs().max()), scale)
    return worst / scale
