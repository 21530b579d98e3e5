"""Segmentation metrics of the reference (``robosat/metrics.py``) with the counting done on the GPU.

Same class/method names and numbers (including the reference's ``fn``/``fp`` naming, which is swapped relative to the
usual meaning -- harmless for mIoU / IoU / MCC).  The reference adds one sample at a time and synchronises the device
four times per sample (metrics.py:38-41); here ``add`` / ``add_batch`` only enqueue a counting kernel
(``rs_confusion_counts``) and the single device->host copy happens when a score is requested.
"""

import math

import numpy as np
import torch

from . import ops


class Metrics:
    """Tracking mean metrics (binary: label 0 = background, label 1 = foreground, as the reference)."""

    def __init__(self, labels):
        self.labels = labels
        self._counts = None  # int64[4] on the device: tn, fn, fp, tp
        self._host = np.zeros(4, dtype=np.int64)

    def _buf(self, device):
        if self._counts is None:
            self._counts = torch.zeros(4, device=device, dtype=torch.int64)
        return self._counts

    def add(self, actual, predicted):
        """One observation: ``actual`` [H,W] labels, ``predicted`` [C,H,W] scores (reference signature)."""

        self.add_batch(actual.unsqueeze(0), predicted.unsqueeze(0))

    def add_batch(self, actual, predicted):
        """A whole batch: ``actual`` [N,H,W] int64, ``predicted`` [N,C,H,W] float32."""

        ops.confusion_counts(predicted.detach().float().contiguous(), actual.contiguous(), self._buf(predicted.device))

    def _sync(self):
        if self._counts is not None:
            self._host += self._counts.cpu().numpy()
            self._counts.zero_()
        return [int(v) for v in self._host]

    @property
    def tn(self):
        return self._sync()[0]

    @property
    def fn(self):
        return self._sync()[1]

    @property
    def fp(self):
        return self._sync()[2]

    @property
    def tp(self):
        return self._sync()[3]

    def get_miou(self):
        tn, fn, fp, tp = self._sync()
        try:
            return float(np.nanmean([tn / (tn + fn + fp), tp / (tp + fn + fp)]))
        except ZeroDivisionError:
            return float("NaN")

    def get_fg_iou(self):
        tn, fn, fp, tp = self._sync()
        try:
            return tp / (tp + fn + fp)
        except ZeroDivisionError:
            return float("NaN")

    def get_mcc(self):
        tn, fn, fp, tp = self._sync()
        try:
            return (tp * tn - fp * fn) / math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
        except ZeroDivisionError:
            return float("NaN")
