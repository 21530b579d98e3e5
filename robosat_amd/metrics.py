"""Segmentation metrics of the reference (``robosat/metrics.py``) with the counting done on the GPU, for any class count.

Same class / method names and -- for the binary models the reference supports -- the same numbers, bit for bit
(including its ``fn`` / ``fp`` naming, which is swapped relative to the usual meaning; harmless for mIoU / IoU / MCC).  The
reference adds one sample at a time and synchronises the device four times per sample (metrics.py:38-41); here
``add`` / ``add_batch`` only enqueue a counting kernel and the single device->host copy happens when a score is requested.

The reference's own ``Todo`` (metrics.py:87-88) is the multi-class case: its four counters come from the ratio
``argmax / actual``, which is only meaningful for labels {0, 1}.  Here the kernel (``rs_confusion_matrix``) accumulates the
full C x C confusion matrix ``M[actual][predicted]``; for C = 2 the reference's counters are its four entries
(tn = M00, "fn" = M01, "fp" = M10, tp = M11) and every score below is evaluated with the reference's own expression, so
nothing changes for binary models.  For C > 2:

  * ``get_miou``   mean over classes of IoU_c = M_cc / (row_c + col_c - M_cc), NaNs (absent classes) skipped -- which for
                   C = 2 IS the reference's ``nanmean([tn/(tn+fn+fp), tp/(tp+fn+fp)])``;
  * ``get_fg_iou`` IoU of class 1, the class whose name the training log prints (tools/train.py:136);
  * ``get_mcc``    the multi-class Matthews coefficient R_K (Gorodkin 2004), which reduces to the binary formula at C = 2.
"""

import math

import numpy as np
import torch

from . import ops


class Metrics:
    """Tracking mean metrics over the classes ``labels`` (label 0 = background)."""

    def __init__(self, labels):
        self.labels = labels
        self.num_classes = len(labels)
        assert 2 <= self.num_classes <= 8, "2..8 classes"
        self._counts = None  # int64 [C*C] on the device: M[actual][predicted]
        self._host = np.zeros(self.num_classes * self.num_classes, dtype=np.int64)

    def _buf(self, device):
        if self._counts is None:
            self._counts = torch.zeros(self.num_classes * self.num_classes, device=device, dtype=torch.int64)
        return self._counts

    def add(self, actual, predicted):
        """One observation: ``actual`` [H,W] labels, ``predicted`` [C,H,W] scores (reference signature)."""

        self.add_batch(actual.unsqueeze(0), predicted.unsqueeze(0))

    def add_batch(self, actual, predicted):
        """A whole batch: ``actual`` [N,H,W] int64, ``predicted`` [N,C,H,W] float32."""

        assert predicted.size(1) == self.num_classes
        ops.confusion_matrix(predicted.detach().float().contiguous(), actual.contiguous(), self._buf(predicted.device))

    def _sync(self):
        if self._counts is not None:
            self._host += self._counts.cpu().numpy()
            self._counts.zero_()
        return self._host

    def confusion_matrix(self):
        """int64 [C,C], rows = actual class, columns = predicted class."""

        return self._sync().reshape(self.num_classes, self.num_classes).copy()

    # the reference's four counters (binary view: class 0 vs class 1 entries of the matrix)
    @property
    def tn(self):
        return int(self.confusion_matrix()[0, 0])

    @property
    def fn(self):
        return int(self.confusion_matrix()[0, 1])

    @property
    def fp(self):
        return int(self.confusion_matrix()[1, 0])

    @property
    def tp(self):
        return int(self.confusion_matrix()[1, 1])

    def get_class_ious(self):
        """IoU per class (NaN for a class that neither occurs nor is predicted)."""

        m = self.confusion_matrix()
        out = []
        for c in range(self.num_classes):
            inter = int(m[c, c])
            union = int(m[c, :].sum()) + int(m[:, c].sum()) - inter
            out.append(inter / union if union else float("NaN"))
        return out

    def get_miou(self):
        m = self.confusion_matrix()
        if self.num_classes == 2:  # the reference's expression, verbatim (metrics.py:49-53)
            tn, fn, fp, tp = (int(v) for v in m.reshape(-1))
            try:
                return float(np.nanmean([tn / (tn + fn + fp), tp / (tp + fn + fp)]))
            except ZeroDivisionError:
                return float("NaN")
        ious = self.get_class_ious()
        if all(math.isnan(v) for v in ious):
            return float("NaN")
        return float(np.nanmean(ious))

    def get_fg_iou(self):
        m = self.confusion_matrix()
        if self.num_classes == 2:
            tn, fn, fp, tp = (int(v) for v in m.reshape(-1))
            try:
                return tp / (tp + fn + fp)
            except ZeroDivisionError:
                return float("NaN")
        return self.get_class_ious()[1]

    def get_mcc(self):
        m = self.confusion_matrix()
        if self.num_classes == 2:
            tn, fn, fp, tp = (int(v) for v in m.reshape(-1))
            try:
                return (tp * tn - fp * fn) / math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
            except ZeroDivisionError:
                return float("NaN")
        # R_K: (c*s - sum_k p_k t_k) / sqrt((s^2 - sum p_k^2)(s^2 - sum t_k^2)) on exact Python integers
        t = [int(v) for v in m.sum(axis=1)]  # occurrences of each class
        p = [int(v) for v in m.sum(axis=0)]  # predictions of each class
        c, s = int(np.trace(m)), int(m.sum())
        try:
            return (c * s - sum(a * b for a, b in zip(p, t))) / math.sqrt((s * s - sum(a * a for a in p)) * (s * s - sum(b * b for b in t)))
        except ZeroDivisionError:
            return float("NaN")
