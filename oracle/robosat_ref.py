"""Plain-PyTorch CPU (fp32) restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): this is the checker the
HIP path is compared with; the product never imports it.

Every function cites the reference file:line it follows (paths are relative to
``/root/reference``).  The ResNet-50 encoder is third-party arithmetic that is
NOT in the reference tree: torchvision 0.3.0 (``setup.py:38`` pins
``torchvision~=0.3``; call sites ``robosat/unet.py:15,94,122-130``); its
published architecture is restated in ``ResNet50`` below.

Pinning: ``tests/test_oracle_pin.py`` checks this file against the UNMODIFIED
reference modules (imported through ``oracle/refshim.py``) wherever
``/root/reference`` exists, and against the committed ``tests/golden/*.npz``
(which were produced by the reference itself) everywhere.
"""

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# torchvision 0.3.0 ResNet-50 (third-party; restated from its published definition)
# --------------------------------------------------------------------------------------


class Bottleneck(nn.Module):
    """torchvision 0.3.0 ``Bottleneck``: 1x1 -> 3x3 (carries the stride) -> 1x1 (x4), BN after each,
    residual add then ReLU.  Used through ``robosat/unet.py:127-130``."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet50(nn.Module):
    """torchvision 0.3.0 ``resnet50()``: layers [3, 4, 6, 3]; kept whole (incl. the unused avgpool/fc)
    because ``robosat/unet.py:94`` stores the whole module, so ``fc`` is in every checkpoint."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4),
            )
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)


# --------------------------------------------------------------------------------------
# robosat/unet.py
# --------------------------------------------------------------------------------------


class _Conv3x3Relu(nn.Module):
    """``ConvRelu`` (robosat/unet.py:18-44): 3x3 conv, padding 1, no bias, then ReLU."""

    def __init__(self, cin, cout):
        super().__init__()
        self.block = nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False)

    def forward(self, x):
        return F.relu(self.block(x))


class _UpConv(nn.Module):
    """``DecoderBlock`` (robosat/unet.py:47-73): nearest x2 upsample, then ``ConvRelu``."""

    def __init__(self, cin, cout):
        super().__init__()
        self.block = _Conv3x3Relu(cin, cout)

    def forward(self, x):
        return self.block(F.interpolate(x, scale_factor=2, mode="nearest"))


class UNetRef(nn.Module):
    """``UNet`` (robosat/unet.py:76-141).  Same sub-module names => same 329 state-dict keys.

    ``in_channels`` is an extension (the reference hard-codes 3, robosat/unet.py:92); with 3 it is the
    reference architecture exactly."""

    def __init__(self, num_classes, num_filters=32, in_channels=3):
        super().__init__()
        self.resnet = ResNet50()
        if in_channels != 3:
            self.resnet.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        nf = num_filters
        self.center = _UpConv(2048, nf * 8)  # unet.py:99
        self.dec0 = _UpConv(2048 + nf * 8, nf * 8)  # unet.py:101
        self.dec1 = _UpConv(1024 + nf * 8, nf * 8)  # unet.py:102
        self.dec2 = _UpConv(512 + nf * 8, nf * 2)  # unet.py:103
        self.dec3 = _UpConv(256 + nf * 2, nf * 2 * 2)  # unet.py:104
        self.dec4 = _UpConv(nf * 2 * 2, nf)  # unet.py:105
        self.dec5 = _Conv3x3Relu(nf, nf)  # unet.py:106
        self.final = nn.Conv2d(nf, num_classes, kernel_size=1)  # unet.py:108

    def forward(self, x, taps=None):
        """robosat/unet.py:110-141.  ``taps`` (optional dict) receives every intermediate feature map."""

        assert x.size(-1) % 32 == 0 and x.size(-2) % 32 == 0, "image resolution has to be divisible by 32 for resnet"
        r = self.resnet
        stem = r.relu(r.bn1(r.conv1(x)))  # unet.py:122-124
        enc0 = r.maxpool(stem)  # unet.py:125
        enc1 = r.layer1(enc0)  # unet.py:127
        enc2 = r.layer2(enc1)
        enc3 = r.layer3(enc2)
        enc4 = r.layer4(enc3)  # unet.py:130
        center = self.center(F.max_pool2d(enc4, kernel_size=2, stride=2))  # unet.py:132
        dec0 = self.dec0(torch.cat([enc4, center], dim=1))  # unet.py:134
        dec1 = self.dec1(torch.cat([enc3, dec0], dim=1))
        dec2 = self.dec2(torch.cat([enc2, dec1], dim=1))
        dec3 = self.dec3(torch.cat([enc1, dec2], dim=1))  # unet.py:137
        dec4 = self.dec4(dec3)
        dec5 = self.dec5(dec4)
        out = self.final(dec5)  # unet.py:141
        if taps is not None:
            taps.update(
                stem=stem, enc0=enc0, enc1=enc1, enc2=enc2, enc3=enc3, enc4=enc4, center=center,
                dec0=dec0, dec1=dec1, dec2=dec2, dec3=dec3, dec4=dec4, dec5=dec5, logits=out,
            )
        return out


def predict_probs(net, images):
    """Per-pixel class probabilities as ``rs predict`` computes them (robosat/tools/predict.py:84-87)."""

    with torch.no_grad():
        return F.softmax(net(images), dim=1)


def quantize_probs(foreground):
    """8-bit quantisation of the foreground probability (robosat/tools/predict.py:102-103).

    ``np.digitize`` bins are 1-based and ``p == 1.0`` gives 256, which wraps to 0 in uint8 (reference quirk)."""

    anchors = np.linspace(0, 1, 256)
    return np.digitize(foreground, anchors).astype(np.uint8)


# --------------------------------------------------------------------------------------
# robosat/losses.py
# --------------------------------------------------------------------------------------


def cross_entropy2d(logits, targets, weight=None):
    """``CrossEntropyLoss2d`` (robosat/losses.py:8-25): weighted NLL of log_softmax over dim 1,
    'mean' reduction = sum(w[t] * -logp[t]) / sum(w[t])."""

    return F.nll_loss(F.log_softmax(logits, dim=1), targets, weight=weight)


def focal2d(logits, targets, gamma=2, weight=None):
    """``FocalLoss2d`` (robosat/losses.py:28-50)."""

    penalty = (1 - F.softmax(logits, dim=1)) ** gamma
    return F.nll_loss(penalty * F.log_softmax(logits, dim=1), targets, weight=weight)


def onehot(targets, num_classes):
    """The ``zeros(...).scatter_(1, targets, 1)`` one-hot of robosat/losses.py:76,99."""

    n, h, w = targets.shape
    return torch.zeros(n, num_classes, h, w).scatter_(1, targets.view(n, 1, h, w), 1)


def miou2d(logits, targets, weight=None):
    """``mIoULoss2d`` (robosat/losses.py:53-83): soft IoU per (class, image), mean over both; the
    reference returns Python ``max(miou, nll)`` i.e. whichever branch is larger (line 83)."""

    n, c, h, w = logits.shape
    softs = F.softmax(logits, dim=1).permute(1, 0, 2, 3)
    masks = onehot(targets, c).permute(1, 0, 2, 3)
    inters = softs * masks
    unions = (softs + masks) - (softs * masks)
    miou = 1.0 - (inters.reshape(c, n, -1).sum(2) / unions.reshape(c, n, -1).sum(2)).mean()
    nll = F.nll_loss(F.log_softmax(logits, dim=1), targets, weight=weight)
    return max(miou, nll)


def lovasz2d(logits, targets):
    """``LovaszLoss2d`` (robosat/losses.py:86-119).

    Per image, over the flattened C*H*W vector with the one-hot mask as binary labels:
    hinge errors 1 - (2m-1)*x sorted descending, Jaccard-index deltas from two cumulative sums,
    dot(relu(errors), deltas); mean over the batch.  No softmax, no class weights (reference behaviour)."""

    n, c, h, w = logits.shape
    masks = onehot(targets, c).view(n, -1)
    flat = logits.reshape(n, -1)
    total = 0.0
    for i in range(n):
        m = masks[i]
        err = 1.0 - (m * 2 - 1) * flat[i]
        err_sorted, order = torch.sort(err, descending=True)
        lab = m[order]
        gts = lab.sum()
        inter = gts - lab.cumsum(0)
        union = gts + (1.0 - lab).cumsum(0)
        jac = 1.0 - inter / union
        if jac.numel() > 1:
            jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        total = total + torch.dot(F.relu(err_sorted), jac)
    return total / n


LOSSES = {"CrossEntropy": cross_entropy2d, "Focal": focal2d, "mIoU": miou2d, "Lovasz": lovasz2d}


# --------------------------------------------------------------------------------------
# robosat/metrics.py
# --------------------------------------------------------------------------------------


def confusion_counts(actual, predicted):
    """``Metrics.add`` (robosat/metrics.py:27-41) for ONE sample: returns (tn, fn, fp, tp) with the
    reference's own naming (its ``fn`` counts pred=1/actual=0, its ``fp`` counts pred=0/actual=1).

    ``actual`` [H,W] integer labels, ``predicted`` [C,H,W] scores."""

    pred = torch.argmax(predicted, 0).view(-1).float()
    act = actual.view(-1).float()
    q = pred / act
    tn = int(torch.isnan(q).sum())
    fn = int((q == float("inf")).sum())
    fp = int((q == 0).sum())
    tp = int((q == 1).sum())
    return tn, fn, fp, tp


def metric_scores(tn, fn, fp, tp):
    """``get_miou`` / ``get_fg_iou`` / ``get_mcc`` (robosat/metrics.py:43-84)."""

    try:
        miou = float(np.nanmean([tn / (tn + fn + fp), tp / (tp + fn + fp)]))
    except ZeroDivisionError:
        miou = float("nan")
    try:
        fg = tp / (tp + fn + fp)
    except ZeroDivisionError:
        fg = float("nan")
    try:
        mcc = (tp * tn - fp * fn) / math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
    except ZeroDivisionError:
        mcc = float("nan")
    return miou, fg, mcc
