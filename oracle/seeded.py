"""Deterministic, non-trivial U-Net parameters and synthetic inputs (TEST INFRASTRUCTURE ONLY).

A freshly initialised network makes eval-mode BatchNorm an identity (running_mean 0, running_var 1,
gamma 1, beta 0) and would hide most BN bugs, so parity runs use these seeded values instead
(SURVEY.md section 8c/8d).  Values depend only on (key, shape, seed) -- not on construction order --
so the reference model, the oracle restatement and the HIP model all receive identical tensors.
"""

import zlib

import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def seeded_state_dict(template, seed=0, prefix=""):
    """Returns a state dict with the keys/shapes/dtypes of ``template`` (a state dict) and seeded values.

    conv / linear weights ~ N(0, 2/fan_in), biases ~ N(0, 0.1), BN gamma ~ U(0.5, 1.5) (U(0.1, 0.4) for the
    residual-branch ``bn3``), beta ~ N(0, 0.1), running_mean ~ N(0, 0.1), running_var ~ U(0.5, 1.5):
    activations stay O(1) through ~60 layers so the softmax is not saturated."""

    out = type(template)()
    for key, ref in template.items():
        name = key[len(prefix):] if prefix and key.startswith(prefix) else key
        g = _gen(name, seed)
        shape = tuple(ref.shape)
        if name.endswith("num_batches_tracked"):
            val = torch.zeros(shape, dtype=ref.dtype)
        elif name.endswith("running_mean"):
            val = torch.randn(shape, generator=g) * 0.1
        elif name.endswith("running_var"):
            val = torch.rand(shape, generator=g) + 0.5
        elif ref.dim() == 1 and name.endswith("bn3.weight"):
            # last BN of a bottleneck: small gamma keeps the residual stream O(1) over 16 blocks in eval mode
            val = torch.rand(shape, generator=g) * 0.3 + 0.1
        elif ref.dim() == 1 and name.endswith("weight"):  # BN gamma
            val = torch.rand(shape, generator=g) + 0.5
        elif ref.dim() == 1:  # BN beta, conv/linear bias
            val = torch.randn(shape, generator=g) * 0.1
        else:  # conv [O,I,kh,kw] / linear [O,I]
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            val = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        out[key] = val.to(ref.dtype)
    return out


def synthetic_images(n, c, h, w, seed=0):
    """Post-Normalize imagery is roughly N(0,1)."""

    g = torch.Generator()
    g.manual_seed(1000 + seed)
    return torch.randn(n, c, h, w, generator=g)


def synthetic_targets(n, num_classes, h, w, seed=0):
    """Blocky label maps (8x8 blocks) so classes form regions like real masks do."""

    g = torch.Generator()
    g.manual_seed(2000 + seed)
    bh, bw = max(h // 8, 1), max(w // 8, 1)
    coarse = torch.randint(0, num_classes, (n, bh, bw), generator=g)
    return coarse.repeat_interleave(h // bh, 1).repeat_interleave(w // bw, 2).contiguous()
