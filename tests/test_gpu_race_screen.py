"""Race screens for the kernels whose LDS pipelines are fed by LDS-DMA: a pipeline that reads a staged buffer too early, or
restages it too early, passes a parity test whenever the DMA happens to land first (cdna_hip_programming.md, "Read a staged buffer
one phase AFTER the wait that retires it").  So: the same launch twice on the same fresh data, many rounds, with a SECOND STREAM
keeping the machine busy, and the two results compared bit for bit.  A rare wrong tile shows up as one unequal round.

What the second stream runs matters (learnt the hard way in round 5, profiles/r05/wgrad_ring.txt): beside an HBM-bound elementwise
kernel the two counted-wait rings of the bf16 weight gradient were bit-reproducible over 300 rounds; beside a kernel that USES LDS on
the same CUs (a bf16 1x1 convolution, 29 KB per block) 42-96 % of their launches were not (they padded the tail of a split with
out-of-range LDS-DMA pieces, which retire ahead of older loads: the counted wait was satisfied before the last real chunk landed).  Hence:

  shipped kernels, LDS-using neighbour:   conv_wgrad_bf16 (two buffers, drained waits), conv_wgrad_f32_dma, conv_halo_bf16 (3x3 form),
                                          conv_thin_bf16 with its ReLU-mask patch (the one counted wait followed by a same-phase read)
  opt-in kernels, HBM-bound neighbour:    conv_wgrad_bf16<.., RING = 3> against the two-buffer pipeline, conv_wgrad_phase4_bf16 against itself
  the defect, as non-strict expected failures: the same two beside the LDS-using neighbour.  The ring of three drains its tail since the
                                          end of round 5 (0 of 150 unequal in scripts/flaky_ring4.py): it should XPASS; the four-offset
                                          kernel still pads and should fail until it drains too
"""

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
ROUNDS = 300
ROUNDS_LDS = 100

_STATE = {}


def _neighbour(lds):
    """Queues work on the second stream: 4 sweeps over a 256 MB buffer, or 6 launches of a bf16 1x1 convolution on the 128x64 tile."""
    from robosat_amd import ops

    if "side" not in _STATE:
        g = torch.Generator(device=DEV).manual_seed(99)
        _STATE["side"] = torch.cuda.Stream()
        _STATE["buf"] = torch.ones(64 << 20, device=DEV)
        _STATE["nx"] = torch.randn(32, 64, 64, 256, device=DEV, generator=g).to(BF)
        _STATE["nw"] = (torch.randn(64, 1, 1, 256, device=DEV, generator=g) * 0.05).to(BF)
    with torch.cuda.stream(_STATE["side"]):
        if lds:
            for _ in range(6):
                ops.conv2d(_STATE["nx"], _STATE["nw"])
        else:
            for _ in range(4):
                _STATE["buf"].mul_(1.0001)


def _twice(make, fn, rounds, lds):
    bad = []
    for r in range(rounds):
        args = make()
        torch.cuda.synchronize()
        _neighbour(lds)
        one = fn(*args)
        _neighbour(lds)
        two = fn(*args)
        if not torch.equal(one, two):
            bad.append(r)
    torch.cuda.synchronize()
    return bad


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


# ---- shipped kernels beside an LDS-using neighbour -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n,cin,cout,k,stride,h,w", [(16, 128, 128, 3, 2, 64, 64), (8, 256, 512, 1, 1, 32, 32), (6, 64, 256, 1, 1, 28, 30)])
def test_wgrad_bf16_shipped_pipeline_beside_an_lds_user(n, cin, cout, k, stride, h, w):
    from robosat_amd import ops

    assert ops.get_knob("wgrad_ring") == 2 and ops.get_knob("wgrad_phase4") == 0
    g = _gen(5)
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    bad = _twice(lambda: (torch.randn(n, ho, wo, cout, device=DEV, generator=g).to(BF), torch.randn(n, h, w, cin, device=DEV, generator=g).to(BF)),
                 lambda dy, x: ops.conv2d_wgrad(dy, x, k, k, stride=stride, pad=k // 2), ROUNDS_LDS, lds=True)
    assert not bad, bad


def test_wgrad_bf16_phase_form_shipped_beside_an_lds_user():
    from robosat_amd import ops

    g = _gen(6)
    n, c1, c2, cout, h, w = 2, 256, 64, 128, 32, 48  # dec3 at the eager-vs-graphed test's size
    bad = _twice(lambda: (torch.randn(n, 2 * h, 2 * w, cout, device=DEV, generator=g).to(BF), torch.randn(n, h, w, c1, device=DEV, generator=g).to(BF),
                          torch.randn(n, h, w, c2, device=DEV, generator=g).to(BF)),
                 lambda dz, a, b: ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1), ROUNDS_LDS, lds=True)
    assert not bad, bad


def test_wgrad_f32_lds_dma_beside_an_lds_user():
    from robosat_amd import ops

    g = _gen(7)
    n, cin, cout, h, w = 4, 128, 256, 30, 34
    bad = _twice(lambda: (torch.randn(n, h, w, cout, device=DEV, generator=g), torch.randn(n, h, w, cin, device=DEV, generator=g)),
                 lambda dy, x: ops.conv2d_wgrad(dy, x, 3, 3, pad=1), ROUNDS_LDS, lds=True)
    assert not bad, bad


def test_halo_3x3_bf16_beside_an_lds_user():
    from robosat_amd import ops

    g = _gen(8)
    n, c, h, w = 32, 128, 64, 64  # layer2's conv2 at the train leg's batch: the 512-pixel halo patch form
    wt = (torch.randn(c, 3, 3, c, device=DEV, generator=g) * 0.05).to(BF)
    assert "halo" in ops.conv_tile_name(ops.ConvDesc(n, h, w, c, 0, 0, 3, 3, 1, 1, h, w, c, 0, 0), True, False)
    bad = _twice(lambda: (torch.randn(n, h, w, c, device=DEV, generator=g).to(BF),), lambda x: ops.conv2d(x, wt, pad=1), ROUNDS_LDS, lds=True)
    assert not bad, bad


def test_conv_thin_bf16_with_relu_mask_beside_an_lds_user():
    from robosat_amd import ops

    g = _gen(9)
    n, c, s = 8, 32, 512  # dec5's data gradient
    wt = (torch.randn(c, 3, 3, c, device=DEV, generator=g) * 0.05).to(BF)
    assert "thin" in ops.conv_tile_name(ops.ConvDesc(n, s, s, c, 0, 0, 3, 3, 1, 1, s, s, c, 0, 0), True, False)
    bad = _twice(lambda: (torch.randn(n, s, s, c, device=DEV, generator=g).to(BF), torch.randn(n, s, s, c, device=DEV, generator=g).to(BF)),
                 lambda d, m: ops.conv2d(d, wt, pad=1, relu_mask=m), 40, lds=True)
    assert not bad, bad


# ---- the two opt-in rings: fine beside HBM-bound neighbours ... ------------------------------------------------------------------
@pytest.mark.xfail(reason="opt-in kernel rewritten after the last suite run; screened on one tile (profiles/r05/wgrad_ring.txt)", strict=False)
@pytest.mark.parametrize("n,cin,cout,k,stride,h,w", [(8, 256, 512, 1, 1, 32, 32), (4, 128, 128, 3, 2, 40, 36), (32, 256, 1024, 1, 1, 32, 32)])
def test_wgrad_bf16_ring_of_three_against_two_buffers_hbm_bound_neighbour(n, cin, cout, k, stride, h, w):
    from robosat_amd import ops

    g = _gen(5)
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    for r in range(ROUNDS):
        x = torch.randn(n, h, w, cin, device=DEV, generator=g).to(BF)
        dy = torch.randn(n, ho, wo, cout, device=DEV, generator=g).to(BF)
        _neighbour(False)
        with ops.knob("wgrad_ring", 3):
            new = ops.conv2d_wgrad(dy, x, k, k, stride=stride, pad=k // 2)
        old = ops.conv2d_wgrad(dy, x, k, k, stride=stride, pad=k // 2)
        assert torch.equal(new, old), "round {}: {} elements differ".format(r, int((new != old).sum()))
    torch.cuda.synchronize()


def test_wgrad_bf16_phase_four_offsets_is_deterministic_hbm_bound_neighbour():
    from robosat_amd import ops

    g = _gen(6)
    n, c1, c2, cout, h, w = 4, 256, 128, 256, 24, 20
    with ops.knob("wgrad_phase4", 1):
        bad = _twice(lambda: (torch.randn(n, 2 * h, 2 * w, cout, device=DEV, generator=g).to(BF), torch.randn(n, h, w, c1, device=DEV, generator=g).to(BF),
                              torch.randn(n, h, w, c2, device=DEV, generator=g).to(BF)),
                     lambda dz, a, b: ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1), ROUNDS, lds=False)
    assert not bad, bad


# ---- ... and the defect that keeps them opt-in --------------------------------------------------------------------------------------
@pytest.mark.xfail(reason="first version: padded tail, out-of-order retirement (profiles/r05/wgrad_ring.txt); the drained version is expected to XPASS -- made a plain test once a full suite has run with it", strict=False)
def test_wgrad_bf16_ring_of_three_beside_an_lds_user():
    from robosat_amd import ops

    g = _gen(5)
    with ops.knob("wgrad_ring", 3):
        bad = _twice(lambda: (torch.randn(16, 32, 32, 128, device=DEV, generator=g).to(BF), torch.randn(16, 64, 64, 128, device=DEV, generator=g).to(BF)),
                     lambda dy, x: ops.conv2d_wgrad(dy, x, 3, 3, stride=2, pad=1), 60, lds=True)
    assert not bad, bad


@pytest.mark.xfail(reason="conv_wgrad_phase4_bf16 still pads the tail of a split with out-of-range pieces (profiles/r05/wgrad_phase4.txt)", strict=False)
def test_wgrad_bf16_phase_four_offsets_beside_an_lds_user():
    from robosat_amd import ops

    g = _gen(6)
    n, c1, c2, cout, h, w = 8, 256, 64, 128, 64, 64
    with ops.knob("wgrad_phase4", 1):
        bad = _twice(lambda: (torch.randn(n, 2 * h, 2 * w, cout, device=DEV, generator=g).to(BF), torch.randn(n, h, w, c1, device=DEV, generator=g).to(BF),
                              torch.randn(n, h, w, c2, device=DEV, generator=g).to(BF)),
                     lambda dz, a, b: ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1), 40, lds=True)
    assert not bad, bad
