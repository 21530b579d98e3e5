"""Race screens for the kernels whose LDS pipelines are ordered by COUNTED waits (s_waitcnt vmcnt(N) + one barrier) rather than
by a drained queue: a pipeline that reads a staged buffer one step too early, or restages it one step too early, passes a
parity test whenever the DMA happens to land first (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that
retires it").  So: the same launch many times, on fresh data each round, with a second stream keeping the memory system busy,
and every round's result compared BIT FOR BIT with a run of the same arithmetic through the drained two-buffer pipeline (or,
where no such twin exists, with a second run of itself).  A rare wrong tile shows up as one unequal round.

  conv_wgrad_bf16<.., RING = 3>   vs  the two-buffer pipeline at the same split count      (bit-identical by construction)
  conv_wgrad_phase4_bf16          vs  itself                                                (deterministic: no atomics)
  conv_wgrad_f32_dma              vs  itself
  conv_halo_bf16 (3x3 form)       vs  itself
"""

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
ROUNDS = 300


def _noise(stream, buf):
    with torch.cuda.stream(stream):  # an HBM-bound neighbour: perturbs DMA return times
        for _ in range(4):
            buf.mul_(1.0001)


@pytest.mark.parametrize("n,cin,cout,k,stride,h,w", [(8, 256, 512, 1, 1, 32, 32), (4, 128, 128, 3, 2, 40, 36), (6, 64, 256, 1, 1, 28, 30),
                                                     (32, 256, 1024, 1, 1, 32, 32)])  # (last: a layer3 1x1 at the train leg's batch)
def test_wgrad_bf16_ring_of_three_against_two_buffers_many_rounds(n, cin, cout, k, stride, h, w):
    from robosat_amd import ops

    side = torch.cuda.Stream()
    buf = torch.ones(64 << 20, device=DEV)
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    g = torch.Generator(device=DEV).manual_seed(5)
    for r in range(ROUNDS):
        x = torch.randn(n, h, w, cin, device=DEV, generator=g).to(BF)
        dy = torch.randn(n, ho, wo, cout, device=DEV, generator=g).to(BF)
        _noise(side, buf)
        with ops.knob("wgrad_ring", 3):  # (opt-in)
            new = ops.conv2d_wgrad(dy, x, k, k, stride=stride, pad=k // 2)
        old = ops.conv2d_wgrad(dy, x, k, k, stride=stride, pad=k // 2)
        assert torch.equal(new, old), "round {}: {} elements differ".format(r, int((new != old).sum()))
    torch.cuda.synchronize()


def test_wgrad_bf16_phase_four_offsets_is_deterministic_many_rounds():
    from robosat_amd import ops

    side = torch.cuda.Stream()
    buf = torch.ones(64 << 20, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(6)
    n, c1, c2, cout, h, w = 4, 256, 128, 256, 24, 20
    for r in range(ROUNDS):
        a = torch.randn(n, h, w, c1, device=DEV, generator=g).to(BF)
        b = torch.randn(n, h, w, c2, device=DEV, generator=g).to(BF)
        dz = torch.randn(n, 2 * h, 2 * w, cout, device=DEV, generator=g).to(BF)
        with ops.knob("wgrad_phase4", 1):  # (opt-in kernel)
            _noise(side, buf)
            one = ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1)
            _noise(side, buf)
            two = ops.conv2d_wgrad(dz, a, 3, 3, src2=b, ups=1, pad=1)
        assert torch.equal(one, two), "round {}: {} elements differ".format(r, int((one != two).sum()))
    torch.cuda.synchronize()


def test_wgrad_f32_lds_dma_is_deterministic_many_rounds():
    from robosat_amd import ops

    side = torch.cuda.Stream()
    buf = torch.ones(64 << 20, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(7)
    n, cin, cout, h, w = 4, 128, 256, 30, 34
    for r in range(ROUNDS):
        x = torch.randn(n, h, w, cin, device=DEV, generator=g)
        dy = torch.randn(n, h, w, cout, device=DEV, generator=g)
        _noise(side, buf)
        one = ops.conv2d_wgrad(dy, x, 3, 3, pad=1)
        _noise(side, buf)
        two = ops.conv2d_wgrad(dy, x, 3, 3, pad=1)
        assert torch.equal(one, two), "round {}: {} elements differ".format(r, int((one != two).sum()))
    torch.cuda.synchronize()


def test_halo_3x3_bf16_is_deterministic_many_rounds():
    from robosat_amd import ops

    side = torch.cuda.Stream()
    buf = torch.ones(64 << 20, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(8)
    n, c, h, w = 32, 128, 64, 64  # layer2's conv2 at the train leg's batch: the 512-pixel halo patch form
    wt = (torch.randn(c, 3, 3, c, device=DEV, generator=g) * 0.05).to(BF)
    assert "halo" in ops.conv_tile_name(ops.ConvDesc(n, h, w, c, 0, 0, 3, 3, 1, 1, h, w, c, 0, 0), True, False)
    for r in range(ROUNDS):
        x = torch.randn(n, h, w, c, device=DEV, generator=g).to(BF)
        _noise(side, buf)
        one = ops.conv2d(x, wt, pad=1)
        _noise(side, buf)
        two = ops.conv2d(x, wt, pad=1)
        assert torch.equal(one, two), "round {}: {} elements differ".format(r, int((one != two).sum()))
    torch.cuda.synchronize()
