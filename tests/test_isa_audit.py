"""Static check of the built library (no GPU): no kernel may signal an s_barrier while one of its own LDS writes is still pending
(scripts/isa_audit.py, rule R1).  hipcc drops __syncthreads()'s `s_waitcnt lgkmcnt(0)` inside the loops of the kernels whose LDS-DMA waits
are inline asm; round 5's "counted wait" defect was exactly such a write (profiles/r06/dma_order.txt).  The audit has its own positive
control: the one instantiation that keeps the defect on purpose -- the GPU race screen's control -- must be flagged."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "robosat_amd", "librobosat_hip.so")


def _audit():
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "scripts", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def flagged():
    if not os.path.exists(SO):
        pytest.skip("librobosat_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump in this image")
    return _audit().audit_library(SO)


def test_no_kernel_publishes_an_lds_write_through_a_bare_barrier(flagged):
    mod = _audit()
    unexpected = {k: v for k, v in flagged.items() if not any(e in k for e in mod.EXPECTED)}
    assert not unexpected, "s_barrier with an LDS write pending (add rs_lds_writes_done() behind the write): {}".format(unexpected)


def test_the_audit_flags_the_kernel_that_keeps_the_defect(flagged):
    mod = _audit()
    assert any(any(e in k for e in mod.EXPECTED) for k in flagged), "the audit no longer sees the race screen's positive control: it proves nothing"


def test_the_dataflow_on_hand_written_streams():
    """The rule itself, on four tiny instruction streams (so that a change of the parser or the lattice shows up here, not on the GPU)."""
    audit = _audit().audit_kernel
    w, r, b = ("0", "ds_write_b32", "v1, v2"), ("1", "ds_read_b32", "v3, v1"), ("2", "s_barrier", "")
    wait0, wait1 = ("3", "s_waitcnt", "lgkmcnt(0)"), ("4", "s_waitcnt", "lgkmcnt(1)")
    assert audit([w, b]) == [("2", "0")]                      # write, bare barrier
    assert audit([w, wait0, b]) == []                         # waited for
    assert audit([w, r, wait1, b]) == []                      # one younger DS operation outstanding at most: the write has completed
    assert audit([w, r, r, wait1 + (), b]) == []              # (two younger reads, lgkmcnt(1): the write is older than both)
    assert audit([w, ("5", "s_waitcnt", "vmcnt(0)"), b]) == [("2", "0")]  # a vmcnt wait does not cover LDS
    # a loop: the write at the END of the body reaches the barrier at its top through the back edge (round 5's kernels)
    loop = [("label", "L0", ""), ("6", "s_waitcnt", "vmcnt(6)"), b, r, wait0, w, ("7", "s_cbranch_scc1", "L0"), ("8", "s_endpgm", "")]
    assert audit(loop) == [("2", "0")]
    loop_ok = [("label", "L0", ""), ("6", "s_waitcnt", "vmcnt(6)"), b, r, wait0, w, wait0, ("7", "s_cbranch_scc1", "L0"), ("8", "s_endpgm", "")]
    assert audit(loop_ok) == []
