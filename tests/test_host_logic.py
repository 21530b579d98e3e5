"""CPU-only checks of the host side: checkpoint/key contract, weight layout, and that the C-ABI library loads and
exports every symbol declared in include/robosat_hip.h (no kernel is launched here)."""

import ctypes
import os
import re

import pytest
import torch

from oracle import robosat_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_contract_matches_reference_layout():
    from robosat_amd.unet import UNet

    ours, ref = UNet(2, pretrained=False), R.UNetRef(2)
    so, sr = ours.state_dict(), ref.state_dict()
    assert list(so.keys()) == list(sr.keys())  # same 329 keys, same order
    assert len(so) == 329
    for k in so:
        assert so[k].shape == sr[k].shape and so[k].dtype == sr[k].dtype, k
    # parameter ORDER is part of the contract: Adam's state is indexed by it (resume from a reference checkpoint)
    assert [n for n, _ in ours.named_parameters()] == [n for n, _ in ref.named_parameters()]
    assert sum(p.numel() for p in ours.parameters()) == 39390314


def test_checkpoint_roundtrip_both_directions(tmp_path):
    from robosat_amd.unet import UNet

    ours, ref = UNet(3, pretrained=False), R.UNetRef(3)
    # DataParallel-style "module." prefix as the reference writes it (tools/train.py:69,158)
    ck = {"epoch": 1, "state_dict": {"module." + k: v for k, v in ref.state_dict().items()}}
    path = str(tmp_path / "checkpoint-00001-of-00010.pth")
    torch.save(ck, path)
    sd = torch.load(path, map_location="cpu")["state_dict"]
    ours.load_state_dict({k[len("module."):]: v for k, v in sd.items()})
    for k, v in ours.state_dict().items():
        assert torch.equal(v, ref.state_dict()[k]), k
    ref2 = R.UNetRef(3)
    ref2.load_state_dict(ours.state_dict())  # and back
    assert torch.equal(ref2.state_dict()["dec0.block.block.weight"], ref.state_dict()["dec0.block.block.weight"])


def test_conv_weights_are_krsc_in_memory():
    from robosat_amd.unet import UNet

    net = UNet(2, pretrained=False)
    w = net.dec3.block.block.weight
    assert w.shape == (128, 320, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    k = net.dec3.block.block.krsc()
    assert k.shape == (128, 3, 3, 320) and k.is_contiguous() and k.data_ptr() == w.data_ptr()
    net.load_state_dict(R.UNetRef(2).state_dict())  # loading NCHW tensors keeps our layout
    assert net.dec3.block.block.weight.is_contiguous(memory_format=torch.channels_last)


def test_library_exports_every_declared_symbol():
    from robosat_amd import _lib

    header = open(os.path.join(ROOT, "include", "robosat_hip.h")).read()
    declared = set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("librobosat_hip.so not built (run __graft_entry__.build())")
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.rs_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    from robosat_amd.unet import UNet

    net = UNet(2, pretrained=False).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 64, 64))

def test_struct_layouts_match_the_c_header(tmp_path):
    """The ctypes mirrors of the two structs that cross the C ABI (rs_conv_desc, rs_wprep_item) have the size and field
    offsets gcc gives the declarations in include/robosat_hip.h (the header must also compile as plain C)."""
    import ctypes
    import subprocess

    from robosat_amd import _lib, ops

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"rs_conv_desc": _lib.ConvDesc, "rs_wprep_item": ops._WPrepItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "robosat_hip.h"', "int main(void) {"]
    for cname, mirror in structs.items():
        lines.append('  printf("{0} size %zu\\n", sizeof({0}));'.format(cname))
        for fname, _ in mirror._fields_:
            lines.append('  printf("{0} {1} %zu\\n", offsetof({0}, {1}));'.format(cname, fname))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, field, value = line.split()
        got[(cname, field)] = int(value)
    for cname, mirror in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(mirror), cname
        for fname, _ in mirror._fields_:
            assert got[(cname, fname)] == getattr(mirror, fname).offset, (cname, fname)
