"""CPU-only checks of the host side: checkpoint/key contract, weight layout, and that the C-ABI library loads and
exports every symbol declared in include/robosat_hip.h (no kernel is launched here)."""

import ctypes
import os
import sys
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

def test_dispatcher_choices_for_the_benchmark_layers():
    """rs_conv2d_config is pure host logic: pin the (tile, K-chunk row bytes) the dispatcher picks for representative
    layers of the two benchmark configurations, i.e. the measured heuristics of pick_tile / pick_rowb.
    A deliberate re-tune changes this table together with the measurement that justifies it."""
    from robosat_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("librobosat_hip.so not built (run __graft_entry__.build())")
    lib = _lib.lib()

    def cfg(n, hs, ws, c1, c2, ups, k, stride, pad, ho, wo, cout, es, phase=0):
        d = _lib.ConvDesc(n, hs, ws, c1, c2, ups, k, k, stride, pad, ho, wo, cout, 0, 0)
        tile, rowb = ctypes.c_int(0), ctypes.c_int(0)
        assert lib.rs_conv2d_config(ctypes.byref(d), es, phase, ctypes.byref(tile), ctypes.byref(rowb)) == 0
        name = (lib.rs_conv2d_tile_name_bf16 if es == 2 else lib.rs_conv2d_tile_name)(tile.value).decode()
        if "<" not in name:
            return ("halo" if tile.value == 8 else "thin"), rowb.value & 0xFFF  # (halo-once forms report their N tile in `rowb`)
        return name[name.index("<") + 1:-1], rowb.value

    fp32 = [  # predict bs 16, fp32 (es 4)
        ((16, 128, 128, 64, 0, 0, 1, 1, 0, 128, 128, 256, 4), ("128x64", 64)),       # layer1 conv3: K = 64, output bound
        ((16, 64, 64, 128, 0, 0, 1, 1, 0, 64, 64, 512, 4), ("128x128", 64)),         # layer2 conv3: short K
        ((16, 32, 32, 256, 0, 0, 1, 1, 0, 32, 32, 1024, 4), ("128x128", 64)),        # layer3 conv3: nk128 = 8
        ((16, 32, 32, 1024, 0, 0, 1, 1, 0, 32, 32, 256, 4), ("128x64", 128)),        # layer3 conv1: long K, small grid
        ((16, 16, 16, 512, 0, 0, 3, 1, 1, 16, 16, 512, 4), ("64x64", 128)),          # layer4 conv2
        ((16, 128, 128, 256, 64, 1, 3, 1, 1, 256, 256, 128, 4, 1), ("128x128", 64)), # dec3, phase form: large grid
        ((16, 16, 16, 2048, 256, 1, 3, 1, 1, 32, 32, 256, 4, 1), ("128x64", 128)),   # dec0, phase form
    ]
    bf16 = [  # train bs 32, bf16 (es 2)
        ((32, 128, 128, 64, 0, 0, 1, 1, 0, 128, 128, 256, 2), ("128x128", 64)),      # short K: occupancy
        ((32, 32, 32, 1024, 0, 0, 1, 1, 0, 32, 32, 256, 2), ("128x128", 128)),       # nk128 = 16, 512 blocks
        ((32, 128, 128, 64, 0, 0, 3, 1, 1, 128, 128, 64, 2), ("128x64", 128)),       # layer1 conv2: the 64-cout halo form ties / loses
        ((32, 64, 64, 128, 0, 0, 3, 1, 1, 64, 64, 128, 2), ("halo", 128)),           # layer2 conv2: halo-once 3x3, 512-pixel patches (-17 %)
        ((32, 32, 32, 256, 0, 0, 3, 1, 1, 32, 32, 256, 2), ("halo", 128)),           # layer3 conv2: 128 patches x 2 N tiles (-12 %)
        ((32, 16, 16, 512, 0, 0, 3, 1, 1, 16, 16, 512, 2), ("128x64", 128)),         # layer4 conv2: 16 px rows do not tile into 8 x 32 patches
        ((32, 32, 32, 1024, 256, 1, 3, 1, 1, 64, 64, 256, 2, 1), ("256x256", 128)),  # dec1, phase form: 8-wave tile (halo form: +10 %)
        ((32, 64, 64, 512, 256, 1, 3, 1, 1, 128, 128, 64, 2, 1), ("halo", 64)),      # dec2, phase form (-10 %)
        ((32, 128, 128, 256, 64, 1, 3, 1, 1, 256, 256, 128, 2, 1), ("halo", 128)),   # dec3, phase form (-13 %)
        ((32, 256, 256, 128, 0, 0, 4, 2, 1, 128, 128, 320, 2), ("halo", 128)),       # dec3 data gradient: ragged N (320) (-10 %)
        ((32, 64, 64, 256, 0, 0, 4, 2, 1, 32, 32, 1280, 2), ("halo", 128)),          # dec1 data gradient (-8 %)
        ((32, 16, 16, 2048, 256, 1, 3, 1, 1, 32, 32, 256, 2, 1), ("128x128", 128)),  # dec0: 16 px rows -> implicit GEMM
        ((32, 128, 128, 64, 0, 0, 4, 2, 1, 64, 64, 768, 2), ("256x256", 128)),       # dec2 data gradient: 16 short steps, the 8-wave tile wins
        ((32, 512, 512, 32, 0, 0, 3, 1, 1, 512, 512, 32, 2), ("thin", 64)),          # dec5: all-taps kernel (conv_thin_bf16.hip)
        ((32, 256, 256, 128, 0, 1, 3, 1, 1, 512, 512, 32, 2, 1), ("thin", 256)),     # dec4, phase form: all-taps kernel
        ((32, 512, 512, 32, 0, 0, 4, 2, 1, 256, 256, 128, 2), ("thin", 64)),         # dec4 data gradient: all-taps kernel
    ]
    for args, want in fp32 + bf16:
        assert cfg(*args) == want, (args, cfg(*args), want)


def test_profiler_symbols_map_to_the_bench_names():
    """bench.py looks `roofline.traffic` up in profiles/pmc_traffic.json by ITS kernel names; scripts/pmc_traffic.py derives
    them from the symbols rocprofv3 prints (demangled or not).  A kernel whose template list changes must keep mapping to the
    name the bench reports, or the traffic field silently becomes null."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    try:
        from pmc_traffic import bench_name
    finally:
        sys.path.pop(0)

    ns = "void (anonymous namespace)::"
    table = {
        ns + "conv_wino_f32_kernel<8, 8, 1, 4>((anonymous namespace)::WinoArgs)": "conv_wino_f32<phase,p8,128x64>",
        ns + "conv_wino_f32_kernel<8, 4, 2, 2>((anonymous namespace)::WinoArgs)": "conv_wino_f32<phase,p8,64x64>",
        ns + "conv_wino_f32_kernel<8, 8, 1, 2>((anonymous namespace)::WinoArgs)": "conv_wino_f32<phase,p8,128x32>",
        ns + "conv_wino_f32_kernel<4, 4, 2, 2>((anonymous namespace)::WinoArgs)": "conv_wino_f32<phase,p4,64x64>",
        ns + "conv_wino33_f32_kernel<4, 2, false>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3,p8,64x32>",
        ns + "conv_wino33_f32_kernel<8, 1, false>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3,p8,128x16>",
        ns + "conv_wino33_f32_kernel<4, 2, true>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3+final,p8,64x32>",
        ns + "conv_wino33_f32_kernel<4, 2, 0>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3,p8,64x32>",
        ns + "conv_wino33_f32_kernel<4, 2, 1>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3+final,p8,64x32>",
        ns + "conv_wino33_f32_kernel<4, 2, 3>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3+final,p8,64x32>",
        ns + "conv_wino33_f32_kernel<8, 1, 4>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3+stats,p8,128x16>",
        ns + "conv_wino33_f32_kernel<4, 2, 5>((anonymous namespace)::Wino33Args)": "conv_wino_f32<3x3+bwd,p8,64x32>",
        ns + "conv_wino_f32_kernel<8, 8, 1, 4, false>((anonymous namespace)::WinoArgs)": "conv_wino_f32<phase,p8,128x64>",
        ns + "conv_wino_f32_kernel<8, 4, 2, 2, true>((anonymous namespace)::WinoArgs)": "conv_wino_f32<dgrad4x4,p8,64x64>",
        "_ZN12_GLOBAL__N_120conv_wino_f32_kernelILi8ELi8ELi1ELi4ELb1EEEvNS_8WinoArgsE": "conv_wino_f32<dgrad4x4,p8,128x64>",
        ns + "conv_igemm_dma<float, 128, 128, 2, 2, 64, false, 0>(ConvArgsT<float>)": "conv_igemm_f32<128x128,r64>",
        ns + "conv_igemm_dma<float, 64, 64, 2, 2, 128, true, 0>(ConvArgsT<float>)": "conv_igemm_f32<phase,64x64,r128>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi128ELi128ELi2ELi2ELi64ELb0ELi2EEEv9ConvArgsTIT_E": "conv_igemm_bf16<128x128,r64>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi256ELi256ELi2ELi4ELi128ELb1ELi0EEEv9ConvArgsTIT_E": "conv_igemm_bf16<phase,256x256,r128>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi256ELi128ELi4ELi2ELi128ELb1ELi0ELi2EEEv9ConvArgsTIT_E": "conv_halo_bf16<phase,256x128>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi256ELi64ELi4ELi2ELi128ELb0ELi1ELi1EEEv9ConvArgsTIT_E": "conv_halo_bf16<3x3,256x64>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi512ELi128ELi4ELi2ELi64ELb1ELi0ELi2ELi0EEEv9ConvArgsTIT_E": "conv_halo_bf16<phase,512x128>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi512ELi128ELi4ELi2ELi64ELb0ELi0ELi3ELi0EEEv9ConvArgsTIT_E": "conv_halo_bf16<dgrad4x4,512x128>",
        "_ZN12_GLOBAL__N_114conv_igemm_dmaIDF16bLi128ELi128ELi2ELi2ELi64ELb0ELi2ELi0ELi0EEEv9ConvArgsTIT_E": "conv_igemm_bf16<128x128,r64>",
        ns + "conv_igemm_dma<__bf16, 256, 128, 4, 2, 128, false, 0, 3>(ConvArgsT<__bf16>)": "conv_halo_bf16<dgrad4x4,256x128>",
        ns + "conv_igemm_dma<__bf16, 256, 128, 4, 2, 128, false, 0, 3, 0>(ConvArgsT<__bf16>)": "conv_halo_bf16<dgrad4x4,256x128>",
        ns + "conv_igemm_dma<float, 128, 128, 2, 2, 64, false, 0, 0, 0>(ConvArgsT<float>)": "conv_igemm_f32<128x128,r64>",
        ns + "conv_igemm_f32<128, 64, 2, 2, 1>((anonymous namespace)::ConvArgs)": "conv_igemm_f32<128x64,stem>",
        ns + "stem_conv_f32<3>((anonymous namespace)::StemArgs)": "stem_conv_f32<128x64>",
        "_ZN12_GLOBAL__N_113stem_conv_f32ILi4EEEvNS_8StemArgsE": "stem_conv_f32<128x64>",
        "void (anonymous namespace)::bottleneck_tail_f32<true>((anonymous namespace)::TailArgs)": "bottleneck_tail_f32",
        "void (anonymous namespace)::bottleneck_tail_f32<false>((anonymous namespace)::TailArgs)": "conv1x1_wave_f32",
        "_ZN12_GLOBAL__N_119bottleneck_tail_f32ILb1EEEvNS_8TailArgsE": "bottleneck_tail_f32",
        "_ZN12_GLOBAL__N_119bottleneck_tail_f32ILb0EEEvNS_8TailArgsE": "conv1x1_wave_f32",
        ns + "conv_thin_bf16<1>((anonymous namespace)::ThinConvArgs)": "conv_thin_bf16<phase>",
        ns + "conv_wgrad_thin_bf16<4, 1>((anonymous namespace)::ThinArgs)": "conv_wgrad_thin_bf16<128,ups>",
        ns + "conv_wgrad_bf16<256, 128, 4, 2, 64, false>((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<256x128>",
        ns + "conv_wgrad_bf16<256, 128, 4, 2, 64, false, 3>((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<256x128>",
        ns + "conv_wgrad_bf16<128, 64, 2, 2, 64, true, 2>((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<phase,128x64>",
        ns + "conv_wgrad_bf16<128, 128, 2, 2, 64, true>((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<phase,128x128>",
        ns + "conv_wgrad_bf16<128, 64, 2, 2, 64, true, 2, 0>((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<phase,128x64>",
        ns + "conv_wgrad_bf16<256, 128, 4, 2, 64, false, 3, 0>((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<256x128>",
        "_ZN12_GLOBAL__N_115conv_wgrad_bf16ILi128ELi64ELi2ELi2ELi64ELb0ELi3ELi0EEEvNS_10WgradArgsBE": "conv_wgrad_bf16<128x64>",
        "(anonymous namespace)::conv_wgrad_phase4_bf16((anonymous namespace)::WgradArgsB)": "conv_wgrad_bf16<phase4,128x128>",
        "_ZN12_GLOBAL__N_122conv_wgrad_phase4_bf16ENS_10WgradArgsBE": "conv_wgrad_bf16<phase4,128x128>",
        ns + "conv_wgrad_f32_dma<128, 128, 2, 2, false, 32>(WgradArgs)": "conv_wgrad_f32_dma",
        ns + "conv_wgrad_wino_f32<2, 2, 8, 3>((anonymous namespace)::WinoWgradArgs)": "conv_wgrad_wino_f32",
        ns + "conv_wgrad_wino33_f32<2, 2>((anonymous namespace)::Wino33WgradArgs)": "conv_wgrad_wino33_f32",
        "_ZN12_GLOBAL__N_126bn_bwd_apply_stream_kernelIDF16bLb0ELb0EEEvPKT_S3_S3_PKfS5_PS1_S6_li": "bn_bwd_apply_stream_kernel",
        "(anonymous namespace)::reduce_lanes_kernel(float const*, float*, long, int, int)": "reduce_lanes_kernel",
    }
    for symbol, want in table.items():
        assert bench_name(symbol) == want, (symbol, bench_name(symbol), want)


def test_winograd_forms_are_chosen_by_geometry_pinned():
    """rs_conv2d_phase_wino_ok / _name, rs_conv2d_wino33_ok / _name, rs_conv2d_wino33_head_ok are pure host logic: pin which
    fp32 layers of the predict pass run which Winograd block shape, that a layer's FORM never depends on
    the batch size (a tile's probabilities must not depend on its batch neighbours) and that the one batch-dependent choice
    -- 128 x 64 vs 64 x 64 blocks, which accumulate identically -- follows the two-work-items-per-CU rule (256 CUs assumed
    when no device is visible)."""
    from robosat_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("librobosat_hip.so not built (run __graft_entry__.build())")
    lib = _lib.lib()

    def phase(n, hs, c1, c2, cout):
        d = _lib.ConvDesc(n, hs, hs, c1, c2, 1, 3, 3, 1, 1, 2 * hs, 2 * hs, cout, 1, 0)
        return lib.rs_conv2d_phase_wino_ok(ctypes.byref(d)), lib.rs_conv2d_phase_wino_name(ctypes.byref(d)).decode()

    # the DecoderBlocks of a 512^2 pass at bs 16 (ok: 1 = runs and should, 2 = runs but the generic kernel is the better choice)
    assert phase(16, 8, 2048, 0, 256) == (2, "conv_wino_f32<phase,p4,64x64>")       # center: < 8 tiles per side
    assert phase(16, 16, 2048, 256, 256) == (1, "conv_wino_f32<phase,p8,64x64>")    # dec0: 1 024 items -> the 64-tile block
    assert phase(16, 32, 1024, 256, 256) == (1, "conv_wino_f32<phase,p8,128x64>")   # dec1
    assert phase(16, 64, 512, 256, 64) == (1, "conv_wino_f32<phase,p8,128x64>")     # dec2
    assert phase(16, 128, 256, 64, 128) == (1, "conv_wino_f32<phase,p8,128x64>")    # dec3
    assert phase(16, 256, 128, 0, 32) == (1, "conv_wino_f32<phase,p8,128x32>")      # dec4: 32 couts
    for hs, c1, c2, cout in ((16, 2048, 256, 256), (64, 512, 256, 64), (256, 128, 0, 32), (8, 2048, 0, 256)):
        assert len({phase(n, hs, c1, c2, cout)[0] for n in (1, 2, 16, 64)}) == 1      # the form: geometry only
    assert phase(1, 64, 512, 256, 64)[1] == "conv_wino_f32<phase,p8,64x64>"         # one tile: too few items for the wide block
    assert phase(16, 2, 64, 0, 64)[0] == 0 and phase(16, 16, 24, 0, 64)[0] == 0      # tiny layer / Cin % 16 != 0: not runnable

    def w33(n, h, c, cout, stride=1):
        d = _lib.ConvDesc(n, h, h, c, 0, 0, 3, 3, stride, 1, h // stride, h // stride, cout, 1, 0)
        return lib.rs_conv2d_wino33_ok(ctypes.byref(d)), lib.rs_conv2d_wino33_name(ctypes.byref(d)).decode(), d

    for h, c in ((128, 64), (64, 128), (32, 256), (16, 512), (512, 32)):           # layer1-4 conv2, dec5
        assert w33(16, h, c, c)[:2] == (1, "conv_wino_f32<3x3,p8,64x32>") and w33(1, h, c, c)[0] == 1
    assert w33(16, 128, 64, 48)[:2] == (1, "conv_wino_f32<3x3,p8,128x16>")          # Cout % 32 != 0
    assert w33(16, 128, 128, 128, stride=2)[0] == 0 and w33(16, 8, 512, 512)[0] == 0 and w33(16, 64, 16, 64)[0] == 0
    # dec5 + final as one launch: the 32-cout layer only, up to 8 classes
    assert lib.rs_conv2d_wino33_head_ok(ctypes.byref(w33(16, 512, 32, 32)[2]), 2) == 1
    assert lib.rs_conv2d_wino33_head_ok(ctypes.byref(w33(1, 64, 32, 32)[2]), 8) == 1
    assert lib.rs_conv2d_wino33_head_ok(ctypes.byref(w33(16, 512, 32, 32)[2]), 9) == 0
    assert lib.rs_conv2d_wino33_head_ok(ctypes.byref(w33(16, 128, 64, 64)[2]), 2) == 0
    assert lib.rs_conv2d_wino33_head_name().decode() == "conv_wino_f32<3x3+final,p8,64x32>"


def test_pretrained_encoder_loads_a_legacy_torchvision_state_dict(tmp_path, monkeypatch):
    """resnet50-19c8e357.pth (torchvision 0.3.0's download, reference unet.py:94) has no ``num_batches_tracked`` keys:
    the encoder must load it like nn.BatchNorm2d's version shim does, find it through $ROBOSAT_RESNET50_WEIGHTS, and a
    missing file must be an error when the caller requires the ImageNet start (``rs train`` without a checkpoint)."""
    from robosat_amd.unet import UNet

    ref = R.UNetRef(2)
    legacy = {k[len("resnet."):]: v.clone() + 0.25 for k, v in ref.state_dict().items()
              if k.startswith("resnet.") and not k.endswith("num_batches_tracked")}
    assert len(legacy) == 267 and "fc.weight" in legacy
    path = tmp_path / "resnet50-19c8e357.pth"
    torch.save(legacy, str(path))
    monkeypatch.setenv("ROBOSAT_RESNET50_WEIGHTS", str(path))
    net = UNet(2, pretrained="require")
    for k, v in legacy.items():
        assert torch.equal(net.resnet.state_dict()[k], v), k
    assert int(net.resnet.bn1.num_batches_tracked) == 0
    net4 = UNet(2, pretrained=True, in_channels=4)  # 4-band stem: everything but conv1.weight comes from the file
    assert net4.resnet.conv1.weight.shape == (64, 4, 7, 7) and torch.equal(net4.resnet.layer1[0].conv1.weight, legacy["layer1.0.conv1.weight"])
    bad = dict(legacy)
    bad.pop("layer2.0.conv1.weight")
    with pytest.raises(RuntimeError, match="does not fit"):
        UNet(2, pretrained=False).load_pretrained_encoder(bad)
    monkeypatch.setenv("ROBOSAT_RESNET50_WEIGHTS", str(tmp_path / "absent.pth"))
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "nohome"))
    with pytest.raises(FileNotFoundError, match="pretrained = false"):
        UNet(2, pretrained="require")
    with pytest.warns(UserWarning):
        UNet(2, pretrained=True)


def test_checkpointed_optimizer_state_is_a_copy():
    """`rs train` saves the optimizer every epoch and keeps training: the saved state must not alias the live one (the fused
    Adam keeps its step counters where the parameters are; the checkpoint carries them as host tensors, as a stock Adam does)."""
    from robosat_amd.tools.train import _portable_optimizer_state

    p = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    opt = torch.optim.Adam(p, lr=1e-3, fused=True)
    for q in p:
        q.grad = torch.randn_like(q)
    opt.step()
    live = {id(st): st["step"] for st in opt.state.values()}
    saved = _portable_optimizer_state(opt)
    assert set(saved) == {"state", "param_groups"} and len(saved["state"]) == 2
    for st in saved["state"].values():
        assert st["step"].device.type == "cpu" and float(st["step"]) == 1.0
        assert set(st) == {"step", "exp_avg", "exp_avg_sq"}
    for st in opt.state.values():  # the live per-parameter dicts still hold the very same tensors
        assert st["step"] is live[id(st)]
    opt.step()  # and training goes on
    assert all(float(st["step"]) == 2.0 for st in opt.state.values())
    assert all(float(st["step"]) == 1.0 for st in saved["state"].values())
    fresh = torch.optim.Adam(p, lr=1e-3)  # a stock Adam resumes from it
    fresh.load_state_dict(saved)
    assert all(float(st["step"]) == 1.0 for st in fresh.state.values())


def test_mode_switches_and_invalidate_move_the_cache_generation():
    """Derived weight copies are keyed on (data_ptr, _version, generation): `train()` / `eval()` / `invalidate_caches()`
    move the generation, which is what catches optimizers that write parameters without bumping `_version` (fused Adam)."""
    from robosat_amd import unet

    net = unet.UNet(2, pretrained=False)
    g0 = unet._GENERATION[0]
    net.eval()
    g1 = unet._GENERATION[0]
    net.train()
    g2 = unet._GENERATION[0]
    net.invalidate_caches()
    g3 = unet._GENERATION[0]
    assert g0 < g1 < g2 < g3
    w = net.dec5.block.weight
    v = w._version
    opt = torch.optim.Adam([w], lr=1e-3, fused=True)
    w.grad = torch.randn_like(w)
    before = w.detach().clone()
    opt.step()
    assert not torch.equal(before, w.detach())
    if w._version == v:  # (the behaviour this guards against; a torch that bumps the counter makes the generation redundant, not wrong)
        assert unet._GENERATION[0] == g3  # nothing moved by itself: the next training forward / mode switch must


def test_decoded_tile_cache_holds_the_unaugmented_items(tmp_path):
    """The tile cache (filled through a DataLoader, with workers too) holds exactly the items of UnaugmentedTiles, in tile order."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from robosat_amd.datasets import DecodedTileCache, UnaugmentedTiles

    root = synth.make_dataset(str(tmp_path / "ds"), n_train=9, n_val=2, size=96, seed=3)
    img, lab = os.path.join(root, "training", "images"), os.path.join(root, "training", "labels")
    items = UnaugmentedTiles([img], lab, 64, draw=False)
    for workers in (0, 2):
        cache = DecodedTileCache([img], lab, 64, torch.device("cpu"), workers=workers)
        assert len(cache) == len(items) == 9 and tuple(cache.images.shape) == (9, 64, 64, 3) and cache.images.dtype == torch.uint8
        assert tuple(cache.masks.shape) == (9, 64, 64) and cache.masks.dtype == torch.uint8
        for i in range(len(items)):
            image, mask, code, tiles = items[i]
            assert code == 0 and torch.equal(cache.images[i], image) and torch.equal(cache.masks[i], mask)
            assert tuple(cache.tiles[i]) == tuple(tiles[0])


def test_band_layout_defaults_and_validation():
    """robosat_amd.bands: no keys = the reference's single RGB directory with the ImageNet statistics (train.py:246,262);
    extension keys name further sources; inconsistent layouts are errors, not silent 3-band batches."""
    from robosat_amd.bands import bands_from_config, split_per_source

    b = bands_from_config({"common": {"dataset": "/x"}})
    assert b.dirs == ["images"] and b.modes == ["RGB"] and b.channels == 3
    assert b.mean == [0.485, 0.456, 0.406] and b.std == [0.229, 0.224, 0.225]
    b = bands_from_config({"common": {"image_dirs": ["images", "ir"]}}, {"model": {"in_channels": 4}})
    assert b.modes == ["RGB", "L"] and b.channels == 4 and split_per_source(b, b.mean) == [[0.485, 0.456, 0.406], [0.449]]
    b = bands_from_config({"common": {"image_dirs": ["rgbi"], "image_modes": ["RGBA"], "mean": [0.1, 0.2, 0.3, 0.4], "std": [1, 1, 1, 2]}})
    assert b.channels == 4 and b.mean == [0.1, 0.2, 0.3, 0.4] and b.std == [1.0, 1.0, 1.0, 2.0]
    for bad, model in (({"common": {"image_dirs": ["a", "b"], "image_modes": ["RGB"]}}, None),            # modes vs dirs
                       ({"common": {"image_dirs": ["a", "b"], "image_modes": ["RGB", "RGB"]}}, None),      # 6 bands
                       ({"common": {"image_modes": ["CMYK"]}}, None),                                        # unknown mode
                       ({"common": {"mean": [0.5]}}, None),                                                  # 1 mean for 3 bands
                       ({"common": {}}, {"model": {"in_channels": 4}}),                                      # model wants 4, dataset has 3
                       ({"common": {"image_dirs": []}}, None)):
        with pytest.raises(ValueError):
            bands_from_config(bad, model)


def test_class_count_limits_are_stated_once():
    from robosat_amd.config import check_num_classes

    for tool, ok, bad in (("train", (2, 8), (1, 9)), ("predict", (2, 5), (1, 6))):
        for n in ok:
            check_num_classes(n, tool)
        for n in bad:
            with pytest.raises(ValueError):
                check_num_classes(n, tool)


def test_tools_rebuild_their_command_line_from_the_namespace():
    """The per-GPU relaunch must not replay the HOST program's sys.argv: the command line is rebuilt from the
    namespace main() received, and parses back to it."""
    import argparse

    from robosat_amd.tools import predict, train

    sub = argparse.ArgumentParser().add_subparsers()
    train.add_parser(sub)
    predict.add_parser(sub)
    parser = sub._name_parser_map
    ns = argparse.Namespace(model="m.toml", dataset="d.toml", checkpoint="c.pth", resume=True, workers=3)
    argv = train.argv_from_args(ns)
    back = parser["train"].parse_args(argv[1:])
    assert argv[0] == "train" and (back.model, back.dataset, back.checkpoint, back.resume, back.workers) == ("m.toml", "d.toml", "c.pth", True, 3)
    ns = argparse.Namespace(model="m.toml", dataset="d.toml", checkpoint=None, resume=False, workers=0)
    back = parser["train"].parse_args(train.argv_from_args(ns)[1:])
    assert back.checkpoint is None and back.resume is False
    ns = argparse.Namespace(batch_size=4, checkpoint="c.pth", overlap=16, tile_size=256, workers=2, tiles="t", probs="p", model="m", dataset="d",
                            extra_tiles=["ir"])
    back = parser["predict"].parse_args(predict.argv_from_args(ns)[1:])
    assert (back.tiles, back.probs, back.extra_tiles, back.batch_size, back.overlap) == ("t", "p", ["ir"], 4, 16)
    ns.extra_tiles = []
    back = parser["predict"].parse_args(predict.argv_from_args(ns)[1:])
    assert back.extra_tiles == [] and back.tiles == "t"


def test_nccl_needs_a_device_per_rank():
    from robosat_amd.launch import check_ranks_fit_devices

    check_ranks_fit_devices(2, 2, "nccl")
    check_ranks_fit_devices(2, 1, "gloo")
    with pytest.raises(RuntimeError, match="one device per rank"):
        check_ranks_fit_devices(2, 1, "nccl")


def test_resume_reasserts_this_runs_optimizer_flags():
    """`optimizer.load_state_dict` replaces the param groups with the SAVED ones, which silently dropped
    `[model] graph = true` (capturable) on --resume from a checkpoint saved without it, and forced it the other way round.
    Checkpoints carry stock-Adam flags; after the load the run's own configuration is put back."""
    import copy

    from robosat_amd.graph import capturable
    from robosat_amd.tools.train import _portable_optimizer_state, _reassert_optimizer_flags

    p = [torch.nn.Parameter(torch.randn(4, 3))]
    p[0].grad = torch.randn_like(p[0])
    for saved_capturable in (False, True):
        opt = torch.optim.Adam(p, lr=1e-3)  # (CPU: fused / capturable steps are GPU-only; the flag is set on the groups below)
        opt.step()
        for g in opt.param_groups:
            g["capturable"] = saved_capturable
        saved = _portable_optimizer_state(opt)
        assert all(g["capturable"] is False and g.get("fused") is None for g in saved["param_groups"])
        for want in (False, True):
            fresh = torch.optim.Adam(p, lr=1e-3, capturable=want)
            fresh.load_state_dict(copy.deepcopy(saved))  # (as read from a file: torch's load aliases the step tensors it is given)
            _reassert_optimizer_flags(fresh, want, torch.device("cpu"))
            assert capturable(fresh) == want
            assert all(st["step"].dtype == torch.float32 and float(st["step"]) == 1.0 for st in fresh.state.values())
            if not want:  # (a capturable Adam only steps on the device: tests/test_gpu_cli.py resumes with [model] graph = true)
                fresh.step()
                assert all(float(st["step"]) == 2.0 for st in fresh.state.values())


def test_rank_device_check_counts_local_ranks(monkeypatch):
    """what must fit the visible devices is the NODE-LOCAL rank count, not the job's world size; a launcher that
    isolates one GPU per rank (HIP_VISIBLE_DEVICES) is fine as well."""
    from robosat_amd.launch import check_ranks_fit_devices

    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "LOCAL_WORLD_SIZE", "LOCAL_RANK", "ROBOSAT_RANK_ISOLATED"):
        monkeypatch.delenv(k, raising=False)
    check_ranks_fit_devices(16, 8, "nccl", local_world=8)       # two nodes x 8 GPUs
    check_ranks_fit_devices(8, 8, "nccl")                                       # one node, env absent: world = local world
    check_ranks_fit_devices(2, 1, "gloo")                                       # the shared-device tests
    with pytest.raises(RuntimeError, match="one device per rank"):
        check_ranks_fit_devices(8, 4, "nccl", local_world=8)      # two local ranks per device
    with pytest.raises(RuntimeError, match="one device per rank"):
        check_ranks_fit_devices(2, 1, "nccl")                                   # one visible device, nothing isolates the ranks
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")
    with pytest.raises(RuntimeError):
        check_ranks_fit_devices(8, 1, "nccl")                                   # a single visible device alone says nothing
    monkeypatch.setenv("ROBOSAT_RANK_ISOLATED", "1")
    check_ranks_fit_devices(8, 1, "nccl")                                       # one GPU per rank, and the launcher says so
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    with pytest.raises(RuntimeError):
        check_ranks_fit_devices(8, 2, "nccl")


def test_bench_prints_a_compact_line_and_keeps_the_full_record(tmp_path):
    """the driver keeps an 8 KB tail of the bench's stdout; the line must fit it whole -- the train leg's
    value / ms_per_step / roofline included -- while per-kernel tables and step lists go to a side file."""
    import json

    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(ROOT, "profiles", "r03", "bench_default.json")) as fp:
        full = json.load(fp)  # a real full record (round 3, 12.7 KB)
    assert len(json.dumps(full)) > 8192
    full["train"]["reducer"] = {"backend": "nccl", "world": 1, "forced": True, "collectives_issued": 180, "wire": "fp32"}
    path = bench.write_full_record(full, str(tmp_path / "sub" / "bench_full.json"))
    assert path and json.load(open(path)) == full
    line = bench.compact_line(full, path)
    text = json.dumps(line)
    assert len(text) <= bench.COMPACT_LIMIT, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["config"]["workload"].startswith("rs predict") and "model" not in line["config"]
    t = line["train"]
    assert t["value"] == full["train"]["value"] and t["ms_per_step"] == full["train"]["ms_per_step"]
    assert t["step_ms"] == {k: full["train"]["step_ms"][k] for k in ("min", "median", "max", "n", "stalled_steps")}
    assert t["roofline"]["frac"] == full["train"]["roofline"]["frac"] and "all_convs" in t["roofline"] and "per_kernel" not in t["roofline"]
    assert t["reducer"]["collectives_issued"] == 180
    assert set(line["legs"]) == set(full["legs"]) and all(leg["value"] > 0 for leg in line["legs"].values())
    assert line["full_record"] == path


def test_bench_line_and_rank_environment_at_eight_ranks(tmp_path, monkeypatch):
    """`bench.py --gpus 8` under either launcher.  torchrun's environment (WORLD_SIZE / RANK / LOCAL_RANK) is
    what `launch.dist_env` reads and what stops bench.py from spawning ranks of its own; `spawn_ranks` writes the same
    variables for its children (exercised with eight real processes in tests/test_parallel_gloo.py); and the compact line of
    an 8-rank run still carries `train.reducer` -- backend, world, collectives issued -- so the driver can tell that RCCL saw
    eight ranks, and the new `conv3x3` group, within the 4 KB the driver's tail holds."""
    import json

    sys.path.insert(0, ROOT)
    import bench
    from robosat_amd import launch

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert not launch.under_launcher() and launch.dist_env() == (1, 0, 0)
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "5")
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert launch.under_launcher() and launch.dist_env() == (8, 5, 5)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    args = bench.parse()
    assert args.gpus == 8 and args.scaling == "weak" and args.grad_dtype == "fp32"

    with open(os.path.join(ROOT, "profiles", "r03", "bench_default.json")) as fp:
        full = json.load(fp)
    full["n_gpus"] = 8
    full["value"] = round(full["value"] * 8, 2)
    full["train"]["reducer"] = {"backend": "nccl", "world": 8, "forced": False, "collectives_issued": 6 * 39, "wire": "fp32"}
    for roof in (full["roofline"], full["train"]["roofline"]):
        roof["conv3x3"] = {"executed_tflops": 100.0, "frac": 0.64, "ms": 6.2, "launches": 21, "algorithmic_tflops": 300.0, "what": "x" * 200}
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(json.dumps(line)) <= bench.COMPACT_LIMIT
    assert line["n_gpus"] == 8 and line["train"]["reducer"] == full["train"]["reducer"]
    assert line["roofline"]["conv3x3"]["frac"] == 0.64 and "what" not in line["train"]["roofline"]["conv3x3"]
    assert line["train"]["config"]["parallelism"].startswith("dp")


def test_every_knob_is_documented_with_its_default():
    """robosat_amd/csrc/knobs.hip is the ONE table of measurement switches; INTEGRATION.md's knob table is what
    a maintainer reads.  Pin the two against each other -- every (name, environment seed) pair of the library appears in the
    document's table, in a row whose default column carries the value the library reports (rs_get_knob on a fresh table; no
    GPU involved) -- and that the Python side refuses names the library does not know."""
    import ctypes
    import re

    from robosat_amd import _lib

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(here, "robosat_amd", "csrc", "knobs.hip")).read()
    pairs = re.findall(r'\{"(\w+)", "(\w+)", &RsKnobs::(\w+), -?\d+, -?\d+\}', src)
    assert len(pairs) >= 20 and all(name == field for name, _, field in pairs)
    rows = [r for r in open(os.path.join(here, "INTEGRATION.md")).read().splitlines() if r.startswith("| `")]
    lib = _lib.lib()
    for name, env, _ in pairs:
        assert env not in os.environ, env + " is set: this test reads the defaults"
        row = [r for r in rows if "`{}`".format(name) in r.split("|")[1]]
        assert len(row) == 1, name
        cells = [c.strip() for c in row[0].split("|")[1:-1]]
        names = re.findall(r"`(\w+)`", cells[0])
        assert re.findall(r"`(\w+)`", cells[1])[names.index(name)] == env, (name, env)
        value = ctypes.c_int(12345)
        assert lib.rs_get_knob(name.encode(), ctypes.byref(value)) == 0
        assert int(cells[2].split(",")[names.index(name)]) == value.value, (name, cells[2], value.value)
    assert lib.rs_get_knob(b"no_such_knob", ctypes.byref(ctypes.c_int())) != 0
    assert lib.rs_set_knob(b"no_such_knob", 1) != 0
    # ... and values outside a knob's range (`conv_rowb` = 32 or `wgrad_ring` = 9 used to be taken silently)
    for name, bad in ((b"conv_rowb", 32), (b"wgrad_ring", 9), (b"wgrad_ring", 1), (b"conv_tile", 99), (b"wgrad_blocks", 0), (b"conv1x1_ew", 2)):
        before = ctypes.c_int(0)
        assert lib.rs_get_knob(name, ctypes.byref(before)) == 0
        assert lib.rs_set_knob(name, bad) != 0, (name, bad)
        after = ctypes.c_int(0)
        assert lib.rs_get_knob(name, ctypes.byref(after)) == 0 and after.value == before.value
    assert lib.rs_set_knob(b"conv_rowb", 64) == 0 and lib.rs_set_knob(b"conv_rowb", 0) == 0
