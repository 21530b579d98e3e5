"""ctypes binding of ``librobosat_hip.so`` (C ABI declared in ``include/robosat_hip.h``).

There is NO fallback: if the library is missing or fails to load, importing the compute path raises.  PyTorch is
imported first on purpose -- it loads its bundled ``libamdhip64.so.7``; our library's NEEDED entry has the same
SONAME and therefore binds to the SAME HIP runtime, so torch's streams and device pointers are valid in our kernels.
"""

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_long, c_void_p

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ROBOSAT_HIP_LIB: another build of the same ABI, for A/B runs of a kernel change on one box -- never a different backend)
LIB_PATH = os.environ.get("ROBOSAT_HIP_LIB") or os.path.join(_HERE, "librobosat_hip.so")

RS_EINVAL = -22
ABI_VERSION = 21
RS_F32, RS_BF16 = 0, 1


class ConvDesc(ctypes.Structure):
    """``rs_conv_desc`` (include/robosat_hip.h)."""

    _fields_ = [
        ("N", c_int32), ("Hs", c_int32), ("Ws", c_int32), ("C1", c_int32), ("C2", c_int32), ("ups", c_int32),
        ("kh", c_int32), ("kw", c_int32), ("stride", c_int32), ("pad", c_int32), ("Ho", c_int32), ("Wo", c_int32),
        ("Cout", c_int32), ("relu", c_int32), ("stem", c_int32),
    ]


P = c_void_p  # device pointers and the stream travel as void*

# name -> (restype, argtypes); must list every symbol declared in include/robosat_hip.h
SIGNATURES = {
    "rs_abi_version": (c_int, []),
    "rs_conv2d_fwd": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, P]),
    "rs_conv2d_tile": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_tile_name": (c_char_p, [c_int]),
    "rs_pack_stem_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_nchw_to_nhwc4": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_maxpool2d_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_bn_fold": (c_int, [P, P, P, P, c_float, P, P, c_int, P]),
    "rs_final_conv1x1": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    # training path
    "rs_conv2d_wgrad_workspace_bytes": (c_long, [POINTER(ConvDesc)]),
    "rs_conv2d_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "rs_unpack_stem_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_pack_dgrad_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_bn_workspace_bytes": (c_long, [c_long, c_int]),
    "rs_bn_train_stats": (c_int, [P, c_long, c_int, c_float, c_float, P, P, P, P, P, P, P, P, P, P, P]),
    "rs_bn_apply": (c_int, [P, P, P, P, P, c_long, c_int, c_int, P]),
    "rs_bn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_long, c_int, P, P]),
    "rs_maxpool2d_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_upsample2x_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_final_conv1x1_bwd_workspace_bytes": (c_long, [c_int, c_int]),
    "rs_final_conv1x1_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    # losses / metrics
    "rs_nll_loss_workspace_bytes": (c_long, []),
    "rs_nll_loss_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P]),
    "rs_nll_loss_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "rs_miou_loss_workspace_bytes": (c_long, [c_int, c_int]),
    "rs_miou_loss_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "rs_miou_loss_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "rs_lovasz_workspace_bytes": (c_long, [c_int, c_int, c_int, c_int]),
    "rs_lovasz_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "rs_scale_by_scalar": (c_int, [P, P, P, c_long, P]),
    "rs_confusion_counts": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    # bf16 path (activations typed by a dtype code / bf16 entry points)
    "rs_conv2d_fwd_bf16": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, P]),
    "rs_conv2d_tile_bf16": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_config": (c_int, [POINTER(ConvDesc), c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "rs_conv2d_set_tuning": (c_int, [c_int, c_int]),
    "rs_set_knob": (c_int, [c_char_p, c_int]),
    "rs_get_knob": (c_int, [c_char_p, POINTER(c_int)]),
    "rs_conv2d_tile_name_bf16": (c_char_p, [c_int]),
    "rs_conv2d_wgrad_bf16_workspace_bytes": (c_long, [POINTER(ConvDesc)]),
    "rs_conv2d_wgrad_bf16_form": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_wgrad_form": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_wgrad_bf16_tile": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_wgrad_bf16": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "rs_cast_f32_to_bf16": (c_int, [P, P, c_long, P]),
    "rs_cast_bf16_to_f32_scaled": (c_int, [P, P, c_long, c_float, P]),
    "rs_cast_f32_to_bf16_scaled": (c_int, [P, P, c_long, c_float, P]),
    "rs_weight_prep_bf16": (c_int, [P, c_int, c_int, P]),
    "rs_weight_prep_f32": (c_int, [P, c_int, c_int, P]),
    "rs_pack_dgrad_weight_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_maxpool2d_fwd_dt": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_maxpool2d_bwd_dt": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_final_conv1x1_dt": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_final_conv1x1_bwd_dt": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "rs_bn_train_stats_dt": (c_int, [P, c_int, c_long, c_int, c_float, c_float, P, P, P, P, P, P, P, P, P, P, P]),
    "rs_bn_apply_dt": (c_int, [P, P, P, P, P, c_int, c_long, c_int, c_int, P]),
    "rs_bn_bwd_dt": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_long, c_int, P, P]),
    "rs_upsample2x_bwd_dt": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_conv2d_bnstats_rows": (c_long, [POINTER(ConvDesc)]),
    "rs_conv2d_bnstats_rows_dt": (c_long, [POINTER(ConvDesc), c_int]),
    "rs_conv2d_fwd_bnstats_dt": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P, P]),
    "rs_bn_finalize_stats": (c_int, [P, c_long, c_long, c_int, c_float, c_float, P, P, P, P, P, P, P, P, P, P, P]),
    "rs_pack_phase_weight_dt": (c_int, [P, P, c_int, c_int, c_int, P]),
    "rs_pack_s2_dgrad_phase_weight_dt": (c_int, [P, P, c_int, c_int, c_int, P]),
    "rs_conv2d_fwd_phase_dt": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P, P, P, P, P]),
    "rs_conv2d_phase_wino_ok": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_phase_wino_name": (c_char_p, [POINTER(ConvDesc)]),
    "rs_pack_wino_phase_weight": (c_int, [P, P, c_int, c_int, P]),
    "rs_conv2d_fwd_phase_wino": (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    "rs_conv2d_dgrad_phase_wino_ok": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_dgrad_phase_wino_name": (c_char_p, [POINTER(ConvDesc)]),
    "rs_pack_wino_dgrad_weight": (c_int, [P, P, c_int, c_int, P]),
    "rs_conv2d_dgrad_phase_wino": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_int, P]),
    "rs_conv2d_wino33_ok": (c_int, [POINTER(ConvDesc)]),
    "rs_conv2d_wino33_name": (c_char_p, [POINTER(ConvDesc)]),
    "rs_pack_wino33_weight": (c_int, [P, P, c_int, c_int, P]),
    "rs_conv2d_fwd_wino33": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P]),
    "rs_conv2d_wino33_head_ok": (c_int, [POINTER(ConvDesc), c_int]),
    "rs_conv2d_wino33_head_name": (c_char_p, []),
    "rs_conv2d_fwd_wino33_head": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_int, c_int, P, c_int, P, P, P]),
    "rs_bottleneck_tail_f32": (c_int, [P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, c_int, P]),
    "rs_conv1x1_wave_f32": (c_int, [P, P, P, P, P, c_int, P, c_long, c_int, c_int, P]),
    "rs_conv2d_wino33_stats_rows": (c_long, [POINTER(ConvDesc)]),
    "rs_conv2d_fwd_wino33_stats": (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    "rs_conv2d_dgrad_wino33": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, P, P]),
    "rs_pack_dgrad_phase_weight_dt": (c_int, [P, P, c_int, c_int, c_int, P]),
    "rs_combine_dgrad_phase_weight_dt": (c_int, [P, P, c_int, c_int, c_int, P]),
    "rs_conv2d_fwd_split_dt": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P, P, c_int, P]),
    "rs_cat_split_bwd_dt": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_scatter_add_stride2_dt": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_conv2d_dgrad_bnstats_dt": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P, P, P, P, P, P]),
    "rs_bn_bwd_from_partials_dt": (c_int, [P, P, P, P, P, P, P, P, P, c_long, c_int, c_long, c_int, P, P]),
    "rs_bn_apply_bits_dt": (c_int, [P, P, P, P, P, P, c_int, c_long, c_int, c_int, P]),
    "rs_conv2d_dgrad_bnstats_bits_dt": (c_int, [POINTER(ConvDesc), c_int, P, P, P, P, P, P, P, P, P, P]),
    "rs_nchw_to_nhwc4_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_pack_stem_weight_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "rs_stem_conv_fwd_bf16": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "rs_stem_conv_wgrad_bf16_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "rs_stem_conv_wgrad_bf16": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    # device-side predict input / output (N1)
    "rs_u8_to_nhwc4_norm": (c_int, [P, P, POINTER(c_float), POINTER(c_float), c_int, c_int, c_int, c_int, P]),
    "rs_final_conv1x1_quantize_dt": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "rs_final_conv1x1_argmax_dt": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    # callers either side of the network (N2-N4)
    "rs_confusion_matrix": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "rs_label_histogram_u8": (c_int, [P, c_long, P, P]),
    "rs_softvote_masks": (c_int, [P, P, P, P, c_int, c_long, c_int, P]),
    "rs_augment_tiles": (c_int, [P, P, P, P, POINTER(c_float), POINTER(c_float), P, P, c_int, c_int, c_int, P]),
}

_lib = None


def kernel_source_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h and the ABI header) in name order: identifies the TREE a measurement
    belongs to where there is no .git (the GPU box gets a snapshot).  ``scripts/pmc_traffic.py`` stamps the counter tables with
    it; ``bench.py`` prints ``roofline.traffic`` only when the stamp equals the digest of the sources it runs."""

    import hashlib

    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    names = sorted(n for n in os.listdir(src) if n.endswith((".hip", ".h")))
    for path in [os.path.join(src, n) for n in names] + [os.path.join(os.path.dirname(_HERE), "include", "robosat_hip.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()[:16]


def lib():
    """Returns the loaded library; raises (never falls back) if it is not there."""

    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "robosat_amd: {} is missing -- the MI355X kernels are not built and there is no CPU fallback. "
                "Build with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C robosat_amd/csrc`).".format(LIB_PATH)
            )
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == header and library disagree
            fn.restype, fn.argtypes = restype, argtypes
        got = handle.rs_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("robosat_amd: librobosat_hip.so ABI {} != expected {}".format(got, ABI_VERSION))
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        if rc == RS_EINVAL:
            raise ValueError("{}: invalid arguments (RS_EINVAL)".format(what))
        raise RuntimeError("{}: HIP error {}".format(what, rc))
