"""HBM traffic per launch from the two rocprofv3 --pmc passes of scripts/gpu_profile.sh -> profiles/pmc_traffic.json.

Units and corrections (MI355X_MICROARCH.md "HBM" section, re-checked here on kernels with known byte counts):
  * both counters are reported in KiB;
  * FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950: reads are DOUBLED (calibration: nchw_to_nhwc4 reads
    3 fp32 planes = 50.3 MB at bs 16 and reports 24 576 KiB = 25.2 MB);
  * WRITE_SIZE is exact for 16 B/lane stores (calibration: the stem's bn_apply<float> writes 134.2 MB at bs 8 and
    reports 131 072 KiB).
usage: python scripts/pmc_traffic.py gpurun_out/r02/pmc [profiles/pmc_traffic.json] [tag]
The JSON carries a ``_meta`` record (tag, date, git commit of the tree that was profiled, the commands): bench.py passes it
through as ``roofline.traffic_source`` so a reader can tell which kernels the counters belong to.
"""
import csv
import os
import json
import re
import sys
from collections import defaultdict

HALO_FORMS = {"1": "3x3", "2": "phase", "3": "dgrad4x4"}


def bench_name(k):
    k = k.replace("(anonymous namespace)::", "")
    # halo-once forms: a ninth template argument (1 = 3x3, 2 = phase, 3 = dgrad4x4); bench.py: conv_halo_bf16<form,BMxBN>
    m = re.search(r"conv_igemm_dma<__bf16, (\d+), (\d+), \d+, \d+, \d+, (?:true|false), \d+, ([123])(?:, \d+)*>", k)
    if m:
        return "conv_halo_bf16<{},{}x{}>".format(HALO_FORMS[m.group(3)], m.group(1), m.group(2))
    m = re.search(r"conv_igemm_dmaIDF16bLi(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ELb[01]ELi\d+ELi([123])E(?:Li\d+E)*E", k)
    if m:
        return "conv_halo_bf16<{},{}x{}>".format(HALO_FORMS[m.group(3)], m.group(1), m.group(2))
    m = re.search(r"conv_igemm_dma<(float|__bf16), (\d+), (\d+), \d+, \d+, (\d+), (true|false)(?:, (?:\d+|true|false))*>", k)
    if m:
        return "conv_igemm_{}<{}{}x{},r{}>".format("f32" if m.group(1) == "float" else "bf16",
                                                  "phase," if m.group(5) == "true" else "", m.group(2), m.group(3), m.group(4))
    # rocprofv3 leaves the __bf16 instantiations mangled: conv_igemm_dmaIDF16bLi128ELi128ELi2ELi2ELi64ELb0ELi2ELi0EE
    m = re.search(r"conv_igemm_dmaI(f|DF16b)Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELb([01])E", k)
    if m:
        return "conv_igemm_{}<{}{}x{},r{}>".format("f32" if m.group(1) == "f" else "bf16",
                                                  "phase," if m.group(5) == "1" else "", m.group(2), m.group(3), m.group(4))
    # per-instantiation names, spelled as bench.py reports them (ops.wgrad_kernel_name / conv_tile_name / rs_conv2d_phase_wino_name)
    m = re.search(r"conv_wino_f32_kernel<(\d+), (\d+), (\d+)(?:, (\d+))?(?:, (true|false))?>", k)
    if m:  # (last parameter, round 6: the data-gradient instantiations)
        return "conv_wino_f32<{},p{},{}x{}>".format("dgrad4x4" if m.group(5) == "true" else "phase", m.group(1), 16 * int(m.group(2)),
                                                    16 * int(m.group(4) or 2) * int(m.group(3)))
    m = re.search(r"conv_wino_f32_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", k)
    if m:
        return "conv_wino_f32<{},p{},{}x{}>".format("dgrad4x4" if m.group(5) == "1" else "phase", m.group(1), 16 * int(m.group(2)),
                                                    16 * int(m.group(4)) * int(m.group(3)))
    m = re.search(r"conv_wino33_f32_kernel<(\d+), (\d+)(?:, (true|false|\d+))?>", k)
    if m:  # (third parameter: 0 / absent = the layer alone, 1..3 = + self.final with one of its three output kinds, 4 = + BatchNorm partial sums, 5 = the data gradient)
        kind = "+stats" if m.group(3) == "4" else "+bwd" if m.group(3) == "5" else "" if m.group(3) in (None, "false", "0") else "+final"
        return "conv_wino_f32<3x3{},p8,{}x{}>".format(kind, 16 * int(m.group(1)), 16 * int(m.group(2)))
    m = re.search(r"conv_thin_bf16<(\d)>", k)
    if m:
        return "conv_thin_bf16<{}>".format(("3x3", "phase", "dgrad4x4")[int(m.group(1))])
    m = re.search(r"conv_wgrad_thin_bf16<(\d+), (\d)>", k)
    if m:
        return "conv_wgrad_thin_bf16<{}{}>".format(32 * int(m.group(1)), ",ups" if m.group(2) == "1" else "")
    if "conv_wgrad_phase4_bf16" in k:
        return "conv_wgrad_bf16<phase4,128x128>"
    m = re.search(r"conv_wgrad_bf16<(\d+), (\d+), \d+, \d+, \d+, (true|false)(?:, \d+)*>", k)  # (then: chunk buffers 2 / 3, the race control's variant)
    if m is None:  # (rocprofv3 on the GPU box prints mangled names)
        mm = re.search(r"conv_wgrad_bf16ILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ELb([01])E", k)
        if mm:
            return "conv_wgrad_bf16<{}{}x{}>".format("phase," if mm.group(3) == "1" else "", mm.group(1), mm.group(2))
    if m:  # (a two-launch layer -- "128x128+128x64" in the bench's name -- is looked up by its first tile)
        return "conv_wgrad_bf16<{}{}x{}>".format("phase," if m.group(3) == "true" else "", m.group(1), m.group(2))
    if "bottleneck_tail_f32" in k:  # (bottleneck_tail_f32.hip, round 6: <true> = the chained form, <false> = its first stage alone)
        return "conv1x1_wave_f32" if ("<false>" in k or "ILb0E" in k) else "bottleneck_tail_f32"
    if "stem_conv_f32" in k:  # (stem_f32.hip, round 6; <3> = RGB, <4> = four live bands: one report name)
        return "stem_conv_f32<128x64>"
    m = re.search(r"_ZN\d+_GLOBAL__N_1\d+([a-z_0-9]+?)I", k)
    if m:
        return m.group(1)
    m = re.search(r"conv_igemm_f32<(\d+), (\d+), \d+, \d+, 1>", k)
    if m:  # (rounds 1-5: the stem as an instantiation of the register-staged implicit-GEMM kernel)
        return "conv_igemm_f32<{}x{},stem>".format(m.group(1), m.group(2))
    m = re.search(r"(conv_wgrad_wino33_f32|conv_wgrad_wino_f32|conv_wgrad_f32_dma|conv_wgrad_f32)", k)
    if m:
        return m.group(1)
    if "conv1x1_ew_f32_kernel" in k:
        return "conv1x1_ew_f32<128x64,r64>"
    m = re.match(r"(?:void )?(\w+)", k)
    return m.group(1) if m else k[:40]


def collect(root, tag, counter):
    acc = defaultdict(lambda: [0.0, 0])
    path = "{}/{}_pmc_{}/p_counter_collection.csv".format(root, tag, counter)
    if not os.path.exists(path):
        return {}
    with open(path) as fp:
        for row in csv.DictReader(fp):
            if row["Counter_Name"] != counter:
                continue
            a = acc[bench_name(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}



if __name__ == "__main__":
    root = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_traffic.json"
    result, lines = {}, []
    for tag in ("predict", "trainbf16", "trainf32"):
        f, w = collect(root, tag, "FETCH_SIZE"), collect(root, tag, "WRITE_SIZE")
        lines.append("== {} (per launch, averaged over the launches of a kernel; MB = 1e6 bytes)".format(tag))
        tot_bytes, tot_launches, steps = 0.0, 0, 0
        for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[0] + w.get(k, (0, 0))[0]) * max(f.get(k, (0, 1))[1], 1)):
            rd = 2.0 * f.get(k, (0, 0))[0] * 1024
            wr = w.get(k, (0, 0))[0] * 1024
            n = (f.get(k) or w.get(k))[1]
            lines.append("{:42s} launches {:4d}  read {:9.2f} MB  write {:9.2f} MB  total {:9.2f} MB".format(k, n, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
            if tag == "predict" or k not in result:
                result[k] = round(rd + wr)
            tot_bytes += (rd + wr) * n
            tot_launches += n
            if k in ("stem_conv_f32<128x64>", "conv_igemm_f32<128x64,stem>", "stem_fwd_bf16_kernel", "stem_conv_bf16"):
                steps = n
        if steps:  # (the run's untimed steps are steps like the timed one: per-step figures = totals / launches of the stem kernel)
            lines.append("   => {} steps in the run: {:.1f} launches and {:.2f} GB of HBM traffic per step".format(steps, tot_launches / steps, tot_bytes / steps / 1e9))
    import datetime
    import subprocess

    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from robosat_amd._lib import kernel_source_digest

    result["_meta"] = {
        # digest of the kernel sources of the PROFILED tree: bench.py prints roofline.traffic only when it matches its own
        "csrc_digest": kernel_source_digest(),
        "profiled_csrc_digest": kernel_source_digest(),  # (what bench.py compares; written by this script only -- never edited by hand)
        "tag": sys.argv[3] if len(sys.argv) > 3 else None, "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
        "commit": commit,  # None on the GPU box (the snapshot has no .git): filled in when the file is copied to profiles/
        "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python bench.py --no-cpu-baseline --no-parity "
                   "--steps 1 --warmup 1 [--no-train-leg | --phase train --dtype bf16 --batch 32] (scripts/gpu_round.sh pmc)",
        "corrections": "KiB units; FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64); WRITE_SIZE as reported"}
    with open(out, "w") as fp:
        json.dump(result, fp, indent=1, sort_keys=True)
    print("\n".join(lines))
