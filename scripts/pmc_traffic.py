"""HBM traffic per launch from the two rocprofv3 --pmc passes of scripts/gpu_profile.sh -> profiles/pmc_traffic.json.

Units and corrections (MI355X_MICROARCH.md "HBM" section, re-checked here on kernels with known byte counts):
  * both counters are reported in KiB;
  * FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950: reads are DOUBLED (calibration: nchw_to_nhwc4 reads
    3 fp32 planes = 50.3 MB at bs 16 and reports 24 576 KiB = 25.2 MB);
  * WRITE_SIZE is exact for 16 B/lane stores (calibration: the stem's bn_apply<float> writes 134.2 MB at bs 8 and
    reports 131 072 KiB).
usage: python scripts/pmc_traffic.py gpurun_out/profile_r01 [profiles/pmc_traffic.json]
"""
import csv
import os
import json
import re
import sys
from collections import defaultdict

def bench_name(k):
    k = k.replace("(anonymous namespace)::", "")
    m = re.search(r"conv_igemm_dma<(float|__bf16), (\d+), (\d+), \d+, \d+, (\d+), (true|false)(?:, (?:\d+|true|false))*>", k)
    if m:
        return "conv_igemm_{}<{}{}x{},r{}>".format("f32" if m.group(1) == "float" else "bf16",
                                                  "phase," if m.group(5) == "true" else "", m.group(2), m.group(3), m.group(4))
    # rocprofv3 leaves the __bf16 instantiations mangled: conv_igemm_dmaIDF16bLi128ELi128ELi2ELi2ELi64ELb0ELi2ELi0EE
    m = re.search(r"conv_igemm_dmaI(f|DF16b)Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELb([01])E", k)
    if m:
        return "conv_igemm_{}<{}{}x{},r{}>".format("f32" if m.group(1) == "f" else "bf16",
                                                  "phase," if m.group(5) == "1" else "", m.group(2), m.group(3), m.group(4))
    m = re.search(r"_ZN\d+_GLOBAL__N_1\d+([a-z_0-9]+?)I", k)
    if m:
        return m.group(1)
    m = re.search(r"conv_igemm_f32<(\d+), (\d+), \d+, \d+, 1>", k)
    if m:
        return "conv_igemm_f32<{}x{},stem>".format(m.group(1), m.group(2))
    m = re.search(r"conv_wgrad_bf16<[^>]*, true>", k)
    if m:
        return "conv_wgrad_bf16<phase>"
    m = re.search(r"(conv_wgrad_thin_bf16|conv_wgrad_bf16|conv_wgrad_f32)", k)
    if m:
        return m.group(1)
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
        for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0, 0))[0] + w.get(k, (0, 0))[0]) * max(f.get(k, (0, 1))[1], 1)):
            rd = 2.0 * f.get(k, (0, 0))[0] * 1024
            wr = w.get(k, (0, 0))[0] * 1024
            n = (f.get(k) or w.get(k))[1]
            lines.append("{:42s} launches {:4d}  read {:9.2f} MB  write {:9.2f} MB  total {:9.2f} MB".format(k, n, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
            if tag == "predict" or k not in result:
                result[k] = round(rd + wr)
    with open(out, "w") as fp:
        json.dump(result, fp, indent=1, sort_keys=True)
    print("\n".join(lines))
