"""Summarise rocprofv3 --pmc counter_collection CSVs per (kernel, grid size): averages per dispatch + derived ratios."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else "conv_"
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fp:
        for row in csv.DictReader(fp):
            name = row["Kernel_Name"]
            m = re.search(r"(\w+<[^>]*>|\w+_kernel\w*)", name.replace("(anonymous namespace)::", ""))
            k = m.group(1) if m else name[:40]
            if only not in k:
                continue
            key = (k, int(row["Grid_Size"]) // int(row["Workgroup_Size"]))
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[key][row["Counter_Name"]] += 1
            dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for key in sorted(acc, key=lambda k: -sum(dur[k])):
    a = {c: acc[key][c] / cnt[key][c] for c in acc[key]}
    d = sum(dur[key]) / len(dur[key])
    print("{} blocks={}  avg {:.1f} us  (n={})".format(key[0], key[1], d, max(cnt[key].values())))
    print("   " + "  ".join("{}={:.3g}".format(c, v) for c, v in sorted(a.items())))
    if "SQ_WAVE_CYCLES" in a:
        wc = a["SQ_WAVE_CYCLES"]
        print("   wave-cycle split: wait_any {:.1%}  wait_inst_any {:.1%}  active {:.1%}  wait_inst_lds {:.1%}".format(
            a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_WAIT_INST_LDS", 0) / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
        print("   mfma busy / (gui_active * 1024 SIMDs) = {:.1%}".format(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] * 1024)))
