"""Per-kernel MFMA-busy / LDS / wait summary from scripts/gpu_pmc_mfma.sh -> text table (profiles/r01/pmc_mfma_per_kernel.txt).

GRBM_GUI_ACTIVE is summed over the 8 XCDs (so /8 = shader cycles of the launch); SQ_VALU_MFMA_BUSY_CYCLES counts cycles over
all 1024 SIMDs; SQ_LDS_IDX_ACTIVE over the 256 CUs' LDS arrays; SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY are in quad-cycles.
usage: python scripts/pmc_mfma_summary.py gpurun_out/pmc_mfma_r01
"""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import bench_name  # noqa: E402  (same symbol -> bench-name mapping)

root = sys.argv[1]
for tag in ("predict", "trainbf16"):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    dur = defaultdict(float)
    with open("{}/{}/p_counter_collection.csv".format(root, tag)) as fp:
        for r in csv.DictReader(fp):
            k = bench_name(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("== {} (sums over the launches of a kernel in the run; kernels by shader cycles)".format(tag))
    print("{:42s} {:>5s} {:>10s} {:>9s} {:>9s} {:>9s} {:>9s} {:>8s}".format(
        "kernel", "n", "us total", "GHz", "MFMA busy", "LDS act", "wave wait", "bank cf"))
    for k in sorted(acc, key=lambda k: -acc[k]["GRBM_GUI_ACTIVE"])[:24]:
        a = acc[k]
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0:
            continue
        print("{:42s} {:5d} {:10.1f} {:9.2f} {:8.1f}% {:8.1f}% {:8.1f}% {:7.2f}%".format(
            k, n[k], dur[k], cyc / dur[k] / 1e3 if dur[k] else 0.0, 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
            100 * a["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 100 * a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0),
            100 * a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1.0)))
