"""Where the wall time of one overlapped step goes, from a `rocprofv3 --kernel-trace` CSV (measurement tool).

usage: python scripts/trace_gaps.py <p_kernel_trace.csv> [first-kernel-of-a-step substring, default "stem_fwd"] [step index]

Splits the trace into steps at every launch of the step's first kernel, takes the chosen step (default: the middle one) and the one before it, and prints per step:
wall time, the union of kernel intervals (GPU busy), idle time (no kernel resident), time with >= 2 queues busy, the busy
time of every queue, the kernels with the largest total duration, the largest idle gaps with the kernels either side, and
the copies (`__amd_rocclr_copyBuffer`) with their grid sizes."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "stem_fwd"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"], int(r["Grid_Size_X"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if first in r[3]]
if len(marks) < 3:
    raise SystemExit("fewer than 3 launches of %r in the trace" % first)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:70]


def analyse(seg, t0, t1):
    ev = []
    for s, e, q, n, g in seg:
        ev.append((s, 1, q))
        ev.append((e, -1, q))
    ev.sort()
    depth, busy, multi, last = 0, 0, 0, t0
    qd = defaultdict(int)
    for t, d, q in ev:
        if depth > 0:
            busy += t - last
        if sum(1 for v in qd.values() if v > 0) >= 2:
            multi += t - last
        last = t
        depth += d
        qd[q] += d
    return busy, multi


steps = []
for a, b in zip(marks[:-1], marks[1:]):
    steps.append((rows[a:b], rows[a][0], rows[b][0]))
print("walls of all steps (ms):", [round((t1 - t0) / 1e6, 2) for _, t0, t1 in steps])
pick = int(sys.argv[3]) if len(sys.argv) > 3 else len(steps) // 2
steps = steps[max(0, pick - 1):pick + 1]
for seg, t0, t1 in steps:
    busy, multi = analyse(seg, t0, t1)
    perq = defaultdict(int)
    for s, e, q, n, g in seg:
        perq[q] += e - s
    print("step: wall %.3f ms  busy(union) %.3f  idle %.3f  >=2 queues %.3f  launches %d  per-queue busy %s" % (
        (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, multi / 1e6, len(seg),
        {q: round(v / 1e6, 2) for q, v in sorted(perq.items())}))

seg, t0, t1 = steps[-1]
# idle gaps: walk the union
iv = sorted((s, e, n) for s, e, q, n, g in seg)
gaps = []
cur_end, cur_name = iv[0][1], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n))
    if e > cur_end:
        cur_end, cur_name = e, n
hist = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    hist[k][0] += 1
    hist[k][1] += g
print("idle gaps of the last step:", {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
print("largest gaps (us, after -> before):")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print("  %7.1f  %s -> %s" % (g / 1e3, short(a), short(b)))
after = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    after[short(a)][0] += 1
    after[short(a)][1] += g
print("idle time by the kernel that precedes the gap:")
for k, v in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %7.1f us in %3d gaps after %s" % (v[1] / 1e3, v[0], k))
tot = defaultdict(lambda: [0, 0])
for s, e, q, n, g in seg:
    tot[(q, short(n))][0] += 1
    tot[(q, short(n))][1] += e - s
print("kernels of the last step by total time (queue, name, launches, ms):")
for (q, n), v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  q%d %-72s %4d %8.3f" % (q, n, v[0], v[1] / 1e6))
print("copies (with the kernel launched before / after each on the same queue):")
byq = defaultdict(list)
for r in seg:
    byq[r[2]].append(r)
for q, rs in byq.items():
    for i, (s, e, _, n, g) in enumerate(rs):
        if "copyBuffer" in n or "fillBuffer" in n:
            prev = short(rs[i - 1][3])[:44] if i else "-"
            nxt = short(rs[i + 1][3])[:44] if i + 1 < len(rs) else "-"
            print("  q%d %-24s grid %9d  %6.1f us  at +%7.3f ms | after %s | before %s" % (q, short(n)[:24], g, (e - s) / 1e3, (s - t0) / 1e6, prev, nxt))
