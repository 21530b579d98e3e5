"""One screen of a bench.py JSON line: every leg's tiles/s, mean and min / median / max step time, roofline fractions,
parity and mIoU.  ``python scripts/bench_brief.py <bench.json>``"""

import json
import sys


def leg(name, d):
    sm = d.get("step_ms") or {}
    r = d.get("roofline") or {}
    extra = ""
    if r:
        extra = " | dominant {} {} frac {} ({} {}), all convs executed_frac {} roofline_frac {}".format(
            r.get("kernel"), r.get("bound"), r.get("frac"), r.get("achieved"), r.get("unit"),
            r.get("all_convs", {}).get("executed_frac"), r.get("all_convs", {}).get("roofline_frac"))
    par = d.get("parity")
    if par:
        extra += " | parity max|dp| {:.2e}".format(par["max_abs_vs_oracle"])
    stall = " STALLED x{} (slowest: step {})".format(sm.get("stalled_steps"), sm.get("slowest_step_index")) if sm.get("stalled_steps") else ""
    print("{:34s} {:9.1f} tiles/s  {:8.3f} ms/step (min {} med {} max {}, n {}){}{}".format(
        name, d["value"], d["ms_per_step"], sm.get("min"), sm.get("median"), sm.get("max"), sm.get("n"), stall, extra))


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    leg("headline " + d["config"].get("phase", "") + " " + d["dtype"], d)
    if "train" in d:
        leg("train bf16", d["train"])
    for k, v in (d.get("legs") or {}).items():
        leg(k, v)
    if "miou" in d:
        print("miou", d["miou"]["gpu"], "vs cpu_ref", d["miou"]["cpu_ref"])
    for k in ("cpu_baseline",):
        if k in d:
            print(k, d[k]["value"], d[k]["unit"], "cores", d[k]["cores"])
        if "train" in d and k in d["train"]:
            print("train", k, d["train"][k]["value"], d["train"][k]["unit"], "cores", d["train"][k]["cores"])


if __name__ == "__main__":
    main()
