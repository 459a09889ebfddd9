"""Condense rocprofv3 CSV output (kernel stats and/or PMC counters) into a small text table.
    python tools/summarize_prof.py <dir> [<dir> ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.match(r"void (\w+)<(.*)>", name)
    if m:
        args = m.group(2).replace("__hip_bfloat16", "bf16").replace("_Float16", "f16").replace("__bf16", "bf16")
        return f"{m.group(1)}<{args}>"
    return name.replace("void ", "")


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
            print(f"== {f}")
            rows = list(csv.DictReader(open(f)))
            print(f"{'kernel':70s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'pct':>6s}")
            for r in rows:
                print(f"{short(r['Name'])[:70]:70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e3:10.1f} "
                      f"{float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            print(f"== {f}")
            agg = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k].add(r["Dispatch_Id"])
            for k, cs in agg.items():
                n = max(1, len(cnt[k]))
                print(f"{k[:90]}  dispatches={n}")
                print("    " + "  ".join(f"{c}={v/n:.4g}" for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main()
