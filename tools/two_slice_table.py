"""Counters of the PRODUCTION schedule (two batch slices on two streams) next to the one-stream profile pass:
    python tools/two_slice_table.py gpurun_out/r04x > profiles/r04_two_slice_counters.md
rocprofv3 --pmc collects per dispatch, so the two-slice run needs no trace domain; what it cannot give is a per-launch
duration (a kernel trace perturbs the interleave of the two streams: DESIGN.md 9.7c) -- so the columns are ratios of
counters of the same dispatches: MFMA-pipe busy cycles per wave-resident cycle (SQ_VALU_MFMA_BUSY_CYCLES / (4 x
SQ_WAVE_CYCLES): SQ_WAVE_CYCLES counts quad-cycles, the conv kernels keep one wave per SIMD), issue-stalled and parked
shares of the wave cycles (SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]


def fam(k):
    for key, name in (("conv_w4", "conv_w4"), ("conv_r4", "conv_r4"), ("conv_gstag", "conv_gstag"), ("conv_mfma", "conv_mfma"),
                      ("stem_pool", "stem_pool"), ("fc1", "fc1")):
        if key in k:
            return name
    return None


def load(sub):
    s = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
    for r in csv.DictReader(open(glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)[0])):
        f = fam(r["Kernel_Name"])
        if f:
            s[f][r["Counter_Name"]] += float(r["Counter_Value"]); n[f].add(r["Dispatch_Id"])
    return s, n


one, n1 = load("pmc_sq")
two, n2 = load("pmc_sq_two_slices")
print("| kernel family | schedule | dispatches | MFMA busy / wave-resident cycle | issue-stalled (WAIT_INST_ANY) | parked (WAIT_ANY) | wave quad-cycles per dispatch |")
print("|---|---|---|---|---|---|---|")
for k in ("conv_w4", "conv_r4", "stem_pool", "conv_gstag", "conv_mfma"):
    for tag, s, n in (("one stream (profile pass)", one, n1), ("two slices (production)", two, n2)):
        if k not in s:
            continue
        c = s[k]; w = c["SQ_WAVE_CYCLES"]
        print(f"| {k} | {tag} | {len(n[k])} | {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * w):.1%} | {c['SQ_WAIT_INST_ANY'] / w:.1%} | {c['SQ_WAIT_ANY'] / w:.1%} | {w / len(n[k]):.3g} |")
