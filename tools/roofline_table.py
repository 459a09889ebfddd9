"""Per-kernel roofline table from one profiling run directory (rocprofv3 kernel trace + PMC passes):
    python tools/roofline_table.py gpurun_out/r47 > profiles/r01_v3_roofline.md
Columns: launches per forward, average duration, algorithmic GFLOP per launch (conv / FC MACs x 2), TFLOP/s,
fraction of the 2.5 PFLOP/s dense f16 MFMA peak, MFMA-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES over duration x 2.4 GHz x
1024 SIMDs), HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md correction) and GB/s vs 8 TB/s."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
FWD = 20.0   # forwards in the kernel-trace run (tools/profile_target.py 20), 4 in the PMC runs


def fam(k):
    if "conv_w4" in k: return "conv_w4<256x128> (layers 3-4, 3x3 s1, 4 waves; two of six carry the folded 1x1 shortcut)", 59.19 + 3.29 / 3
    if "conv_r4" in k: return "conv_r4<8 rows x 56> (layer 1, 4 waves)", 59.19
    if "conv_stag" in k and "Li64E" in k: return "conv_stag<8 rows x 64> (layer 1)", 59.19
    if "conv_stag" in k: return "conv_stag<256x128> (layers 2-4, 3x3 s1; three of nine carry the folded 1x1 shortcut)", 59.19 + 3.29 / 3
    if "conv_gstag" in k: return "conv_gstag<256x128,s2> (3x3 s2, Cin >= 128)", 29.59
    if "conv_s1r" in k: return "conv_s1r<4 rows x 28> (layer 2, 3x3 s1, 128 -> 128: 2.0.conv2 + folded shortcut, 2.1.conv1, 2.1.conv2; K split over wave pairs, weights in registers)", 60.29
    if "conv_s2r" in k: return "conv_s2r<4 rows x 28> (layer2.0.conv1: 3x3 s2, 64 -> 128; 8 waves, weights in registers)", 29.59
    if "conv_mfma" in k: return "conv_mfma<128x128,gather> (3x3 s2, Cin = 64)", 29.59
    if "stem_pool" in k: return "stem_pool (conv1 7x7 s2 + bn + relu + maxpool)", 60.42
    if "fc1" in k: return "fc1 (fc.0 + ReLU, f32 MFMA)", 0.537
    if "fc2" in k: return "fc2_procrustes", 0.0094
    if "avgpool" in k: return "avgpool", 0.0
    return None, 0


dur = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(glob.glob(f"{d}/kt/runc/*kernel_stats.csv")[0])):
    f, _ = fam(r["Name"])
    if f:
        dur[f][0] += float(r["TotalDurationNs"]); dur[f][1] += int(r["Calls"])


def pmc(sub, name):
    s = defaultdict(float); n = defaultdict(set)
    for r in csv.DictReader(open(glob.glob(f"{d}/{sub}/runc/*counter_collection.csv")[0])):
        if r["Counter_Name"] != name: continue
        f, _ = fam(r["Kernel_Name"])
        if f:
            s[f] += float(r["Counter_Value"]); n[f].add(r["Dispatch_Id"])
    return {k: s[k] / len(n[k]) for k in s}


fetch, write, mfma = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE"), pmc("pmc_sq", "SQ_VALU_MFMA_BUSY_CYCLES")
gf = {}
for k in list(dur):
    for probe in ("conv_w4", "conv_r4", "conv_stag_kernelIDF16_Li5ELi64E", "conv_stag_kernelIDF16_Li4ELi128E", "conv_gstag", "conv_s1r", "conv_s2r", "conv_mfma", "stem_pool", "fc1", "fc2", "avgpool"):
        f, g = fam(probe)
        if f == k: gf[k] = g
print("| kernel | launches / forward | avg µs | GFLOP / launch | TFLOP/s | of 2.5 PF | MFMA pipe busy | HBM MB / launch | HBM GB/s (of 8 TB/s) |")
print("|---|---|---|---|---|---|---|---|---|")
for k, (ns, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    us = ns / calls / 1e3
    tf = gf[k] / us * 1e3 if us else 0
    hb = 2 * 1024 * fetch.get(k, 0) + 1024 * write.get(k, 0)
    busy = mfma.get(k, 0) / (us * 1e-6 * 2.4e9 * 1024) if us else 0
    print(f"| {k} | {calls / FWD:.0f} | {us:.1f} | {gf[k]:.2f} | {tf:.0f} | {tf / 2500:.1%} | {busy:.1%} | {hb / 1e6:.0f} | {hb / us / 1e3:.0f} ({hb / us / 1e3 / 8000:.0%}) |")
