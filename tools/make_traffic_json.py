"""HBM bytes per launch of every conv kernel family from the PMC passes of tools/collect_profiles.sh:
    python tools/make_traffic_json.py gpurun_out/r02 > profiles/r02_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes:
/opt/skills/guides/MI355X_MICROARCH.md §HBM; checked again for this round by tools/calib/fetch_calib.hip).  The digest of the
kernel sources ties the figures to a build: bench.py only reports `roofline.traffic` when it matches."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_digest  # noqa: E402

d = sys.argv[1]


def label(k):
    if "conv_w4" in k: return "conv_w4_kernel<256x128>"
    if "conv_r4" in k: return "conv_r4_kernel<8rows x56>"
    if "conv_stag" in k and "Li64E" in k: return "conv_stag_kernel<8rows x64>"
    if "conv_stag" in k: return "conv_stag_kernel<256x128>"
    if "conv_gstag" in k: return "conv_gstag_kernel<256x128,s2>"
    if "conv_s1r" in k: return "conv_s1r_kernel<4rows x28>"
    if "conv_s2r" in k: return "conv_s2r_kernel<4rows x28>"
    if "conv_mfma" in k: return "conv_mfma_kernel<128x128,gather,ring2>"
    if "stem_pool" in k: return "stem_pool_kernel"
    return None


def pmc(sub, name):
    s, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"] != name:
            continue
        k = label(r["Kernel_Name"])
        if k:
            s[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: (s[k] / len(n[k]), len(n[k])) for k in s}


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FLOPE_OPTS=streams=1, tools/profile_target.py 4 "
                 "(B=256, 224x224, f16); KiB units x1024; FETCH_SIZE doubled per MI355X_MICROARCH.md",
       "source_digest": source_digest(), "kernels": {}}
for k in fetch:
    f = int(2 * 1024 * fetch[k][0]); w = int(1024 * write.get(k, (0, 0))[0])
    out["kernels"][k] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w, "dispatches": fetch[k][1]}
print(json.dumps(out, indent=1))
