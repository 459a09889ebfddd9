"""Persistent stem kernel, diagnostic build: cycles per tile and workgroup in each phase.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_stem.py [streams]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, S = int(os.environ.get("B", 256)), int(os.environ.get("S", 224))
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
e = PoseEngine(S, S, B, "f16")
e.set_option("streams", streams)
stem_r = int(os.environ.get("STEM_R", 1))
e.set_option("stem_r", stem_r)
e.load_state_dict(synthetic_state_dict(0))
e.set_option("dbg", 64)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.uint64)       # up to 4 workgroups per CU
rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(30 * 1048576), C.c_size_t(buf.nbytes))
assert rc == 0
r = buf.reshape(-1, 8).astype(np.int64)
r_all = r
r = r[(r[:, 5] > 0) & (r[:, 6] > 0) & (r[:, 6] < 10**9)]
tiles = r[:, 5]
names = (["loads issue + MFMA phase (+ ReLU, vertical max, 8 LDS writes)", "barrier 1", "window write + pool + store", "barrier 2"] if stem_r else
         ["loads issue + acc init + MFMA phase", "barrier 1", "window write + ReLU -> Cs", "barrier 2", "pool + store"])
if stem_r:
    print(f"shader clock over the workgroups' lifetime: {np.median(r[:, 6] / np.maximum(r[:, 4], 1)) * 0.1:.3f} GHz; lifetime {np.median(r[:, 4]) / 100:.1f} us")
print(f"B={B} S={S} streams={streams}: {len(r)} workgroups, {int(tiles.sum())} tiles; cycles per tile and workgroup (median over workgroups)")
tot = 0
for i, n in enumerate(names):
    v = np.median(r[:, i] / tiles)
    tot += v
    print(f"  {n:40s} {v:8.0f}")
if not stem_r:
    print(f"  {'(of phase 0: tile decode + loads issue)':40s} {np.median(r[:, 7] / tiles):8.0f}")
if stem_r:
    ab = np.zeros(1024 * 4, dtype=np.uint64)
    rc = e.lib.flope_debug_read_ws(e.handle, ab.ctypes.data_as(C.c_void_p), C.c_size_t(30 * 1048576 + 16384 * 8), C.c_size_t(ab.nbytes))
    assert rc == 0
    ab = ab.reshape(-1, 4).astype(np.int64)
    nwg = int((ab[:, 0] > 0).sum())
    ab = ab[:nwg]
    t0_ = ab[:, 0].min()
    ex = (ab[:, 2] - t0_) / 100.0
    print(f"  wall inside the launch (100 MHz clock): first entry -> last exit {ex.max():.1f} us; entries spread over "
          f"{(ab[:, 0].max() - t0_) / 100:.1f} us; entry -> loop start median {np.median(ab[:, 1] - ab[:, 0]) / 100:.1f} us")
    print("  exit time percentiles (us): " + " ".join(f"p{q}={np.percentile(ex, q):.1f}" for q in (0, 10, 25, 50, 75, 90, 99, 100)))
    for x in range(8):
        m = (ab[:, 3] & 15) == x
        if m.any():
            print(f"    XCC {x}: {int(m.sum())} workgroups, exit median {np.median(ex[m]):.1f} max {ex[m].max():.1f} us; tiles {int(r[:nwg][m, 5].sum()) if len(r) >= nwg else -1}")
    late = np.argsort(ex)[-8:]
    print("  latest workgroups (block, XCC, tiles, exit us): " + " ".join(f"({b},{int(ab[b,3]&15)},{int(r[b,5]) if b < len(r) else -1},{ex[b]:.0f})" for b in late))
print(f"  {'sum':40s} {tot:8.0f}   (MFMA floor per tile and wave: 140 MFMAs x 16 = 2240)")
