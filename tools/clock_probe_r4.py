"""conv_r4 tile timeline (diagnostic build): cycles of the phases of each workgroup's third tile and the in-kernel clock.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_r4.py [streams] [opts]"""
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
sd = synthetic_state_dict(0)
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
e = PoseEngine(S, S, B, "f16")
e.set_option("streams", streams)
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.load_state_dict(sd)
e.set_option("dbg", 64 | int(os.environ.get("ABL", 0)))   # ABL=1: the next-tile burst re-reads the own (L2-warm) patch: timing only
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(50):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
buf = np.zeros(256 * 16, dtype=np.uint64)
names = ["sub 0-7", "wait+bar 1", "sub 8-16", "wait+bar 2", "sub 17", "epilogue"]
print(f"B={B} S={S} streams={streams}  conv_r4, third tile of every workgroup, median cycles (MFMA floor: 448 per sub-step, 8064 per tile)")
for i in range(4):
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576), C.c_size_t(buf.nbytes))
    assert rc == 0
    r = buf.reshape(-1, 16).astype(np.int64)
    ok = (r[:, 0] > 0) & (r[:, 6] > r[:, 0]) & (r[:, 6] - r[:, 0] < 10**7) & (r[:, 8] > r[:, 7])
    if not ok.any():
        print(i, "no records")
        continue
    d = r[ok]
    seg = [np.median(d[:, k + 1] - d[:, k]) for k in range(6)]
    clk = (d[:, 6] - d[:, 0]) / (d[:, 8] - d[:, 7]) * 0.1
    if d[:, 9].any():
        print(f"          sub 0 {np.median(d[:, 11] - d[:, 0]):.0f}  sub 1-7 {np.median(d[:, 1] - d[:, 11]):.0f} | sub 8 {np.median(d[:, 9] - d[:, 2]):.0f}  sub 9-12 {np.median(d[:, 10] - d[:, 9]):.0f}  sub 13-16 {np.median(d[:, 3] - d[:, 10]):.0f}")
    print(f"conv {i}: {int(ok.sum())} wg  " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, seg)) + f"  | tile {np.median(d[:, 6] - d[:, 0]):.0f}  clock {np.median(clk):.3f} GHz")
e.close()
