"""In-kernel clock and cycles per double step of every conv_stag launch (diagnostic build only):
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe.py [streams]
MI355X_MICROARCH.md "DVFS give-back" item 6: clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, stamped around the first tile's
main loop after >= 2 s of back-to-back forwards on random data; median over workgroups.  The MFMA floor of a double step is 1024
cycles per SIMD (2 waves x 32 MFMAs x 16 cycles): floor / measured = share of the loop the matrix pipe is issuing."""
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
e.set_option("dbg", 64)
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(50):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
names = [n for n, _, _ in e.launch_info(B)]
plan = e.describe_plan().splitlines()
buf = np.zeros(2048 * 2 * 4, dtype=np.uint64)
print(f"B={B} S={S} streams={streams}")
print(f"{'conv':4s} {'workgroups':>10s} {'GHz(med)':>9s} {'GHz(min-max)':>14s} {'cyc/loop':>10s} {'us/loop':>8s}")
for i in range(20):
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576), C.c_size_t(buf.nbytes))
    assert rc == 0
    rec = buf.reshape(-1, 4).astype(np.int64)
    ok = (rec[:, 1] > rec[:, 0]) & (rec[:, 3] > rec[:, 2]) & (rec[:, 1] - rec[:, 0] < 10**9)
    if not ok.any():
        continue
    d = rec[ok]
    clk = (d[:, 1] - d[:, 0]) / (d[:, 3] - d[:, 2]) * 0.1
    cyc = np.median(d[:, 1] - d[:, 0])
    print(f"{i:4d} {int(ok.sum()):10d} {np.median(clk):9.3f} {clk.min():6.2f}-{clk.max():5.2f} {cyc:10.0f} {np.median(d[:, 3] - d[:, 2]) / 100:8.2f}")
print("\n".join(l for l in plan if "conv_stag" in l))
e.close()
