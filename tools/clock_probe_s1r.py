"""conv_s1r band timeline (diagnostic build): cycles between the stamps of each workgroup's second band, wave 0.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_s1r.py [streams] [opts]"""
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
B = int(os.environ.get("B", 256))
x = torch.rand(B, 224, 224, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
e = PoseEngine(224, 224, B, "f16")
e.set_option("streams", streams)
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.load_state_dict(synthetic_state_dict(0))
e.set_option("dbg", 64)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
buf = np.zeros(256 * 16, dtype=np.uint64)
names = ["sub-tile A: 18 steps x 8 MFMAs (+ 6 pieces)", "A: swap write, residual issue, wait", "A: barrier", "A: swap read + epilogue (2 tiles)", "(gap)",
         "sub-tile B: 18 steps x 6 MFMAs", "B: swap write, residual issue, waits", "B: barrier", "B: swap read + epilogue"]
for ci in (6, 7):       # layer2.1.conv1, layer2.1.conv2 in the engine's conv list
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(ci * 1048576), C.c_size_t(buf.nbytes))
    assert rc == 0
    r = buf.reshape(-1, 16).astype(np.int64)
    ok = (r[:, 0] > 0) & (r[:, 9] > r[:, 0]) & (r[:, 9] - r[:, 0] < 10**7)
    if ok.sum() < 100:
        print("region", ci, "no records"); continue
    d = r[ok]
    print(f"region {ci}: {int(ok.sum())} workgroups; median cycles, second band, wave 0 (MFMA floor of a SIMD's pair: 4608 for A, 3456 for B)")
    for k, n in enumerate(names):
        print(f"  {n:52s} {np.median(d[:, k + 1] - d[:, k]):8.0f}")
    print(f"  {'band':52s} {np.median(d[:, 9] - d[:, 0]):8.0f}")
    life, pro = d[:, 12] - d[:, 10], d[:, 11] - d[:, 10]
    clk = life / np.maximum(d[:, 14] - d[:, 13], 1) * 0.1
    t0_ = d[:, 13].min()
    print(f"  workgroup lifetime {np.median(life):.0f} cycles = {np.median(d[:, 14] - d[:, 13]) / 100:.1f} us (clock {np.median(clk):.2f} GHz); entry -> loop start {np.median(pro):.0f} cycles; "
          f"entries spread over {(d[:, 13].max() - t0_) / 100:.1f} us; first entry -> last exit {(d[:, 14].max() - t0_) / 100:.1f} us")
    wb = np.zeros(256 * 16, dtype=np.uint64)
    rc = e.lib.flope_debug_read_ws(e.handle, wb.ctypes.data_as(C.c_void_p), C.c_size_t(ci * 1048576 + 8192 * 8), C.c_size_t(wb.nbytes))
    assert rc == 0
    w = wb.reshape(-1, 16).astype(np.int64)[ok]
    for nm, off, ref in (("A", 0, 0), ("B", 8, 5)):
        rel = w[:, off:off + 8] - d[:, ref:ref + 1]
        print(f"  arrival at barrier {nm} after wave 0's sub-tile start, waves 0..7 (median): " + " ".join(f"{np.median(rel[:, k]):.0f}" for k in range(8)))
e.close()
