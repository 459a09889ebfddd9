"""Diagnostic build: real-time clock at the start of each slice's stem launch in an un-profiled run (is the second slice late?).
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/probes/slice_lag.py"""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
B, S = 256, 224
e = PoseEngine(S, S, B, "f16"); e.load_state_dict(synthetic_state_dict(0))
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    k, v = kv.split("="); e.set_option(k, int(v))
e.set_option("dbg", 64)
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
lags = []
buf = np.zeros(1024, dtype=np.uint64)
for it in range(40):
    for _ in range(6):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(30 * 1048576 + 8192 * 8), C.c_size_t(buf.nbytes))
    assert rc == 0
    a, b = int(buf[96]), int(buf[160])
    lags.append((b - a) / 100.0)           # 100 MHz -> us
lags = np.array(lags)
print(f"second slice's stem starts {np.median(lags):.1f} us (median; {lags.min():.1f} .. {lags.max():.1f}) after the first slice's, last step of 40 un-profiled runs")
