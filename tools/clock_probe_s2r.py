"""conv_s2r tile timeline (diagnostic build): cycles between the stamps of each workgroup's second tile, wave 0.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_s2r.py [streams] [opts]"""
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
idx = [i for i, (layer, k, _) in enumerate(e.launch_info(B)) if "layer2.0.conv1" in layer][0]
conv = [n for n in range(40)]
buf = np.zeros(512 * 16, dtype=np.uint64)
names = ["steps 0-8 (+ next patch pieces)", "steps 9-17", "wait for the pieces", "epilogue", "barrier"]
for ci in range(20):
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(ci * 1048576), C.c_size_t(buf.nbytes))
    assert rc == 0
    r = buf.reshape(-1, 16).astype(np.int64)
    ok = (r[:, 0] > 0) & (r[:, 5] > r[:, 0]) & (r[:, 5] - r[:, 0] < 10**7) & (r[:, 6] == 0)
    if ok.sum() < 100:
        continue
    d = r[ok]
    print(f"region {ci}: {int(ok.sum())} workgroups; median cycles, second tile, wave 0 (wave 0: 8 MFMAs per step = 128 cycles alone; the SIMD's two waves together 14 = 224)")
    for k, n in enumerate(names):
        print(f"  {n:40s} {np.median(d[:, k + 1] - d[:, k]):8.0f}")
    print(f"  {'tile':40s} {np.median(d[:, 5] - d[:, 0]):8.0f}")
e.close()
