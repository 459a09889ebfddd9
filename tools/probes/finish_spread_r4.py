"""Finish-time spread of the workgroups of a conv_r4 launch (diagnostic build): entry / exit on the 100 MHz clock per workgroup.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/probes/finish_spread_r4.py [streams]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, S = int(os.environ.get("B", 256)), int(os.environ.get("S", 224))
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
e = PoseEngine(S, S, B, "f16")
e.set_option("streams", streams)
e.load_state_dict(synthetic_state_dict(0))
e.set_option("dbg", 64)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(20):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
for i in range(4):
    ab = np.zeros(256 * 4, dtype=np.uint64)
    rc = e.lib.flope_debug_read_ws(e.handle, ab.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576 + 65536), C.c_size_t(ab.nbytes))
    assert rc == 0
    ab = ab.reshape(-1, 4).astype(np.int64)
    ab = ab[ab[:, 0] > 0]
    t0_ = ab[:, 0].min()
    ex = (ab[:, 1] - t0_) / 100.0
    print(f"conv {i}: {len(ab)} workgroups, tiles per workgroup {int(ab[:, 2].min())}..{int(ab[:, 2].max())}; exit (us) " +
          " ".join(f"p{q}={np.percentile(ex, q):.1f}" for q in (0, 10, 25, 50, 75, 90, 100)) +
          "; by XCC median " + " ".join(f"{np.median(ex[(ab[:, 3] & 15) == x_]):.0f}" for x_ in range(8) if ((ab[:, 3] & 15) == x_).any()))
e.close()
