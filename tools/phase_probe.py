"""Where a double step of conv_stag spends its cycles (diagnostic build only):
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/phase_probe.py [streams] [opts]
Four shader-clock stamps per double step D of the first tile of every workgroup, wave 0 of each wave group:
    t0 start (DMA issue + 16 ds_read_b128 + waits follow) | t1 before barrier 1 | t2 after barrier 1 (32 MFMAs follow) |
    t3 MFMAs issued, before barrier 2
Printed per conv launch, median over workgroups, in cycles: load = t1 - t0, bar1 = t2 - t1, mfma = t3 - t2, bar2 = t0' - t3."""
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
KD = 27
sd = synthetic_state_dict(0)
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
e = PoseEngine(S, S, B, "f16")
e.set_option("streams", streams)
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.load_state_dict(sd)
e.set_option("dbg", 128)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
which = [int(v) for v in os.environ.get("CONVS", "0,1,6,7,8,11,12,16,17").split(",")]
nwg = 1024
buf = np.zeros(nwg * 2 * KD * 4, dtype=np.uint64)
for i in which:
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576 + 65536), C.c_size_t(buf.nbytes))
    assert rc == 0
    rec = buf.reshape(nwg, 2, KD, 4).astype(np.int64)
    print(f"conv {i}  (streams={streams})")
    for g in (0, 1):
        r = rec[:, g]
        ok = (r[:, 0, 0] > 0) & (r[:, 1, 0] > r[:, 0, 0]) & (r[:, 1, 0] - r[:, 0, 0] < 10**7)
        if not ok.any():
            continue
        r = r[ok]
        nd = int(((r[:, :, 0] > 0) & (r[:, :, 3] >= r[:, :, 2])).all(0).sum())
        load = np.median(r[:, :nd, 1] - r[:, :nd, 0], 0)
        bar1 = np.median(r[:, :nd, 2] - r[:, :nd, 1], 0)
        mfma = np.median(r[:, :nd, 3] - r[:, :nd, 2], 0)
        bar2 = np.median(r[:, 1:nd, 0] - r[:, :nd - 1, 3], 0)
        tot = np.median(r[:, 1:nd, 0] - r[:, :nd - 1, 0], 0)
        print(f"  group {g}: {int(ok.sum())} workgroups, {nd} double steps stamped; medians per double step (cycles)")
        print("    D    " + " ".join(f"{d:5d}" for d in range(nd - 1)))
        print("    load " + " ".join(f"{v:5.0f}" for v in load[:nd - 1]))
        print("    bar1 " + " ".join(f"{v:5.0f}" for v in bar1[:nd - 1]))
        print("    mfma " + " ".join(f"{v:5.0f}" for v in mfma[:nd - 1]))
        print("    bar2 " + " ".join(f"{v:5.0f}" for v in bar2))
        print("    step " + " ".join(f"{v:5.0f}" for v in tot) + f"   mean {tot.mean():.0f}")
e.close()
