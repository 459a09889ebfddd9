"""conv_w4 timeline per workgroup (diagnostic build): cycles before / inside / after the main loop and the in-kernel clock.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_w4.py [streams] [opts]"""
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
e.set_option("dbg", 64 | int(os.environ.get("ABL", 0)))   # ABL=1: no weight DMA in the loop, 2: no patch DMA (timing only), 256: per-double-step stamps of a FIRST body
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(50):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.uint64)
print(f"B={B} S={S} streams={streams}  conv_w4 workgroups: cycles entry->loop | loop | loop->exit, clock")
for i in (6, 7, 8, 11, 12, 13, 16, 17, 18):
    rc = e.lib.flope_debug_read_ws(e.handle, buf.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576), C.c_size_t(buf.nbytes))
    assert rc == 0
    r = buf.reshape(-1, 8).astype(np.int64)
    ok = (r[:, 1] > r[:, 0]) & (r[:, 2] > r[:, 1]) & (r[:, 3] > r[:, 2]) & (r[:, 5] > r[:, 4]) & (r[:, 3] - r[:, 0] < 10**8)
    if not ok.any():
        print(i, "no records")
        continue
    d = r[ok]
    clk = (d[:, 2] - d[:, 1]) / (d[:, 5] - d[:, 4]) * 0.1
    pre, loop, post = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    buf2 = np.zeros(1024 * 4, dtype=np.uint64)
    e.lib.flope_debug_read_ws(e.handle, buf2.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576 + 65536), C.c_size_t(buf2.nbytes))
    q = buf2.reshape(-1, 4).astype(np.int64)[: r.shape[0]][ok]
    seg = [np.median(q[:, 0] - d[:, 0]), np.median(q[:, 1] - q[:, 0]), np.median(q[:, 2] - q[:, 1]), np.median(q[:, 3] - q[:, 2]), np.median(d[:, 1] - q[:, 3])]
    print("          pre split: entry->first DMA %.0f | DMA issue + psrc %.0f | tables + acc init %.0f | wait + barrier %.0f | shortcut + first reads %.0f" % tuple(seg))
    buf3 = np.zeros(1024 * 8, dtype=np.uint64)
    e.lib.flope_debug_read_ws(e.handle, buf3.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576 + 131072), C.c_size_t(buf3.nbytes))
    bb = buf3.reshape(-1, 8).astype(np.int64)[: r.shape[0]][ok]
    two = (bb[:, 4] > bb[:, 0]) & (bb[:, 5] > bb[:, 4])
    if two.any():
        t = bb[two]
        print("          boundary (class walk, first): epilogue %.0f | pointer bumps %.0f | accumulator init %.0f | shortcut + fragment reload %.0f | second tile loop %.0f   (%d wg)"
              % (np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(t[:, 4] - t[:, 3]), np.median(t[:, 5] - t[:, 4]), int(two.sum())))
    buf4 = np.zeros(1024 * 32, dtype=np.uint32)
    e.lib.flope_debug_read_ws(e.handle, buf4.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576 + 196608), C.c_size_t(buf4.nbytes))
    dv = buf4.reshape(-1, 32)[: r.shape[0]][ok].astype(np.int64)
    if dv[:, 1].any():
        wait = (dv[:, 1:18:2] - dv[:, 0:18:2]) & 0xffffffff          # per double step: DMA wait + barrier
        work = (dv[:, 2:18:2] - dv[:, 1:17:2]) & 0xffffffff          # behind barrier D .. in front of wait D + 1: 64 MFMA (floor 1024)
        print("          %s body, per double step D=0..8: wait+barrier " % ("first" if int(os.environ.get("ABL", 0)) & 256 else "last") + " ".join("%4d" % v for v in np.median(wait, axis=0)))
        print("                                    B(D) + A(D+1) 64 MFMA " + " ".join("%4d" % v for v in np.median(work, axis=0)))
    if int(os.environ.get("ABL", 0)) & 512:
        buf5 = np.zeros(1024 * 16, dtype=np.uint32)
        e.lib.flope_debug_read_ws(e.handle, buf5.ctypes.data_as(C.c_void_p), C.c_size_t(i * 1048576 + 327680), C.c_size_t(buf5.nbytes))
        gw = buf5.reshape(-1, 16)[: r.shape[0]][ok].astype(np.int64)
        ref = dv[:, 13:14]                                           # behind barrier 6
        rel = (gw - ref) & 0xffffffff
        print("          D = 6 second sub-step, cycles since barrier 6 behind MFMA group 0..7: " + " ".join("%5d" % v for v in np.median(rel[:, :8], axis=0)))
        print("          D = 7 first  sub-step,                                  group 0..7: " + " ".join("%5d" % v for v in np.median(rel[:, 8:], axis=0)))
    print(f"conv {i:2d}: {int(ok.sum()):4d} wg  pre {np.median(pre):7.0f}  loop {np.median(loop):8.0f}  post {np.median(post):7.0f}  "
          f"clock {np.median(clk):.3f} GHz ({clk.min():.2f}-{clk.max():.2f})")
e.close()
