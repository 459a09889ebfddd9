"""Encoder GEMM kernel (tf_gemm_mfma), diagnostic build: in-kernel time line of workgroup 0 per launch (one layer of the cfg5 shape).
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_tf.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# run the bench (1 layer) in this process so that the stamps are ours
sys.argv = [sys.argv[0], "f16", "256", "257", "1"]
exec(open(os.path.join(ROOT, "tools", "bench_tf.py")).read())
from flope_amd import _lib  # noqa: E402
lib = _lib.load()
lib.flope_tfdbg_read.restype = C.c_int
buf = np.zeros(512 * 8, dtype=np.uint64)
n = lib.flope_tfdbg_read(buf.ctypes.data_as(C.c_void_p), 512)
r = buf[: n * 8].reshape(-1, 8).astype(np.int64)[-4:]
print("last four tf_gemm_mfma launches, workgroup 0 wave 0: cycles entry->first DMA issued | ->first chunk landed | ->K loop done | ->epilogue issued ; in-kernel us, clock")
for d in r:
    rt = (d[6] - d[5]) * 0.01
    print(f"K={d[7] & 0xffffffff:5d} N={d[7] >> 32:5d}  {d[1]-d[0]:6d} {d[2]-d[1]:6d} {d[3]-d[2]:6d} {d[4]-d[3]:6d}   {rt:6.2f} us  {(d[4]-d[0]) / max(rt, 1e-3) / 1e3:5.2f} GHz   MFMA floor of the K loop: {(d[7] & 0xffffffff) // 32 * 16 * 16} cycles")
