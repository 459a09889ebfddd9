"""Detector conv kernels (yconv_body: the small-map / split-K path), diagnostic build: in-kernel time line of workgroup (0, 0) per launch.
    make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe_yolo.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd import _lib  # noqa: E402
from flope_amd.yolo import YoloSeg  # noqa: E402
from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict  # noqa: E402

y = YoloSeg(1080, 1920, 1280, "f16")
y.load_state_dict(synthetic_yolo_state_dict(0))
y.set_option("batch", 0)                                   # one launch per op, program order
img = synthetic_frame(3, 1080, 1920)
for _ in range(30):
    y.forward(img)
torch.cuda.synchronize()
lib = _lib.load()
lib.flope_ydbg_read.restype = C.c_int
buf = np.zeros(512 * 8, dtype=np.uint64)
lib.flope_ydbg_read(buf.ctypes.data_as(C.c_void_p), 512)   # reset
y.forward(img)
n = lib.flope_ydbg_read(buf.ctypes.data_as(C.c_void_p), 512)
r = buf[: n * 8].reshape(-1, 8).astype(np.int64)
print(f"{n} conv launches of one frame (one launch per op), workgroup (0,0): small-map kernel: cycles entry->loads issued | ->K loop done | ->combined | ->epilogue issued; tile kernel: entry->patch loads issued+written | ->barrier | ->K loop done | ->epilogue issued; in-kernel us, clock")
for i, d in enumerate(r):
    rt = (d[6] - d[5]) * 0.01
    cyc = d[4] - d[0]
    tile = "tile " if d[7] < 0 else "small"
    print(f"{i:3d} {tile} M={d[7] & 0xffffffff:6d} ksteps={(d[7] >> 32) & 0x7fffffff:3d}  {d[1]-d[0]:6d} {d[2]-d[1]:6d} {d[3]-d[2]:6d} {d[4]-d[3]:6d}   {rt:6.2f} us  {cyc / max(rt, 1e-3) / 1e3:5.2f} GHz")
y.close()
