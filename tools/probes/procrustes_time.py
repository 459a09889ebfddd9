"""Device time of the special Procrustes alone (one matrix per lane, 64-thread blocks) and of fc_rot + Procrustes, per launch."""
import ctypes as C
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd import _lib
lib = _lib.load()
for n in (64, 256, 4096):
    M = (torch.randn(n, 9, device="cuda") * 0.3).contiguous()
    R = torch.empty_like(M)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(50):
            lib.flope_procrustes(C.c_void_p(M.data_ptr()), C.c_void_p(R.data_ptr()), n, C.c_void_p(st))
        ev[1].record(); torch.cuda.synchronize()
    print(f"procrustes_kernel n={n}: {ev[0].elapsed_time(ev[1]) / 50 * 1e3:.1f} us per launch")
