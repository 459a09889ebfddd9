"""Per-step GPU times of the bench loop in issue order (HIP events around every step), after a synchronize -- where the
timed region's mean differs from its median.   python tools/step_series.py [steps] [opts]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 256
e = PoseEngine(224, 224, B, "f16")
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.load_state_dict(synthetic_state_dict(0))
x = torch.rand(B, 224, 224, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
xyz = torch.zeros(B, 3, device="cuda")
poses = torch.empty(n, B, 16, device="cuda")
for _ in range(5):
    e.forward_poses_into(x, 2, xyz, True, poses[0], R)
for rep in range(3):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(n):
        e.forward_poses_into(x, 2, xyz, True, poses[i], R)
        ev[i + 1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(f"rep {rep}: wall {wall * 1e3:.3f} ms = {wall / n * 1e3:.4f} per step; GPU steps: " + " ".join(f"{m:.3f}" for m in ms))
    print(f"        host enqueue done at (ms): " + " ".join(f"{h * 1e3:.2f}" for h in host))
e.close()
