"""Is the host ahead of the GPU in the bench loop?  Time to ENQUEUE n steps vs time until they have RUN."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
B, S = 256, 224
eng = PoseEngine(S, S, B, "f16"); eng.load_state_dict(synthetic_state_dict(0))
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    k, v = kv.split("="); eng.set_option(k, int(v))
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda"); xyz = torch.zeros(B, 3, device="cuda"); poses = torch.empty(B, 16, device="cuda")
for _ in range(10):
    eng.forward_poses_into(x, 2, xyz, True, poses, R)
torch.cuda.synchronize()
for n in (1, 2, 5, 50):
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_poses_into(x, 2, xyz, True, poses, R)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n:3d} steps: enqueued in {(t1 - t0) / n * 1e3:.3f} ms/step, done in {(t2 - t0) / n * 1e3:.3f} ms/step")
