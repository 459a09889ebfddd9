"""Minimal rocprofv3 target: the bench workload (B=256, 224x224, 16-bit NHWC resident) for a few steps.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/profile_target.py [steps] [dtype] [crop] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
crop = int(sys.argv[3]) if len(sys.argv) > 3 else 224
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
eng = PoseEngine(crop, crop, B, dtype)
for kv in os.environ.get("FLOPE_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
eng.load_state_dict(synthetic_state_dict(0))
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, crop, crop, 3, generator=g).to(torch.float16 if dtype == "f16" else torch.bfloat16).cuda()
R = torch.empty(B, 9, device="cuda")
fmt = 2 if dtype == "f16" else 1
for _ in range(steps):
    eng.forward_into(x, fmt, None, R)
torch.cuda.synchronize()
print("done", steps, dtype, crop, B)
