"""Secondary numbers SURVEY §8(d) asks for beside the headline: B=256 @512x512 (the reference-true crop size) and the
strict fp32 mode.  One JSON line each.   python tools/bench_modes.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

sd = synthetic_state_dict(0)
for dtype, S, B, iters in (("f16", 512, 256, 6), ("bf16", 512, 256, 6), ("f32", 224, 64, 3), ("f16", 224, 16, 50), ("f16", 224, 1, 50)):
    eng = PoseEngine(S, S, B, dtype)
    eng.load_state_dict(sd)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]
    x = torch.rand(B, S, S, 3).to(tdt).cuda() if dtype != "f32" else torch.rand(B, 3, S, S).cuda()
    fmt = {"f16": 2, "bf16": 1, "f32": 0}[dtype]
    R = torch.empty(B, 9, device="cuda")
    for _ in range(2):
        eng.forward_into(x, fmt, None, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.forward_into(x, fmt, None, R)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(json.dumps({"dtype": dtype, "crop": S, "batch": B, "ms_per_batch": round(dt * 1e3, 3), "poses_per_s": round(B / dt, 1),
                      "tflops": round(eng.flops(B) / dt / 1e12, 1)}), flush=True)
    eng.close()
    del x
    torch.cuda.empty_cache()
