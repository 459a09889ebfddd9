"""Determinism soak: N forwards of the bench workload (and of a small split-K batch) must reproduce the first result bit
for bit -- a race in the staggered / persistent / split kernels would show up as a flipped bit sooner or later.
    python tools/soak.py [iterations]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sd = synthetic_state_dict(0)
bad = 0
for B, S, dt in ((256, 224, "f16"), (256, 224, "bf16"), (5, 224, "f16"), (12, 512, "f16")):
    e = PoseEngine(S, S, B, dt)
    e.load_state_dict(sd)
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, S, S, 3, generator=g).to(torch.float16 if dt == "f16" else torch.bfloat16).cuda()
    r9 = torch.empty(B, 9, device="cuda"); R = torch.empty(B, 9, device="cuda")
    e.forward_into(x, 2 if dt == "f16" else 1, r9, R)
    ref9, refR = r9.clone(), R.clone()
    mism = 0
    for i in range(n):
        e.forward_into(x, 2 if dt == "f16" else 1, r9, R)
        if not (torch.equal(r9, ref9) and torch.equal(R, refR)):
            mism += 1
    print(f"B={B} {S}x{S} {dt}: {n} forwards, {mism} mismatches, finite={bool(torch.isfinite(R).all())}", flush=True)
    bad += mism
    e.close()
sys.exit(1 if bad else 0)
