"""Determinism + parity soak of the register-weight layer-2 kernels (conv_s2r / conv_s1r): for several batch sizes and both schedules,
N forwards must reproduce the first bit for bit, and the first must agree with the conv_w4 / gathered-tile path within rounding.
    python tools/probes/soak_layer2.py [iterations]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sd = synthetic_state_dict(0)
bad = 0
for B, streams, dt in ((1, 1, "f16"), (7, 1, "bf16"), (37, 1, "f16"), (128, 2, "f16"), (203, 2, "f16"), (256, 2, "bf16"), (256, 1, "f16")):
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 224, 224, 3, generator=g).to(torch.float16 if dt == "f16" else torch.bfloat16).cuda()
    fmt = 2 if dt == "f16" else 1
    ref = None
    for new in (0, 1):
        e = PoseEngine(224, 224, B, dt)
        e.set_option("streams", streams); e.set_option("s1r", new); e.set_option("s2r", new)
        e.load_state_dict(sd)
        r9 = torch.empty(B, 9, device="cuda"); R = torch.empty(B, 9, device="cuda")
        e.forward_into(x, fmt, r9, R)
        first9, firstR = r9.clone(), R.clone()
        if new == 0:
            ref = first9
        else:
            mism = 0
            for i in range(n):
                e.forward_into(x, fmt, r9, R)
                if not (torch.equal(r9, first9) and torch.equal(R, firstR)):
                    mism += 1
            rel = float((first9 - ref).norm() / ref.norm())
            ok = mism == 0 and rel < (2e-3 if dt == "f16" else 1e-2) and bool(torch.isfinite(R).all())
            print(f"B={B} streams={streams} {dt}: {n} forwards, {mism} mismatches; r9 vs the conv_w4 path rel {rel:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
            bad += 0 if ok else 1
        e.close()
sys.exit(1 if bad else 0)
