"""Small-batch latency of the network alone (the reference's live loop sees 1-31 crops per frame):
    python tools/latency.py ["opt=val,..."]      B x S in {1x224, 16x224, 4x512, 16x512, 31x512}"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

sd = synthetic_state_dict(0)
specs = sys.argv[1:] or [""]
for spec in specs:
    opts = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if "=" in kv)}
    for B, S in ((1, 224), (16, 224), (4, 512), (16, 512), (31, 512)):
        e = PoseEngine(S, S, B, "f16")
        for k, v in opts.items():
            e.set_option(k, v)
        e.load_state_dict(sd)
        x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
        R = torch.empty(B, 9, device="cuda")
        for _ in range(5):
            e.forward_into(x, 2, None, R)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            e.forward_into(x, 2, None, R)
            torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / n
        print(f"{opts} B={B} {S}x{S}: {lat*1e3:.3f} ms per forward (synchronous), {B/lat:,.0f} poses/s", flush=True)
        e.close()
