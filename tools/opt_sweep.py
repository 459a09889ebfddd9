"""Same-run A/B of engine options on the bench workload (B=256, 224x224, f16 NHWC resident): box-to-box spread on the
pool is up to +-4 %, so only pairs measured inside one process are comparable.
    python tools/opt_sweep.py                       # default vs a few knobs
    python tools/opt_sweep.py "split=50" "dsfuse=0" "streams=1,stag=1"
    FLOPE_AMD_LIB=/path/to/other/libflope_amd.so python tools/opt_sweep.py      # A/B of two builds: run twice"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402


def parse(spec):
    return {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if "=" in kv)}


specs = sys.argv[1:] or ["", "split=50", "dsfuse=0", "stag=1", "streams=1", ""]
sd = synthetic_state_dict(0)
B = 256
x = torch.rand(B, 224, 224, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
for spec in specs:
    opts = parse(spec)
    e = PoseEngine(224, 224, B, "f16")
    for k, v in opts.items():
        e.set_option(k, v)
    e.load_state_dict(sd)
    for _ in range(int(os.environ.get("WARM", 5))):      # WARM=100: past the ~30 ms the device needs to reach its sustained clock
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = int(os.environ.get("N", 40))
    for _ in range(n):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(opts, f"{dt*1e3:.4f} ms  {B/dt:,.0f} poses/s", flush=True)
    e.close()
