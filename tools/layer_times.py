"""Per-launch GPU times of one forward (profile mode), for a list of option sets.
    python tools/layer_times.py "patch=1,bm256=1,dma=1" "patch=1,bm256=1,dma=1,dbg=1" ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

B, S = int(os.environ.get("B", 256)), int(os.environ.get("S", 224))
sd = synthetic_state_dict(0)
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
cols = {}
info = None
for spec in sys.argv[1:]:
    eng = PoseEngine(S, S, B, "f16")
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
    eng.load_state_dict(sd)
    eng.set_option("profile", 1)
    info = eng.launch_info(B)
    acc = [0.0] * len(info)
    n = 8
    for it in range(n + 2):
        eng.forward_into(x, 2, None, R)
        ms = eng.profile_read()
        if it >= 2:
            acc = [a + m for a, m in zip(acc, ms)]
    cols[spec] = [a / n for a in acc]
    eng.close()
print(f"{'layer':34s} {'kernel':36s} {'GF':>6s} " + " ".join(f"{s[-22:]:>22s}" for s in cols))
for i, (layer, kern, fl) in enumerate(info):
    print(f"{layer:34s} {kern:36s} {fl/1e9:6.1f} " + " ".join(f"{cols[s][i]*1e3:14.1f}us{(fl/cols[s][i]/1e9 if cols[s][i] > 0 else 0):6.0f}T" for s in cols))
print(f"{'TOTAL':78s} " + " ".join(f"{sum(cols[s])*1e3:14.1f}us      " for s in cols))
