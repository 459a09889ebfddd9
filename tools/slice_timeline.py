"""The two-slice step as it runs in production (no profiler attached): HIP events on every slice's own stream around every launch
(engine option profile = 2), milliseconds since the fork.
    python tools/slice_timeline.py ["opt=v,..."]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402

B, S = int(os.environ.get("B", 256)), int(os.environ.get("S", 224))
e = PoseEngine(S, S, B, "f16")
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []):
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.load_state_dict(synthetic_state_dict(0))
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
for _ in range(20):
    e.forward_into(x, 2, None, R)
torch.cuda.synchronize()
e.set_option("profile", 1)
names = [k for _, k, _ in e.launch_info(B)]            # launch order of one slice
e.set_option("profile", 2)
runs = []
for _ in range(12):
    for _ in range(3):
        e.forward_into(x, 2, None, R)
    ms = (C.c_float * 256)(); sl = (C.c_int * 256)()
    n = e.lib.flope_profile_timeline(e.handle, ms, sl, 256)
    assert n > 2, n
    runs.append((np.array(ms[:n]), np.array(sl[:n])))
t = np.median(np.stack([r[0] for r in runs]), axis=0) * 1e3          # us
sl = runs[0][1]
print(f"B={B} S={S}: events on the slices' own streams, median of {len(runs)} steps, microseconds since the fork")
for s in sorted(set(sl[1:])):
    idx = [i for i in range(1, len(sl)) if sl[i] == s]
    print(f"slice {s}: first launch eligible at {t[idx[0]]:.1f}, last launch done at {t[idx[-1]]:.1f}")
    for j in range(len(idx) - 1):
        nm = names[j] if j < len(names) else "?"
        print(f"   {t[idx[j]]:8.1f} -> {t[idx[j + 1]]:8.1f}  {t[idx[j + 1]] - t[idx[j]]:6.1f} us  {nm[:40]}")
print(f"step: {t[1:].max():.1f} us")
