"""Which kernels the e2e shape (B = 15 crops of 512 x 512) really launches, with per-launch times (profile mode)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
B, S = int(os.environ.get("B", 15)), int(os.environ.get("S", 512))
x = torch.rand(B, S, S, 3).half().cuda()
R = torch.empty(B, 9, device="cuda")
for spec in sys.argv[1:] or [""]:
    e = PoseEngine(S, S, 32, "f16")
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("="); e.set_option(k, int(v))
    e.load_state_dict(synthetic_state_dict(0))
    for _ in range(5):
        e.forward_into(x, 2, None, R)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50):
        e.forward_into(x, 2, None, R)
    t1.record(); torch.cuda.synchronize()
    print(f"[{spec}] forward {t0.elapsed_time(t1) / 50 * 1e3:.1f} us")
    e.set_option("profile", 1)
    acc = None
    for it in range(6):
        e.forward_into(x, 2, None, R)
        ms = e.profile_read()
        if it >= 2:
            acc = ms if acc is None else [a + m for a, m in zip(acc, ms)]
    for (layer, kern, fl), a in zip(e.launch_info(B), acc):
        print(f"   {layer:44s} {kern:34s} {a / 4 * 1e3:7.1f} us")
    e.close()
