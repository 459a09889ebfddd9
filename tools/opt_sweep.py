import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
B = 256
x = torch.rand(B, 224, 224, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda")
combos = [dict(), dict(), dict()]
for opts in combos:
    e = PoseEngine(224, 224, B, "f16")
    for k, v in opts.items(): e.set_option(k, v)
    e.load_state_dict(sd)
    for _ in range(5): e.forward_into(x, 2, None, R)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n): e.forward_into(x, 2, None, R)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(opts, f"{dt*1e3:.4f} ms  {B/dt:,.0f} poses/s", flush=True)
    e.close()
