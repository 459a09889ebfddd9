import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
B=256
e=PoseEngine(224,224,B,"f16"); e.load_state_dict(synthetic_state_dict(0))
x=torch.rand(B,224,224,3).to(torch.float16).cuda()
R=torch.empty(B,9,device="cuda"); xyz=torch.zeros(B,3,device="cuda")
def series(tag, n=20):
    poses=torch.empty(n,B,16,device="cuda")
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(n+1)]
    t0=time.perf_counter(); ev[0].record()
    for i in range(n):
        e.forward_poses_into(x,2,xyz,True,poses[i],R); ev[i+1].record()
    torch.cuda.synchronize(); wall=time.perf_counter()-t0
    print(tag, f"wall/step {wall/n*1e3:.4f}", " ".join(f"{ev[i].elapsed_time(ev[i+1]):.3f}" for i in range(n)), flush=True)
mode=sys.argv[1]
if mode=="tune":
    print(e.autotune(x,2))
elif mode=="groups":
    for g in range(26):
        for _ in range(5): e.forward_into(x,2,None,R)
        torch.cuda.synchronize()
elif mode=="cont":
    for _ in range(130): e.forward_into(x,2,None,R)
    torch.cuda.synchronize()
poses=torch.empty(20,B,16,device="cuda")
for i in range(5): e.forward_poses_into(x,2,xyz,True,poses[i],R)
series("first ")
series("second")
time.sleep(0.5)
series("after 0.5 s idle")
