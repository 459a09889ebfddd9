"""One bench step (B=256, 224^2, f16) launched directly vs replayed from a captured hipGraph (torch.cuda.CUDAGraph around the
engine call: the engine's two slice streams fork from / join to the capturing stream, so the capture holds both)."""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
B, S = 256, 224
eng = PoseEngine(S, S, B, "f16"); eng.load_state_dict(synthetic_state_dict(0))
x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
R = torch.empty(B, 9, device="cuda"); xyz = torch.zeros(B, 3, device="cuda"); poses = torch.empty(B, 16, device="cuda")
st = torch.cuda.Stream()
def direct(n):
    for _ in range(n):
        eng.forward_poses_into(x, 2, xyz, True, poses, R)
with torch.cuda.stream(st):
    direct(5); torch.cuda.synchronize()
    ref = poses.clone()
    for rep in range(3):
        t0 = time.perf_counter(); direct(50); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"direct launches: {(t1 - t0) / 50 * 1e3:.4f} ms / step")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        eng.forward_poses_into(x, 2, xyz, True, poses, R)
    g.replay(); torch.cuda.synchronize()
    print("graph result identical:", bool(torch.equal(ref, poses)))
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"graph replay:    {(t1 - t0) / 50 * 1e3:.4f} ms / step")
