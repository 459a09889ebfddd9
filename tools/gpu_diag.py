"""Stage-by-stage parity + timing report for one GPU box run (never aborts on a mismatch).
Usage: python tools/gpu_diag.py [--quick]   -> prints a table; meant to be redirected to gpurun_out/."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402
from oracle import posenet_ref as O  # noqa: E402

STAGES = ["stem", "pool"] + [f"layer{li}.{bi}" for li in range(1, 5) for bi in range(2)] + ["feat", "hidden"]
TDT = {"f16": torch.float16, "bf16": torch.bfloat16}


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def parity(sd, H, W, B, dtype, opts):
    torch.manual_seed(11)
    x = torch.rand(B, 3, H, W)
    ref = O.forward_stages(sd, x) if dtype == "f32" else O.forward_stages_emulated(sd, x, TDT[dtype])
    e = PoseEngine(H, W, B, dtype)
    for k, v in opts.items():
        e.set_option(k, v)
    e.load_state_dict(sd)
    r9, R = e.forward(x.cuda())
    torch.cuda.synchronize()
    line = []
    for s in STAGES:
        if s == "stem" and dtype != "f32" and opts.get("fuse_stem", 1):
            continue
        got = e.read_stage(s, B).cpu()
        line.append(f"{s}={rel(got, ref[s]):.1e}")
    line.append(f"r9={rel(r9.cpu(), ref['r9']):.1e}")
    Rf = O.procrustes_to_rotmat(O.forward(sd, x))
    line.append(f"Rerr_vs_fp32={float((R.cpu() - Rf).abs().max()):.1e}")
    print(f"[parity] {dtype} {H}x{W} B={B} {opts}: " + " ".join(line), flush=True)
    e.close()


def timing(sd, H, W, B, dtype, opts, iters=20):
    e = PoseEngine(H, W, B, dtype)
    for k, v in opts.items():
        e.set_option(k, v)
    e.load_state_dict(sd)
    x = torch.rand(B, H, W, 3).to(TDT[dtype]).cuda()
    fmt = 1 if dtype == "bf16" else 2
    r9 = torch.empty(B, 9, device="cuda"); R = torch.empty(B, 9, device="cuda")
    for _ in range(5):
        e.forward_into(x, fmt, r9, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        e.forward_into(x, fmt, r9, R)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    tf = e.flops(B) / dt / 1e12
    print(f"[timing] {dtype} {H}x{W} B={B} {opts}: {dt*1e3:.3f} ms/batch  {B/dt:,.0f} poses/s  {tf:.1f} TFLOP/s", flush=True)
    e.close()


def safe(fn, *a, **k):
    try:
        fn(*a, **k)
    except Exception as exc:   # keep going: this is a report, not a gate
        print(f"[error] {fn.__name__}{a[1:]}: {type(exc).__name__}: {exc}", flush=True)


def main():
    quick = "--quick" in sys.argv
    print(torch.cuda.get_device_name(0), flush=True)
    sd = synthetic_state_dict(0)
    if "--rows" in sys.argv:           # the row-band layer-1 kernel (stag=3) only
        e = PoseEngine(224, 224, 256, "f16"); e.set_option("stag", 3); print(e.describe_plan()); e.close()
        for hw in ((224, 224, 48), (224, 224, 5), (96, 80, 3), (65, 71, 2), (64, 256, 3)):
            for st in (1, 2):
                safe(parity, sd, hw[0], hw[1], hw[2], "f16", dict(fuse_stem=1, stag=3, streams=st))
        safe(parity, sd, 224, 224, 7, "bf16", dict(fuse_stem=1, stag=3))
        for opts in [dict(streams=2, stag=1), dict(streams=2, stag=3), dict(streams=1, stag=3), dict(streams=1, stag=1)]:
            safe(timing, sd, 224, 224, 256, "f16", opts)
        return
    e = PoseEngine(224, 224, 256, "f16"); print(e.describe_plan()); e.close()
    safe(parity, sd, 96, 80, 3, "f32", {})
    safe(parity, sd, 224, 224, 48, "f16", dict(fuse_stem=1, stag=2, persist=1))
    safe(parity, sd, 224, 224, 48, "f16", dict(fuse_stem=1, stag=1, persist=1))
    safe(parity, sd, 224, 224, 48, "f16", dict(fuse_stem=1, stag=2, persist=0))
    for hw in ((224, 224, 5), (65, 71, 2), (130, 50, 3)):
        safe(parity, sd, hw[0], hw[1], hw[2], "f16", dict(fuse_stem=1, stag=2))
    for dtype in ("f16", "bf16"):
        for opts in [dict(patch=a, bm256=b, nbuf=c, fuse_stem=a, stag=2 * b) for a in (0, 1) for b in (0, 1) for c in (2, 3)]:
            safe(parity, sd, 96, 80, 3, dtype, opts)
            if not quick:
                safe(parity, sd, 224, 224, 5, dtype, opts)
    for opts in [dict(streams=2, stag=1, persist=0), dict(streams=2, stag=2, persist=0), dict(streams=2, stag=2, persist=1), dict(streams=1, stag=2, persist=1), dict(streams=3, stag=1, persist=0)]:
        safe(timing, sd, 224, 224, 256, "f16", opts)
    safe(timing, sd, 224, 224, 256, "bf16", dict(patch=1, bm256=1))
    safe(timing, sd, 512, 512, 64, "f16", dict(patch=1, bm256=1), iters=5)


if __name__ == "__main__":
    main()
