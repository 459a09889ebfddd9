"""Where a frame of the live loop spends its time (host side, perf_counter around each phase with a device sync after it):
developer probe for FastPosePredictor.iter_flower_poses.   python tools/probe_pipeline.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flope_amd")]
from flope_amd.weights import synthetic_state_dict  # noqa: E402
from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict  # noqa: E402
from sunflower.predictor import fast_pose_predictor as F  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    ckpt, intr, yolo_f = (os.path.join(tmp, n) for n in ("posenet.pth", "intrinsics.yaml", "yolo.pth"))
    torch.save(synthetic_state_dict(0), ckpt)
    open(intr, "w").write(yaml.safe_dump(dict(fx=1400.0, fy=1400.0, cx=960.0, cy=540.0, h=1080, w=1920)))
    torch.save({**synthetic_yolo_state_dict(0), "imgsz": torch.tensor(1280)}, yolo_f)
    rgb = synthetic_frame(0)
    depth = (400 + np.random.default_rng(0).normal(0, 4, rgb.shape[:2])).astype(np.uint16)
    ydt = os.environ.get("YOLO_DTYPE", "f32")
    pred = F.FastPosePredictor("cuda", yolo_f, ckpt, intr, yolo_dtype=ydt)
    print(f"detector dtype: {ydt}")
    dev = torch.device("cuda")
    for _ in range(5):
        pred.get_flower_poses(rgb, depth)
    acc = {}

    def phase(name, fn, sync=True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        if sync:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        a = acc.setdefault(name, [0.0, 0.0])
        a[0] += t1 - t0; a[1] += t2 - t0
        return out

    it = 30
    y = pred.yolo
    for _ in range(it):
        fd = phase("frame H2D (pageable numpy -> device buffer)", lambda: y._frame(rgb))
        det, count, mask_d, frame_d = phase("detector (frame on the device)", lambda: y.detect_device(fd))
        depth_d = phase("depth H2D", lambda: F.upload_depth(depth, dev))
        ctx = pred._frame_ctx()
        phase("flope_frame_select (device-side box selection; count -> pinned memory)", lambda: ctx.select(0, det, count))
        phase("flope_frame_enqueue (count wait, depth lift, crops, network, Rt, copies)", lambda: ctx.enqueue(0, frame_d, mask_d, depth_d, pred.K, 1000.0))
        phase("flope_frame_finish (wait + reliability filter -> float64 [n,4,4])", lambda: ctx.finish(0))
    print(f"{'phase':70s} host-only ms   host+device ms")
    for k, (h, hd) in acc.items():
        print(f"{k:70s} {h / it * 1e3:9.3f}     {hd / it * 1e3:9.3f}")
    print(f"{'sum':70s} {sum(v[0] for v in acc.values()) / it * 1e3:9.3f}     {sum(v[1] for v in acc.values()) / it * 1e3:9.3f}")


if __name__ == "__main__":
    main()
