"""BASELINE cfg3: end-to-end frames/s of the live_pose-style path on synthetic 1080p frames held in host memory.
  * "given": synthetic detector (boxes + mask given): frame/mask/depth H2D -> depth lift -> crop+Lanczos -> PoseResNet
    -> Procrustes -> yaw-null -> Rt -> D2H                                  python tools/bench_e2e.py [n_flowers ...]
  * "yolo": the whole FastPosePredictor with the built-in YOLO11n-seg detector (synthetic weights, imgsz 1280):
    frame/depth H2D -> letterbox -> network -> decode -> NMS -> masks -> boxes D2H -> the path above
                                                                              python tools/bench_e2e.py yolo"""
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
from sunflower.predictor.fast_pose_predictor import FastPosePredictor  # noqa: E402


def scene(n, H=1080, W=1920, seed=0):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    depth = (400 + rng.normal(0, 4, (H, W))).astype(np.uint16)
    yy, xx = np.mgrid[:H, :W]
    boxes = []
    for i in range(n):
        r = int(rng.integers(40, 90)); cx = int(rng.integers(r + 12, W - r - 12)); cy = int(rng.integers(r + 12, H - r - 12))
        mask[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
        boxes.append([cx - r - 3, cy - r, cx + r + 4, cy + r + 2])
    return rgb, mask, depth, np.array(boxes, dtype=np.int16)


def main_yolo(tmp, ckpt, intr):
    from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict
    yolo_f = os.path.join(tmp, "yolo11n_seg.pth")
    torch.save({**synthetic_yolo_state_dict(0), "imgsz": torch.tensor(1280)}, yolo_f)
    rgb = synthetic_frame(0)
    depth = (400 + np.random.default_rng(0).normal(0, 4, rgb.shape[:2])).astype(np.uint16)
    ydt = os.environ.get("YOLO_DTYPE", "f32")                        # r05: the exact-float32 detector is the predictor's default
    pred = FastPosePredictor("cuda", yolo_f, ckpt, intr, yolo_dtype=ydt)
    print(f"detector dtype: {ydt}", flush=True)
    for _ in range(3):
        Rt = pred.get_flower_poses(rgb, depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); it = 20
    for _ in range(it):
        Rt = pred.get_flower_poses(rgb, depth)
    dt = (time.perf_counter() - t0) / it
    t0 = time.perf_counter()
    for _ in range(it):
        bb, mask = pred.get_bbox_mask(rgb)
    dd = (time.perf_counter() - t0) / it
    print(f"[e2e+yolo] 1080p frame -> {bb.shape[0]} detections -> {0 if Rt is None else Rt.shape[0]} poses (512x512 crops): "
          f"{dt*1e3:.2f} ms/frame, {1/dt:.1f} frames/s; get_bbox_mask alone (numpy in, numpy out) {dd*1e3:.2f} ms", flush=True)
    frames = [(rgb, depth)] * 120
    for nd in (1, 2, 3):
        list(pred.iter_flower_poses(frames[:12], detectors=nd))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = list(pred.iter_flower_poses(frames, detectors=nd))
        dp = (time.perf_counter() - t0) / len(frames)
        print(f"[e2e+yolo, pipelined, {nd} detector instance{'s' if nd > 1 else ''}] uploads, detector(s) and pose network of consecutive "
              f"frames on their own streams: {dp*1e3:.2f} ms/frame, {1/dp:.1f} frames/s, "
              f"{sum(0 if o is None else o.shape[0] for o in outs)/len(frames)/dp:.0f} poses/s", flush=True)


def main():
    args = [a for a in sys.argv[1:] if a != "yolo"]
    counts = [int(a) for a in args] or ([] if "yolo" in sys.argv[1:] else [4, 16, 31])
    tmp = tempfile.mkdtemp()
    ckpt, intr = os.path.join(tmp, "posenet.pth"), os.path.join(tmp, "intrinsics.yaml")
    torch.save(synthetic_state_dict(0), ckpt)
    open(intr, "w").write(yaml.safe_dump(dict(fx=1400.0, fy=1400.0, cx=960.0, cy=540.0, h=1080, w=1920)))
    if "yolo" in sys.argv[1:]:
        main_yolo(tmp, ckpt, intr)
    for n in counts:
        rgb, mask, depth, boxes = scene(n)
        pred = FastPosePredictor("cuda", lambda img: (boxes, mask), ckpt, intr)
        for _ in range(3):
            Rt = pred.get_flower_poses(rgb, depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); it = 20
        for _ in range(it):
            Rt = pred.get_flower_poses(rgb, depth)
        dt = (time.perf_counter() - t0) / it
        print(f"[e2e] 1080p, {n} boxes -> {0 if Rt is None else Rt.shape[0]} poses (512x512 crops): {dt*1e3:.2f} ms/frame, "
              f"{1/dt:.1f} frames/s, {(0 if Rt is None else Rt.shape[0])/dt:.0f} poses/s", flush=True)


if __name__ == "__main__":
    main()
