"""Soak of the multi-frame live loop: N different 1080p frames through FastPosePredictor.iter_flower_poses (1 and 2 detector
instances, several frames in flight, per-slot buffers, graph replay) must give, frame by frame, exactly what the sequential
get_flower_poses gives -- a slot-reuse or stream-ordering race would show up as a mismatch sooner or later.
    python tools/soak_pipeline.py [frames]"""
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flope_amd")]
from flope_amd.weights import synthetic_state_dict  # noqa: E402
from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict  # noqa: E402
from sunflower.predictor.fast_pose_predictor import FastPosePredictor  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tmp = tempfile.mkdtemp()
ckpt, intr, yolo_f = (os.path.join(tmp, f) for f in ("posenet.pth", "intrinsics.yaml", "yolo.pth"))
torch.save(synthetic_state_dict(0), ckpt)
open(intr, "w").write(yaml.safe_dump(dict(fx=1400.0, fy=1400.0, cx=960.0, cy=540.0, h=1080, w=1920)))
torch.save({**synthetic_yolo_state_dict(0), "imgsz": torch.tensor(1280)}, yolo_f)
pred = FastPosePredictor("cuda", yolo_f, ckpt, intr)
base = [synthetic_frame(s) for s in range(8)]
rng = np.random.default_rng(0)
frames = []
for i in range(n):
    img = base[i % 8]
    if i % 11 == 5:
        img = np.zeros_like(img)                                   # nothing to detect
    elif i % 3 == 1:
        img = np.ascontiguousarray(np.roll(img, int(rng.integers(1, 600)), axis=1))   # different content every time
    frames.append((img, (400 + rng.normal(0, 4, img.shape[:2])).astype(np.uint16)))
seq = [pred.get_flower_poses(rgb, d) for rgb, d in frames]
print(f"sequential: {sum(r is not None for r in seq)} of {n} frames with poses, {sum(0 if r is None else r.shape[0] for r in seq)} poses", flush=True)
for nd in (1, 2, 2):
    bad = 0
    for a, b in zip(seq, pred.iter_flower_poses(frames, detectors=nd)):
        if (a is None) != (b is None) or (a is not None and not np.array_equal(a, b)):
            bad += 1
    print(f"pipelined, {nd} detector instance(s): {bad} mismatching frames of {n}", flush=True)
