"""cProfile of the host side of FastPosePredictor.get_flower_poses (200 frames): where the Python / ctypes / torch time goes."""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flope_amd")]
from flope_amd.weights import synthetic_state_dict  # noqa: E402
from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict  # noqa: E402
from sunflower.predictor import fast_pose_predictor as F  # noqa: E402

tmp = tempfile.mkdtemp()
ckpt, intr, yolo_f = (os.path.join(tmp, n) for n in ("posenet.pth", "intrinsics.yaml", "yolo.pth"))
torch.save(synthetic_state_dict(0), ckpt)
open(intr, "w").write(yaml.safe_dump(dict(fx=1400.0, fy=1400.0, cx=960.0, cy=540.0, h=1080, w=1920)))
torch.save({**synthetic_yolo_state_dict(0), "imgsz": torch.tensor(1280)}, yolo_f)
rgb = synthetic_frame(0)
depth = (400 + np.random.default_rng(0).normal(0, 4, rgb.shape[:2])).astype(np.uint16)
pred = F.FastPosePredictor("cuda", yolo_f, ckpt, intr)
for _ in range(5):
    pred.get_flower_poses(rgb, depth)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    pred.get_flower_poses(rgb, depth)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
