"""Counterparts of the reference's two pose harnesses, minus their file/GUI plumbing.

* ``detection_rows`` / ``write_detection_file`` -- ``scripts/test_posenet.py:104-161``: squarify + in-frame
  filter, crop batch, PoseResNet, Procrustes, one 15-column row per flower
  ``[xmin, ymin, xmax, ymax, cx, cy, R00..R22]`` written with ``fmt='%.7f'`` (empty file when nothing survives).
  No yaw-nullification and no depth here, exactly like the reference script.
* ``live_pose_loop`` -- ``scripts/live_pose.py:31-41``: frames -> ``predictor.get_flower_poses`` -> ``[N,4,4] | None``.
Detector / segmenter outputs (boxes, mask) are inputs: GroundingDINO and SAM are hub models (out of scope).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import engine as _engine


def detection_rows(posenet, frame: np.ndarray, mask: np.ndarray, boxes, crop_size: int = 512, device="cuda"):
    """-> float64 [N,15] (N may be 0)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from sunflower.utils.mvg import bb_in_frame, squarify_bb
    keep, sq = [], []
    for bb in np.asarray(boxes).reshape(-1, 4):
        s = squarify_bb(bb)
        if bb_in_frame(s, frame.shape):
            keep.append([int(v) for v in bb])
            sq.append(s)
    if not keep:
        return np.zeros((0, 15))
    dev = torch.device(device)
    crops = _engine.crop_resize_mask(torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8)).to(dev),
                                     torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)).to(dev),
                                     torch.tensor(sq, dtype=torch.int32, device=dev), crop_size, _lib.IN_F32_NCHW)
    _, R = posenet.predict_rotations(crops)
    R = R.detach().cpu().numpy().astype(np.float64)
    kb = np.asarray(keep, dtype=np.float64)
    centre = np.stack([(kb[:, 0] + kb[:, 2]) / 2, (kb[:, 1] + kb[:, 3]) / 2], axis=1)
    return np.concatenate([kb, centre, R.reshape(-1, 9)], axis=1)


def write_detection_file(path, rows: np.ndarray) -> None:
    """np.savetxt(..., fmt='%.7f') as scripts/test_posenet.py:161 (and :83 for the empty case)."""
    np.savetxt(path, np.asarray(rows) if len(rows) else np.array([]), fmt="%.7f")


def live_pose_loop(predictor, frames):
    """frames: iterable of (rgb uint8 [H,W,3], depth uint16 [H,W]) -> list of float64 [N,4,4] | None."""
    return [predictor.get_flower_poses(rgb, depth) for rgb, depth in frames]
