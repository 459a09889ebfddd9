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


def live_pose_loop(predictor, frames, pipelined: bool = False):
    """frames: iterable of (rgb uint8 [H,W,3], depth uint16 [H,W]) -> list of float64 [N,4,4] | None.
    pipelined: use the predictor's software pipeline (FastPosePredictor.iter_flower_poses: uploads of frame t + 1, detector
    of frame t and pose network of frame t - 1 on three streams); same results, higher frame rate."""
    if pipelined and hasattr(predictor, "iter_flower_poses"):
        return list(predictor.iter_flower_poses(frames))
    return [predictor.get_flower_poses(rgb, depth) for rgb, depth in frames]


# ---- the files either side of the path: depth_val/*.txt and points_3d/*.txt -----------------------
def depth_val_rows(depth_raw: np.ndarray, mask: np.ndarray, boxes, depth_div: float = 1000.0, near: float = 0.1,
                   far: float = 3.0, device="cuda") -> np.ndarray:
    """``scripts/extract_depth.py:25-57``: per detection box the masked mean depth (m) and its reliability flag,
    as the 2 x N array the script writes (row 0 depth_val, row 1 depth_reliable as 0/1); shape (0,) when there is no
    detection.  The reduction runs on the GPU (flope_depth_lift); boxes are the UN-squared detector boxes."""
    boxes = np.asarray(boxes).reshape(-1, 4)
    if boxes.shape[0] == 0:
        return np.array([])
    dev = torch.device(device)
    d = np.ascontiguousarray(depth_raw)
    dt = torch.from_numpy(d.view(np.int16) if d.dtype == np.uint16 else d.astype(np.float32)).to(dev)
    dv, rel, _ = _engine.depth_lift(dt, torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)).to(dev),
                                    torch.from_numpy(boxes.astype(np.int32)).to(dev), (1.0, 1.0, 0.0, 0.0),
                                    depth_div if d.dtype == np.uint16 else 1.0, near, far)
    return np.vstack((dv.cpu().numpy().astype(np.float64), rel.cpu().numpy().astype(np.float64)))


def write_depth_val_file(path, rows: np.ndarray) -> None:
    np.savetxt(path, rows)                        # default '%.18e', as extract_depth.py:55


def world_points_from_files(det: np.ndarray, depth_info: np.ndarray, cam_pose7: np.ndarray, K: np.ndarray):
    """``scripts/align_measurements.py:196-247``: detection rows [N,15] + depth_val [2,N] + camera pose
    [tx ty tz qx qy qz qw] -> (world translations [M,3], scalar-last quaternions [M,4]) of the detections whose depth
    is reliable, or (None, None) when nothing usable remains (the script writes an empty points_3d file then)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from sunflower.predictor.flower_model import cam_pose_to_matrix
    from sunflower.utils.conversion import get_pose_mat, rotmat2qvec
    from sunflower.utils.mvg import get_points3d, pose_cam_to_world
    det, depth_info = np.asarray(det, dtype=np.float64), np.asarray(depth_info, dtype=np.float64)
    if det.shape[0] == 0 or depth_info.shape[0] == 0:
        return None, None
    if depth_info.ndim == 1:
        depth_info = depth_info[None].T
    depth_val, depth_reliable = depth_info
    if det.ndim == 1:
        det = det[None]
    ok = depth_reliable > 0.5
    depth_val, uv, rot = depth_val[ok], det[ok, 4:6], det[ok, 6:]
    if depth_val.shape[0] == 0:
        return None, None
    pts_cam = get_points3d(uv, depth_val, K)
    world = pose_cam_to_world(get_pose_mat(np.hstack((pts_cam, rot))), cam_pose_to_matrix(cam_pose7))
    return world[:, :3, 3], rotmat2qvec(world[:, :3, :3])
