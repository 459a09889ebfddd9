"""Hot-path subset of the reference's ``sunflower/utils/mvg.py``: box geometry
(:324-362), back-projection (:387-408), yaw nullification (:240-251) and
cam->world (:416-421).  Integer box logic is host code as in the reference; the
rotation work is the closed form the head kernel also uses."""
import numpy as np


def squarify_bb(bb):
    """Grow the short side of [xmin,ymin,xmax,ymax] to the long side: the min edge moves
    by ceil(d/2), the max edge by floor(d/2); no clamping (mvg.py:324-343)."""
    xmin, ymin, xmax, ymax = (int(v) for v in bb)
    w, h = xmax - xmin, ymax - ymin
    d = abs(w - h)
    lo, hi = (d + 1) // 2, d // 2
    if w > h:
        ymin, ymax = ymin - lo, ymax + hi
    elif h > w:
        xmin, xmax = xmin - lo, xmax + hi
    return [xmin, ymin, xmax, ymax]


def bb_in_frame(bb, img_shape):
    """True iff the box lies inside the image; xmax == w and ymax == h are accepted (mvg.py:345-351)."""
    h, w = img_shape[0], img_shape[1]
    return not (bb[0] < 0 or bb[1] < 0 or bb[2] > w or bb[3] > h)


def filter_very_large_bb(bb_dino):
    """Drop boxes whose area exceeds 5x the median area (mvg.py:354-362)."""
    bb = np.array(bb_dino)
    area = (bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1])
    return bb[~(area > 5 * np.median(area))]


def get_points3d(uv, Zray, K):
    """Pixels + ray lengths -> camera-frame points: xyz = n * d / |n|, n = K^-1 [u,v,1]^T
    (depth is distance along the ray, not Z; mvg.py:387-408)."""
    uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
    rays = np.linalg.solve(np.asarray(K, dtype=np.float64), np.c_[uv, np.ones(len(uv))].T).T
    return rays * (np.asarray(Zray, dtype=np.float64) / np.linalg.norm(rays, axis=1))[:, None]


def nullify_yaw_batch(rotmat):
    """Zero the first 'zyx' Euler angle of each rotation (mvg.py:240-251).  Closed form of
    the scipy round trip: R' = R Rz(a)^T with a = atan2(-R01, R00)."""
    R = np.asarray(rotmat, dtype=np.float64)
    c, s = R[:, 0, 0], -R[:, 0, 1]
    n = np.hypot(c, s)
    safe = n > 0
    c = np.where(safe, c / np.where(safe, n, 1), 1.0)
    s = np.where(safe, s / np.where(safe, n, 1), 0.0)
    out = R.copy()
    out[:, :, 0] = R[:, :, 0] * c[:, None] - R[:, :, 1] * s[:, None]
    out[:, :, 1] = R[:, :, 0] * s[:, None] + R[:, :, 1] * c[:, None]
    return out


def nullify_yaw(Rmatrix):
    return nullify_yaw_batch(np.asarray(Rmatrix)[None])[0]


def pose_cam_to_world(obj_pose, cam_pose):
    """(N,4,4) object poses in the camera frame -> world frame (mvg.py:416-421)."""
    return np.einsum("ij,njk->nik", np.asarray(cam_pose), np.asarray(obj_pose))
