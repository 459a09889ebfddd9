"""Oracle: the non-network steps of the pose pipeline (numpy / scipy, CPU).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Follows, step by step:
  * ``sunflower/utils/mvg.py:324-343`` squarify_bb, ``:345-351`` bb_in_frame,
    ``:354-362`` filter_very_large_bb, ``:387-408`` get_points3d,
    ``:240-251`` nullify_yaw_batch (through scipy 'zyx' Euler angles exactly as
    ``sunflower/utils/conversion.py:45-51`` does)
  * ``sunflower/utils/image_manipulation.py:21-36`` shrink_mask, ``:39-96`` get_depth_value
  * ``sunflower/predictor/fast_pose_predictor.py:60-156`` get_flower_poses (from given
    detections), ``scripts/test_posenet.py:124-161`` the 15-column detection rows
  * OpenCV 4.10 ``cv2.resize(INTER_LANCZOS4)`` on uint8 and ``cv2.erode`` with
    ``getStructuringElement(MORPH_ELLIPSE,(10,10))`` -- third-party, absent here, no golden
    vectors in the reference: restated from the published algorithm, PARITY UNPINNED.
Known-answer values for the pure box/point functions are hand-derived from the source
 and checked in tests/test_oracle.py.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


# ---- mvg.py -----------------------------------------------------------------------

def squarify_bb(bb):
    xmin, ymin, xmax, ymax = bb
    xrange, yrange = xmax - xmin, ymax - ymin
    diff = abs(xrange - yrange)
    if diff % 2 == 0:
        dec, inc = diff / 2, diff / 2
    else:
        dec, inc = (diff + 1) / 2, (diff - 1) / 2
    if xrange > yrange:
        ymin, ymax = ymin - dec, ymax + inc
    elif xrange < yrange:
        xmin, xmax = xmin - dec, xmax + inc
    return [int(xmin), int(ymin), int(xmax), int(ymax)]


def bb_in_frame(bb, img_shape):
    h, w = img_shape[0], img_shape[1]
    xmin, ymin, xmax, ymax = bb
    return not (xmin < 0 or ymin < 0 or xmax > w or ymax > h)


def filter_very_large_bb(bb):
    bb = np.array(bb)
    area = (bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1])
    return bb[np.logical_not(area > 5 * np.median(area))]


def get_points3d(uv, Zray, K):
    n = uv.shape[0]
    uv1 = np.hstack((uv, np.ones(n).reshape(-1, 1)))
    rays = (np.linalg.inv(K) @ uv1.T).T
    Z = Zray / np.linalg.norm(rays, axis=1)
    return rays * Z.reshape(-1, 1)


def nullify_yaw_batch(rotmat):
    e = Rotation.from_matrix(rotmat).as_euler("zyx", degrees=True)
    e[:, 0] = 0.0
    return Rotation.from_euler("zyx", e, degrees=True).as_matrix()


# ---- cv2 restatements ----------------------------------------------------------------

def ellipse_kernel(ksize=10):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (ksize, ksize))."""
    r, c = ksize // 2, ksize // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    k = np.zeros((ksize, ksize), np.uint8)
    for i in range(ksize):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            j1, j2 = max(c - dx, 0), min(c + dx + 1, ksize)
            k[i, j1:j2] = 1
    return k


def erode(mask_bool, ksize=10):
    """cv2.erode(mask, ellipse, iterations=1): anchor at (ksize//2, ksize//2); taps that fall
    outside the image are ignored (erode's default border value is +inf)."""
    k = ellipse_kernel(ksize)
    a = ksize // 2
    H, W = mask_bool.shape
    padded = np.ones((H + ksize, W + ksize), bool)
    padded[a:a + H, a:a + W] = mask_bool
    out = np.ones((H, W), bool)
    for i in range(ksize):
        for j in range(ksize):
            if k[i, j]:
                out &= padded[i:i + H, j:j + W]
    return out


def shrink_mask(mask, kernel_size=3):
    return erode(np.asarray(mask, bool), kernel_size)


def _lanczos4_coeffs(x):
    """OpenCV interpolateLanczos4 (imgproc/src/resize.cpp) with C's evaluation types: ``x + 3`` and
    ``x + 3 - i`` are float32 operations (left to right), the angles and the rotation recurrence double,
    the normalisation float32."""
    import math
    s45 = 0.70710678118654752440084436210485
    cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
    x = np.float32(x)
    xp3 = np.float32(x + np.float32(3))
    y0 = -float(xp3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    co = np.zeros(8, np.float32)
    total = np.float32(0)
    for i in range(8):
        y0_ = np.float32(xp3 - np.float32(i))
        if abs(y0_) >= np.float32(1e-6):
            y = -float(y0_) * math.pi * 0.25
            co[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        else:
            co[i] = np.float32(1e30)
        total = np.float32(total + co[i])
    inv = np.float32(np.float32(1.0) / total)
    return (co * inv).astype(np.float32)


def _resize_scale(n_src, n_dst):
    """cv::resize with an explicit dsize: inv_scale = (double)dst / src; scale = 1. / inv_scale."""
    return 1.0 / (float(n_dst) / float(n_src))


def _axis_table(n_src, n_dst):
    """per destination index: (clamped source indices [8], int16-range fixed-point weights [8])"""
    idx, wt, _ = _axis_table_raw(n_src, n_dst)
    return idx, wt


def _axis_table_raw(n_src, n_dst):
    """-> (clamped source indices [n_dst,8], weights [n_dst,8], first-tap position s [n_dst])"""
    scale = _resize_scale(n_src, n_dst)
    idx = np.zeros((n_dst, 8), np.int64)
    wt = np.zeros((n_dst, 8), np.int64)
    s_all = np.zeros(n_dst, np.int64)
    for d in range(n_dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        co = _lanczos4_coeffs(f)
        q = np.rint(co.astype(np.float32) * np.float32(2048)).astype(np.int64)   # cvRound: round half to even
        wt[d] = np.clip(q, -32768, 32767)
        idx[d] = np.clip(np.arange(s - 3, s + 5), 0, n_src - 1)
        s_all[d] = s
    return idx, wt, s_all


def resize_lanczos4_u8(img, size):
    """cv2.resize(img, (size, size), interpolation=cv2.INTER_LANCZOS4) for uint8 HxW[xC]."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w, _ = img.shape
    if (h, w) == (size, size):
        out = img.copy()
        return out[:, :, 0] if squeeze else out
    xi, xw = _axis_table(w, size)
    yi, yw = _axis_table(h, size)
    src = img.astype(np.int64)
    hor = np.einsum("hdkc,dk->hdc", src[:, xi, :], xw)            # [h, size, C] int, no rounding
    ver = np.einsum("dkwc,dk->dwc", hor[yi, :, :], yw)            # [size, size, C]
    out = np.clip((ver + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def _linear_axis_table(src: int, dst: int):
    """cv2 resize.cpp, INTER_LINEAR coefficient tables for one axis of an 8-bit image: fixed point, 11 bits
    (INTER_RESIZE_COEF_BITS), float32 fraction exactly as `fx = (float)((dx+0.5)*scale - 0.5); sx = cvFloor(fx); fx -= sx`."""
    scale = _resize_scale(src, dst)                             # double
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    fr = (f - s0.astype(np.float32)).astype(np.float32)
    lo = s0 < 0
    fr[lo] = 0; s0[lo] = 0
    hi = s0 >= src - 1
    fr[hi] = 0; s0[hi] = src - 1
    a1 = np.rint(fr * np.float32(2048)).astype(np.int64)        # saturate_cast<short>(float) = round half to even
    a0 = np.rint((np.float32(1) - fr) * np.float32(2048)).astype(np.int64)
    s1 = np.minimum(s0 + 1, src - 1)                            # weight 0 wherever this clamp bites
    return s0, s1, a0, a1


def resize_linear_u8(img, size_wh):
    """cv2.resize(img, (W, H)) (default INTER_LINEAR) for a uint8 HxW image (fast_pose_predictor.py:54):
    horizontal pass in 32-bit ints at scale 2048, vertical pass
        dst = ((b0 * (S0 >> 4) >> 16) + (b1 * (S1 >> 4) >> 16) + 2) >> 2
    (resize.cpp VResizeLinear<uchar,int,short,...>).  cv2 is not installed here: restated from the published source,
    PARITY UNPINNED against cv2 itself.  (cv2 switches an exact 2x downscale to INTER_AREA; not restated.)"""
    img = np.asarray(img, dtype=np.uint8)
    h, w = img.shape
    W, H = size_wh
    if (h, w) == (H, W):
        return img.copy()
    x0, x1, a0, a1 = _linear_axis_table(w, W)
    y0, y1, b0, b1 = _linear_axis_table(h, H)
    src = img.astype(np.int64)
    hor = src[:, x0] * a0 + src[:, x1] * a1                     # [h, W]
    S0, S1 = hor[y0], hor[y1]
    out = (((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def merge_masks(masks, size_wh):
    """fast_pose_predictor.py:50-54: sum the instance masks, clip to [0,1], x255, uint8, resize to the frame."""
    m = np.clip(np.asarray(masks, dtype=np.float32).sum(axis=0), 0, 1) * 255
    return resize_linear_u8(m.astype(np.uint8), size_wh)


# ---- image_manipulation.py -----------------------------------------------------------------

def get_depth_value(bbox, depth, seg_mask, scale=None, near_plane=0.1, far_plane=3.0):
    depth = np.array(depth, dtype=np.float32, copy=True)
    if scale:
        depth *= scale
    good = np.logical_and(depth > near_plane, depth < far_plane)
    m = np.logical_and(seg_mask > 128, good)
    m = erode(m, 10)
    depth *= 1000
    vals, rel = [], []
    for wmin, hmin, wmax, hmax in bbox:
        g = depth[hmin:hmax, wmin:wmax][m[hmin:hmax, wmin:wmax]]
        rel.append(g.shape[0] >= 50)
        vals.append(0 if g.shape[0] == 0 else np.mean(g))
    return np.array(vals) / 1000, np.array(rel)


# ---- predictors --------------------------------------------------------------------------

def crop_batch(rgb, mask, sq_bb, size=512):
    """fast_pose_predictor.py:108-121 -> float64 [N, size, size, 3] in [0,1]."""
    out = []
    for xmin, ymin, xmax, ymax in sq_bb:
        ic = resize_lanczos4_u8(rgb[ymin:ymax, xmin:xmax], size)
        mc = resize_lanczos4_u8(mask[ymin:ymax, xmin:xmax], size)
        out.append(ic * (mc.reshape(size, size, 1) / 255.0))
    return np.array(out) / 255.0


def select_boxes(boxes, frame_shape):
    uv, sq, good = [], [], []
    for bb in boxes:
        xmin, ymin, xmax, ymax = bb
        s = squarify_bb(bb)
        if not bb_in_frame(s, frame_shape):
            continue
        uv.append([(xmax + xmin) / 2, (ymax + ymin) / 2])
        sq.append(s)
        good.append(bb)
    return np.array(uv), np.array(sq), np.array(good).astype(np.int16)


def get_flower_poses(forward_fn, procrustes_fn, rgb, depth_raw, boxes, mask, K, depth_div=1000.0, size=512):
    """fast_pose_predictor.py:60-156 from given detections.  forward_fn: float32 NCHW -> r9."""
    import torch
    uv, sq_bb, good_bb = select_boxes(boxes, rgb.shape)
    if good_bb.shape[0] == 0:
        return None
    depth = depth_raw.astype(np.float32) / depth_div
    dv, rel = get_depth_value(good_bb, depth, mask, near_plane=0.1, far_plane=2.5)
    dv, uv, sq_bb = dv[rel], uv[rel], sq_bb[rel]
    if sq_bb.shape[0] == 0:
        return None
    xyz = get_points3d(uv, dv, K)
    batch = torch.as_tensor(crop_batch(rgb, mask, sq_bb, size), dtype=torch.float32).permute(0, 3, 1, 2)
    R = procrustes_fn(forward_fn(batch)).detach().cpu().numpy()
    R = nullify_yaw_batch(R)
    Rt = np.repeat(np.eye(4)[None], R.shape[0], axis=0)
    Rt[:, :3, :3] = R
    Rt[:, :3, 3] = xyz
    return Rt


def detection_rows(boxes, R):
    """scripts/test_posenet.py:150-161: [xmin,ymin,xmax,ymax,cx,cy,R.flatten()] per flower."""
    rows = []
    for bb, r in zip(boxes, R):
        xmin, ymin, xmax, ymax = bb
        rows.append(list(map(float, bb)) + [(xmin + xmax) / 2, (ymin + ymax) / 2] + np.asarray(r).flatten().tolist())
    return np.array(rows)
