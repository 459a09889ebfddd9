"""FramePoses: Python owner of one ``flope_frame_handle`` (include/flope_amd.h, csrc/frame.hip) -- everything
``FastPosePredictor.get_flower_poses`` does behind the detector (fast_pose_predictor.py:55-56, 60-156) as one C call per stage:
box selection on the device, then depth lift, crops, PoseResNet, Procrustes, yaw-null, Rt and the reliability filter.
PyTorch only lends the device pointers and the current stream."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import PoseEngine, _stream_ptr


class FramePoses:
    def __init__(self, engine: PoseEngine, frame_h: int, frame_w: int, max_boxes: int = 300, slots: int = 1):
        self.lib = _lib.load()
        self.engine = engine                     # kept alive: the handle borrows it
        self.device = engine.device
        self.frame_h, self.frame_w, self.max_boxes, self.slots = int(frame_h), int(frame_w), int(max_boxes), int(slots)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.flope_frame_create(engine.handle, self.frame_h, self.frame_w, self.max_boxes, self.slots, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"flope_frame_create: {(self.lib.flope_frame_last_error(None) or b'').decode()}")
        self.handle = h
        self._out = np.empty((self.max_boxes, 4, 4), dtype=np.float64)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.flope_frame_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise RuntimeError((self.lib.flope_frame_last_error(self.handle) or b"").decode() or f"flope_frame error {rc}")
        return rc

    @staticmethod
    def _depth_format(depth_d: torch.Tensor) -> int:
        if depth_d.dtype == torch.float32:
            return 1
        if depth_d.dtype in (torch.int16, torch.uint16):     # int16 storage of uint16 bits is accepted
            return 0
        raise ValueError(f"depth must be uint16 or float32, got {depth_d.dtype}")

    def _check_inputs(self, det, count, frame_d, mask_d, depth_d):
        dev = self.device
        ok = (det.is_cuda and det.device == dev and det.dtype == torch.float32 and det.is_contiguous() and det.dim() == 2 and det.shape[1] == 8
              and count.device == dev and count.dtype == torch.int32 and count.numel() >= 1)
        if not ok:
            raise ValueError("det must be contiguous float32 [max_det, 8] and count int32 [1] on the engine's device")
        if frame_d is not None:
            if not (frame_d.device == dev and frame_d.dtype == torch.uint8 and frame_d.is_contiguous() and tuple(frame_d.shape) == (self.frame_h, self.frame_w, 3)):
                raise ValueError(f"frame must be contiguous uint8 [{self.frame_h},{self.frame_w},3] on {dev}")
            if not (mask_d.device == dev and mask_d.dtype == torch.uint8 and mask_d.is_contiguous() and tuple(mask_d.shape) == (self.frame_h, self.frame_w)):
                raise ValueError(f"mask must be contiguous uint8 [{self.frame_h},{self.frame_w}] on {dev}")
            if not (depth_d.device == dev and depth_d.is_contiguous() and tuple(depth_d.shape) == (self.frame_h, self.frame_w)):
                raise ValueError(f"depth must be contiguous [{self.frame_h},{self.frame_w}] on {dev}")

    # -- the three stages (several frames in flight: one slot each) -----------------------------------------------
    def select(self, slot: int, det: torch.Tensor, count: torch.Tensor) -> None:
        self._check_inputs(det, count, None, None, None)
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_frame_select(self.handle, slot, det.data_ptr(), count.data_ptr(), det.shape[0], _stream_ptr(self.device)))

    def enqueue(self, slot: int, frame_d, mask_d, depth_d, K, depth_div: float = 1000.0, near: float = 0.1, far: float = 2.5) -> int:
        k4 = (C.c_float * 4)(float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
        with torch.cuda.device(self.device):
            return self._check(self.lib.flope_frame_enqueue(self.handle, slot, frame_d.data_ptr(), mask_d.data_ptr(), depth_d.data_ptr(),
                                                            self._depth_format(depth_d), float(depth_div), k4, float(near), float(far),
                                                            _stream_ptr(self.device)))

    def finish(self, slot: int):
        """-> float64 [N,4,4] or None (fast_pose_predictor.py:86-87, :101-102)"""
        n = self._check(self.lib.flope_frame_finish(self.handle, slot, self._out.ctypes.data, self.max_boxes))
        return self._out[:n].copy() if n else None

    def to_poses(self, det, count, frame_d, mask_d, depth_d, K, depth_div: float = 1000.0, near: float = 0.1, far: float = 2.5):
        """Sequential form (one frame): -> float64 [N,4,4] or None."""
        self._check_inputs(det, count, frame_d, mask_d, depth_d)
        k4 = (C.c_float * 4)(float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
        with torch.cuda.device(self.device):
            n = self._check(self.lib.flope_frame_to_poses(self.handle, det.data_ptr(), count.data_ptr(), det.shape[0], frame_d.data_ptr(),
                                                          mask_d.data_ptr(), depth_d.data_ptr(), self._depth_format(depth_d), float(depth_div), k4,
                                                          float(near), float(far), self._out.ctypes.data, self.max_boxes, _stream_ptr(self.device)))
        return self._out[:n].copy() if n else None

    def read_boxes(self, slot: int = 0):
        """test hook: (good_bb int32 [n,4], sq_bb int32 [n,4]) of the last select() of this slot"""
        good = np.empty((self.max_boxes, 4), dtype=np.int32)
        sq = np.empty((self.max_boxes, 4), dtype=np.int32)
        n = self._check(self.lib.flope_frame_read_boxes(self.handle, slot, good.ctypes.data, sq.ctypes.data, self.max_boxes))
        return good[:n].copy(), sq[:n].copy()
