"""YOLO11-seg detector on the GPU: Python owner of one ``flope_yolo_handle``.

Replaces ``self.yolo = YOLO(yolo_path)`` / ``results = self.yolo(image)`` of the reference
(``sunflower/predictor/fast_pose_predictor.py:36,44-57``; network = ultralytics 8.3.27, which the
reference does not vendor).  PyTorch is plumbing (device buffers, the current stream); letterbox,
the whole network, DFL decode, NMS and the mask assembly run in libflope_amd.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_DT = {"f16": _lib.DT_F16, "bf16": _lib.DT_BF16, "f32": _lib.DT_F32}   # f32: strict parity mode (yolo_f32.hip)
MAX_DET = 300


def load_yolo_checkpoint(path: str, allow_pickle: bool = False):
    """-> (state_dict of float tensors, imgsz or None).

    Accepts a plain ``state_dict`` file (``torch.save(model.state_dict(), f)`` -- keys ``model.<i>. ...``; readable
    with ``weights_only=True``, which executes nothing from the file; an optional ``imgsz`` entry carries the training
    size).  An ultralytics ``.pt`` is a pickled model object: opening it EXECUTES code from the file, so that route is
    taken only on request (``allow_pickle=True``) and only where ultralytics is installed; its weights then still run on
    this build's kernels."""
    import pickle
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as exc:         # the safe loader refused the file: a pickled model object, or a hostile file
        if not allow_pickle:
            raise RuntimeError(
                f"{path}: not a plain state_dict file (the weights_only loader refused it). Export the weights once on a "
                "machine that has ultralytics: tools/export_yolo_state_dict.py <model.pt> <out.pth> -- or, for a file you "
                "trust, pass allow_pickle=True where ultralytics is installed.") from exc
        try:
            from ultralytics import YOLO
        except ImportError:
            raise RuntimeError(f"{path}: a pickled ultralytics model needs ultralytics, which is not installed") from exc
        y = YOLO(path)
        sd = {k: v.detach().float().cpu() for k, v in y.model.state_dict().items()}
        imgsz = (getattr(y.model, "args", None) or {}).get("imgsz") if isinstance(getattr(y.model, "args", None), dict) else None
        return sd, imgsz
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = {**obj["state_dict"], **({"imgsz": obj["imgsz"]} if "imgsz" in obj else {})}
    if not isinstance(obj, dict) or "model.0.conv.weight" not in obj:
        raise RuntimeError(f"{path}: expected a yolo11-seg state_dict (keys 'model.<i>. ...')")
    imgsz = obj.get("imgsz")
    imgsz = int(imgsz) if imgsz is not None else None
    sd = {k: v for k, v in obj.items() if isinstance(v, torch.Tensor) and k != "imgsz"}
    return sd, imgsz


class YoloSeg:
    def __init__(self, frame_h: int, frame_w: int, imgsz: int = 1280, dtype: str = "f16", device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("flope_amd: no HIP device visible; the product path has no CPU fallback")
        self.lib = _lib.load()
        idx = None if device is None else torch.device(device).index          # "cuda" without an index = the current device
        self.device = torch.device("cuda", torch.cuda.current_device() if idx is None else idx)
        self.dtype = dtype
        self.frame_h, self.frame_w, self.imgsz = int(frame_h), int(frame_w), int(imgsz)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.flope_yolo_create(self.device.index, self.frame_h, self.frame_w, self.imgsz, _DT[dtype], C.byref(h))
        self._check(rc, None)
        self.handle = h
        ih, iw = C.c_int(), C.c_int()
        self._check(self.lib.flope_yolo_input_size(self.handle, C.byref(ih), C.byref(iw)))
        self.input_hw = (ih.value, iw.value)
        # persistent device buffers: the captured launch graph bakes their addresses in
        self._det = torch.zeros((MAX_DET, 8), dtype=torch.float32, device=self.device)
        self._count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._frame_buf = torch.zeros((self.frame_h, self.frame_w, 3), dtype=torch.uint8, device=self.device)
        self._mask = torch.zeros((self.frame_h, self.frame_w), dtype=torch.uint8, device=self.device)
        # graph capture needs a real stream: when the caller works on the (uncapturable) default stream the detector runs
        # on this one, ordered after / before the caller's stream with two events
        self._own_stream = torch.cuda.Stream(self.device)

    def _check(self, rc, handle="self"):
        if rc != 0:
            h = self.handle if handle == "self" else handle
            msg = self.lib.flope_yolo_last_error(h)
            raise RuntimeError(f"flope_amd yolo error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.flope_yolo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: dict) -> None:
        items = [(k, v.detach().to("cpu", torch.float32).contiguous()) for k, v in sd.items()
                 if isinstance(v, torch.Tensor) and v.is_floating_point() and not k.endswith("num_batches_tracked")]
        n = len(items)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in items])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in items])
        ndims = (C.c_int * n)(*[t.dim() for _, t in items])
        shape_arrs = [(C.c_int64 * max(t.dim(), 1))(*t.shape) for _, t in items]
        shapes = (C.c_void_p * n)(*[C.cast(a, C.c_void_p).value for a in shape_arrs])
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_yolo_load_weights(self.handle, n, names, ptrs, ndims, shapes))

    def _frame(self, frame) -> torch.Tensor:
        """host or device frame -> the engine's own frame buffer (one H2D / D2D copy, no allocation)"""
        if isinstance(frame, np.ndarray):
            frame = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))
        if frame.dtype != torch.uint8 or tuple(frame.shape) != (self.frame_h, self.frame_w, 3):
            raise ValueError(f"expected a uint8 BGR frame [{self.frame_h},{self.frame_w},3], got {frame.dtype} {tuple(frame.shape)}")
        if frame.data_ptr() != self._frame_buf.data_ptr():
            self._frame_buf.copy_(frame, non_blocking=True)
        return self._frame_buf

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, frame) -> None:
        """letterbox + network only (parity taps via read_tensor)."""
        f = self._frame(frame)
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_yolo_forward(self.handle, f.data_ptr(), self._stream()))

    def new_outputs(self):
        """a private (det, count, mask) buffer set for detect_device(out=...): several frames in flight"""
        return (torch.zeros((MAX_DET, 8), dtype=torch.float32, device=self.device),
                torch.zeros(1, dtype=torch.int32, device=self.device),
                torch.zeros((self.frame_h, self.frame_w), dtype=torch.uint8, device=self.device))

    def detect_device(self, frame, conf: float = 0.25, iou: float = 0.7, max_det: int = MAX_DET, out=None, in_place: bool = False):
        """-> (det float32 [max_det,8], count int32 [1], mask uint8 [H,W], frame uint8 [H,W,3]): the engine's own device
        buffers, valid until the next call -- or the buffers of `out` (from new_outputs()).  in_place: `frame` is a device
        tensor that stays untouched until the detector has run; it is read where it is.  Nothing is synchronised."""
        if in_place:
            if not (isinstance(frame, torch.Tensor) and frame.is_cuda and frame.dtype == torch.uint8 and frame.is_contiguous()
                    and tuple(frame.shape) == (self.frame_h, self.frame_w, 3)):
                raise ValueError(f"in_place needs a contiguous uint8 device frame [{self.frame_h},{self.frame_w},3]")
            f = frame
        else:
            f = self._frame(frame)
        det, count, mask = out if out is not None else (self._det, self._count, self._mask)
        if out is not None and not (det.is_cuda and det.dtype == torch.float32 and tuple(det.shape) == (MAX_DET, 8) and det.is_contiguous()
                                    and count.dtype == torch.int32 and count.numel() == 1 and count.is_cuda
                                    and mask.is_cuda and mask.dtype == torch.uint8 and mask.is_contiguous()
                                    and tuple(mask.shape) == (self.frame_h, self.frame_w)):
            raise ValueError("out must come from new_outputs()")
        cur = torch.cuda.current_stream(self.device)
        st = cur
        if cur.cuda_stream == 0:
            st = self._own_stream
            st.wait_stream(cur)
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_yolo_detect(self.handle, f.data_ptr(), float(conf), float(iou), int(max_det),
                                                   det.data_ptr(), count.data_ptr(), mask.data_ptr(), st.cuda_stream))
        if st is not cur:
            cur.wait_stream(st)
        return det, count, mask, f

    def detect(self, frame, conf: float = 0.25, iou: float = 0.7, max_det: int = MAX_DET):
        """-> (boxes float32 [n,4] frame xyxy, conf [n], cls [n], anchor [n], mask uint8 [H,W]) as numpy arrays."""
        det, count, mask, _ = self.detect_device(frame, conf, iou, max_det)
        n = int(count.item())
        d = det[:n].cpu().numpy()
        return d[:, :4].copy(), d[:, 4].copy(), d[:, 5].astype(np.int64), d[:, 6].astype(np.int64), mask.cpu().numpy()

    def get_bbox_mask(self, image, conf: float = 0.25, iou: float = 0.7):
        """fast_pose_predictor.py:44-57: -> (bbox int16 [N,4] xyxy, mask uint8 [H,W])."""
        boxes, _, _, _, mask = self.detect(image, conf, iou)
        return boxes.astype(np.int16), mask

    def read_tensor(self, name: str) -> torch.Tensor:
        dims = (C.c_int64 * 3)()
        self._check(self.lib.flope_yolo_read_tensor(self.handle, name.encode(), None, dims, None))     # size query
        c, h, w = (int(d) for d in dims)
        buf = torch.empty((c, h, w), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_yolo_read_tensor(self.handle, name.encode(), buf.data_ptr(), dims, self._stream()))
        return buf

    def set_option(self, name: str, value: int) -> int:
        rc = self.lib.flope_yolo_set_option(self.handle, name.encode(), int(value))
        if rc < 0:
            self._check(rc)
        return rc

    def profile(self, frame, iters: int = 20) -> str:
        """developer aid: mean microseconds of every launch of the graph, as a text table"""
        f = self._frame(frame)
        buf = C.create_string_buffer(1 << 16)
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_yolo_profile(self.handle, f.data_ptr(), int(iters), buf, len(buf), self._own_stream.cuda_stream))
        return buf.value.decode()

    def flops(self) -> float:
        return float(self.lib.flope_yolo_flops(self.handle))

    def launches(self) -> int:
        return int(self.lib.flope_yolo_launches(self.handle))

    def graph_cache_size(self) -> int:
        return int(self.lib.flope_yolo_graph_cache_size(self.handle))
