"""PoseEngine: Python owner of one ``flope_handle`` (one GPU, one crop size).

PyTorch is plumbing here: it owns device buffers and the current HIP stream; all
compute is in libflope_amd.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .weights import validate_state_dict

_DTYPES = {"bf16": _lib.DT_BF16, "f16": _lib.DT_F16, "f32": _lib.DT_F32,
           torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16, torch.float32: _lib.DT_F32}

STAGES = {"stem": _lib.STAGE_STEM, "pool": _lib.STAGE_POOL, "feat": _lib.STAGE_FEAT,
          "hidden": _lib.STAGE_HIDDEN,
          **{f"layer{li}.{bi}": _lib.STAGE_LAYER(li, bi) for li in range(1, 5) for bi in range(2)}}


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("flope_amd: no HIP device visible; the product path has no CPU fallback")


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def input_format(x: torch.Tensor) -> int:
    """Map a crop-batch tensor onto a FLOPE_IN_* code (include/flope_amd.h)."""
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D crop batch, got shape {tuple(x.shape)}")
    if x.dtype == torch.float32 and x.shape[1] == 3:
        return _lib.IN_F32_NCHW
    if x.shape[-1] == 3 and x.dtype == torch.bfloat16:
        return _lib.IN_BF16_NHWC
    if x.shape[-1] == 3 and x.dtype == torch.float16:
        return _lib.IN_F16_NHWC
    if x.shape[-1] == 3 and x.dtype == torch.uint8:
        return _lib.IN_U8_NHWC
    raise ValueError(f"unsupported crop batch: dtype {x.dtype}, shape {tuple(x.shape)} "
                     "(float32 [B,3,H,W] or bf16/f16/uint8 [B,H,W,3])")


class PoseEngine:
    def __init__(self, height: int, width: int, max_batch: int, dtype="f16", device=None,
                 backbone_out_dim: int = 2048):
        _require_gpu()
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        self.height, self.width, self.max_batch = int(height), int(width), int(max_batch)
        self.dtype_code = _DTYPES[dtype]
        self.backbone_out_dim = int(backbone_out_dim)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.flope_create(self.device.index, self.height, self.width, self.max_batch,
                                       self.dtype_code, self.backbone_out_dim, C.byref(h))
        _lib.check(rc)
        self.handle = h
        self._keep = None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.flope_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------
    def load_state_dict(self, sd: dict) -> None:
        validate_state_dict(sd, self.backbone_out_dim)
        items = [(k, v.detach().to("cpu", torch.float32).contiguous()) for k, v in sd.items()
                 if not k.endswith("num_batches_tracked")]
        n = len(items)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in items])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in items])
        ndims = (C.c_int * n)(*[t.dim() for _, t in items])
        shape_arrs = [(C.c_int64 * max(t.dim(), 1))(*t.shape) for _, t in items]
        shapes = (C.c_void_p * n)(*[C.cast(a, C.c_void_p).value for a in shape_arrs])
        with torch.cuda.device(self.device):
            rc = self.lib.flope_load_weights(self.handle, n, names, ptrs, ndims, shapes)
        _lib.check(rc, self.handle)

    # -- hot path --------------------------------------------------------------
    def _check_input(self, x: torch.Tensor) -> int:
        fmt = input_format(x)
        if not x.is_cuda or x.device != self.device:
            raise RuntimeError(f"crop batch must live on {self.device} (got {x.device}); no CPU path")
        hw = tuple(x.shape[2:]) if fmt == _lib.IN_F32_NCHW else tuple(x.shape[1:3])
        if hw != (self.height, self.width):
            raise ValueError(f"engine built for {self.height}x{self.width} crops, got {hw}")
        if x.shape[0] > self.max_batch:
            raise ValueError(f"batch {x.shape[0]} exceeds max_batch {self.max_batch}")
        return fmt

    def forward(self, x: torch.Tensor, want_r9: bool = True, want_R: bool = True):
        """-> (r9 [B,9] | None, R [B,3,3] | None), float32 on the engine's device."""
        fmt = self._check_input(x)
        x = x.contiguous()
        B = x.shape[0]
        r9 = torch.empty((B, 9), dtype=torch.float32, device=self.device) if want_r9 else None
        R = torch.empty((B, 3, 3), dtype=torch.float32, device=self.device) if want_R else None
        rc = self.lib.flope_forward(self.handle, x.data_ptr(), fmt, B,
                                    r9.data_ptr() if want_r9 else None,
                                    R.data_ptr() if want_R else None, _stream_ptr(self.device))
        _lib.check(rc, self.handle)
        self._keep = x
        return r9, R

    def _check_into(self, x: torch.Tensor, fmt: int, **outs) -> None:
        """Host-only checks of the allocation-free entry points (no device work): device, layout, size of every
        caller-owned buffer -- a wrong one would be a silent out-of-bounds access on the device."""
        if self._check_input(x) != fmt:
            raise ValueError(f"fmt {fmt} does not describe a {x.dtype} tensor of shape {tuple(x.shape)}")
        if not x.is_contiguous():
            raise ValueError("crop batch must be contiguous")
        B = x.shape[0]
        for name, (t, per) in outs.items():
            if t is None:
                continue
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < B * per:
                raise ValueError(f"{name}: need a contiguous float32 buffer of >= {B * per} elements on {self.device}")

    def forward_into(self, x: torch.Tensor, fmt: int, r9: torch.Tensor | None, R: torch.Tensor | None) -> None:
        """Allocation-free variant for timed loops (buffers owned by the caller)."""
        self._check_into(x, fmt, r9=(r9, 9), R=(R, 9))
        rc = self.lib.flope_forward(self.handle, x.data_ptr(), fmt, x.shape[0],
                                    r9.data_ptr() if r9 is not None else None,
                                    R.data_ptr() if R is not None else None, _stream_ptr(self.device))
        _lib.check(rc, self.handle)

    def forward_poses_into(self, x: torch.Tensor, fmt: int, xyz: torch.Tensor | None, nullify: bool, Rt: torch.Tensor,
                           R: torch.Tensor | None = None) -> None:
        """Allocation-free: crops -> [B,16] poses (Procrustes, optional yaw-null, Rt assembly in the head kernel)."""
        self._check_into(x, fmt, xyz=(xyz, 3), Rt=(Rt, 16), R=(R, 9))
        rc = self.lib.flope_forward_poses(self.handle, x.data_ptr(), fmt, x.shape[0],
                                          xyz.data_ptr() if xyz is not None else None, int(bool(nullify)), None,
                                          R.data_ptr() if R is not None else None, Rt.data_ptr(), _stream_ptr(self.device))
        _lib.check(rc, self.handle)

    def extract_features(self, x: torch.Tensor) -> torch.Tensor:
        fmt = self._check_input(x)
        x = x.contiguous()
        out = torch.empty((x.shape[0], self.backbone_out_dim), dtype=torch.float32, device=self.device)
        rc = self.lib.flope_extract_features(self.handle, x.data_ptr(), fmt, x.shape[0], out.data_ptr(),
                                             _stream_ptr(self.device))
        _lib.check(rc, self.handle)
        self._keep = x
        return out

    # -- introspection ---------------------------------------------------------
    def read_stage(self, name: str, batch: int) -> torch.Tensor:
        code = STAGES[name]
        dims = (C.c_int64 * 4)()
        # size the destination generously from the engine geometry: ask once with a probe buffer
        cap = batch * 64 * ((self.height + 1) // 2) * ((self.width + 1) // 2)
        cap = max(cap, batch * max(512, self.backbone_out_dim))
        buf = torch.empty(cap, dtype=torch.float32, device=self.device)
        rc = self.lib.flope_read_stage(self.handle, code, batch, buf.data_ptr(), dims, _stream_ptr(self.device))
        _lib.check(rc, self.handle)
        shape = [int(d) for d in dims]
        n = shape[0] * shape[1] * shape[2] * shape[3]
        out = buf[:n].view(shape)
        return out.reshape(shape[0], shape[1]) if code in (_lib.STAGE_FEAT, _lib.STAGE_HIDDEN) else out

    def set_option(self, name: str, value: int) -> int:
        rc = self.lib.flope_set_option(self.handle, name.encode(), int(value))
        if rc < 0:
            _lib.check(rc, self.handle)
        return rc

    # schedule candidates of autotune(): engine options whose best value differs from box to box and from batch to batch
    # (same-run pairs move by up to +-2 %, and boxes of one pool differ by up to 12 % in clock)
    # (one slice instead of two is not a candidate: below ~130 crops it would switch split-K on, which sums in another order)
    # "split38" / "split716": slice 0 = 3/8 or 7/16 of the batch as an ABSOLUTE image count on a multiple of 8 (engine option
    # split = 100 + images), so that every layer's tiles stay whole in both slices -- a percentage would give 97 / 159 at B = 256
    # (ADVICE r4): resolved per batch by tune_candidates()
    TUNE_CANDIDATES = ({}, {"w4cw": 4, "w4cwf": 1}, {"lag": 0}, {"split": "3/8"}, {"split": "7/16"})

    @classmethod
    def tune_candidates(cls, batch: int):
        """TUNE_CANDIDATES with the slice fractions resolved for `batch` (candidates that coincide with the default are dropped)."""
        out = []
        for c in cls.TUNE_CANDIDATES:
            c = dict(c)
            if isinstance(c.get("split"), str):
                num, den = (int(v) for v in c["split"].split("/"))
                first = (batch * num // den) & ~7
                if first < 8 or first >= batch or first == ((batch // 2) & ~7):
                    continue
                c["split"] = 100 + first
            if c not in out:
                out.append(c)
        return out

    def autotune(self, x: torch.Tensor, fmt: int, rounds: int = 5, per_round: int = 5, candidates=None) -> dict:
        """Pick the launch schedule for THIS device and THIS batch by measurement: every candidate option set runs `per_round`
        forwards per round, candidates interleaved over `rounds` rounds (so that none of them owns the slow steps of a GPU that
        is still ramping its clock up), timed with events on the caller's stream; the set with the smallest median step is left
        in force.  ~rounds x per_round x len(candidates) forwards (a hundred milliseconds at B = 256).  Results do not depend on
        the choice: every candidate is bit-identical (tests/test_gpu_parity.py).  If a forward fails the options the engine had
        on entry are restored before the exception propagates."""
        cands = [dict(c) for c in (candidates if candidates is not None else self.tune_candidates(x.shape[0]))]
        R = torch.empty((x.shape[0], 9), dtype=torch.float32, device=self.device)
        self._check_into(x, fmt, R=(R, 9))
        names = sorted({k for c in cands for k in c})
        base = {k: self.set_option(k, 0) for k in names}                   # current values (set_option returns the previous one)
        for k, v in base.items():
            self.set_option(k, v)
        times = [[] for _ in cands]
        chosen = None
        try:
            with torch.cuda.device(self.device):
                for _ in range(rounds):
                    for ci, c in enumerate(cands):
                        for k in names:
                            self.set_option(k, c.get(k, base[k]))
                        self.forward_into(x, fmt, None, R)                      # (first forward of a new option set: not timed)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(per_round):
                            self.forward_into(x, fmt, None, R)
                        e1.record()
                        e1.synchronize()
                        times[ci].append(e0.elapsed_time(e1) / per_round)
            med = [sorted(t)[len(t) // 2] for t in times]
            chosen = cands[min(range(len(cands)), key=lambda i: med[i])]
        finally:
            for k in names:
                self.set_option(k, (chosen if chosen is not None else base).get(k, base[k]))
        return {"chosen": chosen, "median_ms": {str(c): round(m, 4) for c, m in zip(cands, med)}}

    def flops(self, batch: int) -> float:
        return float(self.lib.flope_forward_flops(self.handle, batch))

    def launches(self) -> int:
        return int(self.lib.flope_forward_launches(self.handle))

    def launch_info(self, batch: int):
        """[(layer, kernel, flops)] for every launch of one forward."""
        out = []
        for i in range(self.launches()):
            name = C.create_string_buffer(160)
            fl = C.c_double()
            _lib.check(self.lib.flope_launch_info(self.handle, i, batch, name, 160, C.byref(fl)), self.handle)
            layer, kern = name.value.decode().split("|")
            out.append((layer, kern, fl.value))
        return out

    def profile_read(self):
        """per-launch GPU milliseconds of the last forward (needs set_option('profile', 1))."""
        n = self.launches()
        ms = (C.c_float * n)()
        rc = self.lib.flope_profile_read(self.handle, ms, n)
        if rc < 0:
            _lib.check(rc, self.handle)
        return [float(ms[i]) for i in range(rc)]

    def describe_plan(self) -> str:
        buf = C.create_string_buffer(8192)
        _lib.check(self.lib.flope_describe_plan(self.handle, buf, 8192), self.handle)
        return buf.value.decode()


# ---- handle-less kernels ---------------------------------------------------------

def _as_dev_f32(t: torch.Tensor, cols: int):
    _require_gpu()
    src_dev = t.device
    d = t.detach().reshape(-1, cols).to("cuda" if not t.is_cuda else t.device, torch.float32).contiguous()
    return d, src_dev


def procrustes(M: torch.Tensor) -> torch.Tensor:
    """special_procrustes on the GPU: [...,9] or [...,3,3] -> [N,3,3] (same dtype/device as M)."""
    d, src = _as_dev_f32(M, 9)
    out = torch.empty_like(d)
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().flope_procrustes(d.data_ptr(), out.data_ptr(), d.shape[0], _stream_ptr(d.device)))
    return out.view(-1, 3, 3).to(device=src, dtype=M.dtype if M.is_floating_point() else torch.float32)


def nullify_yaw(R: torch.Tensor) -> torch.Tensor:
    d, src = _as_dev_f32(R, 9)
    out = torch.empty_like(d)
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().flope_nullify_yaw(d.data_ptr(), out.data_ptr(), d.shape[0], _stream_ptr(d.device)))
    return out.view(-1, 3, 3).to(device=src, dtype=R.dtype)


def compose_pose(R: torch.Tensor, xyz: torch.Tensor | None, nullify: bool) -> torch.Tensor:
    d, _ = _as_dev_f32(R, 9)
    x = None if xyz is None else xyz.detach().reshape(-1, 3).to(d.device, torch.float32).contiguous()
    out = torch.empty((d.shape[0], 4, 4), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().flope_compose_pose(d.data_ptr(), x.data_ptr() if x is not None else None, d.shape[0],
                                                  int(bool(nullify)), out.data_ptr(), _stream_ptr(d.device)))
    return out


def crop_resize_mask(frame: torch.Tensor, mask: torch.Tensor, boxes: torch.Tensor, size: int,
                     out_format: int = _lib.IN_F32_NCHW) -> torch.Tensor:
    """frame uint8 [H,W,3], mask uint8 [H,W], boxes int32 [N,4] (all on one GPU)."""
    _require_gpu()
    assert frame.is_cuda and frame.dtype == torch.uint8 and mask.dtype == torch.uint8
    H, W = mask.shape
    boxes = boxes.to(frame.device, torch.int32).contiguous()
    n = boxes.shape[0]
    if out_format == _lib.IN_F32_NCHW:
        out = torch.empty((n, 3, size, size), dtype=torch.float32, device=frame.device)
    else:
        dt = torch.bfloat16 if out_format == _lib.IN_BF16_NHWC else torch.float16
        out = torch.empty((n, size, size, 3), dtype=dt, device=frame.device)
    mask = mask.to(frame.device)
    with torch.cuda.device(frame.device):
        _lib.check(_lib.load().flope_crop_resize_mask(frame.contiguous().data_ptr(), mask.contiguous().data_ptr(), H, W,
                                                      boxes.data_ptr(), n, size, out_format, out.data_ptr(),
                                                      _stream_ptr(frame.device)))
    return out


def merge_masks_resize(masks: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Instance masks float32 [n,h,w] on the GPU -> uint8 [H,W] frame mask (fast_pose_predictor.py:50-54:
    sum, clip to [0,1], x255, uint8, cv2.resize default bilinear), all on the device."""
    _require_gpu()
    if masks.dim() != 3:
        raise ValueError(f"expected masks [n,h,w], got {tuple(masks.shape)}")
    if not masks.is_cuda:
        raise RuntimeError("masks must live on the GPU; no CPU path")
    m = masks.to(torch.float32).contiguous()
    n, h, w = m.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=m.device)
    scratch = torch.empty(h * w, dtype=torch.uint8, device=m.device)
    with torch.cuda.device(m.device):
        _lib.check(_lib.load().flope_merge_masks_resize(m.data_ptr() if n else None, n, h, w, scratch.data_ptr(),
                                                        out.data_ptr(), H, W, _stream_ptr(m.device)))
    return out


def lanczos4_table(n_src: int, n_dst: int, device="cuda"):
    """Test hook: one axis of the crop kernel's Lanczos-4 tables -> (first tap position int32 [n_dst], int16 weights
    [n_dst,8]), evaluated on the device exactly as flope_crop_resize_mask does."""
    _require_gpu()
    dev = torch.device(device)
    s0 = torch.empty(n_dst, dtype=torch.int32, device=dev)
    co = torch.empty((n_dst, 8), dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().flope_lanczos4_table(int(n_src), int(n_dst), s0.data_ptr(), co.data_ptr(), _stream_ptr(dev)))
    return s0, co


def depth_lift(depth_raw: torch.Tensor, mask: torch.Tensor, boxes: torch.Tensor, K4, depth_div: float,
               near: float, far: float):
    """depth uint16 (raw) or float32 [H,W], mask uint8 [H,W], boxes int32 [N,4]; metres = depth / depth_div
    -> (depth_val [N], reliable [N] bool, xyz [N,3])."""
    _require_gpu()
    dev = depth_raw.device
    H, W = mask.shape
    boxes = boxes.to(dev, torch.int32).contiguous()
    n = boxes.shape[0]
    scratch = torch.empty(H * W + 16 + 512 * n, dtype=torch.uint8, device=dev)   # valid mask + strip partials
    dv = torch.empty(n, dtype=torch.float32, device=dev)
    rel = torch.empty(n, dtype=torch.int32, device=dev)
    xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    k = (C.c_float * 4)(*[float(v) for v in K4])
    d16 = depth_raw.contiguous()
    if d16.dtype == torch.float32:
        fmt = 1
    elif d16.dtype in (torch.uint16, torch.int16):      # int16 storage of uint16 bits is accepted
        fmt = 0
    else:
        raise ValueError(f"depth must be uint16 or float32, got {d16.dtype}")
    mask = mask.to(dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().flope_depth_lift(d16.data_ptr(), fmt, mask.contiguous().data_ptr(), H, W, float(depth_div),
                                                float(near), float(far), boxes.data_ptr(), n, k, scratch.data_ptr(),
                                                dv.data_ptr(), rel.data_ptr(), xyz.data_ptr(), _stream_ptr(dev)))
    return dv, rel.bool(), xyz
