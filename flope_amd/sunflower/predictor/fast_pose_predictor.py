"""Drop-in for the reference's ``sunflower/predictor/fast_pose_predictor.py``
(``FastPosePredictor``, :19-156): detect -> squarify -> depth -> crop batch -> PoseResNet
-> Procrustes -> yaw-null -> Rt.  Everything after the detector runs on the GPU in one
stream with a single device->host copy of the final ``[N,4,4]`` poses.

Detector: the reference constructs ``ultralytics.YOLO(yolo_path)`` (:36) and calls it per frame
(:49).  Here ``yolo_path`` names the same network's weights -- a plain ``state_dict`` file of the
ultralytics model (``tools/export_yolo_state_dict.py``), or the ``.pt`` itself where ultralytics
is installed to unpickle it -- and the detector runs on this build's own kernels
(``flope_amd/yolo.py``: letterbox, YOLO11-seg graph, DFL decode, NMS, mask assembly; C-ABI
``flope_yolo_*``).  ``yolo_path`` may also be a callable ``image -> (bbox int16 [N,4], mask uint8
[H,W])`` (the contract of ``get_bbox_mask``) to plug in any other detector.
"""
from pathlib import Path

import numpy as np
import torch

from flope_amd import _lib
from flope_amd import engine as _engine
from sunflower.models.posenet import PoseResNet
from sunflower.utils.io import read_intrinsics_yaml_to_K_h_w
from sunflower.utils.mvg import bb_in_frame, squarify_bb


def select_boxes(boxes, frame_shape):
    """fast_pose_predictor.py:65-83: keep boxes whose squarified version lies inside the frame.
    -> (uv [n,2] float64 box centres, sq_bb [n,4] int, good_bb [n,4] int16)"""
    uv, sq, good = [], [], []
    for bb in boxes:
        xmin, ymin, xmax, ymax = (int(v) for v in bb)
        s = squarify_bb(bb)
        if not bb_in_frame(s, frame_shape):
            continue
        uv.append([(xmax + xmin) / 2, (ymax + ymin) / 2])
        sq.append(s)
        good.append([xmin, ymin, xmax, ymax])
    return (np.array(uv, dtype=np.float64).reshape(-1, 2), np.array(sq, dtype=np.int64).reshape(-1, 4),
            np.array(good).astype(np.int16).reshape(-1, 4))


def upload_depth(depth, dev):
    """depth image -> device tensor (uint16 bits travel unchanged in an int16 tensor; anything else as float32 metres)"""
    depth_np = np.ascontiguousarray(depth)
    if depth_np.dtype == np.uint16:
        return torch.from_numpy(depth_np.view(np.int16)).to(dev)
    return torch.from_numpy(depth_np.astype(np.float32)).to(dev)


def enqueue_poses(posenet, frame_shape, boxes, K, depth_div, frame_d, mask_d, depth_d, crop_size=512, near=0.1, far=2.5,
                  device=None):
    """Device half of the shared tail of both predictors (fast_pose_predictor.py:65-144, pose_predictor.py:90-174): box
    selection on the host, then depth lift, crops, PoseResNet, Procrustes, yaw-null and Rt assembly enqueued on the
    current stream.  -> float32 [n, 17] on the device (16 pose entries + the depth-reliable flag) or None when no box
    survives squarify / in-frame; nothing is synchronised."""
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("flope_amd predictors run on HIP devices only")
    uv, sq_bb, good_bb = select_boxes(boxes, frame_shape)
    if good_bb.shape[0] == 0:
        return None
    K4 = (K[0][0], K[1][1], K[0][2], K[1][2])
    # both box lists in ONE host -> device copy (each small pageable copy is a synchronous ~30 us call)
    n_box = good_bb.shape[0]
    both = torch.from_numpy(np.concatenate([good_bb.astype(np.int32), sq_bb.astype(np.int32)], axis=0)).to(dev)
    _, reliable, xyz = _engine.depth_lift(depth_d, mask_d, both[:n_box], K4, depth_div, near, far)
    # Every in-frame box goes through the network and the unreliable ones are dropped at the very end: one device ->
    # host round trip per frame instead of two (the reference filters first; crops are independent, so the surviving rows are
    # the same poses up to fp32 summation order).
    sq_all = both[n_box:]
    # crops in the trunk's own 16-bit NHWC layout when there is one: the stem would round the float32 crop to that type
    # anyway (bit-identical result), and the crop tensor is a third of the size
    fmt = {"f16": _lib.IN_F16_NHWC, "bf16": _lib.IN_BF16_NHWC}.get(getattr(posenet, "compute_dtype", "f32"), _lib.IN_F32_NCHW)
    crops = _engine.crop_resize_mask(frame_d, mask_d, sq_all, crop_size, fmt)
    _, R = posenet.predict_rotations(crops)
    Rt = _engine.compose_pose(R, xyz, nullify=True)
    return torch.cat([Rt.reshape(-1, 16), reliable.to(torch.float32).reshape(-1, 1)], dim=1)


def finish_poses(packed):
    """Host half: one device -> host copy, drop the flowers without reliable depth (fast_pose_predictor.py:97-102),
    float64 [N,4,4] or None (:86-87, :101-102)."""
    if packed is None:
        return None
    packed = packed.cpu().numpy()
    keep = packed[:, 16] > 0.5
    if not keep.any():
        return None
    return packed[keep, :16].reshape(-1, 4, 4).astype(np.float64)


def poses_from_detections(posenet, rgb, depth, boxes, mask, K, depth_div, crop_size=512, near=0.1, far=2.5,
                          device=None, frame_d=None, mask_d=None, depth_d=None):
    """Shared tail of both predictors (fast_pose_predictor.py:65-156, pose_predictor.py:90-186).
    frame_d / mask_d / depth_d: inputs that are already on the device (the built-in detector keeps frame and mask there
    and uploads the depth image while the detector is still running)."""
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("flope_amd predictors run on HIP devices only")
    if frame_d is None:
        frame_d = torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.uint8)).to(dev)
    if mask_d is None:
        mask_d = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)).to(dev)
    if depth_d is None:
        depth_d = upload_depth(depth, dev)
    return finish_poses(enqueue_poses(posenet, rgb.shape, boxes, K, depth_div, frame_d, mask_d, depth_d, crop_size, near, far, dev))


class FastPosePredictor:
    def __init__(self, device: str, yolo_path, posenet_path: str, intrin_path: str, debug: bool = False,
                 imgsz: int = None, yolo_dtype: str = "f32"):
        """yolo_dtype: "f32" (default, r05) = the exact-float32 detector on the matrix cores -- the int16 boxes and the uint8 mask
        that reach the pose path (reference :49-56) equal the float32 network's; "f16" / "bf16" = the 16-bit detector (0.3 ms per 1080p
        frame faster; boxes within 2 px, a candidate within 0.01 of the confidence threshold may flip)."""
        self.device = device
        self.debug = debug
        self.crop_size = 512                       # fast_pose_predictor.py:115-116
        self.posenet = PoseResNet().to(device)
        self.posenet.load_state_dict(torch.load(posenet_path, weights_only=True))
        print(f"Model loaded: {Path(posenet_path).name}")
        self.K, self.height, self.width = read_intrinsics_yaml_to_K_h_w(intrin_path)
        if callable(yolo_path):
            self.yolo = None
            self._detector = yolo_path
        else:
            from flope_amd.yolo import YoloSeg, load_yolo_checkpoint
            # a `.pt` is the reference's own argument (`YOLO(yolo_path)`, :36, unpickles it): same trust, explicit by suffix;
            # anything else must be a plain state_dict that the weights_only loader accepts
            sd, ck_imgsz = load_yolo_checkpoint(str(yolo_path), allow_pickle=str(yolo_path).endswith(".pt"))
            # ultralytics predicts at the size the checkpoint was trained with (the reference's is `yolo11nseg_1280.pt`, :177)
            self._yolo_args = (sd, int(imgsz or ck_imgsz or 1280), yolo_dtype)
            self._yolo_more = []               # further detector instances of the pipelined loop, built on first use
            self.yolo = YoloSeg(int(self.height), int(self.width), self._yolo_args[1], yolo_dtype, device=device)
            self.yolo.load_state_dict(sd)
            self._detector = self.yolo.get_bbox_mask
            print(f"YOLO loaded: {Path(str(yolo_path)).name}")
        print("FastPosePredictor initialized!")

    def get_bbox_mask(self, image):
        """-> (bbox int16 [N,4] xyxy, mask uint8 [H,W])   (fast_pose_predictor.py:44-57; no detection: empty bbox and
        an all-zero mask, where the reference raises on `results[0].masks.data`)"""
        return self._detector(image)

    def _frame_ctx(self, slots: int = 1):
        """The flope_frame handle behind this predictor's detector (csrc/frame.hip): rebuilt when the PoseResNet engine it
        borrows was rebuilt (larger batch) or when more slots are needed; parameters re-uploaded after load_state_dict."""
        eng = self.posenet.engine_for(self.device, (self.crop_size, self.crop_size))
        ctx = getattr(self, "_fctx", None)
        if ctx is None or ctx.engine is not eng or ctx.slots < slots:
            from flope_amd.frame import FramePoses
            from flope_amd.yolo import MAX_DET
            if ctx is not None:
                ctx.close()
            ctx = self._fctx = FramePoses(eng, self.yolo.frame_h, self.yolo.frame_w, MAX_DET, max(slots, ctx.slots if ctx is not None else 1))
        return ctx

    def get_flower_poses(self, rgb, depth):
        """rgb uint8 [H,W,3], depth uint16 [H,W] (millimetres) -> float64 [N,4,4] | None"""
        if self.yolo is not None:
            # frame, mask and detections stay on the device; everything behind the detector is ONE C call (flope_frame_to_poses:
            # box selection on the device, a 4-byte count read-back, depth lift, crops, network, Rt, reliability filter)
            det, count, mask_d, frame_d = self.yolo.detect_device(rgb)
            depth_d = upload_depth(depth, torch.device(self.device))   # host copy runs while the GPU is busy with the detector
            return self._frame_ctx().to_poses(det, count, frame_d, mask_d, depth_d, self.K, depth_div=1000.0)
        bb, mask = self.get_bbox_mask(rgb)
        return poses_from_detections(self.posenet, rgb, depth, bb, mask, self.K, depth_div=1000.0,
                                     device=self.device)

    def iter_flower_poses(self, frames, detectors: int = 2):
        """`get_flower_poses` over a stream of (rgb, depth) frames, software-pipelined.  The detector is a chain of ~75 short,
        narrow launches that leaves most of the GPU idle, so `detectors` (1..4, default 2) instances of it work on consecutive frames
        on their own HIP streams (replaying captured hipGraphs: one host call per frame) while crops -> PoseResNet ->
        Procrustes of an earlier frame run on another stream and the uploads of the next frame on a third; the host only ever
        waits for the oldest detector.  Yields exactly what get_flower_poses returns, frame by frame, in order,
        `detectors + 1` frames late.  The reference loop (scripts/live_pose.py:31-41) is sequential; this is the same
        computation with several frames in flight."""
        if self.yolo is None:
            for rgb, depth in frames:
                yield self.get_flower_poses(rgb, depth)
            return
        dev = torch.device(self.device)
        nd = max(1, min(int(detectors), 4))
        while len(self._yolo_more) < nd - 1:
            from flope_amd.yolo import YoloSeg
            sd, imgsz, dt = self._yolo_args
            extra = YoloSeg(self.yolo.frame_h, self.yolo.frame_w, imgsz, dt, device=self.device)
            extra.load_state_dict(sd)
            self._yolo_more.append(extra)
        dets = [self.yolo] + self._yolo_more[:nd - 1]
        s_io, s_pose = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        s_det = [torch.cuda.Stream(dev) for _ in range(nd)]
        H, W = self.yolo.frame_h, self.yolo.frame_w
        NS = 2 * (nd + 1)                               # frames in flight: nd detecting, one in the pose stage, one being uploaded
        slots = [dict(frame=torch.empty((H, W, 3), dtype=torch.uint8, device=dev), out=self.yolo.new_outputs(), depth=None,
                      ready=None, pose_done=None, shape=None) for _ in range(NS)]
        # the slot buffers were filled on the caller's stream: every side stream starts behind that fill
        cur = torch.cuda.current_stream(dev)
        for s_ in [s_io, s_pose] + s_det:
            s_.wait_stream(cur)
        prev_graph = [d.set_option("graph", 1) for d in dets]

        def stage_detect(t, rgb, depth):
            sl = slots[t % NS]
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            if rgb.shape != (H, W, 3):
                raise ValueError(f"expected a uint8 BGR frame [{H},{W},3], got {rgb.shape}")
            with torch.cuda.stream(s_io):
                if sl["pose_done"] is not None:
                    s_io.wait_event(sl["pose_done"])        # the pose stage of frame t - NS has read this slot
                sl["frame"].copy_(torch.from_numpy(rgb), non_blocking=True)
                sl["depth"] = upload_depth(depth, dev)
                sl["depth"].record_stream(s_pose)
                up = torch.cuda.Event()
                up.record(s_io)
            sl["shape"] = rgb.shape
            st = s_det[t % nd]
            with torch.cuda.stream(st):
                st.wait_event(up)
                dets[t % nd].detect_device(sl["frame"], out=sl["out"], in_place=True)
                fctx.select(t % NS, sl["out"][0], sl["out"][1])      # box selection on the device, behind the detector; the count goes to pinned memory
                sl["ready"] = torch.cuda.Event()
                sl["ready"].record(st)

        fctx = self._frame_ctx(NS)

        def stage_pose(t):
            sl = slots[t % NS]
            with torch.cuda.stream(s_pose):
                s_pose.wait_event(sl["ready"])
                fctx.enqueue(t % NS, sl["frame"], sl["out"][2], sl["depth"], self.K, 1000.0)   # waits (host) for frame t's box count only
                sl["pose_done"] = torch.cuda.Event()
                sl["pose_done"].record(s_pose)
            return t % NS

        def finish(slot):
            return fctx.finish(slot)

        try:
            detecting, posing = [], []                 # frame indices / frame-handle slots in flight per stage (oldest first)
            for t, (rgb, depth) in enumerate(frames):
                stage_detect(t, rgb, depth)
                detecting.append(t)
                if len(detecting) > nd:                 # the oldest detector ran while nd newer frames were uploaded and queued
                    posing.append(stage_pose(detecting.pop(0)))   # its pose work queues up BEHIND the frame before it ...
                    if len(posing) > 1:
                        yield finish(posing.pop(0))     # ... whose results the host now waits for (its own event: not the whole stream)
            for u in detecting:
                posing.append(stage_pose(u))
                if len(posing) > 1:
                    yield finish(posing.pop(0))
            for slot in posing:
                yield finish(slot)
        finally:
            torch.cuda.synchronize(dev)
            for d, g in zip(dets, prev_graph):
                d.set_option("graph", g)
