"""Oracle: YOLO11-seg detector behind ``FastPosePredictor.get_bbox_mask`` (CPU, torch functional + numpy).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Follows the reference call sites ``sunflower/predictor/fast_pose_predictor.py:36`` (``YOLO(yolo_path)``),
``:44-57`` (``get_bbox_mask``: ``results = self.yolo(image)``, ``masks.data`` summed / clipped / x255 / uint8 /
``cv2.resize((W,H))``, ``boxes.xyxy -> int16``).  The network, its pre- and post-processing live in
``ultralytics==8.3.27`` (``environment.yml:231``), which is NOT vendored by the reference, NOT installed here and
ships no weights or golden vectors with the reference: **PARITY UNPINNED**.  What is restated below is the
published YOLO11-seg algorithm of that package as the build's author knows it:

  * model graph ``yolo11-seg.yaml`` (Conv = conv + BatchNorm(eps 1e-3) + SiLU; C3k2 / C3k / Bottleneck; SPPF;
    C2PSA with position-sensitive attention; nearest 2x upsample + concat neck; Segment head = Detect head with
    DFL (16 bins) + 32 mask coefficients + Proto), modules named as in its ``state_dict`` (``model.<i>. ...``);
    repeat counts and the C3k switch are read from the key set, channel widths from the tensor shapes;
  * ``LetterBox(auto=True, stride=32)`` + BGR->RGB + /255 pre-processing;
  * ``Detect._inference`` (anchors at cell centres, DFL expectation, ``dist2bbox`` xywh, x stride, sigmoid classes);
  * ``ops.non_max_suppression`` (single label per box, class-offset boxes, torchvision-style greedy NMS in score
    order, ``max_det``), ``ops.process_mask`` (coef . proto, crop at proto resolution, bilinear upsample to the
    letterboxed input, ``> 0``) and ``ops.scale_boxes`` (undo the letterbox, clip).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import pipeline_ref as P

BN_EPS = 1e-3            # ultralytics initialises every BatchNorm2d with eps = 1e-3
REG_MAX = 16
NM = 32


# ---- building blocks -----------------------------------------------------------------------------
# forward_layers(..., emulate=torch.float16 / torch.bfloat16) restates WHERE the 16-bit device path (flope_amd/csrc/yolo*.hip) rounds:
# BatchNorm folded into the conv weights in float32 and the folded weights stored in the 16-bit type (depthwise weights and every
# bias stay float32), float32 accumulation, activation (+ residual) in float32, ONE rounding when a map is stored.  It pins the
# bf16 mode far tighter than the float32 forward can (8 mantissa bits over 23 layers): tests/test_gpu_yolo.py.
_EMU = None


def _r(t):
    return t if _EMU is None else t.to(_EMU).float()


def conv(sd, p, x, k=1, s=1, act=True, groups=1, res=None):
    """ultralytics ``Conv``: Conv2d(bias=False, padding=k//2) -> BatchNorm2d -> SiLU (+ res: a shortcut added behind the activation)."""
    w = sd[p + ".conv.weight"]
    if _EMU is None:
        y = F.conv2d(x, w, None, stride=s, padding=w.shape[-1] // 2, groups=groups)
        y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                         training=False, eps=BN_EPS)
    else:
        scale = sd[p + ".bn.weight"] / torch.sqrt(sd[p + ".bn.running_var"] + BN_EPS)
        wf = w * scale[:, None, None, None]
        if groups == 1:
            wf = _r(wf)
        y = F.conv2d(x, wf, sd[p + ".bn.bias"] - sd[p + ".bn.running_mean"] * scale, stride=s, padding=w.shape[-1] // 2, groups=groups)
    y = F.silu(y) if act else y
    return _r(y if res is None else y + res)


def bottleneck(sd, p, x):
    """Bottleneck(c, c, shortcut=True, k=(3,3)): x + cv2(cv1(x))."""
    return conv(sd, p + ".cv2", conv(sd, p + ".cv1", x), res=x)


def c3k(sd, p, x):
    """C3k (a C3 with 3x3 bottlenecks, e = 1.0 inside): cv3(cat(m(cv1(x)), cv2(x)))."""
    y = conv(sd, p + ".cv1", x)
    i = 0
    while f"{p}.m.{i}.cv1.conv.weight" in sd:
        y = bottleneck(sd, f"{p}.m.{i}", y)
        i += 1
    return conv(sd, p + ".cv3", torch.cat([y, conv(sd, p + ".cv2", x)], 1))


def c3k2(sd, p, x):
    """C3k2 (C2f whose inner modules are Bottlenecks or C3k blocks)."""
    y = list(conv(sd, p + ".cv1", x).chunk(2, 1))
    i = 0
    while f"{p}.m.{i}.cv1.conv.weight" in sd:
        inner = f"{p}.m.{i}"
        y.append(c3k(sd, inner, y[-1]) if f"{inner}.cv3.conv.weight" in sd else bottleneck(sd, inner, y[-1]))
        i += 1
    return conv(sd, p + ".cv2", torch.cat(y, 1))


def sppf(sd, p, x):
    y = [conv(sd, p + ".cv1", x)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    return conv(sd, p + ".cv2", torch.cat(y, 1))


def attention(sd, p, x):
    """ultralytics ``Attention(dim, num_heads = dim // 64, attn_ratio = 0.5)``."""
    B, C, H, W = x.shape
    nh = C // 64
    hd = C // nh
    kd = hd // 2
    N = H * W
    qkv = conv(sd, p + ".qkv", x, act=False)
    q, k, v = qkv.view(B, nh, kd * 2 + hd, N).split([kd, kd, hd], dim=2)
    attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = _r(_r((v @ attn.transpose(-2, -1)).view(B, C, H, W)) + conv(sd, p + ".pe", v.reshape(B, C, H, W), act=False, groups=C))
    return conv(sd, p + ".proj", y, act=False)


def c2psa(sd, p, x):
    c = sd[p + ".cv1.conv.weight"].shape[0] // 2
    a, b = conv(sd, p + ".cv1", x).split((c, c), 1)
    i = 0
    while f"{p}.m.{i}.attn.qkv.conv.weight" in sd:
        q = f"{p}.m.{i}"
        b = _r(b + attention(sd, q + ".attn", b))
        b = conv(sd, q + ".ffn.1", conv(sd, q + ".ffn.0", b), act=False, res=b)
        i += 1
    return conv(sd, p + ".cv2", torch.cat([a, b], 1))


def _plain(sd, p, x):
    return F.conv2d(x, _r(sd[p + ".weight"]), sd[p + ".bias"])      # head rows: float32 outputs


def forward_layers(sd: dict, x: torch.Tensor, emulate=None) -> dict:
    """fp32 eval-mode forward of the 24-entry graph.  -> {layer index: output, 'box','cls','coef' per level, 'proto'}
    emulate = a 16-bit torch dtype: with the device path's rounding points (see _EMU above)."""
    global _EMU
    _EMU = emulate
    try:
        return _forward_layers(sd, x)
    finally:
        _EMU = None


def _forward_layers(sd: dict, x: torch.Tensor) -> dict:
    sd = {k: v.float() for k, v in sd.items() if v.is_floating_point()}
    o = {}
    o[0] = conv(sd, "model.0", _r(x.float()), s=2)
    o[1] = conv(sd, "model.1", o[0], s=2)
    o[2] = c3k2(sd, "model.2", o[1])
    o[3] = conv(sd, "model.3", o[2], s=2)
    o[4] = c3k2(sd, "model.4", o[3])
    o[5] = conv(sd, "model.5", o[4], s=2)
    o[6] = c3k2(sd, "model.6", o[5])
    o[7] = conv(sd, "model.7", o[6], s=2)
    o[8] = c3k2(sd, "model.8", o[7])
    o[9] = sppf(sd, "model.9", o[8])
    o[10] = c2psa(sd, "model.10", o[9])
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    o[13] = c3k2(sd, "model.13", torch.cat([up(o[10]), o[6]], 1))
    o[16] = c3k2(sd, "model.16", torch.cat([up(o[13]), o[4]], 1))
    o[17] = conv(sd, "model.17", o[16], s=2)
    o[19] = c3k2(sd, "model.19", torch.cat([o[17], o[13]], 1))
    o[20] = conv(sd, "model.20", o[19], s=2)
    o[22] = c3k2(sd, "model.22", torch.cat([o[20], o[10]], 1))
    h = "model.23"
    for i, f in enumerate((o[16], o[19], o[22])):
        o[f"box{i}"] = _plain(sd, f"{h}.cv2.{i}.2", conv(sd, f"{h}.cv2.{i}.1", conv(sd, f"{h}.cv2.{i}.0", f)))
        t = conv(sd, f"{h}.cv3.{i}.0.1", conv(sd, f"{h}.cv3.{i}.0.0", f, groups=f.shape[1]))
        t = conv(sd, f"{h}.cv3.{i}.1.1", conv(sd, f"{h}.cv3.{i}.1.0", t, groups=t.shape[1]))
        o[f"cls{i}"] = _plain(sd, f"{h}.cv3.{i}.2", t)
        o[f"coef{i}"] = _plain(sd, f"{h}.cv4.{i}.2", conv(sd, f"{h}.cv4.{i}.1", conv(sd, f"{h}.cv4.{i}.0", f)))
    t = conv(sd, h + ".proto.cv1", o[16])
    t = _r(F.conv_transpose2d(t, _r(sd[h + ".proto.upsample.weight"]), sd[h + ".proto.upsample.bias"], stride=2))
    o["proto_up"] = t
    o["proto"] = conv(sd, h + ".proto.cv3", conv(sd, h + ".proto.cv2", t))
    return o


def decode(o: dict, strides=(8, 16, 32)):
    """Detect._inference + the mask-coefficient rows: -> pred float32 [A, 4 + nc + 32] (xywh in letterbox pixels,
    sigmoid class scores, coefficients), anchors in level-major / row-major order."""
    rows = []
    for i, s in enumerate(strides):
        box, cls, coef = o[f"box{i}"][0], o[f"cls{i}"][0], o[f"coef{i}"][0]
        _, h, w = box.shape
        sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5,
                                indexing="ij")
        anc = torch.stack([sx, sy], -1).view(-1, 2)
        d = box.view(4, REG_MAX, h * w).softmax(1)
        d = (d * torch.arange(REG_MAX, dtype=torch.float32).view(1, -1, 1)).sum(1).t()        # [A,4] l t r b
        x1y1, x2y2 = anc - d[:, :2], anc + d[:, 2:]
        xywh = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], 1) * s
        rows.append(torch.cat([xywh, cls.view(cls.shape[0], -1).t().sigmoid(), coef.view(NM, -1).t()], 1))
    return torch.cat(rows, 0)


def nms_numpy(boxes, scores, iou_thres):
    """torchvision.ops.nms (CPU kernel): stable sort by descending score, greedy suppression with
    inter / (area_i + area_j - inter) > thr, float32 arithmetic.  -> kept indices in score order."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    order = np.argsort(-scores, kind="stable")
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    dead = np.zeros(len(boxes), bool)
    keep = []
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(int(i))
        rest = order[a + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = (w * h).astype(np.float32)
        ovr = inter / (areas[i] + areas[rest] - inter)
        dead[rest[ovr > np.float32(iou_thres)]] = True
    return np.array(keep, np.int64)


def candidates(pred, nc):
    """First half of ops.non_max_suppression for one image: xywh2xyxy, best class per anchor.
    -> (box float32 [A,4] xyxy, conf float32 [A], cls int64 [A])"""
    pred = np.asarray(pred, np.float32)
    xy, wh = pred[:, :2], pred[:, 2:4] / np.float32(2)
    box = np.concatenate([xy - wh, xy + wh], 1).astype(np.float32)
    return box, pred[:, 4:4 + nc].max(1), pred[:, 4:4 + nc].argmax(1)


def nms_candidates(box, conf, cls, conf_thres=0.25, iou_thres=0.7, max_det=300, max_wh=7680, max_nms=30000):
    """Second half: confidence filter (anchor order kept), class-offset boxes, torchvision-style NMS, max_det.
    -> kept anchor indices in NMS order.  max_nms: ultralytics keeps the 30,000 most confident candidates; the device
    path keeps 4,096 (pass max_nms=4096 to restate it: the most confident ones, ties at the cut in anchor order)."""
    idx = np.nonzero(conf > np.float32(conf_thres))[0]
    if idx.shape[0] == 0:
        return idx
    if idx.shape[0] > max_nms:
        idx = np.sort(idx[np.argsort(-conf[idx], kind="stable")[:max_nms]])
    off = (cls[idx].astype(np.float32) * np.float32(max_wh))[:, None]
    keep = nms_numpy(box[idx] + off, conf[idx], iou_thres)[:max_det]
    return idx[keep]


def non_max_suppression(pred, nc, conf_thres=0.25, iou_thres=0.7, max_det=300, max_wh=7680, max_nms=30000):
    """ops.non_max_suppression for one image, single label per box.  pred [A, 4+nc+32] as `decode` returns.
    -> (det float32 [n, 6 + 32]: xyxy, conf, cls, coefficients; anchor indices [n])"""
    pred = np.asarray(pred, np.float32)
    box, conf, cls = candidates(pred, nc)
    keep = nms_candidates(box, conf, cls, conf_thres, iou_thres, max_det, max_wh, max_nms)
    det = np.concatenate([box[keep], conf[keep, None], cls[keep, None].astype(np.float32), pred[keep, 4 + nc:]], 1).astype(np.float32)
    return det.reshape(-1, 6 + NM), keep


def process_mask(proto, coef, boxes, shape, return_float=False):
    """ops.process_mask(protos [32,mh,mw], masks_in [n,32], bboxes [n,4] letterbox xyxy, shape (ih,iw), upsample=True)
    -> float32 {0,1} [n, ih, iw]   (return_float: the interpolated values before the `> 0` threshold, for tests that
    need to know how close to the threshold a pixel is)"""
    proto = torch.as_tensor(proto, dtype=torch.float32)
    c, mh, mw = proto.shape
    ih, iw = shape
    m = (torch.as_tensor(coef, dtype=torch.float32) @ proto.view(c, -1)).view(-1, mh, mw)
    b = torch.as_tensor(boxes, dtype=torch.float32).clone()
    b[:, 0] *= mw / iw
    b[:, 2] *= mw / iw
    b[:, 1] *= mh / ih
    b[:, 3] *= mh / ih
    r = torch.arange(mw, dtype=torch.float32)[None, None, :]
    cc = torch.arange(mh, dtype=torch.float32)[None, :, None]
    x1, y1, x2, y2 = (b[:, i, None, None] for i in range(4))
    m = m * ((r >= x1) * (r < x2) * (cc >= y1) * (cc < y2))
    m = F.interpolate(m[None], (ih, iw), mode="bilinear", align_corners=False)[0]
    return m if return_float else (m > 0).float()


def letterbox_geometry(h0, w0, imgsz=1280, stride=32):
    """LetterBox(new_shape=(imgsz, imgsz), auto=True, scaleup=True, stride): -> (new_w, new_h, top, bottom, left, right)"""
    r = min(imgsz / h0, imgsz / w0)
    nw, nh = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = (imgsz - nw) % stride, (imgsz - nh) % stride
    dw, dh = dw / 2, dh / 2
    return nw, nh, int(round(dh - 0.1)), int(round(dh + 0.1)), int(round(dw - 0.1)), int(round(dw + 0.1))


def letterbox(img_bgr, imgsz=1280, stride=32):
    """uint8 BGR [H,W,3] -> uint8 BGR letterboxed [h,w,3] (cv2.resize INTER_LINEAR + copyMakeBorder(114))."""
    h0, w0 = img_bgr.shape[:2]
    nw, nh, top, bottom, left, right = letterbox_geometry(h0, w0, imgsz, stride)
    im = img_bgr
    if (w0, h0) != (nw, nh):
        im = np.stack([P.resize_linear_u8(img_bgr[:, :, c], (nw, nh)) for c in range(3)], -1)
    out = np.full((nh + top + bottom, nw + left + right, 3), 114, np.uint8)
    out[top:top + nh, left:left + nw] = im
    return out


def preprocess(img_bgr, imgsz=1280):
    """BasePredictor.preprocess: letterbox -> BGR to RGB -> CHW -> float32 / 255.  -> [1,3,h,w]"""
    lb = letterbox(img_bgr, imgsz)
    return torch.from_numpy(np.ascontiguousarray(lb[..., ::-1].transpose(2, 0, 1))).float()[None] / 255


def scale_boxes(img1_shape, boxes, img0_shape):
    """ops.scale_boxes(..., padding=True) + clip_boxes: letterbox xyxy -> original-frame xyxy (float32)."""
    boxes = np.array(boxes, np.float32, copy=True)
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    boxes[:, [0, 2]] -= np.float32(pad[0])
    boxes[:, [1, 3]] -= np.float32(pad[1])
    boxes[:, :4] /= np.float32(gain)
    boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, img0_shape[1])
    boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, img0_shape[0])
    return boxes


def detect(sd, img_bgr, imgsz=1280, conf=0.25, iou=0.7, max_det=300):
    """`self.yolo(image)[0]` as far as get_bbox_mask uses it: -> (boxes.xyxy float32 [n,4] in frame pixels, conf [n],
    masks.data float32 {0,1} [n, h, w] at the letterboxed resolution)."""
    x = preprocess(img_bgr, imgsz)
    o = forward_layers(sd, x)
    nc = o["cls0"].shape[1]
    det, _ = non_max_suppression(decode(o).numpy(), nc, conf, iou, max_det)
    if det.shape[0] == 0:
        return np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), np.zeros((0,) + tuple(x.shape[2:]), np.float32)
    masks = process_mask(o["proto"][0], det[:, 6:], det[:, :4], tuple(x.shape[2:])).numpy()
    return scale_boxes(tuple(x.shape[2:]), det[:, :4], img_bgr.shape), det[:, 4], masks


def get_bbox_mask(sd, img_bgr, imgsz=1280, conf=0.25, iou=0.7, max_det=300):
    """fast_pose_predictor.py:44-57 on top of `detect`: -> (bbox int16 [n,4], mask uint8 [H,W])."""
    H, W = img_bgr.shape[:2]
    boxes, _, masks = detect(sd, img_bgr, imgsz, conf, iou, max_det)
    if masks.shape[0] == 0:             # the reference raises here (results[0].masks is None); the build returns "nothing found"
        return np.zeros((0, 4), np.int16), np.zeros((H, W), np.uint8)
    return boxes.astype(np.int16), P.merge_masks(masks, (W, H))
