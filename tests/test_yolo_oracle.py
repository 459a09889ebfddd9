"""CPU: the YOLO11-seg oracle (oracle/yolo_ref.py) against hand-derived known answers of the published ultralytics
8.3.27 algorithm, and the host-side pieces of the detector front end.  ultralytics is absent and the reference holds no
detector fixtures: PARITY UNPINNED against ultralytics itself -- these tests pin the restatement's own arithmetic."""
import numpy as np
import pytest
import torch

from oracle import pipeline_ref as P
from oracle import yolo_ref as Y


@pytest.fixture(scope="module")
def ysd():
    from flope_amd.yolo_weights import synthetic_yolo_state_dict
    return synthetic_yolo_state_dict(0)


def test_letterbox_geometry_kats():
    # 1080p at imgsz 1280: r = 2/3 -> 1280 x 720, dh = (1280 - 720) % 32 / 2 = 8 -> 736 x 1280 (LetterBox auto=True)
    assert Y.letterbox_geometry(1080, 1920, 1280) == (1280, 720, 8, 8, 0, 0)
    assert Y.letterbox_geometry(480, 640, 640) == (640, 480, 0, 0, 0, 0)          # no resize, no padding
    assert Y.letterbox_geometry(300, 500, 320) == (320, 192, 0, 0, 0, 0)
    nw, nh, t, b, l, r = Y.letterbox_geometry(333, 517, 640)
    assert (nh + t + b) % 32 == 0 and (nw + l + r) % 32 == 0 and abs(t - b) <= 1
    img = np.full((1080, 1920, 3), 200, np.uint8)
    lb = Y.letterbox(img, 1280)
    assert lb.shape == (736, 1280, 3) and (lb[:8] == 114).all() and (lb[-8:] == 114).all() and (lb[8:-8] == 200).all()
    x = Y.preprocess(img, 1280)
    assert x.shape == (1, 3, 736, 1280) and x.dtype == torch.float32 and float(x.max()) == pytest.approx(200 / 255)
    bgr = np.zeros((64, 64, 3), np.uint8); bgr[..., 0] = 255                       # pure blue in BGR -> channel 2 after BGR->RGB
    assert float(Y.preprocess(bgr, 64)[0, 2].min()) == 1.0 and float(Y.preprocess(bgr, 64)[0, 0].max()) == 0.0


def test_nms_known_answers():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], np.float32)
    # IoU(0,1) = 81/119 = 0.68 < 0.7 kept; IoU(0,3) = 100/105 = 0.95 suppressed; ties keep input order (stable sort)
    assert Y.nms_numpy(boxes, scores, 0.7).tolist() == [0, 1, 2]
    assert Y.nms_numpy(boxes, scores, 0.5).tolist() == [0, 2]
    assert Y.nms_numpy(boxes[:0], scores[:0], 0.5).tolist() == []


def test_decode_known_answer():
    """One level, 2 x 3 cells, stride 8: DFL logits peaked at bin k give distance k; box = anchor -+ distances."""
    o = {}
    H, W = 2, 3
    box = torch.full((1, 64, H, W), -50.0)
    for s, k in enumerate((2, 1, 3, 4)):                  # l t r b
        box[0, s * 16 + k] = 50.0
    o["box0"], o["cls0"], o["coef0"] = box, torch.zeros(1, 1, H, W), torch.ones(1, 32, H, W)
    for i in (1, 2):
        o[f"box{i}"], o[f"cls{i}"], o[f"coef{i}"] = torch.zeros(1, 64, 0, 0), torch.zeros(1, 1, 0, 0), torch.zeros(1, 32, 0, 0)
    pred = Y.decode(o).numpy()
    assert pred.shape == (6, 4 + 1 + 32)
    # anchor (x=1.5, y=0.5): x1 = -0.5, y1 = -0.5, x2 = 4.5, y2 = 4.5 -> cx 2, cy 2, w 5, h 5, times 8
    np.testing.assert_allclose(pred[1, :4], [16, 16, 40, 40], atol=1e-5)
    np.testing.assert_allclose(pred[:, 4], 0.5)
    det, idx = Y.non_max_suppression(pred, 1, conf_thres=0.25, iou_thres=0.7)
    assert idx.tolist() == sorted(idx.tolist()) and det.shape[1] == 38 and (det[:, 5] == 0).all()


def test_scale_boxes_and_process_mask_kats():
    b = Y.scale_boxes((736, 1280), np.array([[0, 8, 1280, 728], [640, 368, 1400, 900]], np.float32), (1080, 1920, 3))
    np.testing.assert_allclose(b, [[0, 0, 1920, 1080], [960, 540, 1920, 1080]], atol=1e-3)
    proto = torch.zeros(32, 8, 8); proto[0] = 1.0
    coef = torch.zeros(2, 32); coef[0, 0] = 1.0; coef[1, 0] = -1.0
    m = Y.process_mask(proto, coef, np.array([[8, 8, 24, 24], [0, 0, 32, 32]], np.float32), (32, 32)).numpy()
    assert m.shape == (2, 32, 32) and m[1].sum() == 0                      # negative everywhere -> empty
    ys, xs = np.nonzero(m[0])
    assert ys.min() >= 6 and ys.max() <= 25 and m[0][12:20, 12:20].all()   # inside the box, bled by < one proto pixel


def test_forward_shapes_and_module_wiring(ysd):
    from flope_amd.yolo_weights import synthetic_frame
    img = synthetic_frame(1, 256, 320)
    x = Y.preprocess(img, 320)
    o = Y.forward_layers(ysd, x)
    assert o[0].shape == (1, 16, 128, 160) and o[22].shape == (1, 256, 8, 10) and o["proto"].shape == (1, 32, 64, 80)
    assert o["box1"].shape == (1, 64, 16, 20) and o["cls2"].shape == (1, 1, 8, 10) and o["coef0"].shape == (1, 32, 32, 40)
    pred = Y.decode(o)
    assert pred.shape == (32 * 40 + 16 * 20 + 8 * 10, 37) and torch.isfinite(pred).all()
    # C3k2 = C2f wiring: recompute layer 2 by hand from its parts
    sd = {k: v.float() for k, v in ysd.items() if v.is_floating_point()}
    y = Y.conv(sd, "model.2.cv1", o[1])
    a, b = y.chunk(2, 1)
    bb = b + Y.conv(sd, "model.2.m.0.cv2", Y.conv(sd, "model.2.m.0.cv1", b))
    np.testing.assert_allclose(Y.conv(sd, "model.2.cv2", torch.cat([a, b, bb], 1)).numpy(), o[2].numpy(), atol=1e-6)


def test_get_bbox_mask_contract(ysd):
    from flope_amd.yolo_weights import synthetic_frame
    img = synthetic_frame(2, 360, 640)
    bb, mask = Y.get_bbox_mask(ysd, img, imgsz=640)
    assert bb.dtype == np.int16 and bb.ndim == 2 and bb.shape[1] == 4 and mask.shape == (360, 640) and mask.dtype == np.uint8
    assert bb.shape[0] >= 1 and set(np.unique(mask)) <= set(range(256)) and (mask == 255).any()
    bb0, mask0 = Y.get_bbox_mask(ysd, img, imgsz=640, conf=0.9999)          # nothing found: defined behaviour of the build
    assert bb0.shape == (0, 4) and not mask0.any()


class _NotATensor:
    """stands in for the model object of an ultralytics .pt: a class the weights_only unpickler does not know"""


def test_synthetic_checkpoint_key_set(ysd, tmp_path):
    """ultralytics naming: `model.<i>.` prefixes, Conv = .conv/.bn, plain head convs = .weight/.bias, DFL buffer."""
    keys = set(ysd)
    for k in ("model.0.conv.weight", "model.2.m.0.cv1.bn.running_var", "model.6.m.0.m.1.cv2.conv.weight", "model.9.cv2.conv.weight",
              "model.10.m.0.attn.pe.conv.weight", "model.10.m.0.ffn.1.bn.bias", "model.23.cv2.0.2.bias", "model.23.cv3.2.0.0.conv.weight",
              "model.23.cv4.1.2.weight", "model.23.proto.upsample.weight", "model.23.dfl.conv.weight"):
        assert k in keys, k
    assert ysd["model.10.m.0.attn.pe.conv.weight"].shape == (128, 1, 3, 3) and ysd["model.23.proto.upsample.weight"].shape == (64, 64, 2, 2)
    assert sum(v.numel() for k, v in ysd.items() if k.endswith(("conv.weight", ".weight", ".bias")) and "bn" not in k or ".bn.weight" in k or ".bn.bias" in k) > 2.5e6
    from flope_amd.yolo import load_yolo_checkpoint
    f = tmp_path / "yolo.pth"
    torch.save({**ysd, "imgsz": torch.tensor(640)}, f)
    sd, imgsz = load_yolo_checkpoint(str(f))
    assert imgsz == 640 and set(sd) == keys
    torch.save({"not": "a model"}, f)
    with pytest.raises(RuntimeError, match="state_dict"):
        load_yolo_checkpoint(str(f))
    # a pickled object (what an ultralytics .pt is) is refused by the safe loader and NOT escalated to a code-executing unpickle
    torch.save({"model": _NotATensor()}, f)
    with pytest.raises(RuntimeError, match="allow_pickle"):
        load_yolo_checkpoint(str(f))
