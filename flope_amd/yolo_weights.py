"""YOLO11-seg checkpoints for the detector front end (`flope_amd/yolo.py`).

The reference loads ``YOLO(yolo_path)`` from an ultralytics ``.pt`` it does not ship
(``fast_pose_predictor.py:36,177``: ``yolo11nseg_1280.pt``).  This build consumes the
network's plain ``state_dict`` -- keys ``model.<i>. ...`` exactly as ultralytics names them --
stored with ``torch.save`` so that ``torch.load(path, weights_only=True)`` can read it
(``tools/export_yolo_state_dict.py`` converts a ``.pt`` on a machine that has ultralytics).

``synthetic_yolo_state_dict`` builds a seeded random-init checkpoint of the ``n`` scale
(widths 16/32/64/128/256, one repeat per C3k2) for tests and benches: there is no network
here to fetch real weights from and the reference ships none.
"""
from __future__ import annotations

import math

import torch

BN_KEYS = ("weight", "bias", "running_mean", "running_var")
# BatchNorm gamma multipliers that keep the random-init activations O(1) through the neck (measured with the CPU oracle)
_DAMPED = {"model.9.cv2": 0.45, "model.10.cv2": 0.6, "model.13.cv2": 0.6, "model.16.cv2": 0.8, "model.19.cv2": 0.7,
           "model.22.cv2": 0.7, "model.8.cv2": 0.8, "model.6.cv2": 0.9}


def conv_specs(widths=(16, 32, 64, 128, 256), nc: int = 1, nm: int = 32):
    """Ordered list of (prefix, kind, cout, cin, k, groups) for every parameterised module of yolo11{n,s}-seg.
    kind: 'conv' (Conv2d no bias + BatchNorm), 'plain' (Conv2d with bias), 'deconv' (ConvTranspose2d 2x2 s2)."""
    w0, w1, w2, w3, w4 = widths
    out = []

    def conv(p, cout, cin, k=1, g=1):
        out.append((p, "conv", cout, cin, k, g))

    def bottleneck(p, c, e):
        c_ = int(c * e)
        conv(p + ".cv1", c_, c, 3)
        conv(p + ".cv2", c, c_, 3)

    def c3k(p, c):
        c_ = c // 2
        conv(p + ".cv1", c_, c)
        conv(p + ".cv2", c_, c)
        conv(p + ".cv3", c, 2 * c_)
        for j in range(2):
            bottleneck(f"{p}.m.{j}", c_, 1.0)

    def c3k2(p, cin, cout, is_c3k, e):
        c = int(cout * e)
        conv(p + ".cv1", 2 * c, cin)
        conv(p + ".cv2", cout, 3 * c)
        if is_c3k:
            c3k(p + ".m.0", c)
        else:
            bottleneck(p + ".m.0", c, 0.5)

    conv("model.0", w0, 3, 3)
    conv("model.1", w1, w0, 3)
    c3k2("model.2", w1, w2, False, 0.25)
    conv("model.3", w2, w2, 3)
    c3k2("model.4", w2, w3, False, 0.25)
    conv("model.5", w3, w3, 3)
    c3k2("model.6", w3, w3, True, 0.5)
    conv("model.7", w4, w3, 3)
    c3k2("model.8", w4, w4, True, 0.5)
    conv("model.9.cv1", w4 // 2, w4)
    conv("model.9.cv2", w4, 2 * w4)
    c = w4 // 2
    conv("model.10.cv1", 2 * c, w4)
    conv("model.10.cv2", w4, 2 * c)
    conv("model.10.m.0.attn.qkv", 2 * c, c)
    conv("model.10.m.0.attn.proj", c, c)
    conv("model.10.m.0.attn.pe", c, 1, 3, c)
    conv("model.10.m.0.ffn.0", 2 * c, c)
    conv("model.10.m.0.ffn.1", c, 2 * c)
    c3k2("model.13", w4 + w3, w3, False, 0.5)
    c3k2("model.16", w3 + w3, w2, False, 0.5)
    conv("model.17", w2, w2, 3)
    c3k2("model.19", w2 + w3, w3, False, 0.5)
    conv("model.20", w3, w3, 3)
    c3k2("model.22", w3 + w4, w4, True, 0.5)
    ch = (w2, w3, w4)
    c2, c3, c4 = max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100)), max(ch[0] // 4, nm)
    h = "model.23"
    for i, x in enumerate(ch):
        conv(f"{h}.cv2.{i}.0", c2, x, 3)
        conv(f"{h}.cv2.{i}.1", c2, c2, 3)
        out.append((f"{h}.cv2.{i}.2", "plain", 64, c2, 1, 1))
        conv(f"{h}.cv3.{i}.0.0", x, 1, 3, x)
        conv(f"{h}.cv3.{i}.0.1", c3, x)
        conv(f"{h}.cv3.{i}.1.0", c3, 1, 3, c3)
        conv(f"{h}.cv3.{i}.1.1", c3, c3)
        out.append((f"{h}.cv3.{i}.2", "plain", nc, c3, 1, 1))
        conv(f"{h}.cv4.{i}.0", c4, x, 3)
        conv(f"{h}.cv4.{i}.1", c4, c4, 3)
        out.append((f"{h}.cv4.{i}.2", "plain", nm, c4, 1, 1))
    npr = ch[0]
    conv(h + ".proto.cv1", npr, ch[0], 3)
    out.append((h + ".proto.upsample", "deconv", npr, npr, 2, 1))
    conv(h + ".proto.cv2", npr, npr, 3)
    conv(h + ".proto.cv3", nm, npr)
    return out


def synthetic_yolo_state_dict(seed: int = 0, nc: int = 1, cls_bias: float = -4.5, widths=(16, 32, 64, 128, 256)) -> dict:
    """Seeded random-init yolo11n-seg ``state_dict`` (CPU generator: identical on every machine; ``widths`` =
    (32, 64, 128, 256, 512) gives the ``s`` scale).
    He-scaled convolutions, non-trivial BatchNorm statistics, and a class-head bias that lets a small
    fraction of the anchors pass the default 0.25 confidence threshold on a noise frame."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = {}
    for p, kind, cout, cin, k, groups in conv_specs(widths=widths, nc=nc):
        fan_in = cin * k * k
        if kind == "conv":
            # linear (act=False) convs and the closing conv of a residual branch get a smaller gain: nothing
            # contracts their output, and a random-init residual stream would otherwise grow layer after layer
            linear = any(p.endswith(t) for t in (".attn.qkv", ".attn.proj", ".attn.pe", ".ffn.1"))
            closing = p.endswith(".cv2") and ".m." in p.rsplit(".cv2", 1)[0][-6:]
            gain, gamma = (1.0, 0.5) if linear else ((2.0, 0.5) if closing else (2.0, 1.0))
            if p in _DAMPED:                        # block outputs fed by wide concatenations of correlated maps
                gamma *= _DAMPED[p]
            sd[p + ".conv.weight"] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(gain / fan_in)
            sd[p + ".bn.weight"] = (0.9 + 0.4 * torch.rand(cout, generator=g)) * gamma
            sd[p + ".bn.bias"] = 0.1 * torch.randn(cout, generator=g)
            sd[p + ".bn.running_mean"] = 0.1 * torch.randn(cout, generator=g)
            sd[p + ".bn.running_var"] = 0.6 + 0.8 * torch.rand(cout, generator=g)
            sd[p + ".bn.num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)
        elif kind == "plain":
            sd[p + ".weight"] = torch.randn(cout, cin, 1, 1, generator=g) * math.sqrt(1.0 / fan_in)
            sd[p + ".bias"] = 0.1 * torch.randn(cout, generator=g)
            if ".cv3." in p:                      # class logits ~ N(cls_bias, ~1.5^2): a few percent of the anchors pass 0.25
                sd[p + ".bias"] = torch.full((cout,), float(cls_bias))
                sd[p + ".weight"] = sd[p + ".weight"] * 3.0
        else:                                       # ConvTranspose2d weight [cin, cout, 2, 2]
            sd[p + ".weight"] = torch.randn(cin, cout, 2, 2, generator=g) * math.sqrt(2.0 / cin)
            sd[p + ".bias"] = 0.1 * torch.randn(cout, generator=g)
    sd["model.23.dfl.conv.weight"] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)
    return sd


def synthetic_frame(seed: int = 0, H: int = 1080, W: int = 1920):
    """A noise frame with bright discs (uint8 BGR): structure for the detector to respond to."""
    import numpy as np
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 96, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(24):
        r = int(rng.integers(H // 40, H // 12))
        cx, cy = int(rng.integers(r, W - r)), int(rng.integers(r, H - r))
        col = rng.integers(150, 256, 3)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = col
    return img
