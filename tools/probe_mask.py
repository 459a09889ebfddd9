"""Detector and pose network on CU-masked streams: alone and side by side, for several splits of the 256 CUs.
Developer probe for FastPosePredictor.iter_flower_poses.   python tools/probe_mask.py"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flope_amd")]
from flope_amd import _lib  # noqa: E402
from flope_amd.weights import synthetic_state_dict  # noqa: E402
from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict  # noqa: E402
from sunflower.predictor import fast_pose_predictor as F  # noqa: E402


def masked_stream(lib, cus, n_total=256):
    """cus: iterable of CU indices"""
    words = (n_total + 31) // 32
    m = (C.c_uint32 * words)()
    for c in cus:
        m[c // 32] |= 1 << (c % 32)
    h = C.c_void_p()
    rc = lib.flope_stream_create_cu_mask(0, m, words, C.byref(h))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


def main():
    lib = _lib.load()
    tmp = tempfile.mkdtemp()
    ckpt, intr, yolo_f = (os.path.join(tmp, n) for n in ("posenet.pth", "intrinsics.yaml", "yolo.pth"))
    torch.save(synthetic_state_dict(0), ckpt)
    open(intr, "w").write(yaml.safe_dump(dict(fx=1400.0, fy=1400.0, cx=960.0, cy=540.0, h=1080, w=1920)))
    torch.save({**synthetic_yolo_state_dict(0), "imgsz": torch.tensor(1280)}, yolo_f)
    rgb = synthetic_frame(0)
    depth = (400 + np.random.default_rng(0).normal(0, 4, rgb.shape[:2])).astype(np.uint16)
    pred = F.FastPosePredictor("cuda", yolo_f, ckpt, intr)
    dev = torch.device("cuda")
    y = pred.yolo
    for _ in range(3):
        pred.get_flower_poses(rgb, depth)
    fd = y._frame(rgb)
    det, count, mask_d, frame_d = y.detect_device(fd)
    depth_d = F.upload_depth(depth, dev)
    bb = det[:int(count.item()), :4].cpu().numpy().astype(np.int16)
    fb, mb = frame_d.clone(), mask_d.clone()
    torch.cuda.synchronize()

    def run_det(st, n):
        with torch.cuda.stream(st):
            for _ in range(n):
                y.detect_device(fd)

    def run_pose(st, n):
        with torch.cuda.stream(st):
            for _ in range(n):
                F.enqueue_poses(pred.posenet, rgb.shape, bb, pred.K, 1000.0, fb, mb, depth_d, device=dev)

    def timed(fn, n=30):
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    plain_a, plain_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    print(f"unmasked: detector {timed(lambda n: run_det(plain_a, n)):.3f} ms, pose stage {timed(lambda n: run_pose(plain_b, n)):.3f} ms, "
          f"side by side {timed(lambda n: (run_det(plain_a, n), run_pose(plain_b, n))):.3f} ms per (frame, frame)", flush=True)
    hi = torch.cuda.Stream(dev, priority=-1)
    print(f"detector on a high-priority stream, side by side {timed(lambda n: (run_det(hi, n), run_pose(plain_b, n))):.3f} ms", flush=True)
    for nd, layout in ((32, "low"), (64, "low"), (96, "low"), (128, "low"), (64, "stride4"), (64, "xcd01")):
        if layout == "low":
            dcus = list(range(nd))
        elif layout == "stride4":
            dcus = list(range(0, 256, 4))
        else:
            dcus = [c for c in range(256) if c % 8 < 2]
        pcus = [c for c in range(256) if c not in set(dcus)]
        sd_, sp_ = masked_stream(lib, dcus), masked_stream(lib, pcus)
        td = timed(lambda n: run_det(sd_, n))
        tp = timed(lambda n: run_pose(sp_, n))
        tb = timed(lambda n: (run_det(sd_, n), run_pose(sp_, n)))
        print(f"detector on {len(dcus):3d} CUs ({layout}) {td:.3f} ms | pose stage on {len(pcus):3d} CUs {tp:.3f} ms | side by side {tb:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
