#!/usr/bin/env python3
"""Detector front end timing (BASELINE configs[2], detector leg): one 1080p frame resident on the device ->
flope_yolo_detect (letterbox, YOLO11n-seg at imgsz 1280, decode, NMS, masks, resize).  Synthetic weights and frame."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--profile-iters", type=int, default=0, help="run only this many detects (for rocprofv3)")
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1, help="0: one launch per op in program order")
    ap.add_argument("--xcd", type=int, default=-1, help="XCD band remap of conv workgroups: 0 never, 1 3x3 convs (default), 2 also 1x1")
    ap.add_argument("--opt", action="append", default=[], help="name=value engine option (repeatable)")
    ap.add_argument("--per-launch", action="store_true", help="print the per-launch time table (HIP events) and exit")
    args = ap.parse_args()
    from flope_amd.yolo import YoloSeg
    from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict
    y = YoloSeg(1080, 1920, 1280, args.dtype)
    y.load_state_dict(synthetic_yolo_state_dict(0))
    y.set_option("graph", args.graph)
    y.set_option("batch", args.batch)
    if args.xcd >= 0:
        y.set_option("xcd", args.xcd)
    for kv in args.opt:
        k, v = kv.split("=")
        y.set_option(k, int(v))
    frame = torch.from_numpy(synthetic_frame(0)).cuda()
    if args.per_launch:
        print(y.profile(frame, 20))
        return
    n = args.profile_iters or args.iters
    for _ in range(3 if args.profile_iters else 10):
        y.detect_device(frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        det, count, mask, _ = y.detect_device(frame)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"detector_ms_per_frame": round(dt * 1e3, 4), "frames_per_s": round(1 / dt, 1), "detections": int(count.item()),
                      "launches": y.launches(), "gflop_per_frame": round(y.flops() / 1e9, 2),
                      "tflops": round(y.flops() / dt / 1e12, 2), "input": list(y.input_hw), "dtype": args.dtype, "graph": args.graph, "batch": args.batch}))


if __name__ == "__main__":
    main()
