#!/usr/bin/env python3
"""Convert an ultralytics YOLO11-seg checkpoint (.pt, a pickled model object) into the plain state_dict file
flope_amd's detector loads with ``torch.load(..., weights_only=True)``.  Needs ultralytics (not installed in the
build container): run it once wherever the reference's own environment (environment.yml) exists.

    python tools/export_yolo_state_dict.py yolo11nseg_1280.pt yolo11nseg_1280.state_dict.pth
"""
import sys

import torch


def main():
    src, dst = sys.argv[1], sys.argv[2]
    from ultralytics import YOLO          # third-party; reference fast_pose_predictor.py:12,36
    y = YOLO(src)
    sd = {k: v.detach().float().cpu() for k, v in y.model.state_dict().items()}
    args = getattr(y.model, "args", None)
    if isinstance(args, dict) and args.get("imgsz"):
        sd["imgsz"] = torch.tensor(int(args["imgsz"]))
    torch.save(sd, dst)
    print(f"wrote {dst}: {len(sd)} entries")


if __name__ == "__main__":
    main()
