"""Intrinsics reader with the reference's semantics (``sunflower/utils/io.py:86-98``):
yaml keys fx, fy, cx, cy, h, w -> (K 3x3, h, w)."""
import numpy as np
import yaml


def read_intrinsics_yaml(filepath: str):
    with open(filepath, "r") as f:
        return yaml.safe_load(f)


def read_intrinsics_yaml_to_K_h_w(filepath: str):
    d = read_intrinsics_yaml(filepath)
    K = np.array([[d["fx"], 0, d["cx"]], [0, d["fy"], d["cy"]], [0, 0, 1]])
    return K, d["h"], d["w"]
