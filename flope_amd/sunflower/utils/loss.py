"""Drop-in for the reference's ``sunflower/utils/loss.py`` (:3-18): the rotation-error metric."""
import math

import torch


def diff_quats(partices: torch.Tensor, gt: torch.Tensor):
    """(N,4),(N,4) unit quaternions -> (dot in [-1,1], angle error in degrees in [0,180])"""
    dot = (partices * gt).sum(dim=-1).clamp(-1.0, 1.0)
    return dot, torch.acos(dot.abs()) * (360.0 / math.pi)
