"""Drop-in for the reference's ``sunflower/predictor/pose_predictor.py`` (``PosePredictor``,
:40-186).  Same pipeline as FastPosePredictor with the teacher front end
(GroundingDINO boxes -> ``filter_very_large_bb`` -> SAM mask) and the RealSense-D405 depth
scale (/10000, :118).  GroundingDINO and SAM are Hugging Face hub models fetched by name
in the reference (models/grounding_dino.py:8-10, models/sam.py:10-11); they cannot be
loaded offline, so the front end is injected: ``detector(rgb) -> boxes [N,4]`` and
``segmenter(rgb, boxes) -> mask uint8 [H,W]``.
"""
import numpy as np
import torch

from sunflower.models.posenet import PoseResNet
from sunflower.predictor.fast_pose_predictor import poses_from_detections
from sunflower.utils.io import read_intrinsics_yaml_to_K_h_w
from sunflower.utils.mvg import filter_very_large_bb


class PosePredictor:
    def __init__(self, device: str, posenet_path: str, intrin_path: str, debug: bool = False,
                 detector=None, segmenter=None):
        self.device = device
        self.debug = debug
        self.posenet = PoseResNet().to(device)
        self.posenet.load_state_dict(torch.load(posenet_path, weights_only=True))
        if detector is None or segmenter is None:
            raise RuntimeError("PosePredictor: GroundingDINO / SAM are hub-fetched models that are not "
                               "available offline; pass detector= and segmenter= callables")
        self.detector, self.segmenter = detector, segmenter
        self.K, self.height, self.width = read_intrinsics_yaml_to_K_h_w(intrin_path)

    def get_flower_poses(self, rgb, depth):
        """rgb uint8 [H,W,3], depth uint16 [H,W] (1e-4 m units) -> float64 [N,4,4] | None"""
        bb = np.asarray(self.detector(rgb))
        if bb.shape[0] == 0:
            return None
        bb = filter_very_large_bb(bb)
        mask = self.segmenter(rgb, bb.tolist())
        return poses_from_detections(self.posenet, rgb, depth, bb, mask, self.K, depth_div=10000.0,
                                     device=self.device)
