"""Drop-in for the reference's ``sunflower/predictor/pose_predictor.py`` (``PosePredictor``,
:40-186).  Same pipeline as FastPosePredictor with the teacher front end
(GroundingDINO boxes -> ``filter_very_large_bb`` -> SAM mask) and the RealSense-D405 depth
scale (/10000, :118).

Front end.  The reference builds ``GroundingDINO(device, 'white flower.', box_th=0.3,
text_th=0.3, obj_filter='white flower')`` and ``SAM(device)`` itself (:55-60); both are thin
wrappers over Hugging Face hub models (``sunflower/models/grounding_dino.py:8-10``,
``sunflower/models/sam.py:10-11``) and out of this build's scope.  The mirror
does the same thing the same way: with the reference call shape
``PosePredictor(device, posenet_path, intrin_path, debug)`` it imports
``sunflower.models.grounding_dino.GroundingDINO`` and ``sunflower.models.sam.SAM`` -- the
``sunflower`` packages are namespace packages (no ``__init__.py`` on either side), so with
the reference tree on ``PYTHONPATH`` *behind* this mirror those two modules resolve to the
reference's own files -- and constructs them with the reference's arguments.  Only if that
import fails does the constructor raise.  ``detector=`` / ``segmenter=`` inject callables
instead (``detector(rgb) -> boxes [N,4]``, ``segmenter(rgb, boxes) -> mask uint8 [H,W]``).
"""
import logging
from pathlib import Path

import numpy as np
import torch

from sunflower.models.posenet import PoseResNet
from sunflower.predictor.fast_pose_predictor import poses_from_detections
from sunflower.utils.io import read_intrinsics_yaml_to_K_h_w
from sunflower.utils.mvg import filter_very_large_bb

log = logging.getLogger(__name__)


class PosePredictor:
    def __init__(self, device: str, posenet_path: str, intrin_path: str, debug: bool = False,
                 detector=None, segmenter=None):
        self.device = device
        self.debug = debug
        self.posenet = PoseResNet().to(device)
        self.posenet.load_state_dict(torch.load(posenet_path, weights_only=True))
        log.info(f"Model loaded: {Path(posenet_path).name}")
        self.gdino = self.sam = None
        if detector is None or segmenter is None:
            try:                                   # pose_predictor.py:29-30,55-60
                from sunflower.models.grounding_dino import GroundingDINO
                from sunflower.models.sam import SAM
            except ImportError as exc:
                raise ImportError(
                    "PosePredictor: sunflower.models.grounding_dino / sunflower.models.sam are not importable. They are "
                    "the reference's Hugging Face wrappers (GroundingDINO-tiny, SAM ViT-H), not part of flope_amd: put the "
                    "reference tree on PYTHONPATH behind flope_amd/ (they resolve through the `sunflower` namespace "
                    "package), or pass detector= and segmenter= callables") from exc
        if detector is None:
            self.gdino = GroundingDINO(device, 'white flower.', box_th=0.3, text_th=0.3, obj_filter='white flower')
            log.info("Grounding DINO loaded")
            detector = self.gdino.detect
        if segmenter is None:
            self.sam = SAM(device)
            log.info("SAM loaded")
            segmenter = self._sam_mask
        self.detector, self.segmenter = detector, segmenter
        self.K, self.height, self.width = read_intrinsics_yaml_to_K_h_w(intrin_path)
        log.info("PosePredictor initialized!")

    def _sam_mask(self, rgb, boxes):
        from PIL import Image                      # pose_predictor.py:87-88
        return self.sam.get_segmentation_mask(Image.fromarray(rgb), boxes)

    def get_flower_poses(self, rgb, depth):
        """rgb uint8 [H,W,3], depth uint16 [H,W] (1e-4 m units) -> float64 [N,4,4] | None"""
        bb = np.asarray(self.detector(rgb))
        if bb.shape[0] == 0:                       # :76-78 (the reference's detector returns shape (0,) then)
            return None
        bb = filter_very_large_bb(bb)              # :83
        mask = self.segmenter(rgb, bb.tolist())    # :87-88
        return poses_from_detections(self.posenet, rgb, depth, bb, mask, self.K, depth_div=10000.0,
                                     device=self.device)
