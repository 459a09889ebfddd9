"""Drop-in for the reference's ``sunflower/models/posenet.py`` (PoseResNet, :5-34).

Same constructor, same 124-entry ``state_dict`` key set, same call signature
(float32 ``[B,3,H,W]`` in -> ``[B,9]`` out), but the forward runs the hand-written
gfx950 kernels of libflope_amd.so instead of torchvision/cuDNN.  Differences, all
deliberate:
  * construction never downloads ImageNet weights (reference posenet.py:10 does);
    parameters start at torch's default random init until ``load_state_dict``.
  * inference always uses eval-mode semantics (BatchNorm running statistics,
    dropout = identity) whatever ``self.training`` says: the reference's scripts
    forget ``.eval()`` (D9), which makes its outputs random; eval mode is the only
    deterministic reading and the one its accuracy numbers were produced in.
  * HIP only: a CPU tensor raises instead of silently computing somewhere else.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from flope_amd.engine import PoseEngine

_STAGES = ((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2))


class _Block(nn.Module):
    """Parameter container with torchvision BasicBlock's attribute names."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class _Trunk(nn.Module):
    """Parameter container with torchvision ResNet's attribute names (resnet18)."""

    def __init__(self, out_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        for i, (cin, cout, s) in enumerate(_STAGES, start=1):
            setattr(self, f"layer{i}", nn.Sequential(_Block(cin, cout, s), _Block(cout, cout, 1)))
        self.fc = nn.Sequential(nn.Linear(512, out_dim), nn.ReLU())


class PoseResNet(nn.Module):
    def __init__(self, backbone_out_dim=2048, dropout=0.5, compute_dtype=None, max_batch=64):
        super().__init__()
        self.base = _Trunk(backbone_out_dim)
        self.fc_rot = nn.Linear(backbone_out_dim, 9)
        self.dropout = dropout                       # kept for API parity; identity at inference
        self.backbone_out_dim = backbone_out_dim
        self.compute_dtype = compute_dtype or os.environ.get("FLOPE_DTYPE", "f16")
        self._max_batch = max_batch
        self._engines = {}                           # (H, W) -> [engine, weights_version]
        self._version = 0

    # -- weight bookkeeping ----------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._version += 1
        return out

    def refresh(self):
        """Re-upload parameters after they were modified in place."""
        self._version += 1

    def _engine(self, x: torch.Tensor, hw) -> PoseEngine:
        if not x.is_cuda:
            raise RuntimeError("flope_amd PoseResNet runs on HIP devices only: move the crop batch to "
                               "'cuda' (there is no CPU fallback)")
        return self.engine_for(x.device, hw, int(x.shape[0]))

    def engine_for(self, device, hw, batch: int = 1) -> PoseEngine:
        """The engine for crops of hw on `device` with room for `batch` crops and the module's current parameters (built on
        first use, rebuilt when a larger batch arrives, re-loaded after load_state_dict / refresh)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("flope_amd PoseResNet runs on HIP devices only (there is no CPU fallback)")
        index = device.index if device.index is not None else torch.cuda.current_device()
        key = (index, int(hw[0]), int(hw[1]))
        slot = self._engines.get(key)
        if slot is None or slot[0].max_batch < batch:
            if slot is not None:
                slot[0].close()
            mb = max(self._max_batch, int(batch))
            slot = [PoseEngine(hw[0], hw[1], mb, self.compute_dtype, torch.device("cuda", index), self.backbone_out_dim), -1]
            self._engines[key] = slot
        if slot[1] != self._version:
            slot[0].load_state_dict(self.state_dict())
            slot[1] = self._version
        return slot[0]

    @staticmethod
    def _hw(x):
        return tuple(x.shape[2:]) if x.dtype == torch.float32 else tuple(x.shape[1:3])

    # -- reference API -----------------------------------------------------------
    def extract_features(self, x):
        """posenet.py:24-29 -> [B, backbone_out_dim]."""
        return self._engine(x, self._hw(x)).extract_features(x)

    def forward(self, x):
        """posenet.py:31-34 -> unconstrained [B,9] (row-major 3x3)."""
        r9, _ = self._engine(x, self._hw(x)).forward(x, want_r9=True, want_R=False)
        return r9

    # -- fused extension -----------------------------------------------------------
    def predict_rotations(self, x):
        """model(x) followed by procrustes_to_rotmat in one launch sequence:
        -> (r9 [B,9], R [B,3,3])."""
        return self._engine(x, self._hw(x)).forward(x, want_r9=True, want_R=True)
