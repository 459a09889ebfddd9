"""Oracle: PoseResNet forward + special Procrustes (CPU, torch functional ops).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates, in eval-mode semantics:
  * reference ``sunflower/models/posenet.py:5-34``  (PoseResNet: torchvision
    resnet18 trunk with ``avgpool -> AdaptiveAvgPool2d(1)``, ``fc ->
    Linear(512,2048)+ReLU`` (:12-16), ``F.relu`` (:26), dropout = identity in
    eval (:27-28), ``fc_rot = Linear(2048,9)`` (:19,:33)).
  * torchvision 0.20.1 ``ResNet._forward_impl`` / ``BasicBlock.forward``
    (third-party, not vendored; public topology: 7x7-s2-p3 stem, BN eps 1e-5,
    3x3-s2-p1 max-pool, BasicBlock x [2,2,2,2], 1x1-s2 conv + BN downsample).
  * reference ``sunflower/utils/conversion.py:54-58`` -> roma 1.5.1
    ``special_procrustes``: M = U S V^T, R = U diag(1,1,det(U)det(V)) V^T.

PARITY UNPINNED for the torchvision / roma boundaries (no golden vectors in
the reference, packages absent here).

``act_dtype`` lets the oracle emulate the device path's 16-bit activation /
weight storage (round-to-nearest-even after every fused conv epilogue, BN
folded into the weights before rounding) so a kernel can be checked at
accumulation-order precision, independent of the 16-bit rounding error.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LAYER_CFG = [(1, 64, 64, 1), (2, 64, 128, 2), (3, 128, 256, 2), (4, 256, 512, 2)]


def _bn(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=BN_EPS)


def forward_stages(sd: dict, x: torch.Tensor) -> dict:
    """fp32 eval-mode forward; returns every stage the device path can expose.

    Keys: stem (post conv1+bn1+relu), pool, layer{1..4}.{0,1}, feat (post
    avg-pool), hidden (post fc.0 + ReLU [+ReLU, dropout=identity]), r9.
    """
    out = {}
    sd = {k: v.float() for k, v in sd.items() if v.is_floating_point()}
    x = x.float()
    y = F.conv2d(x, sd["base.conv1.weight"], None, stride=2, padding=3)
    y = F.relu(_bn(y, sd, "base.bn1"))
    out["stem"] = y
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    out["pool"] = y
    for li, cin, cout, stride in LAYER_CFG:
        for bi in range(2):
            p = f"base.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            idt = y
            z = F.conv2d(y, sd[p + ".conv1.weight"], None, stride=s, padding=1)
            z = F.relu(_bn(z, sd, p + ".bn1"))
            out[f"layer{li}.{bi}.mid"] = z
            z = F.conv2d(z, sd[p + ".conv2.weight"], None, stride=1, padding=1)
            z = _bn(z, sd, p + ".bn2")
            if (p + ".downsample.0.weight") in sd:
                idt = F.conv2d(y, sd[p + ".downsample.0.weight"], None, stride=s, padding=0)
                idt = _bn(idt, sd, p + ".downsample.1")
            y = F.relu(z + idt)
            out[f"layer{li}.{bi}"] = y
    feat = y.mean(dim=(2, 3))                      # AdaptiveAvgPool2d(1) + flatten
    out["feat"] = feat
    hid = F.relu(F.linear(feat, sd["base.fc.0.weight"], sd["base.fc.0.bias"]))
    hid = F.relu(hid)                              # posenet.py:26 (idempotent)
    out["hidden"] = hid                            # dropout: identity in eval
    out["r9"] = F.linear(hid, sd["fc_rot.weight"], sd["fc_rot.bias"])
    return out


def forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] f32 in [0,1] -> [B,9] f32 (reference posenet.py:31-34)."""
    return forward_stages(sd, x)["r9"]


# --- 16-bit storage emulation ------------------------------------------------

def fold_bn(w, sd, prefix):
    """Eval-mode BN folded into the preceding bias-free conv: w' = w*g/sqrt(v+eps),
    b' = beta - mean*g/sqrt(v+eps).  fp64 internally, returned as fp32."""
    g = sd[prefix + ".weight"].double()
    b = sd[prefix + ".bias"].double()
    m = sd[prefix + ".running_mean"].double()
    v = sd[prefix + ".running_var"].double()
    scale = g / torch.sqrt(v + BN_EPS)
    return (w.double() * scale.view(-1, 1, 1, 1)).float(), (b - m * scale).float()


def forward_stages_emulated(sd: dict, x: torch.Tensor, act_dtype=torch.bfloat16) -> dict:
    """Same network, emulating the device data path: BN folded, weights and
    every stored activation rounded to ``act_dtype``; accumulation, bias,
    residual add, ReLU and the whole head in fp32."""
    rd = lambda t: t.to(act_dtype).float()
    out = {}
    sd = {k: v.float() for k, v in sd.items() if v.is_floating_point()}
    x = rd(x.float())
    w, b = fold_bn(sd["base.conv1.weight"], sd, "base.bn1")
    y = rd(F.relu(F.conv2d(x, rd(w), b, stride=2, padding=3)))
    out["stem"] = y
    y = F.max_pool2d(y, 3, 2, 1)
    out["pool"] = y
    for li, cin, cout, stride in LAYER_CFG:
        for bi in range(2):
            p = f"base.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            idt = y
            w, b = fold_bn(sd[p + ".conv1.weight"], sd, p + ".bn1")
            z = rd(F.relu(F.conv2d(y, rd(w), b, stride=s, padding=1)))
            out[f"layer{li}.{bi}.mid"] = z
            if (p + ".downsample.0.weight") in sd:
                w, b = fold_bn(sd[p + ".downsample.0.weight"], sd, p + ".downsample.1")
                idt = rd(F.conv2d(y, rd(w), b, stride=s, padding=0))
            w, b = fold_bn(sd[p + ".conv2.weight"], sd, p + ".bn2")
            y = rd(F.relu(F.conv2d(z, rd(w), b, stride=1, padding=1) + idt))
            out[f"layer{li}.{bi}"] = y
    feat = y.mean(dim=(2, 3))
    out["feat"] = feat
    hid = F.relu(F.linear(feat, sd["base.fc.0.weight"], sd["base.fc.0.bias"]))
    out["hidden"] = hid
    out["r9"] = F.linear(hid, sd["fc_rot.weight"], sd["fc_rot.bias"])
    return out


# --- special Procrustes ------------------------------------------------------

def special_procrustes(M: torch.Tensor) -> torch.Tensor:
    """R = argmin_{R in SO(3)} ||R - M||_F = U diag(1,1,det(U)det(V)) V^T,
    M = U S V^T (roma 1.5.1 special_procrustes; called at reference
    conversion.py:58 and train_posenet.py:37).  fp64 SVD, result in M.dtype."""
    Md = M.reshape(-1, 3, 3).double()
    U, _, Vh = torch.linalg.svd(Md)
    d = torch.det(U) * torch.det(Vh)
    D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), d], -1))
    return (U @ D @ Vh).to(M.dtype)


def procrustes_to_rotmat(inp: torch.Tensor) -> torch.Tensor:
    """reference sunflower/utils/conversion.py:54-58."""
    return special_procrustes(inp.reshape(-1, 3, 3))


def singular_values(M: torch.Tensor) -> torch.Tensor:
    return torch.linalg.svdvals(M.reshape(-1, 3, 3).double())


# --- rotation error metric (reference sunflower/utils/loss.py:3-18) -----------

def rotmat_to_unitquat(R: torch.Tensor) -> torch.Tensor:
    """xyzw unit quaternion via scipy (roma.rotmat_to_unitquat convention, used at
    train_posenet.py:134-137 only to feed diff_quats)."""
    from scipy.spatial.transform import Rotation
    import numpy as np
    q = Rotation.from_matrix(R.reshape(-1, 3, 3).double().numpy()).as_quat()
    return torch.from_numpy(np.asarray(q))


def diff_quats(q1: torch.Tensor, q2: torch.Tensor):
    """angle[deg] = 2*acos(|clamp(q1.q2,-1,1)|)*180/pi  (loss.py:15-17)."""
    dot = torch.einsum("nd,nd->n", q1, q2).clamp(-1, 1)
    return dot, 2 * torch.arccos(dot.abs()) * (180 / torch.pi)


def geodesic_deg(R1: torch.Tensor, R2: torch.Tensor) -> torch.Tensor:
    return diff_quats(rotmat_to_unitquat(R1), rotmat_to_unitquat(R2))[1]
