"""Checkpoint handling for PoseResNet: key inventory, validation and a seeded
synthetic state_dict.

The reference ships no checkpoint (SURVEY.md §5 "Checkpoint / resume"); a real
one is a 124-entry ``state_dict`` saved by ``scripts/train_posenet.py:186`` and
loaded with ``torch.load(path, weights_only=True)`` (``scripts/test_posenet.py:51``).
``expected_keys()`` spells that inventory out so ``load_state_dict`` can reject
anything else, and ``synthetic_state_dict`` produces random-init weights of the
exact architecture (the bench / test fixture: there is no network for the
ImageNet initialisation the reference fetches at ``posenet.py:10``).
"""
from __future__ import annotations

import math

import torch

LAYERS = [(1, 64, 64, 1), (2, 64, 128, 2), (3, 128, 256, 2), (4, 256, 512, 2)]
BN_FIELDS = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")


def expected_keys(backbone_out_dim: int = 2048) -> dict:
    """name -> shape of the 124 state_dict entries (posenet.py:10-19)."""
    keys = {"base.conv1.weight": (64, 3, 7, 7)}

    def bn(prefix, c):
        for f in BN_FIELDS:
            keys[f"{prefix}.{f}"] = () if f == "num_batches_tracked" else (c,)

    bn("base.bn1", 64)
    for li, cin, cout, stride in LAYERS:
        for bi in range(2):
            p = f"base.layer{li}.{bi}"
            keys[p + ".conv1.weight"] = (cout, cin if bi == 0 else cout, 3, 3)
            bn(p + ".bn1", cout)
            keys[p + ".conv2.weight"] = (cout, cout, 3, 3)
            bn(p + ".bn2", cout)
            if bi == 0 and (stride != 1 or cin != cout):
                keys[p + ".downsample.0.weight"] = (cout, cin, 1, 1)
                bn(p + ".downsample.1", cout)
    keys["base.fc.0.weight"] = (backbone_out_dim, 512)
    keys["base.fc.0.bias"] = (backbone_out_dim,)
    keys["fc_rot.weight"] = (9, backbone_out_dim)
    keys["fc_rot.bias"] = (9,)
    return keys


def validate_state_dict(sd: dict, backbone_out_dim: int = 2048) -> None:
    exp = expected_keys(backbone_out_dim)
    missing = [k for k in exp if k not in sd]
    unexpected = [k for k in sd if k not in exp]
    if missing or unexpected:
        raise RuntimeError(f"Error(s) in loading state_dict for PoseResNet: "
                           f"missing keys {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                           f"unexpected keys {unexpected[:4]}{'...' if len(unexpected) > 4 else ''}")
    for k, shp in exp.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)}, model {tuple(shp)}")


def _rot_from_seed(g: torch.Generator) -> torch.Tensor:
    q = torch.randn(4, generator=g, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]],
                        dtype=torch.float64)


def synthetic_state_dict(seed: int = 0, backbone_out_dim: int = 2048,
                         rot_gain: float = 0.35) -> dict:
    """Seeded random-init weights of the PoseResNet architecture (CPU generator,
    so identical on every machine with the same torch).

    He-scaled convolutions, non-trivial BN running statistics, and a
    well-conditioned rotation head: ``fc_rot.bias = vec(R0)`` for a seeded
    rotation and ``fc_rot.weight`` scaled so that the data-dependent part of the
    3x3 output has entries of order ``rot_gain`` -- the unconstrained matrix then
    has singular values near 1, like a trained network, instead of the ~1e-2
    values of an untrained head (SURVEY.md §7 "Hard parts").
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k):
        std = math.sqrt(2.0 / (cin * k * k))
        sd[name] = torch.randn(cout, cin, k, k, generator=g) * std

    def bn(prefix, c, gamma_scale=1.0):
        sd[prefix + ".weight"] = (0.75 + 0.5 * torch.rand(c, generator=g)) * gamma_scale
        sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)

    conv("base.conv1.weight", 64, 3, 7)
    bn("base.bn1", 64)
    for li, cin, cout, stride in LAYERS:
        for bi in range(2):
            p = f"base.layer{li}.{bi}"
            conv(p + ".conv1.weight", cout, cin if bi == 0 else cout, 3)
            bn(p + ".bn1", cout)
            conv(p + ".conv2.weight", cout, cout, 3)
            bn(p + ".bn2", cout, gamma_scale=0.5)        # keeps the residual stream bounded
            if bi == 0 and (stride != 1 or cin != cout):
                conv(p + ".downsample.0.weight", cout, cin, 1)
                bn(p + ".downsample.1", cout)
    sd["base.fc.0.weight"] = torch.randn(backbone_out_dim, 512, generator=g) * math.sqrt(2.0 / 512)
    sd["base.fc.0.bias"] = 0.05 * torch.randn(backbone_out_dim, generator=g)
    R0 = _rot_from_seed(g)
    # hidden features are ReLU outputs with rms ~= FEAT_RMS (measured once with the
    # oracle on uniform[0,1] crops); centre the rows so the mean feature does not
    # shift M away from R0.
    FEAT_RMS = 2.35
    w = torch.randn(9, backbone_out_dim, generator=g)
    w = w - w.mean(dim=1, keepdim=True)
    sd["fc_rot.weight"] = w * (rot_gain / (FEAT_RMS * math.sqrt(backbone_out_dim)))
    sd["fc_rot.bias"] = R0.reshape(9).float()
    return sd
