"""Regenerates tests/golden/*.npz.  Run from the repo root in the build container:

    python tests/golden/make_goldens.py

  posenet_cfg1.npz   BASELINE cfg1: 16 seeded uniform[0,1] 224x224 crops through the CPU
                     oracle (oracle/posenet_ref.py, fp32, eval mode) with the seeded
                     synthetic weights (flope_amd.weights.synthetic_state_dict(0)); holds
                     r9, R, the 15-column detection rows and strided slices of every stage.
                     The reference itself cannot produce these (torchvision / roma absent,
                     SURVEY.md §8c): the oracle is the restatement, parity unpinned.
  reference_fixtures.npz
                     outputs of the two reference files that ARE importable here
                     (sunflower/utils/loss.py diff_quats, scripts/tf_encoder.py
                     TransformerEncoder): inputs + expected outputs, produced by importing
                     them from /root/reference.  Data only -- no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from flope_amd.weights import synthetic_state_dict  # noqa: E402
from oracle import pipeline_ref, posenet_ref  # noqa: E402


def cfg1():
    torch.set_num_threads(8)
    sd = synthetic_state_dict(0)
    torch.manual_seed(0)
    x = torch.rand(16, 3, 224, 224)
    st = posenet_ref.forward_stages(sd, x)
    R = posenet_ref.procrustes_to_rotmat(st["r9"])
    boxes = np.tile(np.array([[0, 0, 224, 224]]), (16, 1))
    out = {"r9": st["r9"].numpy(), "R": R.numpy(), "rows": pipeline_ref.detection_rows(boxes, R.numpy()),
           "x_checksum": np.array([x.double().sum().item()])}
    for k, v in st.items():
        if v.dim() == 4:
            out["stage_" + k] = v[:2, ::7, ::5, ::5].numpy()          # small strided slice
            out["mean_" + k] = v.double().mean(dim=(1, 2, 3)).numpy()
        elif k != "r9":
            out["stage_" + k] = v[:, ::16].numpy()
    np.savez_compressed(os.path.join(HERE, "posenet_cfg1.npz"), **out)
    print("posenet_cfg1.npz:", {k: v.shape for k, v in out.items() if k in ("r9", "R", "rows")})


def reference_fixtures():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference not present: skipping reference_fixtures.npz")
        return
    out = {}
    sys.path.insert(0, ref)
    sys.path.insert(0, os.path.join(ref, "scripts"))
    # diff_quats (sunflower/utils/loss.py) -- a namespace-package import of the reference tree
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_loss", os.path.join(ref, "sunflower/utils/loss.py"))
    ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_loss)
    g = torch.Generator().manual_seed(7)
    q1 = torch.nn.functional.normalize(torch.randn(64, 4, generator=g, dtype=torch.float64), dim=1)
    q2 = torch.nn.functional.normalize(torch.randn(64, 4, generator=g, dtype=torch.float64), dim=1)
    q2[:4] = q1[:4]
    q2[4:8] = -q1[4:8]
    dot, ang = ref_loss.diff_quats(q1, q2)
    out.update(dq_q1=q1.numpy(), dq_q2=q2.numpy(), dq_dot=dot.numpy(), dq_angle=ang.numpy())
    # TransformerEncoder toy (scripts/tf_encoder.py)
    spec = importlib.util.spec_from_file_location("ref_tf", os.path.join(ref, "scripts/tf_encoder.py"))
    ref_tf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_tf)
    torch.manual_seed(11)
    enc = ref_tf.TransformerEncoder(16, 32, 9, 4, 2, 64, 0.1).eval()
    x = torch.randn(8, 10, 16)
    with torch.no_grad():
        y = enc(x)
    out["tf_x"] = x.numpy()
    out["tf_y"] = y.numpy()
    for k, v in enc.state_dict().items():
        out["tf_sd::" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "reference_fixtures.npz"), **out)
    print("reference_fixtures.npz:", len(out), "arrays; tf_y", y.shape)


if __name__ == "__main__":
    cfg1()
    reference_fixtures()
