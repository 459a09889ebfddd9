"""conv_s2r against conv_mfma<gather>: where do the layer2.0 outputs differ (image, row band, column, channel group)?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flope_amd.engine import PoseEngine
from flope_amd.weights import synthetic_state_dict
B = int(os.environ.get("B", 5))
sd = synthetic_state_dict(0)
torch.manual_seed(11)
x = torch.rand(B, 3, 224, 224).cuda()
outs = []
for s2r in (0, 1, 1, 1):
    e = PoseEngine(224, 224, B, "f16")
    e.set_option("s2r", s2r); e.set_option("streams", 1); e.set_option("dsfuse", 0)
    e.load_state_dict(sd)
    e.forward(x)
    outs.append(e.read_stage("layer2.0", B).float().cpu())
    e.close()
ref = outs[0]
print("shape", tuple(ref.shape))
for k, o in enumerate(outs[1:]):
    d = (o - ref).abs()
    bad = d > 0.05 * ref.abs().max()
    print(f"run {k}: max diff {d.max():.4f} (ref max {ref.abs().max():.3f}); bad elements {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        # layout of read_stage: NCHW
        print("  per image:", bad.sum(dim=(1, 2, 3)).tolist())
        print("  per channel group of 8:", bad.sum(dim=(0, 2, 3)).view(-1, 8).sum(1).tolist())
        print("  per row:", bad.sum(dim=(0, 1, 3)).tolist())
        print("  per col:", bad.sum(dim=(0, 1, 2)).tolist())
