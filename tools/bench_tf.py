"""Throughput of the TransformerEncoder path on the BUILD-DEFINED cfg5 shape (the reference only has a
toy; this shape is not from the reference): L=257, d=384, heads=6, layers=12, ff=1536, B=256, f16.
    python tools/bench_tf.py [dtype] [B] [L] [layers] [generic]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flope_amd.tf_encoder import TransformerEncoder, expected_keys  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = int(sys.argv[3]) if len(sys.argv) > 3 else 257
layers = int(sys.argv[4]) if len(sys.argv) > 4 else 12
generic = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dims = (32, 384, 9, 6, layers, 1536)

rng = np.random.default_rng(0)
shapes = {"embedding.weight": (384, 32), "embedding.bias": (384,), "out_layer.weight": (9, 384), "out_layer.bias": (9,)}
for i in range(layers):
    p = f"transformer_encoder.layers.{i}."
    shapes.update({p + "self_attn.in_proj_weight": (1152, 384), p + "self_attn.in_proj_bias": (1152,),
                   p + "self_attn.out_proj.weight": (384, 384), p + "self_attn.out_proj.bias": (384,),
                   p + "linear1.weight": (1536, 384), p + "linear1.bias": (1536,), p + "linear2.weight": (384, 1536),
                   p + "linear2.bias": (384,), p + "norm1.weight": (384,), p + "norm1.bias": (384,),
                   p + "norm2.weight": (384,), p + "norm2.bias": (384,)})
sd = {}
for k in expected_keys(layers):
    s = shapes[k]
    if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
        sd[k] = torch.ones(s)
    elif len(s) == 1:
        sd[k] = torch.from_numpy(rng.uniform(-.1, .1, s).astype(np.float32))
    else:
        sd[k] = torch.from_numpy((rng.uniform(-1, 1, s) / np.sqrt(s[1])).astype(np.float32))
enc = TransformerEncoder(*dims, dtype=dtype, max_tokens=B * L)
enc.load_state_dict(sd)
enc.set_option("generic", generic)
x = torch.randn(B, L, 32, device="cuda")
for _ in range(3):
    y = enc(x)
torch.cuda.synchronize()
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    y = enc(x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = enc.flops(B, L)
print(json.dumps({"workload": f"TransformerEncoder B={B} L={L} d=384 h=6 layers={layers} ff=1536 (build-defined cfg5 shape)",
                  "dtype": dtype, "generic": generic, "ms": round(ms, 3), "tokens_per_s": round(B * L / ms * 1e3),
                  "seqs_per_s": round(B / ms * 1e3, 1), "tflops": round(fl / ms / 1e9, 1), "finite": bool(torch.isfinite(y).all())}))
