"""CPU oracle for the reference's `TransformerEncoder` (scripts/tf_encoder.py:5-27).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and tools that *check* results may import this module; the product path
(flope_amd.tf_encoder -> libflope_amd.so) never does.

Restates, in plain numpy, what the reference module computes in eval mode:
    embedding = Linear(input_dim, model_dim)                                   (tf_encoder.py:9)
    num_layers x nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward,
        dropout, batch_first=True)   -> post-norm (norm_first=False), ReLU     (tf_encoder.py:12-19)
    out_layer = Linear(model_dim, out_dim)                                     (tf_encoder.py:21)
    forward: embedding -> encoder stack -> out_layer, no positional encoding, no mask (tf_encoder.py:23-27)
The layer itself lives in pytorch==2.5.1 (environment.yml:118), restated here from its published definition:
    a = MHA(x): q,k,v = split(x @ in_proj_weight.T + in_proj_bias); per head softmax(q k^T / sqrt(dh)) v;
                concat heads @ out_proj.weight.T + out_proj.bias
    x = LayerNorm1(x + a);  f = linear2(relu(linear1(x)));  x = LayerNorm2(x + f)      (eps 1e-5, biased variance)
Pinned by tests/golden/reference_fixtures.npz (tf_x, tf_y, tf_sd::*), which tests/golden/make_goldens.py produced by
importing the reference file itself in the build container (seed 11, eval mode).
"""
import numpy as np

LN_EPS = 1e-5


def num_layers_of(sd) -> int:
    n = 0
    while f"transformer_encoder.layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def expected_keys(num_layers: int):
    keys = ["embedding.weight", "embedding.bias"]
    for i in range(num_layers):
        p = f"transformer_encoder.layers.{i}."
        keys += [p + s for s in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                                 "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                                 "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]
    return keys + ["out_layer.weight", "out_layer.bias"]


def layer_norm(x, w, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS) * w + b


def attention(x, w_in, b_in, w_out, b_out, heads):
    B, L, d = x.shape
    dh = d // heads
    qkv = x @ w_in.T + b_in
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, L, heads, dh).transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(dh)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, L, d)
    return o @ w_out.T + b_out


def forward(sd, x, num_heads: int, dtype=np.float64, stages=None):
    """sd: {name: array} with the reference's state_dict names; x [B,L,input_dim] -> [B,L,out_dim]."""
    g = lambda k: np.asarray(sd[k], dtype=dtype)
    h = np.asarray(x, dtype=dtype) @ g("embedding.weight").T + g("embedding.bias")
    if stages is not None:
        stages["embed"] = h
    for i in range(num_layers_of(sd)):
        p = f"transformer_encoder.layers.{i}."
        a = attention(h, g(p + "self_attn.in_proj_weight"), g(p + "self_attn.in_proj_bias"),
                      g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"), num_heads)
        h = layer_norm(h + a, g(p + "norm1.weight"), g(p + "norm1.bias"))
        f = np.maximum(h @ g(p + "linear1.weight").T + g(p + "linear1.bias"), 0) @ g(p + "linear2.weight").T + g(p + "linear2.bias")
        h = layer_norm(h + f, g(p + "norm2.weight"), g(p + "norm2.bias"))
        if stages is not None:
            stages[f"layer{i}"] = h
    return h @ g("out_layer.weight").T + g("out_layer.bias")


def synthetic_state_dict(input_dim, model_dim, out_dim, num_layers, ff_dim, seed=0):
    """Seeded weights with the reference's names/shapes (uniform +-1/sqrt(fan_in), LayerNorm gains near 1)."""
    rng = np.random.default_rng(seed)
    u = lambda o, i: rng.uniform(-1, 1, (o, i)).astype(np.float32) / np.sqrt(i)
    sd = {"embedding.weight": u(model_dim, input_dim), "embedding.bias": rng.uniform(-.1, .1, model_dim).astype(np.float32)}
    for l in range(num_layers):
        p = f"transformer_encoder.layers.{l}."
        sd[p + "self_attn.in_proj_weight"] = u(3 * model_dim, model_dim) * 1.5
        sd[p + "self_attn.in_proj_bias"] = rng.uniform(-.1, .1, 3 * model_dim).astype(np.float32)
        sd[p + "self_attn.out_proj.weight"] = u(model_dim, model_dim)
        sd[p + "self_attn.out_proj.bias"] = rng.uniform(-.1, .1, model_dim).astype(np.float32)
        sd[p + "linear1.weight"] = u(ff_dim, model_dim)
        sd[p + "linear1.bias"] = rng.uniform(-.1, .1, ff_dim).astype(np.float32)
        sd[p + "linear2.weight"] = u(model_dim, ff_dim)
        sd[p + "linear2.bias"] = rng.uniform(-.1, .1, model_dim).astype(np.float32)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = (1 + rng.uniform(-.2, .2, model_dim)).astype(np.float32)
            sd[p + n + ".bias"] = rng.uniform(-.1, .1, model_dim).astype(np.float32)
    sd["out_layer.weight"] = u(out_dim, model_dim)
    sd["out_layer.bias"] = rng.uniform(-.1, .1, out_dim).astype(np.float32)
    return sd
