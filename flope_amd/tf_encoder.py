"""Host-side mirror of the reference's `TransformerEncoder` (scripts/tf_encoder.py:5-27) over the
C-ABI `flope_tf_*` (include/flope_amd.h).  Same constructor arguments, same state_dict names, same
call: `enc(x)` with x float32 [B, L, input_dim] on the GPU -> float32 [B, L, out_dim].

Eval-mode semantics only (dropout is identity), as everywhere in this package.  There is no CPU
path: construction fails loudly without a HIP device or without the built library.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .engine import _DTYPES, _require_gpu, _stream_ptr


def expected_keys(num_layers: int) -> list:
    keys = ["embedding.weight", "embedding.bias"]
    for i in range(num_layers):
        p = f"transformer_encoder.layers.{i}."
        keys += [p + s for s in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                                 "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                                 "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]
    return keys + ["out_layer.weight", "out_layer.bias"]


class TransformerEncoder:
    def __init__(self, input_dim, model_dim, out_dim, num_heads, num_layers, ff_dim, dropout=0.1,
                 dtype="f32", max_tokens=4096, device=None):
        _require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError("flope_amd.TransformerEncoder runs on a HIP device only (no CPU path)")
        self.dims = (input_dim, model_dim, out_dim, num_heads, num_layers, ff_dim)
        self.dropout = dropout          # kept for signature parity; eval mode -> identity
        self.dtype = dtype
        self.max_tokens = int(max_tokens)
        self.handle = C.c_void_p()
        idx = self.device.index if self.device.index is not None else 0
        rc = self.lib.flope_tf_create(idx, input_dim, model_dim, out_dim, num_heads, num_layers, ff_dim,
                                      self.max_tokens, _DTYPES[dtype], C.byref(self.handle))
        if rc:
            msg = self.lib.flope_tf_last_error(None)
            raise RuntimeError(f"flope_tf_create failed ({rc}): {msg.decode() if msg else ''}")

    def _check(self, rc):
        if rc:
            msg = self.lib.flope_tf_last_error(self.handle)
            raise RuntimeError(f"flope_amd error {rc}: {msg.decode() if msg else ''}")

    # nn.Module-style no-ops the reference's callers use
    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device and torch.device(device).index not in (None, self.device.index):
            raise RuntimeError("the encoder is bound to the device it was created on")
        return self

    def load_state_dict(self, sd: dict) -> None:
        want = expected_keys(self.dims[4])
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"state_dict is missing {missing[:3]}{'...' if len(missing) > 3 else ''}")
        items = [(k, torch.as_tensor(sd[k]).detach().to("cpu", torch.float32).contiguous()) for k in want]
        n = len(items)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in items])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in items])
        ndims = (C.c_int * n)(*[t.dim() for _, t in items])
        shape_arrs = [(C.c_int64 * max(t.dim(), 1))(*t.shape) for _, t in items]
        shapes = (C.c_void_p * n)(*[C.cast(a, C.c_void_p).value for a in shape_arrs])
        with torch.cuda.device(self.device):
            self._check(self.lib.flope_tf_load_weights(self.handle, n, names, ptrs, ndims, shapes))

    def set_option(self, name: str, value: int) -> int:
        return self.lib.flope_tf_set_option(self.handle, name.encode(), int(value))

    def flops(self, batch: int, seq_len: int) -> float:
        return self.lib.flope_tf_forward_flops(self.handle, batch, seq_len)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.device != self.device:
            raise RuntimeError(f"input must live on {self.device} (got {x.device}); no CPU path")
        if x.dim() != 3 or x.shape[2] != self.dims[0]:
            raise ValueError(f"expected [B, L, {self.dims[0]}], got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        B, L = x.shape[0], x.shape[1]
        y = torch.empty((B, L, self.dims[2]), dtype=torch.float32, device=self.device)
        self._check(self.lib.flope_tf_forward(self.handle, x.data_ptr(), B, L, y.data_ptr(), _stream_ptr(self.device)))
        self._keep = x
        return y

    __call__ = forward

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.flope_tf_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
