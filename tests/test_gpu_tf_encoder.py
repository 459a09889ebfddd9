"""GPU parity of the TransformerEncoder path through the C-ABI (flope_tf_*).

fp32 mode is held to the reference's own output (tests/golden/reference_fixtures.npz: tf_x -> tf_y, produced by
importing scripts/tf_encoder.py in the build container); the 16-bit MFMA kernels are held to the fp64 oracle
(oracle/tf_encoder_ref.py, itself pinned by that fixture) and to the generic kernels of the same dtype.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tf_sd(ref_fixtures):
    return {k[len("tf_sd::"):]: v for k, v in ref_fixtures.items() if k.startswith("tf_sd::")}


def _enc(dims, sd, dtype, max_tokens):
    from flope_amd.tf_encoder import TransformerEncoder
    enc = TransformerEncoder(*dims, dtype=dtype, max_tokens=max_tokens)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return enc


def test_toy_fp32_matches_the_reference_output(ref_fixtures):
    enc = _enc((16, 32, 9, 4, 2, 64), _tf_sd(ref_fixtures), "f32", 128)
    y = enc(torch.from_numpy(ref_fixtures["tf_x"]).cuda()).cpu().numpy()
    assert y.shape == (8, 10, 9)
    assert np.abs(y - ref_fixtures["tf_y"]).max() < 1e-5          # fp32 both sides, different summation order
    # a shorter batch through the same handle, and determinism
    y2 = enc(torch.from_numpy(ref_fixtures["tf_x"][:3]).cuda()).cpu().numpy()
    assert np.array_equal(y2, y[:3])
    enc.close()


@pytest.mark.parametrize("dtype,tol", [("f16", 6e-3), ("bf16", 5e-2)])
def test_toy_16bit_generic_kernels_stay_close(ref_fixtures, dtype, tol):
    enc = _enc((16, 32, 9, 4, 2, 64), _tf_sd(ref_fixtures), dtype, 128)
    y = enc(torch.from_numpy(ref_fixtures["tf_x"]).cuda()).cpu().numpy()
    assert np.abs(y - ref_fixtures["tf_y"]).max() < tol
    enc.close()


# (input, d, out, heads, layers, ff), B, L: MFMA linears + MFMA attention (head_dim 64), ragged M and L
MFMA_CASES = [((16, 128, 9, 2, 2, 256), 5, 50),       # M = 250 (not a multiple of 128), L not a multiple of 32
              ((24, 384, 9, 6, 2, 1536), 3, 257),     # the cfg5 throughput shape, two layers
              ((16, 128, 9, 2, 1, 128), 1, 1),        # a single token
              ((16, 192, 5, 3, 1, 320), 2, 33)]       # d % 128 != 0 for out_proj / lin2 -> generic linears, MFMA attention


@pytest.mark.parametrize("dims,B,L", MFMA_CASES)
@pytest.mark.parametrize("dtype,tol", [("f16", 1.5e-2), ("bf16", 1.2e-1)])
def test_mfma_path_vs_oracle_and_generic(dims, B, L, dtype, tol):
    from oracle import tf_encoder_ref as T
    sd = T.synthetic_state_dict(dims[0], dims[1], dims[2], dims[4], dims[5], seed=5)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, L, dims[0])).astype(np.float32)
    ref = T.forward(sd, x, num_heads=dims[3])
    enc = _enc(dims, sd, dtype, B * L)
    xg = torch.from_numpy(x).cuda()
    y = enc(xg).cpu().numpy()
    assert np.isfinite(y).all()
    assert np.abs(y - ref).max() < tol, np.abs(y - ref).max()
    enc.set_option("generic", 1)
    yg = enc(xg).cpu().numpy()
    enc.set_option("generic", 0)
    assert np.abs(yg - ref).max() < tol
    assert np.abs(y - yg).max() < tol
    # fp32 strict mode on the same weights is tight
    e32 = _enc(dims, sd, "f32", B * L)
    y32 = e32(xg).cpu().numpy()
    assert np.abs(y32 - ref).max() < 2e-4
    # token-permutation equivariance (no positions, no mask) holds on the MFMA path too
    perm = rng.permutation(L)
    yp = enc(xg[:, torch.from_numpy(perm).cuda()]).cpu().numpy()
    assert np.abs(yp - y[:, perm]).max() < tol
    enc.close(); e32.close()


def test_error_paths(ref_fixtures):
    from flope_amd.tf_encoder import TransformerEncoder
    enc = TransformerEncoder(16, 32, 9, 4, 2, 64, dtype="f32", max_tokens=64)
    x = torch.zeros(2, 10, 16, device="cuda")
    with pytest.raises(RuntimeError, match="weights not loaded"):
        enc(x)
    sd = _tf_sd(ref_fixtures)
    bad = dict(sd); bad.pop("out_layer.bias")
    with pytest.raises(KeyError):
        enc.load_state_dict(bad)
    bad = dict(sd); bad["embedding.weight"] = np.zeros((32, 15), np.float32)
    with pytest.raises(RuntimeError, match="wrong shape"):
        enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in bad.items()})
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    with pytest.raises(RuntimeError, match="max_tokens"):
        enc(torch.zeros(8, 10, 16, device="cuda"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.zeros(2, 10, 16))
    assert enc(torch.zeros(0, 10, 16, device="cuda")).shape == (0, 10, 9)
    with pytest.raises(RuntimeError, match="divisible"):
        TransformerEncoder(16, 30, 9, 4, 2, 64)
    enc.close()
