"""CPU: host logic of the product (no GPU compute): C-ABI exports, pose math shared with
the head kernels, 16-bit conversions, MFMA weight packers, checkpoint inventory and the
pure-python parts of the ``sunflower`` mirror."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import pipeline_ref as P
from oracle import posenet_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32p = C.POINTER(C.c_float)


def _f(a):
    return a.ctypes.data_as(f32p)


# ---- the C-ABI library ---------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    from flope_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "flope_amd.h")).read()
    declared = set(re.findall(r"\b(flope_[a-z0-9_]+)\s*\(", header))
    declared.discard("flope_engine")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.flope_version()


def test_create_fails_loudly_without_gpu():
    from flope_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.flope_create(0, 224, 224, 4, 1, 2048, C.byref(h))
    assert rc != 0 and not h.value
    assert "no HIP device" in _lib.last_error(None)
    from flope_amd.engine import PoseEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PoseEngine(224, 224, 4)


def test_bad_arguments_are_rejected():
    from flope_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.flope_create(0, 8, 8, 4, 1, 2048, C.byref(h)) == -1
    assert lib.flope_create(0, 224, 224, 0, 1, 2048, C.byref(h)) == -1
    assert lib.flope_create(0, 224, 224, 4, 7, 2048, C.byref(h)) == -1
    assert lib.flope_procrustes(None, None, 3, None) == -1
    assert lib.flope_procrustes(None, None, 0, None) == 0
    assert lib.flope_forward(None, None, 0, 1, None, None, None) == -1


# ---- pose math compiled from the same header the kernels use ------------------------------------
def test_horn_procrustes_matches_svd_oracle(harness):
    g = torch.Generator().manual_seed(0)
    M = torch.randn(512, 3, 3, generator=g)
    M[:64] = torch.from_numpy(np.linalg.qr(np.random.default_rng(1).normal(size=(64, 3, 3)))[0]).float() * 0.7
    Mn = np.ascontiguousarray(M.numpy().reshape(-1, 9))
    out = np.empty_like(Mn)
    harness.hh_procrustes(_f(Mn), _f(out), Mn.shape[0])
    ref = O.special_procrustes(M.double()).numpy().reshape(-1, 9)
    sv = O.singular_values(M).numpy()
    well = (sv[:, 1] + sv[:, 2]) > 0.05           # away from the gauge-degenerate set
    assert well.sum() > 400
    np.testing.assert_allclose(out[well], ref[well], atol=2e-5)
    R = out.reshape(-1, 3, 3).astype(np.float64)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)


def test_procrustes_fast_path_and_fallback(harness):
    """Newton/adjugate fast path == Jacobi == SVD oracle; degenerate inputs take the Jacobi fallback."""
    g = torch.Generator().manual_seed(8)
    M = torch.randn(4096, 3, 3, generator=g) * torch.logspace(-6, 6, 4096)[:, None, None]     # scale invariance
    Mn = np.ascontiguousarray(M.numpy().reshape(-1, 9))
    a, b = np.empty_like(Mn), np.empty_like(Mn)
    harness.hh_procrustes(_f(Mn), _f(a), Mn.shape[0])
    harness.hh_procrustes_jacobi(_f(Mn), _f(b), Mn.shape[0])
    sv = O.singular_values(M).numpy()
    well = (sv[:, 1] + sv[:, 2]) / sv[:, 0] > 0.05
    ref = O.special_procrustes(M.double()).numpy().reshape(-1, 9)
    np.testing.assert_allclose(a[well], ref[well], atol=2e-5)
    np.testing.assert_allclose(a[well], b[well], atol=2e-5)
    assert harness.hh_procrustes_newton_count(_f(Mn), Mn.shape[0]) > 0.95 * Mn.shape[0]
    # rotations, scaled rotations, reflections, rank-deficient and zero inputs
    from scipy.spatial.transform import Rotation
    Rr = Rotation.random(64, random_state=2).as_matrix().astype(np.float32)
    special = np.concatenate([Rr, 3.5 * Rr, Rr * np.array([1, 1, -1], np.float32)[None, None, :],
                              np.zeros((1, 3, 3), np.float32), np.ones((1, 3, 3), np.float32),
                              np.diag([1, 1, 0]).astype(np.float32)[None], np.diag([2, 0, 0]).astype(np.float32)[None]])
    sp = np.ascontiguousarray(special.reshape(-1, 9))
    out = np.empty_like(sp)
    harness.hh_procrustes(_f(sp), _f(out), sp.shape[0])
    np.testing.assert_allclose(out[:64], Rr.reshape(-1, 9), atol=2e-6)
    np.testing.assert_allclose(out[64:128], Rr.reshape(-1, 9), atol=2e-6)
    Ro = out.reshape(-1, 3, 3).astype(np.float64)
    assert np.all(np.isfinite(Ro))
    np.testing.assert_allclose(np.linalg.det(Ro), 1.0, atol=1e-5)
    np.testing.assert_allclose(Ro @ Ro.transpose(0, 2, 1), np.tile(np.eye(3), (len(Ro), 1, 1)), atol=1e-5)
    ref = O.special_procrustes(torch.from_numpy(special[128:192]).double()).numpy()
    np.testing.assert_allclose(Ro[128:192], ref, atol=2e-5)                                  # reflections: det fixed to +1


def test_nullify_yaw_matches_scipy_oracle(harness):
    from scipy.spatial.transform import Rotation
    R = Rotation.random(256, random_state=4).as_matrix().astype(np.float32).reshape(-1, 9)
    R = np.ascontiguousarray(R)
    out = np.empty_like(R)
    harness.hh_nullify_yaw(_f(R), _f(out), R.shape[0])
    ref = P.nullify_yaw_batch(R.reshape(-1, 3, 3).astype(np.float64)).reshape(-1, 9)
    np.testing.assert_allclose(out, ref, atol=2e-6)


def test_sunflower_mvg_mirror_matches_oracle():
    from sunflower.utils import mvg
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 300, 2); b = a + rng.integers(1, 200, 2)
        bb = [int(a[0]), int(a[1]), int(b[0]), int(b[1])]
        assert mvg.squarify_bb(bb) == P.squarify_bb(bb)
        s = mvg.squarify_bb(bb)
        assert mvg.bb_in_frame(s, (400, 450, 3)) == P.bb_in_frame(s, (400, 450, 3))
    bbs = rng.integers(0, 50, (20, 2)); bbs = np.c_[bbs, bbs + rng.integers(1, 100, (20, 2))]
    assert mvg.filter_very_large_bb(bbs).tolist() == P.filter_very_large_bb(bbs).tolist()
    K = np.array([[910.0, 0, 640.5], [0, 905.0, 360.2], [0, 0, 1]])
    uv = rng.uniform(0, 1000, (16, 2)); d = rng.uniform(0.2, 2, 16)
    np.testing.assert_allclose(mvg.get_points3d(uv, d, K), P.get_points3d(uv, d, K), atol=1e-12)
    from scipy.spatial.transform import Rotation
    R = Rotation.random(32, random_state=9).as_matrix()
    np.testing.assert_allclose(mvg.nullify_yaw_batch(R), P.nullify_yaw_batch(R), atol=1e-12)
    cam = np.eye(4); cam[:3, :3] = R[0]; cam[:3, 3] = [1, 2, 3]
    obj = np.tile(np.eye(4), (3, 1, 1)); obj[:, :3, :3] = R[1:4]
    np.testing.assert_allclose(mvg.pose_cam_to_world(obj, cam), cam @ obj, atol=1e-15)


def test_sunflower_loss_mirror_matches_reference_fixture(ref_fixtures):
    from sunflower.utils.loss import diff_quats
    dot, ang = diff_quats(torch.from_numpy(ref_fixtures["dq_q1"]), torch.from_numpy(ref_fixtures["dq_q2"]))
    np.testing.assert_allclose(dot.numpy(), ref_fixtures["dq_dot"], atol=1e-12)
    np.testing.assert_allclose(ang.numpy(), ref_fixtures["dq_angle"], atol=1e-6)


# ---- 16-bit conversions / packers --------------------------------------------------------------
@pytest.mark.parametrize("dtype,tdt", [(0, torch.bfloat16), (1, torch.float16)])
def test_cvt16_bit_exact_vs_torch(harness, dtype, tdt):
    g = torch.Generator().manual_seed(5)
    vals = torch.cat([torch.randn(20000, generator=g) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 300.0, 7e4)])
    edge = torch.tensor([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e38, -1e38, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5,
                         float("inf"), -float("inf"), 1.0 + 2 ** -11, 1.0 + 2 ** -8, 1.0 + 3 * 2 ** -9])
    v = np.ascontiguousarray(torch.cat([vals, edge]).numpy())
    out = np.empty(v.shape[0], np.uint16)
    harness.hh_cvt16(_f(v), out.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_long(v.shape[0]), dtype)
    ref = torch.from_numpy(v).to(tdt).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(out, ref)


def test_lds_row_permutation_is_a_bijection_with_consecutive_lane_channels(harness):
    rows = [harness.hh_lds_row_to_channel(r) for r in range(128)]
    assert sorted(rows) == list(range(128))
    # lane group g of a wave owns MFMA rows 4g..4g+3 of each of the 4 channel tiles: channels 16g..16g+15
    for rng_ in range(2):
        for g in range(4):
            ch = [rows[rng_ * 64 + ct * 16 + 4 * g + q] for ct in range(4) for q in range(4)]
            assert ch == list(range(rng_ * 64 + 16 * g, rng_ * 64 + 16 * g + 16))


def test_stag_row_permutation_gives_two_runs_of_eight_channels_per_lane(harness):
    """conv_stag / conv_gstag images: lane group g owns MFMA rows 4g..4g+3 of each of the 4 channel tiles = channels
    8g..8g+7 (tiles 0, 1) and 32+8g..32+8g+7 (tiles 2, 3) of the wave's 64-channel range: the two 16-byte stores of a pixel's four
    lane groups are contiguous 64-byte halves of its 128-byte channel block."""
    rows = [harness.hh_stag_row_to_channel(r) for r in range(128)]
    assert sorted(rows) == list(range(128))
    for rng_ in range(2):
        for g in range(4):
            ch = [rows[rng_ * 64 + ct * 16 + 4 * g + q] for ct in range(4) for q in range(4)]
            assert ch == list(range(rng_ * 64 + 8 * g, rng_ * 64 + 8 * g + 8)) + list(range(rng_ * 64 + 32 + 8 * g, rng_ * 64 + 40 + 8 * g))


@pytest.mark.parametrize("cout,cin,k", [(64, 64, 3), (128, 64, 1), (256, 128, 3)])
def test_pack_conv_is_a_permutation_of_the_weights(harness, cout, cin, k):
    n = cout * cin * k * k
    w = (np.arange(n, dtype=np.float32) % 2039) + 1.0          # exactly representable in f16
    w = w.reshape(cout, cin, k, k)
    out = np.empty(n, np.uint16)
    harness.hh_pack_conv(_f(np.ascontiguousarray(w)), cout, cin, k, 1, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    vals = torch.from_numpy(out.view(np.int16)).view(torch.float16).float().numpy()
    assert np.array_equal(np.sort(vals), np.sort(w.reshape(-1)))
    # spot check the documented image order: tile (ntile 0, chunk 0, tap 0), LDS row 0 = channel 0
    BN = 64 if cout == 64 else 128
    row0 = vals[:64]
    logical = np.concatenate([row0[((j ^ 0) * 8):((j ^ 0) * 8 + 8)] for j in range(8)])
    np.testing.assert_array_equal(logical, w[0, :64, 0, 0])
    # row 2 (swizzle (2>>1)&7 = 1): channel 2, slots XOR 1
    r = vals[2 * 64:3 * 64]
    logical = np.concatenate([r[((j ^ 1) * 8):((j ^ 1) * 8 + 8)] for j in range(8)])
    np.testing.assert_array_equal(logical, w[harness.hh_lds_row_to_channel(2), :64, 0, 0])
    assert BN in (64, 128)


def test_pack_stem_layout(harness):
    w = np.random.default_rng(0).integers(-64, 64, (64, 3, 7, 7)).astype(np.float32)
    out = np.empty(7 * 64 * 32, np.uint16)
    harness.hh_pack_stem(_f(np.ascontiguousarray(w)), 1, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    vals = torch.from_numpy(out.view(np.int16)).view(torch.float16).float().numpy().reshape(7, 64, 32)
    h = [0, 2, 3, 1]
    for ky in (0, 3, 6):
        for rl in (0, 5, 17, 63):
            co = harness.hh_lds_row_to_channel(rl)
            row = vals[ky, rl]
            logical = np.concatenate([row[((s ^ h[(rl >> 2) & 3]) * 8):((s ^ h[(rl >> 2) & 3]) * 8 + 8)] for s in range(4)])
            expect = np.zeros(32, np.float32)
            for kx in range(7):
                for c in range(3):
                    expect[kx * 4 + c] = w[co, c, ky, kx]
            np.testing.assert_array_equal(logical, expect)


def _mfma_a_fragment(vals, lane):
    """(row i, k slot) of an A fragment as the MFMA reads it: lane -> matrix row lane & 15, k = 8 (lane >> 4) .. + 7."""
    return lane & 15, 8 * (lane >> 4)


def test_pack_s2r_and_s1r_fragment_images(harness):
    """conv_s2r / conv_s1r keep their weights in registers; a wave-load is one contiguous KiB [lane][8] in A-fragment order.  Every
    weight appears exactly once, and lane (i = lane & 15, g = lane >> 4) of fragment (wave, step = half-chunk * 9 + tap, ct) holds
    W[32 wave + 8 (i >> 2) + 4 ct + (i & 3)][input channel base + 32 hc + 8 g + j][tap] -- so that MFMA output rows 4 g .. 4 g + 3 of
    channel tile ct are eight consecutive channels per lane (one 16-byte store per pixel)."""
    rng = np.random.default_rng(5)
    w2 = rng.integers(-1000, 1000, (128, 64, 3, 3)).astype(np.float32)
    out = np.empty(4 * 18 * 2 * 64 * 8, np.uint16)
    harness.hh_pack_s2r(_f(np.ascontiguousarray(w2)), 1, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    v = torch.from_numpy(out.view(np.int16)).view(torch.float16).float().numpy().reshape(4, 18, 2, 64, 8)
    assert np.array_equal(np.sort(v.reshape(-1)), np.sort(w2.reshape(-1)))
    for wave, st, ct, lane in ((0, 0, 0, 0), (3, 17, 1, 63), (2, 9, 0, 37), (1, 4, 1, 18)):
        i, k0 = _mfma_a_fragment(v, lane)
        hc, tap = divmod(st, 9)
        co = 32 * wave + 8 * (i >> 2) + 4 * ct + (i & 3)
        np.testing.assert_array_equal(v[wave, st, ct, lane], w2[co, 32 * hc + k0:32 * hc + k0 + 8, tap // 3, tap % 3])
    w1 = rng.integers(-1000, 1000, (128, 128, 3, 3)).astype(np.float32)
    out = np.empty(4 * 2 * 18 * 2 * 64 * 8, np.uint16)
    harness.hh_pack_s1r(_f(np.ascontiguousarray(w1)), 1, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    v = torch.from_numpy(out.view(np.int16)).view(torch.float16).float().numpy().reshape(4, 2, 18, 2, 64, 8)
    assert np.array_equal(np.sort(v.reshape(-1)), np.sort(w1.reshape(-1)))
    for cg, kh, st, ct, lane in ((0, 0, 0, 0, 0), (3, 1, 17, 1, 63), (2, 0, 9, 0, 37), (1, 1, 4, 1, 18)):
        i, k0 = _mfma_a_fragment(v, lane)
        hc, tap = divmod(st, 9)
        co = 32 * cg + 8 * (i >> 2) + 4 * ct + (i & 3)
        ci = 64 * kh + 32 * hc + k0
        np.testing.assert_array_equal(v[cg, kh, st, ct, lane], w1[co, ci:ci + 8, tap // 3, tap % 3])
    # the two K halves of a pair partition the input channels: kh 0 never sees a channel >= 64
    np.testing.assert_array_equal(np.sort(v[:, 0].reshape(-1)), np.sort(w1[:, :64].reshape(-1)))


# ---- checkpoint inventory -------------------------------------------------------------------------
def test_state_dict_inventory(state_dict):
    from flope_amd.weights import expected_keys, validate_state_dict
    keys = expected_keys()
    assert len(keys) == 124 and set(keys) == set(state_dict)
    assert sum(int(np.prod(s)) for k, s in keys.items() if "running" not in k and "num_batches" not in k) == 12245577
    validate_state_dict(state_dict)
    bad = dict(state_dict); bad.pop("fc_rot.bias")
    with pytest.raises(RuntimeError, match="missing keys"):
        validate_state_dict(bad)
    bad = dict(state_dict); bad["fc_rot.bias"] = torch.zeros(8)
    with pytest.raises(RuntimeError, match="size mismatch"):
        validate_state_dict(bad)


def test_facade_module_has_the_reference_key_set(state_dict):
    from sunflower.models.posenet import PoseResNet
    m = PoseResNet()
    assert set(m.state_dict()) == set(state_dict)
    m.load_state_dict(state_dict)
    assert sum(p.numel() for p in m.parameters()) == 12245577
    with pytest.raises(RuntimeError, match="HIP devices only"):
        m(torch.rand(1, 3, 64, 64))


def test_tools_and_entry_points_compile():
    """bench.py, __graft_entry__.py and every tool byte-compile (they only run on a GPU box, so nothing else would
    notice a syntax slip before the round-end run)."""
    import glob
    import py_compile
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) >= 10
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            py_compile.compile(f, doraise=True, cfile=os.path.join(tmp, "check.pyc"))


# ---- PosePredictor: the reference's own call shape (pose_predictor.py:41-66) -------------------------------------
_GDINO_STUB = '''
import numpy as np
CALLS = []
class GroundingDINO:
    def __init__(self, device, text_prompt, box_th=0.2, text_th=0.3, obj_filter=None):
        CALLS.append(("gdino", device, text_prompt, box_th, text_th, obj_filter))
    def detect(self, image):
        return np.array([[10, 10, 40, 50], [60, 20, 100, 70]])
'''
_SAM_STUB = '''
import numpy as np
CALLS = []
class SAM:
    def __init__(self, device):
        CALLS.append(("sam", device))
    def get_segmentation_mask(self, image, bounding_boxes):
        CALLS.append(("mask", type(image).__name__, bounding_boxes))
        return np.zeros((image.size[1], image.size[0]), np.uint8)
'''


def _ckpt_and_intrinsics(tmp_path, state_dict, h=480, w=640):
    import yaml
    ckpt, intr = tmp_path / "posenet.pth", tmp_path / "intrinsics.yaml"
    torch.save(state_dict, ckpt)
    intr.write_text(yaml.safe_dump(dict(fx=600.0, fy=600.0, cx=w / 2, cy=h / 2, h=h, w=w)))
    return str(ckpt), str(intr)


def test_pose_predictor_reference_call_shape_builds_gdino_and_sam(tmp_path, state_dict, monkeypatch):
    """`PosePredictor(device, posenet_path, intrin_path, debug)` with no injection -- exactly scripts/live_pose.py:22-28 --
    imports sunflower.models.{grounding_dino,sam} through the namespace package and builds them with the reference's
    arguments (pose_predictor.py:55-60).  The stubs stand in for the reference tree's two hub-model wrappers."""
    import importlib
    import sys
    stub = tmp_path / "reftree" / "sunflower" / "models"
    stub.mkdir(parents=True)
    (stub / "grounding_dino.py").write_text(_GDINO_STUB)
    (stub / "sam.py").write_text(_SAM_STUB)
    monkeypatch.syspath_prepend(str(tmp_path / "reftree"))            # the reference tree, next to the mirror
    for name in ("sunflower.models.grounding_dino", "sunflower.models.sam"):
        sys.modules.pop(name, None)
    importlib.invalidate_caches()
    from sunflower.predictor.pose_predictor import PosePredictor
    ckpt, intr = _ckpt_and_intrinsics(tmp_path, state_dict)
    pred = PosePredictor("cpu", ckpt, intr, True)                     # construction touches no GPU
    gd, sm = sys.modules["sunflower.models.grounding_dino"], sys.modules["sunflower.models.sam"]
    assert gd.CALLS == [("gdino", "cpu", "white flower.", 0.3, 0.3, "white flower")]
    assert sm.CALLS == [("sam", "cpu")]
    assert pred.detector(np.zeros((480, 640, 3), np.uint8)).shape == (2, 4)
    m = pred.segmenter(np.zeros((480, 640, 3), np.uint8), [[10, 10, 40, 50]])
    assert m.shape == (480, 640) and sm.CALLS[-1] == ("mask", "Image", [[10, 10, 40, 50]])
    assert pred.K[0][0] == 600.0 and (pred.height, pred.width) == (480, 640)
    # the posenet mirror inside it is the HIP one: a CPU crop batch must raise, never compute on the host
    with pytest.raises(RuntimeError, match="HIP devices only"):
        pred.posenet(torch.rand(1, 3, 64, 64))
    for name in ("sunflower.models.grounding_dino", "sunflower.models.sam"):
        sys.modules.pop(name, None)


def test_pose_predictor_without_front_end_names_what_is_missing(tmp_path, state_dict):
    import sys
    for name in ("sunflower.models.grounding_dino", "sunflower.models.sam"):
        sys.modules.pop(name, None)
    from sunflower.predictor.pose_predictor import PosePredictor
    ckpt, intr = _ckpt_and_intrinsics(tmp_path, state_dict)
    with pytest.raises(ImportError, match="grounding_dino"):
        PosePredictor("cpu", ckpt, intr)
    # injected callables need no reference modules
    p = PosePredictor("cpu", ckpt, intr, detector=lambda rgb: np.zeros((0,)), segmenter=lambda rgb, bb: None)
    assert p.get_flower_poses(np.zeros((480, 640, 3), np.uint8), np.zeros((480, 640), np.uint16)) is None   # :76-78


def test_integer_relu_clamp_equals_the_float_form_for_every_float16_pattern():
    """common.h pk_relu16<f16_t> (stem, r03): ReLU + saturation of a packed float16 pair as integer max(w, 0) then integer
    min(., 0x7BFF).  Restated here on all 65,536 bit patterns against the float form it replaces (pk_out16<f16_t>(w, true):
    integer max with 0, then fmin(x, 65504) -- which returns the number when x is a NaN -- then fmax(x, -65504))."""
    import numpy as np
    w = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    relu_i = np.maximum(w.view(np.int16), np.int16(0))                     # both forms start with the signed-integer max
    got = np.minimum(relu_i, np.int16(0x7BFF)).view(np.uint16)
    x = relu_i.view(np.float16).astype(np.float32)
    x = np.where(np.isnan(x), np.float32(65504.0), np.minimum(x, np.float32(65504.0)))       # minnum semantics
    x = np.maximum(x, np.float32(-65504.0))
    ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, ref)


def _w4_check(harness, nbd, pw, spread, mt, boundary, res, formula):
    msg = C.create_string_buffer(256)
    rc = harness.flope_host_w4_schedule_check(nbd, pw, spread, mt, boundary, res, formula, msg, 256)
    return rc, msg.value.decode()


def test_conv_w4_wait_counts_cover_every_lds_read(harness):
    """conv_w4's `s_waitcnt vmcnt(N)` counts (flope_amd/csrc/w4_sched.h: a constexpr model of the wave's in-order memory queue, the
    SAME functions the kernel instantiates) replayed against an independent statement of the schedule (tests/host_harness): for every
    instantiated ring depth / patch size / burst spreading / tile height, inside a tile, behind a class walk's tile boundary (epilogue
    stores in the queue) and in a class walk's last body with a residual input (register loads in the queue), every LDS read is
    ordered behind a wait + barrier that covers its producing DMA.  Timing can never show a count that is too lax (the data has
    usually landed anyway): r03 shipped one for a round -- the last two assertions replay exactly that bug."""
    for nbd in (3, 4, 5):
        for pw in (8, 10, 12):
            for spread in (0, 1, 2, 3, 4):
                for mt in (4, 5, 6, 7, 8):
                    for boundary in (0, 1):
                        for res in (0, 1):
                            rc, msg = _w4_check(harness, nbd, pw, spread, mt, boundary, res, 0)
                            assert rc == 0, (nbd, pw, spread, mt, boundary, res, msg)
            # r03's closed form (one burst per half-chunk) describes the same schedule and passes ...
            assert _w4_check(harness, nbd, pw, 0, 7, 0, 0, 1)[0] == 0
    # ... and the expression it replaced lets the burst of double step 0 stay in flight across barrier 3 of a 5-deep ring,
    # behind which its first fragments are read: the check names it (a shallower ring never had the problem)
    rc, msg = _w4_check(harness, 5, 8, 0, 7, 0, 0, 2)
    assert rc != 0 and "barrier 3" in msg and "patch buffer 1" in msg, msg
    assert _w4_check(harness, 4, 8, 0, 7, 0, 0, 2)[0] == 0


def test_conv_w4_queue_model_is_not_stricter_than_the_closed_form(harness):
    """With the cold burst issued whole (spread <= 1) the queue model must reproduce r03's hand-derived counts exactly -- a model
    that over-waits would pass the coverage test and silently cost time."""
    import subprocess, tempfile, textwrap
    src = textwrap.dedent("""
        #include "w4_sched.h"
        #include <cstdio>
        int main() {
          for (int pd = 2; pd <= 4; ++pd) for (int pw = 8; pw <= 12; pw += 2) for (int d = 0; d < 9; ++d)
            if (w4_wait_n(d, pd, pw, 0, 0, 0) != w4_wait_n_r03(d, pd, pw)) { printf("%d %d %d: %d vs %d\\n", pd, pw, d, w4_wait_n(d, pd, pw, 0, 0, 0), w4_wait_n_r03(d, pd, pw)); return 1; }
          static_assert(w4_wait_n(3, 4, 8, 0, 0, 0) == 8 && w4_wait_n(1, 4, 8, 0, 0, 0) == 16 && w4_wait_n(0, 4, 8, 0, 14, 0) == 22, "constexpr");
          return 0;
        }""")
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "m.cpp")
        open(f, "w").write(src)
        exe = os.path.join(td, "m")
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "flope_amd", "csrc"), "-o", exe, f])
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
