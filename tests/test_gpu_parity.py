"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded
inputs.  Tolerances (written here, per BASELINE north_star "within 1e-3"):
  f32 strict mode : every stage |err| <= 2e-4 * max|ref|, R entries <= 1e-4
  f16 / bf16      : vs the oracle that emulates the same 16-bit storage: rel-L2 <= 2e-3 / 1e-2
                    (fp32 accumulation-order differences + occasional 1-ulp storage flips)
  f16 (default)   : R entries within 1e-3 of the fp32 oracle, geodesic angle < 0.1 deg
  bf16            : R entries within 1e-2 of the fp32 oracle (8-bit mantissa; documented miss of 1e-3)
"""
import numpy as np
import pytest
import torch

from oracle import pipeline_ref as P
from oracle import posenet_ref as O

pytestmark = pytest.mark.gpu

STAGES = ["stem", "pool"] + [f"layer{li}.{bi}" for li in range(1, 5) for bi in range(2)] + ["feat", "hidden"]
TDT = {"f16": torch.float16, "bf16": torch.bfloat16}


def _engine(state_dict, H, W, B, dtype, **opts):
    from flope_amd.engine import PoseEngine
    e = PoseEngine(H, W, B, dtype)
    for k, v in opts.items():
        e.set_option(k, v)
    e.load_state_dict(state_dict)
    return e


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def _run(e, x):
    r9, R = e.forward(x.cuda())
    torch.cuda.synchronize()
    return r9.cpu(), R.cpu()


@pytest.mark.parametrize("H,W,B", [(224, 224, 4), (96, 80, 3), (65, 71, 2)])
def test_f32_strict_mode_every_stage(state_dict, H, W, B):
    torch.manual_seed(10)
    x = torch.rand(B, 3, H, W)
    ref = O.forward_stages(state_dict, x)
    e = _engine(state_dict, H, W, B, "f32")
    r9, R = _run(e, x)
    for s in STAGES:
        got = e.read_stage(s, B).cpu()
        assert got.shape == ref[s].shape, s
        assert (got - ref[s]).abs().max() <= 2e-4 * ref[s].abs().max(), s
    assert (r9 - ref["r9"]).abs().max() < 2e-4
    assert (R - O.procrustes_to_rotmat(ref["r9"])).abs().max() < 1e-4
    e.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("opts", [dict(patch=1, bm256=1, nbuf=3, fuse_stem=1, stag=2, stem_persist=2, reslds=0, prio=2), dict(patch=0, bm256=0, nbuf=3, fuse_stem=0, stag=0),
                                  dict(patch=1, bm256=0, nbuf=2, fuse_stem=0, stag=0), dict(patch=0, bm256=1, nbuf=2, fuse_stem=1, stag=1, dsfuse=0, gstag=0, stem_persist=0, skew=0, prio=1)])
@pytest.mark.parametrize("H,W,B", [(224, 224, 5), (96, 80, 3), (65, 71, 2)])
def test_mfma_path_every_stage_vs_emulating_oracle(state_dict, dtype, opts, H, W, B):
    torch.manual_seed(11)
    x = torch.rand(B, 3, H, W)
    emu = O.forward_stages_emulated(state_dict, x, TDT[dtype])
    e = _engine(state_dict, H, W, B, dtype, **opts)
    r9, _ = _run(e, x)
    tol = 2e-3 if dtype == "f16" else 1e-2
    for s in STAGES:
        if s == "stem" and opts["fuse_stem"]:
            with pytest.raises(RuntimeError, match="not materialised"):
                e.read_stage(s, B)
            continue
        got = e.read_stage(s, B).cpu()
        assert got.shape == emu[s].shape, s
        assert _rel(got, emu[s]) <= tol, (s, _rel(got, emu[s]), e.describe_plan())
    assert _rel(r9, emu["r9"]) <= tol
    e.close()


@pytest.mark.parametrize("H,W,B,streams", [(224, 224, 40, 1), (224, 224, 70, 2), (96, 80, 3, 1), (64, 256, 2, 1)])
def test_layer1_row_band_kernel_every_stage(state_dict, H, W, B, streams):
    """stag=3 (default): layer 1 as persistent 8-row bands.  B = 40 x 7 bands = 280 tiles on <= 256 workgroups, so some
    workgroups walk two tiles (next-tile patch + residual prefetch); 64x256 has a 64-wide layer 1 (no padded columns);
    96x80 a 20-wide one (44 padded columns per row)."""
    torch.manual_seed(13)
    x = torch.rand(B, 3, H, W)
    emu = O.forward_stages_emulated(state_dict, x, torch.float16)
    e = _engine(state_dict, H, W, B, "f16", stag=3, streams=streams)
    assert "8-row bands" in e.describe_plan()
    if (H, W) == (224, 224):
        assert e.describe_plan().count("shortcut folded in") == 3     # layer2/3/4 .0.conv2 carry their 1x1 stride-2 shortcut
    r9, _ = _run(e, x)
    for s in STAGES:
        if s == "stem":
            continue
        assert _rel(e.read_stage(s, B).cpu(), emu[s]) <= 2e-3, s
    assert _rel(r9, emu["r9"]) <= 2e-3
    e.close()


@pytest.mark.parametrize("H,W,B,dtype", [(224, 224, 64, "f16"), (224, 224, 19, "f16"), (224, 224, 32, "bf16"), (200, 136, 7, "f16"), (512, 512, 2, "f16")])
def test_conflict_free_patch_image_and_4_wave_kernel_do_not_change_a_bit(state_dict, H, W, B, dtype):
    """conv_w4 (default for the flat 256 x 128 tiles of layers 2-4: four waves, one per SIMD, fragment reads and LDS-DMA issued
    in the gaps of the wave's own MFMAs, one barrier per double step) walks K in the same order per accumulator as conv_stag (two
    staggered 4-wave groups; still the split-K path), folds the shortcut first and adds the residual last, as conv_stag does:
    bit-identical -- at every SHIPPED tile height (w4mt = 8 .. 5 pixel tiles per wave, 0 = chosen per launch) and in the class walk
    (w4cw = tiles per persistent workgroup aimed at, 4 = the default where a launch has the chip to itself: the address table is
    built once, the residual comes by register loads, the boundary is pointer bumps; B = 64 / 32 at 224 x 224 make layers 2, 3
    and 4 / 2 and 3 walk; w4cwf = 3: also with two slices in flight and where a walk leaves CUs idle -- autotune's candidate), with
    line-order stores.  r05: cut to the variants the planner can choose (w8, the r02 image `skew = 0`, 128-pixel
    tiles and the 2- / 16-tile walks are gone).  Odd map widths (25 / 13 / 7 at 200 x 136: ragged last tiles, image-boundary
    crossings inside pixel tiles), the folded shortcut and the 512 x 512 shape are in the set."""
    torch.manual_seed(21)
    x = torch.rand(B, 3, H, W)
    outs, kernels = [], []
    cfgs = [dict(w4mt=7, w4cw=0), dict(w4mt=8), dict(w4mt=0, w4cw=0), dict(w4mt=6), dict(w4mt=5), dict(w4cw=4, w4cwf=2), dict(w4cw=4, w4cwf=3, streams=2), dict(w4=0)]
    if (H, W) != (224, 224):                               # the class walk needs M % 224 == 0 and the small tile heights their patch fit: 224 x 224 only
        cfgs = [dict(w4mt=7, w4cw=0), dict(w4mt=8), dict(w4mt=0, w4cw=0), dict(), dict(w4=0)]
    for cfg in cfgs:
        opts = dict(streams=1, ksplit=0, s1r=0)            # (split-K sends small launches to conv_stag: covered by test_split_k_small_batches; s1r = 0: layer2.1 on conv_w4 too -- conv_s1r has its own test)
        opts.update(cfg)
        e = _engine(state_dict, H, W, B, dtype, **opts)
        r9, R = _run(e, x)
        outs.append([r9, R] + [e.read_stage(s, B).cpu() for s in STAGES if s != "stem"])
        kernels.append(" ".join(f"{layer}|{k}" for layer, k, _ in e.launch_info(B)))
        e.close()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    # the kernels the forwards really launched (flope_launch_info reports the launch-time decision)
    if (H, W) == (224, 224):
        assert kernels[0].count("conv_w4_kernel") == 9 and "[224 px tiles]" in kernels[0], kernels[0]
        assert kernels[1].count("[256 px tiles]") == 9, kernels[1]
        assert "conv_w4" not in kernels[-1] and kernels[-1].count("conv_stag_kernel<256x128>") == 9
        walks = kernels[5].count("walk:")                  # (w4cwf = 2: also where a walk leaves CUs idle, as these small batches do)
        assert walks == (9 if B == 64 else 6 if B == 32 else 0), kernels[5]
        if B == 64:
            assert "walk:" in kernels[6], kernels[6]
    emu = O.forward_stages_emulated(state_dict, x, TDT[dtype])
    assert _rel(outs[0][0], emu["r9"]) <= (2e-3 if dtype == "f16" else 1e-2)


@pytest.mark.parametrize("H,W,B,dtype", [(224, 224, 9, "f16"), (224, 224, 3, "bf16"), (96, 80, 3, "f16"), (65, 71, 2, "f16"), (512, 512, 2, "f16")])
def test_register_weight_stem_does_not_change_a_bit(state_dict, H, W, B, dtype):
    """stem_pool_r_kernel (r05, default: output channels split over the waves, weights in registers, one pixel tile at a time, three
    workgroups per CU, conflict-free conv-output image) against the r02 forms (option stem_r = 0: persistent and one-tile-per-
    workgroup kernels): the same MFMAs in the same order per output -> the pooled map and everything behind it bit for bit, for
    every input format (float32 NCHW, 16-bit NHWC of either type, uint8 NHWC), ragged last tiles (65 x 71 -> 17 x 18 pooled) and
    tiles on all four image borders."""
    torch.manual_seed(31)
    x = (torch.rand(B, 3, H, W) * 255).round() / 255
    nh = x.permute(0, 2, 3, 1).contiguous()
    inputs = [x.cuda(), nh.to(TDT[dtype]).cuda(), nh.to(torch.float16 if dtype == "bf16" else torch.bfloat16).cuda(), (nh * 255).round().to(torch.uint8).cuda()]
    outs = []
    for opts in (dict(stem_r=1), dict(stem_r=0, stem_persist=2), dict(stem_r=0, stem_persist=0)):
        e = _engine(state_dict, H, W, B, dtype, **opts)
        row = []
        for xi in inputs:
            r9, R = e.forward(xi)
            row.append([r9.cpu(), R.cpu(), e.read_stage("pool", B).cpu()])
        outs.append(row)
        e.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            for u, v in zip(a, b):
                assert torch.equal(u, v)
    emu = O.forward_stages_emulated(state_dict, x, TDT[dtype])
    assert _rel(outs[0][0][2], emu["pool"]) <= (2e-3 if dtype == "f16" else 1e-2)


@pytest.mark.parametrize("B,dtype,streams", [(3, "f16", 1), (40, "bf16", 1), (150, "f16", 2)])
def test_stride2_patch_kernel_against_the_gathered_tile_kernel(state_dict, B, dtype, streams):
    """conv_s2r (r05, default for layer2.0.conv1 on 224 x 224 crops: 4-row bands, de-interleaved patch in LDS, weights through
    registers) against conv_mfma<gather> (option s2r = 0).  The K order differs (half-chunk, tap, channel against tap, channel), so
    the two agree within accumulation-order rounding, not bit for bit; both sit inside the stage tolerance of the emulating
    oracle.  B = 3: 21 tiles on 512 workgroups; B = 150: 1050 tiles, two or three per workgroup (next-tile patch prefetch)."""
    torch.manual_seed(17)
    x = torch.rand(B, 3, 224, 224)
    tol = 2e-3 if dtype == "f16" else 1e-2
    emu = O.forward_stages_emulated(state_dict, x[:8], TDT[dtype])
    outs = []
    for s2r in (1, 0):
        e = _engine(state_dict, 224, 224, B, dtype, s2r=s2r, streams=streams)
        r9, R = _run(e, x)
        kernels = [k for _, k, _ in e.launch_info(B)]
        assert any("conv_s2r_kernel" in k for k in kernels) == bool(s2r), kernels
        outs.append((r9, e.read_stage("layer2.0", B).cpu()))
        e.close()
    n = min(B, 8)
    for r9, l2 in outs:
        assert _rel(l2[:n], emu["layer2.0"][:n]) <= tol
        assert _rel(r9[:n], emu["r9"][:n]) <= tol
    assert _rel(outs[0][1], outs[1][1]) <= tol / 2
    # a width the kernel does not take (20-wide layer 2) stays on the gathered-tile kernel
    e = _engine(state_dict, 96, 80, 2, "f16")
    _run(e, torch.rand(2, 3, 96, 80))
    assert not any("conv_s2r_kernel" in k for _, k, _ in e.launch_info(2))
    e.close()


@pytest.mark.parametrize("B,dtype,streams", [(3, "f16", 1), (40, "bf16", 1), (150, "f16", 2)])
def test_register_weight_layer2_kernel_against_conv_w4(state_dict, B, dtype, streams):
    """conv_s1r (r05, default for layer2.0.conv2 / layer2.1.conv1 / conv2 on 224 x 224 crops: K split over wave pairs, all weights in
    registers, partial sums swapped through LDS; 2.1.conv2 with the residual, 2.0.conv2 with the block's 1x1 stride-2 shortcut folded in
    as one more step) against conv_w4 (option s1r = 0): different K order, so equal within accumulation-order rounding; both inside
    the emulating oracle's stage tolerance.  B = 150: 1050 bands on 256 workgroups (four or five per workgroup: patch
    double-buffering, swap-slot reuse).  With dsfuse = 0 the shortcut is its own launch and 2.0.conv2 takes it as a residual."""
    torch.manual_seed(19)
    x = torch.rand(B, 3, 224, 224)
    tol = 2e-3 if dtype == "f16" else 1e-2
    n = min(B, 8)
    emu = O.forward_stages_emulated(state_dict, x[:n], TDT[dtype])
    outs = []
    for s1r in (1, 0):
        e = _engine(state_dict, 224, 224, B, dtype, s1r=s1r, streams=streams)
        r9, R = _run(e, x)
        kernels = [k for _, k, _ in e.launch_info(B)]
        assert sum("conv_s1r_kernel" in k for k in kernels) == (3 if s1r else 0), kernels
        outs.append((r9, e.read_stage("layer2.1", B).cpu(), e.read_stage("layer2.0", B).cpu()))
        e.close()
    for r9, l21, l20 in outs:
        assert _rel(l20[:n], emu["layer2.0"][:n]) <= tol
        assert _rel(l21[:n], emu["layer2.1"][:n]) <= tol
        assert _rel(r9[:n], emu["r9"][:n]) <= tol
    assert _rel(outs[0][1], outs[1][1]) <= tol / 2 and _rel(outs[0][2], outs[1][2]) <= tol / 2
    if B == 3:                                             # the un-folded shortcut: 2.0.conv2 = conv_s1r with a residual
        e = _engine(state_dict, 224, 224, B, dtype, dsfuse=0, streams=streams)
        r9, _ = _run(e, x)
        assert sum("conv_s1r_kernel" in k for _, k, _ in e.launch_info(B)) == 3
        assert _rel(e.read_stage("layer2.0", B).cpu()[:n], emu["layer2.0"][:n]) <= tol and _rel(r9[:n], emu["r9"][:n]) <= tol
        e.close()


@pytest.mark.parametrize("dtype,rtol,deg", [("f16", 1e-3, 0.1), ("bf16", 1e-2, 1.0)])
def test_rotations_vs_fp32_oracle_cfg1(state_dict, golden_cfg1, dtype, rtol, deg):
    """BASELINE cfg1 inputs (16 seeded 224x224 crops) against the committed goldens."""
    torch.manual_seed(0)
    x = torch.rand(16, 3, 224, 224)
    e = _engine(state_dict, 224, 224, 16, dtype)
    r9, R = _run(e, x)
    Rg = torch.from_numpy(golden_cfg1["R"])
    assert (R - Rg).abs().max() <= rtol, float((R - Rg).abs().max())
    assert O.geodesic_deg(R, Rg).max() <= deg
    assert _rel(r9, torch.from_numpy(golden_cfg1["r9"])) <= (2e-3 if dtype == "f16" else 1e-2)
    # input formats: 16-bit NHWC and uint8 NHWC agree with the f32 NCHW path on representable inputs (the persistent stem
    # kernel, the default at this crop size, has one instantiation per format)
    xq = (x * 255).round() / 255
    r_f32, _ = _run(e, xq)
    u8 = (xq * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    r_u8, _ = e.forward(u8.cuda()); r_u8 = r_u8.cpu()
    assert (r_u8 - r_f32).abs().max() <= 2e-3
    nh = xq.permute(0, 2, 3, 1).contiguous().to(TDT[dtype])
    r_nh, _ = e.forward(nh.cuda()); r_nh = r_nh.cpu()
    assert torch.equal(r_nh, _run(e, nh.float().permute(0, 3, 1, 2).contiguous())[0])
    e.close()


def test_packed_16bit_epilogue_helpers_on_every_bit_pattern():
    """csrc/common.h: pk_out16 (every conv epilogue: optional ReLU as a signed-integer max, float16 saturation at +-65504), pk_relu16
    (stem: ReLU + saturation as two integer instructions) and pk_max16_nonneg (max-pool on non-negative values as an unsigned
    integer max) -- the DEVICE code over all 65,536 patterns of both halves of a packed word, against their float
    statements in numpy."""
    import ctypes as C
    from flope_amd import _lib
    lib = _lib.load()
    lo = np.arange(65536, dtype=np.uint32)
    words = torch.from_numpy((lo | (((lo * 40503 + 12345) & 0xffff) << 16)).astype(np.int64)).to(torch.int32).cuda()     # every pattern in the low half, a permutation of them in the high half
    out = torch.empty_like(words)

    def run(which, relu):
        rc = lib.flope_debug_pk16(which, relu, C.c_void_p(words.data_ptr()), C.c_void_p(out.data_ptr()), words.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint32).copy()

    w = words.cpu().numpy().view(np.uint32)
    halves = lambda a: np.stack([a & 0xffff, a >> 16]).astype(np.uint16)
    hw = halves(w)
    # float16
    x = hw.view(np.float16).astype(np.float32)
    finite = np.isfinite(x)
    sat = lambda v: np.clip(v, -65504.0, 65504.0).astype(np.float16).view(np.uint16)
    got = halves(run(1, 0))
    assert np.array_equal(got[finite], hw[finite])                                    # identity on finite values (they are all within +-65504)
    inf = np.isinf(x)
    assert np.array_equal(got[inf], sat(x)[inf])                                      # +-inf saturate
    got_r = halves(run(1, 1))
    ref_r = sat(np.maximum(np.nan_to_num(x, nan=0.0), 0.0))
    pos_nan = np.isnan(x) & (hw < 0x8000)
    ok = ~np.isnan(x)
    assert np.array_equal(got_r[ok], ref_r[ok]) and (got_r[pos_nan] == 0x7BFF).all()  # ReLU + saturation; a positive NaN leaves as 65504, a negative one as 0
    assert (got_r[np.isnan(x) & (hw >= 0x8000)] == 0).all()
    assert np.array_equal(halves(run(3, 0)), got_r)                                   # the stem's integer form: the same bits for EVERY pattern
    # bfloat16: ReLU or identity, no saturation
    assert np.array_equal(run(0, 0), w)
    got_b = halves(run(0, 1))
    assert np.array_equal(got_b, np.where(hw >= 0x8000, 0, hw).astype(np.uint16))
    assert np.array_equal(halves(run(2, 0)), got_b)
    # max of non-negative 16-bit floats = unsigned integer max (the pool works on post-ReLU values)
    nn = torch.from_numpy(((lo & 0x7fff) | ((((lo * 40503 + 12345) & 0x7fff)) << 16)).astype(np.int64)).to(torch.int32).cuda()
    words.copy_(nn)
    a = halves(words.cpu().numpy().view(np.uint32)); b = np.roll(a, -1, axis=1)
    fa, fb = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
    both = ~np.isnan(fa) & ~np.isnan(fb)
    got_m = halves(run(4, 0))
    assert np.array_equal(got_m.view(np.float16).astype(np.float32)[both], np.maximum(fa, fb)[both])


def test_reference_true_shape_512(state_dict):
    torch.manual_seed(12)
    x = torch.rand(2, 3, 512, 512)
    ref = O.forward_stages(state_dict, x)
    e = _engine(state_dict, 512, 512, 2, "f16")
    r9, R = _run(e, x)
    assert (R - O.procrustes_to_rotmat(ref["r9"])).abs().max() <= 1e-3
    emu = O.forward_stages_emulated(state_dict, x, torch.float16)
    for s in ("pool", "layer1.1", "layer2.0", "layer4.1"):
        assert _rel(e.read_stage(s, 2).cpu(), emu[s]) <= 2e-3, s
    e.close()


@pytest.mark.parametrize("H,W,B", [(512, 512, 5), (320, 640, 3), (288, 400, 2)])
def test_layer1_column_segment_bands_vs_emulating_oracle(state_dict, H, W, B):
    """Layer 1 of crops wider than 256 pixels (the reference's own 512 x 512: a 128-wide map) runs as 8-row bands cut
    into 64-column segments (10 x 66-pixel patches, per-lane DMA source mapping).  512x512: two full segments, 16 bands,
    B = 5 -> 160 tiles, some workgroups walk two tiles; 320x640: 160 wide = 2.5 segments (ragged last segment);
    288x400: 72 x 100 map (Ho % 8 == 0, 1.56 segments).  Same result as the flat-tile kernel (rowseg=0) up to fp32
    summation order, and both within the 16-bit tolerance of the emulating oracle."""
    torch.manual_seed(15)
    x = torch.rand(B, 3, H, W)
    emu = O.forward_stages_emulated(state_dict, x, torch.float16)
    e = _engine(state_dict, H, W, B, "f16")
    assert "column segments" in e.describe_plan()
    r9, _ = _run(e, x)
    got = {s: e.read_stage(s, B).cpu() for s in ("pool", "layer1.0", "layer1.1", "layer2.0", "layer4.1")}
    for s, g in got.items():
        assert _rel(g, emu[s]) <= 2e-3, (s, _rel(g, emu[s]))
    assert _rel(r9, emu["r9"]) <= 2e-3
    e2 = _engine(state_dict, H, W, B, "f16", rowseg=0)
    assert "column segments" not in e2.describe_plan()
    r9b, _ = _run(e2, x)
    assert _rel(e2.read_stage("layer1.1", B).cpu(), got["layer1.1"]) <= 3e-4
    assert (r9 - r9b).abs().max() <= 2e-3
    e.close(); e2.close()


def test_full_batch_properties_cfg2(state_dict):
    """B = 256 at 224x224 (the bench workload): size-independent properties."""
    B = 256
    e = _engine(state_dict, 224, 224, B, "f16")
    g = torch.Generator().manual_seed(13)
    x = torch.rand(B, 224, 224, 3, generator=g).to(torch.float16).cuda()
    r9a, Ra = e.forward(x)
    r9b, Rb = e.forward(x)
    assert torch.equal(r9a, r9b) and torch.equal(Ra, Rb)                       # deterministic
    perm = torch.randperm(B, generator=g).cuda()
    r9p, _ = e.forward(x[perm].contiguous())
    assert torch.equal(r9p, r9a[perm])                                         # crops are independent
    eye = torch.eye(3, device="cuda").expand(B, 3, 3)
    assert (Ra @ Ra.transpose(1, 2) - eye).abs().max() < 1e-5
    assert (torch.det(Ra) - 1).abs().max() < 1e-5
    # a 16-crop sub-batch of the big batch equals the same crops run alone (tile boundaries do not leak): bit for bit
    # with the same K order; small batches normally split the K loop over several workgroups (fp32 partial sums added in
    # a different order), which may move the last bits only
    r9k, _ = e.forward(x[:16].contiguous())
    assert _rel(r9k.cpu(), r9a[:16].cpu()) <= 1e-4
    e.set_option("ksplit", 0)
    r9s, _ = e.forward(x[:16].contiguous())
    assert torch.equal(r9s, r9a[:16])
    e.set_option("ksplit", 1)
    # and matches the oracle on a sample
    ref = O.forward_stages_emulated(state_dict, x[:8].float().permute(0, 3, 1, 2).cpu(), torch.float16)["r9"]
    assert _rel(r9a[:8].cpu(), ref) <= 2e-3
    e.close()


@pytest.mark.parametrize("H,W,B", [(224, 224, 1), (224, 224, 4), (128, 96, 9)])
def test_split_k_small_batches(state_dict, H, W, B):
    """Small batches split each tile's K loop over several workgroups (fp32 partials + a finalize kernel that owns bias,
    residual, folded-shortcut bias and ReLU): same numbers as the unsplit kernels up to fp32 summation order, and
    deterministic."""
    torch.manual_seed(B)
    x = torch.rand(B, 3, H, W)
    e = _engine(state_dict, H, W, B, "f16")
    r9a, _ = _run(e, x)
    stages = {s: e.read_stage(s, B).cpu() for s in STAGES if s != "stem"}
    r9b, _ = _run(e, x)
    assert torch.equal(r9a, r9b)
    e.set_option("ksplit", 0)
    r9n, _ = _run(e, x)
    for s in stages:
        assert _rel(stages[s], e.read_stage(s, B).cpu()) <= 1e-3, s       # 16-bit activations: a last-bit flip is 5e-4 of an element
    assert _rel(r9a, r9n) <= 1e-3
    e.close()


@pytest.mark.parametrize("B", [1, 37, 256])
def test_fc1_packed_weights_equal_the_row_major_kernel(state_dict, B):
    """fc.0 on pre-packed (A-fragment order) weights with LDS-staged feature rows runs the same MFMA sequence per output as
    the row-major kernel: `hidden` and the rotations are identical, also for batches that are not multiples of 32."""
    torch.manual_seed(B)
    x = torch.rand(B, 96, 96, 3).to(torch.float16).cuda()
    e = _engine(state_dict, 96, 96, B, "f16")
    r9a, Ra = e.forward(x)
    ha = e.read_stage("hidden", B).clone()
    assert e.set_option("fc1_packed", 0) == 1
    r9b, Rb = e.forward(x)
    assert torch.equal(ha, e.read_stage("hidden", B)) and torch.equal(r9a, r9b) and torch.equal(Ra, Rb)
    e.close()


@pytest.mark.parametrize("B,S", [(1, 96), (37, 96), (256, 224), (13, 256)])
def test_fc_rot_split_over_four_waves_equals_the_one_wave_kernel(state_dict, B, S):
    """fc_rot with K split over the four waves of a workgroup (default) sums in another (fixed) order than the one-wave kernel:
    same rotations within float32 rounding, same bits from run to run."""
    torch.manual_seed(B)
    x = torch.rand(B, S, S, 3).to(torch.float16).cuda()
    e = _engine(state_dict, S, S, B, "f16")
    assert e.set_option("fc2_k4", 0) == 1
    r9a, Ra = e.forward(x)
    assert e.set_option("fc2_k4", 1) == 0                 # (the default)
    r9c, Rc = e.forward(x)
    r9d, Rd = e.forward(x)
    assert torch.equal(r9c, r9d) and torch.equal(Rc, Rd)
    assert (r9a - r9c).abs().max() <= 1e-5 * max(1.0, float(r9c.abs().max())) and (Ra - Rc).abs().max() <= 2e-5
    e.close()


def test_autotune_picks_a_candidate_and_changes_no_bit(state_dict):
    """PoseEngine.autotune (bench.py's set-up step): times the schedule candidates on this device and leaves the fastest in force.
    Every candidate computes the same bits, so the rotations before and after are identical; the report names the choice and a
    median per candidate."""
    torch.manual_seed(5)
    B = 128
    x = torch.rand(B, 224, 224, 3).to(torch.float16).cuda()
    e = _engine(state_dict, 224, 224, B, "f16")
    cands = e.tune_candidates(B)
    r9a, Ra = e.forward(x)
    rep = e.autotune(x, 2, rounds=2, per_round=2)
    assert rep["chosen"] in cands and len(rep["median_ms"]) == len(cands)
    assert all(0.0 < v < 100.0 for v in rep["median_ms"].values())
    for k, v in rep["chosen"].items():
        assert e.set_option(k, v) == v                     # (the chosen values are the ones in force)
    r9b, Rb = e.forward(x)
    assert torch.equal(r9a, r9b) and torch.equal(Ra, Rb)
    e.close()


@pytest.mark.parametrize("B", [256, 203])
def test_every_autotune_candidate_is_bit_identical(state_dict, B):
    """the bench's headline may run ANY of the candidates, so each one is forced in turn -- at the bench's batch and at an
    odd one -- and must reproduce the default schedule's r9 / R bit for bit.  Slice fractions resolve to absolute image counts on
    multiples of 8 (PoseEngine.tune_candidates): 96 / 160 and 112 / 144 at B = 256."""
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 224, 224, 3, generator=g).to(torch.float16).cuda()
    e = _engine(state_dict, 224, 224, 256, "f16")
    cands = e.tune_candidates(B)
    assert {} in cands
    if B == 256:
        assert {"split": 196} in cands and {"split": 212} in cands
    for c in cands:
        if "split" in c:
            assert (c["split"] - 100) % 8 == 0
    r9a, Ra = e.forward(x)
    for c in cands:
        prev = {k: e.set_option(k, v) for k, v in c.items()}
        r9b, Rb = e.forward(x)
        assert torch.equal(r9a, r9b) and torch.equal(Ra, Rb), c
        for k, v in prev.items():
            e.set_option(k, v)
    e.close()


@pytest.mark.parametrize("B", [131, 200, 255])
def test_slice_split_is_invisible(state_dict, B):
    """The internal two-slice split (3/8 : 5/8 on multiples of 8 images, row-band grids proportional to the slice) must
    not change a single bit relative to one slice, for batches that do not divide nicely."""
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 224, 224, 3, generator=g).to(torch.float16).cuda()
    outs = []
    for streams in (1, 2):
        e = _engine(state_dict, 224, 224, 256, "f16", streams=streams, ksplit=0)   # same K order in both runs
        r9, R = e.forward(x)
        outs.append((r9.clone(), R.clone()))
        e.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_forward_poses_equals_forward_plus_compose(state_dict):
    """flope_forward_poses (pose assembly inside the head kernel) == flope_forward followed by flope_compose_pose."""
    from flope_amd import engine as E
    B = 70
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 224, 224, 3, generator=g).to(torch.float16).cuda()
    xyz = torch.rand(B, 3, generator=g).cuda()
    e = _engine(state_dict, 224, 224, B, "f16")
    _, R = e.forward(x)
    for nullify in (True, False):
        ref = E.compose_pose(R, xyz, nullify=nullify)
        Rt = torch.empty(B, 16, device="cuda")
        R2 = torch.empty(B, 9, device="cuda")
        e.forward_poses_into(x, 2, xyz, nullify, Rt, R2)
        assert torch.equal(Rt.view(B, 4, 4), ref) and torch.equal(R2.view(B, 3, 3), R)
    Rt0 = torch.empty(B, 16, device="cuda")
    e.forward_poses_into(x, 2, None, False, Rt0)                      # no translation: zeros
    assert torch.equal(Rt0.view(B, 4, 4)[:, :3, 3], torch.zeros(B, 3, device="cuda"))
    assert torch.equal(Rt0.view(B, 4, 4)[:, :3, :3], R)
    e.close()


def test_procrustes_yaw_compose_kernels():
    from flope_amd import engine as E
    g = torch.Generator().manual_seed(14)
    M = torch.randn(1000, 9, generator=g)
    sv = O.singular_values(M)
    well = (sv[:, 1] + sv[:, 2]) > 0.05
    R = E.procrustes(M.cuda()).cpu()
    ref = O.special_procrustes(M.double()).float()
    assert (R[well] - ref[well]).abs().max() < 5e-5
    assert E.procrustes(M[:0].cuda()).shape == (0, 3, 3)                       # empty input
    Ry = E.nullify_yaw(R.cuda()).cpu()
    np.testing.assert_allclose(Ry[well].numpy(), P.nullify_yaw_batch(R[well].double().numpy()), atol=5e-6)
    xyz = torch.randn(1000, 3, generator=g)
    Rt = E.compose_pose(R.cuda(), xyz.cuda(), True).cpu()
    assert torch.equal(Rt[:, :3, 3], xyz) and torch.equal(Rt[:, 3], torch.tensor([0., 0, 0, 1]).expand(1000, 4))
    np.testing.assert_allclose(Rt[:, :3, :3][well].numpy(), Ry[well].numpy(), atol=1e-6)
    # CPU tensors are moved to the GPU and back (still the HIP kernel)
    assert (E.procrustes(M[:4]) - R[:4]).abs().max() == 0


def _scene(seed, H=480, W=640, n=6):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    depth = (400 + rng.normal(0, 4, (H, W))).astype(np.uint16)
    boxes = []
    yy, xx = np.mgrid[:H, :W]
    for i in range(n):
        r = int(rng.integers(22, 60)); cx = int(rng.integers(r + 5, W - r - 5)); cy = int(rng.integers(r + 5, H - r - 5))
        mask[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
        w2, h2 = r + int(rng.integers(0, 9)), r + int(rng.integers(0, 9))
        boxes.append([cx - w2, cy - h2, cx + w2 + 1, cy + h2])
    boxes.append([W - 30, 10, W - 2, 90])                                      # squarified box leaves the frame
    boxes.append([5, H - 40, 25, H - 20])                                      # no mask -> unreliable depth
    depth[:50, :50] = 0
    return rgb, mask, depth, np.array(boxes, dtype=np.int16)


def test_crop_resize_mask_vs_oracle():
    """Crop + cv2-style INTER_LANCZOS4 + mask multiply (fast_pose_predictor.py:108-123): BIT-EXACT against the oracle --
    the uint8 resized image and mask (integer work) and the float32 product, which the device forms in double like the
    reference's numpy expression and rounds once."""
    from flope_amd import engine as E
    rgb, mask, _, boxes = _scene(20)
    _, sq, _ = P.select_boxes(boxes, rgb.shape)
    for S in (64, 512, 100, 24):         # 100: ragged 16 x 16 output tiles; 24: > 3.5x down-scaling (direct-sum path)
        sel = sq[:3] if S == 512 else sq
        got = E.crop_resize_mask(torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda(),
                                 torch.from_numpy(sel.astype(np.int32)).cuda(), S).cpu().numpy()
        ref = P.crop_batch(rgb, mask, sel, S).transpose(0, 3, 1, 2).astype(np.float32)    # the reference's torch.float32 cast
        assert got.shape == ref.shape and got.dtype == np.float32
        assert np.array_equal(got, ref), (S, int((got != ref).sum()), float(np.abs(got - ref).max()))
    # an all-255 mask exposes the resized uint8 image itself: (float32)(v * (255/255.0) / 255.0) is injective in v
    ones = np.full_like(mask, 255)
    got = E.crop_resize_mask(torch.from_numpy(rgb).cuda(), torch.from_numpy(ones).cuda(),
                             torch.from_numpy(sq.astype(np.int32)).cuda(), 96).cpu().numpy()
    u8 = np.stack([P.resize_lanczos4_u8(rgb[y0:y1, x0:x1], 96) for x0, y0, x1, y1 in sq]).transpose(0, 3, 1, 2)
    assert np.array_equal(np.rint(got * 255).astype(np.uint8), u8)
    # 16-bit NHWC outputs = the float32 crop rounded once more (what the stem would do with it)
    from flope_amd import _lib
    for fmt, tdt in ((_lib.IN_F16_NHWC, torch.float16), (_lib.IN_BF16_NHWC, torch.bfloat16)):
        g16 = E.crop_resize_mask(torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda(),
                                 torch.from_numpy(sq.astype(np.int32)).cuda(), 64, fmt).cpu()
        r16 = torch.from_numpy(P.crop_batch(rgb, mask, sq, 64).astype(np.float32)).to(tdt)
        assert torch.equal(g16, r16)


@pytest.mark.parametrize("n_src,n_dst", [(57, 512), (113, 512), (512, 512), (640, 224), (97, 64), (31, 100), (1000, 24),
                                         (7, 512), (1, 16), (333, 333), (119, 512), (2, 3)])
def test_lanczos_tables_bit_exact_vs_oracle(n_src, n_dst):
    """The int16 x2048 coefficient tables and tap positions the crop kernel evaluates on the device (double sin/cos,
    float normalisation, round-half-even) equal the oracle's for every destination index."""
    from flope_amd import engine as E
    s0, co = E.lanczos4_table(n_src, n_dst)
    _, wt, s = P._axis_table_raw(n_src, n_dst)
    assert np.array_equal(s0.cpu().numpy().astype(np.int64), s)
    assert np.array_equal(co.cpu().numpy().astype(np.int64), wt), int((co.cpu().numpy() != wt).sum())


@pytest.mark.parametrize("n,h,w,H,W", [(5, 160, 288, 1080, 1920), (3, 37, 53, 90, 160), (2, 64, 48, 40, 30), (4, 24, 32, 24, 32),
                                       (0, 16, 16, 20, 28), (1, 7, 5, 480, 640)])
def test_merge_masks_resize_bit_exact_vs_oracle(n, h, w, H, W):
    """get_bbox_mask post-processing (fast_pose_predictor.py:50-54) on the device: bit-exact against the restatement of
    cv2.resize's 8-bit INTER_LINEAR arithmetic (up- and down-scaling, non-integer ratios, unchanged size, no detections)."""
    from flope_amd import engine as E
    rng = np.random.default_rng(n * 1000 + h)
    masks = (rng.random((n, h, w)) < 0.3).astype(np.float32) * rng.choice([1.0, 0.4, 0.7], size=(n, 1, 1)).astype(np.float32)
    got = E.merge_masks_resize(torch.from_numpy(masks).cuda(), H, W).cpu().numpy()
    ref = P.merge_masks(masks, (W, H)) if n else np.zeros((H, W), np.uint8)
    assert got.shape == (H, W) and got.dtype == np.uint8
    assert np.array_equal(got, ref), int((got != ref).sum())


def test_depth_lift_vs_oracle():
    from flope_amd import engine as E
    rgb, mask, depth, boxes = _scene(21)
    _, _, good = P.select_boxes(boxes, rgb.shape)
    K = np.array([[600.0, 0, 320], [0, 600.0, 240], [0, 0, 1]])
    dv_ref, rel_ref = P.get_depth_value(good, depth.astype(np.float32) / 1000, mask, near_plane=0.1, far_plane=2.5)
    dv, rel, xyz = E.depth_lift(torch.from_numpy(depth.view(np.int16)).cuda(), torch.from_numpy(mask).cuda(),
                                torch.from_numpy(good.astype(np.int32)).cuda(), (600.0, 600.0, 320.0, 240.0), 1000.0, 0.1, 2.5)
    assert rel.cpu().numpy().tolist() == rel_ref.tolist() and rel_ref.any() and not rel_ref.all()
    np.testing.assert_allclose(dv.cpu().numpy(), dv_ref, atol=1e-5)
    uv = np.stack([(good[:, 0] + good[:, 2]) / 2, (good[:, 1] + good[:, 3]) / 2], 1)
    np.testing.assert_allclose(xyz.cpu().numpy()[rel_ref], P.get_points3d(uv, dv_ref, K)[rel_ref], atol=1e-5)
    # numpy-signature mirror
    from sunflower.utils.image_manipulation import get_depth_value
    d2, r2, _ = get_depth_value(good, depth.astype(np.float32) / 1000, mask, near_plane=0.1, far_plane=2.5)
    np.testing.assert_allclose(d2, dv_ref, atol=1e-5)
    assert r2.tolist() == rel_ref.tolist()


def test_depth_val_file_rows_vs_oracle(tmp_path):
    """scripts/extract_depth.py:25-57 -> the 2 x N depth_val file, and its round trip into the fusion step."""
    from flope_amd.harness import depth_val_rows, write_depth_val_file
    rgb, mask, depth, boxes = _scene(23)
    dv_ref, rel_ref = P.get_depth_value(boxes, depth.astype(np.float32) / 1000, mask, near_plane=0.1, far_plane=3.0)
    rows = depth_val_rows(depth, mask, boxes, depth_div=1000.0, near=0.1, far=3.0)
    assert rows.shape == (2, len(boxes)) and rows.dtype == np.float64
    np.testing.assert_allclose(rows[0], dv_ref, atol=1e-5)
    assert (rows[1] > 0.5).tolist() == rel_ref.tolist()
    write_depth_val_file(tmp_path / "d.txt", rows)
    np.testing.assert_allclose(np.loadtxt(tmp_path / "d.txt"), rows)
    assert depth_val_rows(depth, mask, np.zeros((0, 4))).shape == (0,)
    # float32 metre depth (the script's 'npy' branch) gives the same numbers
    rows_f = depth_val_rows(depth.astype(np.float32) / 1000, mask, boxes, near=0.1, far=3.0)
    np.testing.assert_allclose(rows_f[0], dv_ref, atol=1e-5)


def test_fast_pose_predictor_end_to_end_vs_oracle(state_dict, tmp_path):
    """BASELINE cfg3 (detections given): frame -> boxes+mask -> crop -> PoseNet -> Procrustes -> yaw-null -> Rt."""
    import yaml
    from sunflower.predictor.fast_pose_predictor import FastPosePredictor
    from sunflower.utils.conversion import procrustes_to_rotmat
    rgb, mask, depth, boxes = _scene(22)
    ckpt, intr = tmp_path / "posenet.pth", tmp_path / "intrinsics.yaml"
    torch.save(state_dict, ckpt)
    intr.write_text(yaml.safe_dump(dict(fx=600.0, fy=600.0, cx=320.0, cy=240.0, h=480, w=640)))
    pred = FastPosePredictor("cuda", lambda img: (boxes, mask), str(ckpt), str(intr))
    Rt = pred.get_flower_poses(rgb, depth)
    K = np.array([[600.0, 0, 320], [0, 600.0, 240], [0, 0, 1]])
    ref = P.get_flower_poses(lambda b: O.forward(state_dict, b), O.procrustes_to_rotmat, rgb, depth, boxes, mask, K)
    assert Rt.dtype == np.float64 and Rt.shape == ref.shape and Rt.shape[0] >= 3
    assert np.abs(Rt[:, :3, :3] - ref[:, :3, :3]).max() <= 1e-3                 # rot err
    assert np.linalg.norm(Rt[:, :3, 3] - ref[:, :3, 3], axis=1).max() <= 1e-5   # trans err (m)
    np.testing.assert_array_equal(Rt[:, 3], np.tile([0, 0, 0, 1.0], (Rt.shape[0], 1)))
    # "None" contracts (fast_pose_predictor.py:86-87,101-102)
    pred2 = FastPosePredictor("cuda", lambda img: (boxes[-2:-1], mask), str(ckpt), str(intr))
    assert pred2.get_flower_poses(rgb, depth) is None                            # no box survives squarify
    pred3 = FastPosePredictor("cuda", lambda img: (boxes[-1:], mask), str(ckpt), str(intr))
    assert pred3.get_flower_poses(rgb, depth) is None                            # no reliable depth
    # reference-style two-call sequence on the facade
    x = torch.rand(3, 3, 128, 128).cuda()
    r9 = pred.posenet(x)
    R = procrustes_to_rotmat(r9)
    assert (R.cpu() - O.procrustes_to_rotmat(O.forward(state_dict, x.cpu()))).abs().max() <= 1e-3
    assert pred.posenet.extract_features(x).shape == (3, 2048)


def test_frame_box_selection_on_the_device_is_integer_exact(state_dict):
    """flope_frame_select (csrc/frame.hip): detector rows -> int16 (numpy astype: truncation) -> squarify_bb -> bb_in_frame, in
    detection order, on the device == the host loop of fast_pose_predictor.py:65-83 (sunflower.utils.mvg mirrors + the oracle's
    select_boxes), integer for integer: the hand-derived KATs of SURVEY App. B 5 ([10,20,50,40] -> [10,10,50,50];
    [0,0,10,5] -> [0,-3,10,7], rejected; xmax == w / ymax == h accepted: the square [w-h,0,w,h]), random rows with fractional coordinates, counts that
    span several 64-lane ballots (0, 1, 64, 65, 300) and a count above the table (clamped to max_det)."""
    from flope_amd.engine import PoseEngine
    from flope_amd.frame import FramePoses
    from sunflower.predictor.fast_pose_predictor import select_boxes
    H, W = 480, 640
    eng = PoseEngine(64, 64, 4, "f16")
    eng.load_state_dict(state_dict)
    ctx = FramePoses(eng, H, W, max_boxes=300, slots=2)
    rng = np.random.default_rng(31)
    kat = np.array([[10, 20, 50, 40], [0, 0, 10, 5], [W - 40, 100, W, 140], [100, H - 40, 140, H], [3, 3, 4, 200], [W - H, 0, W, H],
                    [7, 9, 20, 22], [7, 9, 21, 22], [7, 9, 20, 24], [W - 11, 10, W - 1, 31]], dtype=np.float32)
    for count in (0, 1, 10, 64, 65, 300, 301):
        det = np.zeros((300, 8), np.float32)
        x0 = rng.uniform(0, W - 2, 300); y0 = rng.uniform(0, H - 2, 300)
        det[:, 0] = x0; det[:, 1] = y0
        det[:, 2] = np.minimum(x0 + rng.uniform(1, 220, 300), W); det[:, 3] = np.minimum(y0 + rng.uniform(1, 220, 300), H)
        det[:10, :4] = kat
        det[:, 4] = rng.uniform(0.25, 1, 300)
        n = min(count, 300)
        d_dev = torch.from_numpy(det).cuda()
        c_dev = torch.tensor([count], dtype=torch.int32, device="cuda")
        ctx.select(1, d_dev, c_dev)
        good, sq = ctx.read_boxes(1)
        bb = det[:n, :4].astype(np.int16)                                        # fast_pose_predictor.py:55-56
        _, sq_ref, good_ref = select_boxes(bb, (H, W, 3))
        assert np.array_equal(good, good_ref.astype(np.int32)) and np.array_equal(sq, sq_ref.astype(np.int32)), count
        _, sq_o, good_o = P.select_boxes(bb, (H, W, 3))                          # the oracle's statement of the same loop
        assert np.array_equal(good, np.asarray(good_o, dtype=np.int32).reshape(-1, 4)) and np.array_equal(sq, np.asarray(sq_o, dtype=np.int32).reshape(-1, 4))
        if count >= 10:
            assert [10, 10, 50, 50] in sq.tolist() and [0, -3, 10, 7] not in sq.tolist() and [W - H, 0, W, H] in sq.tolist()
    ctx.close(); eng.close()


def test_frame_to_poses_is_the_host_path_bit_for_bit(state_dict, tmp_path):
    """flope_frame_to_poses (one C call behind the detector: device-side box selection, count read-back, depth lift, crops,
    network, Rt, reliability filter) returns exactly the float64 poses of the r01-r04 host path (`poses_from_detections`: numpy box
    loop + one ctypes call per kernel), for detections given as the detector would leave them on the device -- including the
    frame whose only box leaves the frame (None) and the one without reliable depth (None)."""
    import yaml
    from flope_amd.frame import FramePoses
    from sunflower.predictor.fast_pose_predictor import FastPosePredictor, poses_from_detections, upload_depth
    rgb, mask, depth, boxes = _scene(24)
    ckpt, intr = tmp_path / "posenet.pth", tmp_path / "intrinsics.yaml"
    torch.save(state_dict, ckpt)
    intr.write_text(yaml.safe_dump(dict(fx=600.0, fy=600.0, cx=320.0, cy=240.0, h=480, w=640)))
    pred = FastPosePredictor("cuda", lambda img: (boxes, mask), str(ckpt), str(intr))
    eng = pred.posenet.engine_for("cuda", (512, 512))
    ctx = FramePoses(eng, 480, 640, max_boxes=300)
    frame_d, mask_d, depth_d = torch.from_numpy(rgb).cuda(), torch.from_numpy(mask).cuda(), upload_depth(depth, torch.device("cuda"))
    for sel in (boxes, boxes[-2:-1], boxes[-1:], boxes[:0]):
        det = torch.zeros((300, 8), dtype=torch.float32, device="cuda")
        if len(sel):
            det[:len(sel), :4] = torch.from_numpy(sel.astype(np.float32) + 0.4).cuda()      # (fractions: the int16 cast truncates)
        count = torch.tensor([len(sel)], dtype=torch.int32, device="cuda")
        got = ctx.to_poses(det, count, frame_d, mask_d, depth_d, pred.K)
        ref = poses_from_detections(pred.posenet, rgb, depth, sel, mask, pred.K, depth_div=1000.0, device="cuda")
        assert (got is None) == (ref is None)
        if ref is not None:
            assert got.dtype == np.float64 and np.array_equal(got, ref) and got.shape[0] >= 3
    # more boxes than the engine's batch: several forwards behind one call (engine max_batch 64: 150 boxes = 3 forwards)
    many = np.tile(boxes[:6], (25, 1))
    det = torch.zeros((300, 8), dtype=torch.float32, device="cuda")
    det[:150, :4] = torch.from_numpy(many.astype(np.float32)).cuda()
    got = ctx.to_poses(det, torch.tensor([150], dtype=torch.int32, device="cuda"), frame_d, mask_d, depth_d, pred.K)
    one = ctx.to_poses(det, torch.tensor([6], dtype=torch.int32, device="cuda"), frame_d, mask_d, depth_d, pred.K)
    # (B = 6 sums K in split-K order, B = 64 does not: equal up to float32 summation order in the 16-bit trunk; translations exact)
    rep = np.tile(one, (25, 1, 1))
    assert got.shape == (150, 4, 4) and np.abs(got - rep).max() <= 2e-3 and np.array_equal(got[:, :, 3], rep[:, :, 3])
    with pytest.raises(RuntimeError, match="select"):
        ctx.enqueue(0, frame_d, mask_d, depth_d, pred.K)                             # call order
    ctx.close()


def test_error_paths(state_dict):
    from flope_amd.engine import PoseEngine
    e = PoseEngine(64, 64, 2, "f16")
    with pytest.raises(RuntimeError, match="before flope_load_weights"):
        e.forward(torch.rand(1, 3, 64, 64).cuda())
    e.load_state_dict(state_dict)
    with pytest.raises(ValueError, match="exceeds max_batch"):
        e.forward(torch.rand(3, 3, 64, 64).cuda())
    with pytest.raises(ValueError, match="built for"):
        e.forward(torch.rand(1, 3, 32, 64).cuda())
    with pytest.raises(RuntimeError, match="no CPU path"):
        e.forward(torch.rand(1, 3, 64, 64))
    bad = dict(state_dict); bad["base.bn1.running_var"] = torch.full((64,), float("nan"))
    with pytest.raises(RuntimeError, match="non-finite"):
        e.load_state_dict(bad)
    e.close()


def test_detection_rows_harness_vs_oracle(state_dict, tmp_path):
    """scripts/test_posenet.py:104-161 counterpart: 15-column rows, '%.7f' text file."""
    from flope_amd.harness import detection_rows, write_detection_file
    from sunflower.models.posenet import PoseResNet
    rgb, mask, _, boxes = _scene(23)
    model = PoseResNet().to("cuda")
    model.load_state_dict(state_dict)
    rows = detection_rows(model, rgb, mask, boxes, crop_size=256)
    keep = [bb for bb in boxes if P.bb_in_frame(P.squarify_bb(bb), rgb.shape)]
    sq = [P.squarify_bb(bb) for bb in keep]
    crops = torch.as_tensor(P.crop_batch(rgb, mask, sq, 256), dtype=torch.float32).permute(0, 3, 1, 2)
    Rref = O.procrustes_to_rotmat(O.forward(state_dict, crops)).numpy()
    ref = P.detection_rows(np.array(keep), Rref)
    assert rows.shape == ref.shape == (len(keep), 15)
    np.testing.assert_array_equal(rows[:, :6], ref[:, :6])
    assert np.abs(rows[:, 6:] - ref[:, 6:]).max() <= 1e-3
    f = tmp_path / "frame_00000.txt"
    write_detection_file(f, rows)
    back = np.loadtxt(f).reshape(-1, 15)
    np.testing.assert_allclose(back, rows, atol=5e-8)
    assert all(len(tok.split(".")[1]) == 7 for tok in f.read_text().split())
    write_detection_file(f, detection_rows(model, rgb, mask, boxes[-2:-1], crop_size=256))   # nothing survives
    assert f.read_text() == ""


# ---- predictors and harnesses at the sizes / branches the reference has --------------------
def _write_ckpt(tmp_path, state_dict, h, w, fx=600.0):
    import yaml
    ckpt, intr = tmp_path / "posenet.pth", tmp_path / "intrinsics.yaml"
    torch.save(state_dict, ckpt)
    intr.write_text(yaml.safe_dump(dict(fx=fx, fy=fx, cx=w / 2, cy=h / 2, h=h, w=w)))
    K = np.array([[fx, 0, w / 2], [0, fx, h / 2], [0, 0, 1]])
    return str(ckpt), str(intr), K


def test_pose_predictor_teacher_front_end_vs_oracle(state_dict, tmp_path):
    """PosePredictor (pose_predictor.py:69-186) with the detector / segmenter outputs given: `filter_very_large_bb`
    (:83) drops the oversized box BEFORE squarify, depth is in 1e-4 m units (`/10000`, :118), `None` contracts
    (:76-78, :114-115, :131-132)."""
    from sunflower.predictor.pose_predictor import PosePredictor
    rgb, mask, depth, boxes = _scene(31)
    depth10k = (depth.astype(np.int64) * 10).astype(np.uint16)                # same scene, RealSense-D405 units
    big = np.array([[40, 30, 600, 440]])                                       # area > 5 x median: dropped before squarify
    boxes_all = np.concatenate([boxes[:3], big, boxes[3:]]).astype(np.int64)   # GroundingDINO.detect returns python ints -> int64
    ckpt, intr, K = _write_ckpt(tmp_path, state_dict, 480, 640)
    seen = {}
    def segmenter(img, bb):
        seen["bb"] = bb
        return mask
    pred = PosePredictor("cuda", ckpt, intr, detector=lambda img: boxes_all, segmenter=segmenter)
    Rt = pred.get_flower_poses(rgb, depth10k)
    kept = P.filter_very_large_bb(boxes_all)
    assert len(kept) == len(boxes_all) - 1 and seen["bb"] == kept.tolist()    # SAM sees the filtered boxes (:87-88)
    ref = P.get_flower_poses(lambda b: O.forward(state_dict, b), O.procrustes_to_rotmat, rgb, depth10k, kept, mask, K,
                             depth_div=10000.0)
    assert Rt.dtype == np.float64 and Rt.shape == ref.shape and Rt.shape[0] >= 3
    assert np.abs(Rt[:, :3, :3] - ref[:, :3, :3]).max() <= 1e-3
    assert np.linalg.norm(Rt[:, :3, 3] - ref[:, :3, 3], axis=1).max() <= 1e-5
    # the same frame read with the wrong scale (/1000 semantics) puts every flower beyond the far plane -> None (:131-132)
    far = PosePredictor("cuda", ckpt, intr, detector=lambda img: boxes_all, segmenter=lambda i, b: mask)
    assert far.get_flower_poses(rgb, (depth.astype(np.int64) * 100).clip(0, 65535).astype(np.uint16)) is None
    # no detection at all: the reference's detector returns an array of shape (0,) (:76-78)
    none = PosePredictor("cuda", ckpt, intr, detector=lambda img: np.array([]), segmenter=lambda i, b: mask)
    assert none.get_flower_poses(rgb, depth10k) is None
    # every box leaves the frame once squarified (:114-115)
    edge = PosePredictor("cuda", ckpt, intr, detector=lambda img: boxes[-2:-1], segmenter=lambda i, b: mask)
    assert edge.get_flower_poses(rgb, depth10k) is None


def test_live_pose_loop_over_frames(state_dict, tmp_path):
    """scripts/live_pose.py:31-41: frames -> get_flower_poses -> [N,4,4] | None, frame by frame."""
    from flope_amd.harness import live_pose_loop
    from sunflower.predictor.fast_pose_predictor import FastPosePredictor
    scenes = [_scene(41), _scene(42), _scene(43)]
    ckpt, intr, K = _write_ckpt(tmp_path, state_dict, 480, 640)
    it = iter(scenes)
    cur = {}
    def detector(img):
        return cur["boxes"], cur["mask"]
    pred = FastPosePredictor("cuda", detector, ckpt, intr)
    def frames():
        for i, (rgb, mask, depth, boxes) in enumerate(scenes):
            cur["boxes"], cur["mask"] = (boxes[-1:], mask) if i == 1 else (boxes, mask)   # frame 1: nothing usable
            yield rgb, depth
    out = live_pose_loop(pred, frames())
    assert len(out) == 3 and out[1] is None
    for i in (0, 2):
        rgb, mask, depth, boxes = scenes[i]
        ref = P.get_flower_poses(lambda b: O.forward(state_dict, b), O.procrustes_to_rotmat, rgb, depth, boxes, mask, K)
        assert out[i].shape == ref.shape and np.abs(out[i] - ref)[:, :3, :3].max() <= 1e-3
        assert np.linalg.norm(out[i][:, :3, 3] - ref[:, :3, 3], axis=1).max() <= 1e-5


def test_end_to_end_1080p_31_boxes_vs_oracle(state_dict, tmp_path):
    """BASELINE configs[2] at its own size: one 1080 x 1920 frame, 31 detector boxes, 512 x 512 crops (the
    reference's crop size), detections given (the detector itself is tested in test_gpu_yolo.py)."""
    from sunflower.predictor.fast_pose_predictor import FastPosePredictor
    rgb, mask, depth, boxes = _scene(51, H=1080, W=1920, n=29)
    assert len(boxes) == 31
    ckpt, intr, K = _write_ckpt(tmp_path, state_dict, 1080, 1920, fx=1400.0)
    pred = FastPosePredictor("cuda", lambda img: (boxes, mask), ckpt, intr)
    Rt = pred.get_flower_poses(rgb, depth)
    ref = P.get_flower_poses(lambda b: O.forward(state_dict, b), O.procrustes_to_rotmat, rgb, depth, boxes, mask, K)
    assert Rt.shape == ref.shape and Rt.shape[0] >= 20
    assert np.abs(Rt[:, :3, :3] - ref[:, :3, :3]).max() <= 1e-3                 # rot err
    assert np.linalg.norm(Rt[:, :3, 3] - ref[:, :3, 3], axis=1).max() <= 1e-5   # trans err (m)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_rotation_error_vs_conditioning_of_M(state_dict, dtype):
    """Where the 1e-3 claim holds.  Procrustes amplifies an error dM of the 3x3 head output by ~ |dM| / (s2 + s3)
    (s_i = singular values of M).  The synthetic head is well conditioned by construction (fc_rot.bias = vec(R0),
    s ~ 1.9 / 0.8 / 0.4).  The sweep keeps the trunk (and therefore its 16-bit error dM) and re-biases the head so that
    M_i = a * R0 + (W h_i - mean_j W h_j): singular values ~ a, i.e. (s2 + s3) ~ 2a, down to a = 0.01.
    Asserted: |dR| * (s2 + s3) <= 3 |dM| at every a (the amplification law: measured 1.05-1.2 |dM| down to
    (s2 + s3) = 0.03 and 2.1 - 2.5 |dM| at 0.005, where the first-order law ends -- 2.1 with layer2.0.conv1 on the gathered-tile
    kernel, 2.5 on conv_s2r, whose K order rounds differently), so the claim's domain can be stated:
    f16 meets 1e-3 wherever (s2 + s3) >= 0.5, bf16 only wherever (s2 + s3) >= 3 |dM| / 1e-3 -- more than a rotation-like
    M (s ~ 1, 1, 1) ever has.
    The measured table is printed (pytest -s) and quoted in DESIGN.md."""
    torch.manual_seed(0)
    x = torch.rand(16, 3, 224, 224)
    rows = []
    e = _engine(state_dict, 224, 224, 16, dtype)
    sd0 = dict(state_dict)
    sd0["fc_rot.bias"] = torch.zeros(9)
    mean_wh = O.forward(sd0, x).mean(0)
    for a in (1.0, 0.3, 0.1, 0.03, 0.01):
        sd = dict(state_dict)
        sd["fc_rot.bias"] = state_dict["fc_rot.bias"] * a - mean_wh
        e.load_state_dict(sd)
        r9, R = _run(e, x)
        ref9 = O.forward(sd, x)
        Rref = O.procrustes_to_rotmat(ref9)
        sv = torch.linalg.svdvals(ref9.double().view(-1, 3, 3))
        gap = (sv[:, 1] + sv[:, 2])
        dR = (R - Rref).abs().amax(dim=(1, 2)).double()
        dM = (r9 - ref9).abs().amax(dim=1).double()
        rows.append((a, float(gap.min()), float(gap.median()), float(dM.max()), float(dR.max()), float((dR * gap).max())))
    e.close()
    print(f"\n{dtype}: bias scale a | min(s2+s3) | median | max|dM| | max|dR| | max |dR|*(s2+s3)")
    for r in rows:
        print("   %.2f | %.4f | %.4f | %.2e | %.2e | %.2e" % r)
    tol_M = 1.0e-3 if dtype == "f16" else 8e-3
    assert rows[0][1] > 1.5 and rows[-1][1] < 0.05               # the sweep really spans two decades of conditioning
    for a, gmin, gmed, dM, dR, k in rows:
        assert dM <= tol_M, (a, dM)
        assert k <= 3.0 * dM + 1e-6, (a, k, dM)
        if gmin >= (0.5 if dtype == "f16" else 3 * dM / 1e-3):
            assert dR <= 1e-3, (a, gmin, dR)


def test_float16_activations_saturate_instead_of_overflowing(state_dict):
    """a checkpoint whose residual stream outgrows the float16 range must not turn into inf -> NaN -> a garbage
    rotation.  Scaling one BatchNorm's gamma by 1.5e5 pushes layer1.0 beyond 65504 (the folded weights stay below it): the 16-bit epilogues saturate at
    +-65504 (v_pk_min_f16), every later stage stays finite, and the rotation is still a rotation."""
    sd = dict(state_dict)
    sd["base.layer1.0.bn2.weight"] = state_dict["base.layer1.0.bn2.weight"] * 1.5e5
    fold = sd["base.layer1.0.bn2.weight"] / torch.sqrt(sd["base.layer1.0.bn2.running_var"] + 1e-5)
    assert float((sd["base.layer1.0.conv2.weight"] * fold.view(-1, 1, 1, 1)).abs().max()) < 6e4       # weights representable
    torch.manual_seed(3)
    x = torch.rand(3, 3, 224, 224)
    e = _engine(sd, 224, 224, 3, "f16")
    r9, R = _run(e, x)
    l10 = e.read_stage("layer1.0", 3).cpu()
    assert torch.isfinite(l10).all() and float(l10.max()) == 65504.0
    assert torch.isfinite(r9).all() and torch.isfinite(R).all()
    assert (R @ R.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-4 and (torch.linalg.det(R) - 1).abs().max() < 1e-4
    e.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_engine_on_the_second_device_while_the_first_is_current(state_dict):
    """the forward selects the handle's device itself (hipSetDevice in run_trunk / read_stage)."""
    torch.manual_seed(4)
    x = torch.rand(2, 3, 96, 96)
    from flope_amd.engine import PoseEngine
    e = PoseEngine(96, 96, 2, "f16", device="cuda:1")
    e.load_state_dict(state_dict)
    torch.cuda.set_device(0)
    r9, R = e.forward(x.to("cuda:1"))
    ref = O.procrustes_to_rotmat(O.forward(state_dict, x))
    assert R.device.index == 1 and (R.cpu() - ref).abs().max() <= 1e-3
    e.close()


def test_multi_frame_fusion_on_the_device_path_vs_the_oracle_pipeline(state_dict, tmp_path):
    """SURVEY N3 on the device path: a synthetic multi-frame stream with camera poses goes through
    `FastPosePredictor.iter_flower_poses` (pipelined, built-in detector in strict float32 so that its integer boxes equal the
    oracle detector's) into `FlowerModel` (flower_model.py:146-255: cam -> world, 50 mm association, 7-state Kalman tracks) and
    is compared with `oracle/fusion_ref.py` fed by the ALL-oracle pipeline (oracle detector, crops, PoseResNet, Procrustes,
    depth lift; scipy quaternions): same tracks, same hit counts, filtered states within the pose tolerances."""
    import yaml
    from scipy.spatial.transform import Rotation as Rot
    from flope_amd.yolo_weights import synthetic_frame, synthetic_yolo_state_dict
    from oracle import fusion_ref as F
    from oracle import pipeline_ref as P
    from oracle import yolo_ref as Y
    from sunflower.predictor.fast_pose_predictor import FastPosePredictor
    from sunflower.predictor.flower_model import FlowerModel
    ysd = synthetic_yolo_state_dict(0)
    H, W, imgsz = 1080, 1920, 1280                     # BASELINE configs[2] frame size: a dozen flowers per frame
    yolo_f, ckpt, intr = tmp_path / "yolo.pth", tmp_path / "posenet.pth", tmp_path / "intrinsics.yaml"
    torch.save({**ysd, "imgsz": torch.tensor(imgsz)}, yolo_f)
    torch.save(state_dict, ckpt)
    intr.write_text(yaml.safe_dump(dict(fx=1400.0, fy=1400.0, cx=W / 2, cy=H / 2, h=H, w=W)))
    K = np.array([[1400.0, 0, W / 2], [0, 1400.0, H / 2], [0, 0, 1]])
    pred = FastPosePredictor("cuda", str(yolo_f), str(ckpt), str(intr), yolo_dtype="f32")
    rng = np.random.default_rng(11)
    base = synthetic_frame(7, H, W)
    frames = []
    for i in range(4):                    # the same scene seen again (tracks get a second hit), one empty frame, one different scene
        img = base.copy() if i != 3 else synthetic_frame(6, H, W)     # (r05: four frames instead of five -- the CPU oracle is 3 s per frame)
        if i == 2:
            img = np.zeros_like(img)
        img[:4, :4] = rng.integers(0, 255, (4, 4, 3), dtype=np.uint8)      # frames are not byte-identical
        depth = (400 + rng.normal(0, 1.5, (H, W))).astype(np.uint16)
        cam = np.concatenate([rng.normal(0, 0.002, 3), Rot.from_rotvec(rng.normal(0, 0.002, 3)).as_quat()])
        frames.append((img, depth, cam))
    fm = FlowerModel(dist_th=50, pose_predictor=pred)
    outs = list(fm.add_stream(frames, ignore=True, detectors=2))
    assert len(outs) == 4 and outs[2] == (None, None)
    # ---- all-oracle side
    meas = []
    for (img, depth, cam), (pc, pw) in zip(frames, outs):
        rb, rmask = Y.get_bbox_mask(ysd, img, imgsz)
        ref = P.get_flower_poses(lambda b: O.forward(state_dict, b), O.procrustes_to_rotmat, img, depth, rb, rmask, K) if rb.shape[0] else None
        assert (ref is None) == (pc is None)
        if ref is None:
            continue
        assert pc.shape == ref.shape
        assert np.abs(pc[:, :3, :3] - ref[:, :3, :3]).max() <= 1e-3 and np.linalg.norm(pc[:, :3, 3] - ref[:, :3, 3], axis=1).max() <= 1e-5
        M = np.eye(4); M[:3, :3] = Rot.from_quat(cam[3:]).as_matrix(); M[:3, 3] = cam[:3]
        world = np.einsum("ij,njk->nik", M, ref)                 # mvg.py:416-421
        assert np.abs(pw - world).max() <= 2e-3
        meas.append([list(np.concatenate([w[:3, 3], Rot.from_matrix(w[:3, :3]).as_quat()])) for w in world])
    tracks = F.associate(meas, th=0.05)
    assert len(fm.kfs) == len(tracks) and list(fm.scores) == [len(t) for t in tracks]
    assert len(meas[0]) >= 5 and max(len(t) for t in tracks) >= 2 and len(tracks) > len(meas[0])   # re-observed tracks and newcomers both occur
    for kf, t in zip(fm.kfs, tracks):
        x, p = F.scalar_track(t)
        assert np.abs(kf.x[:3] - np.array(x[:3])).max() <= 1e-5
        assert np.abs(kf.x[3:] - np.array(x[3:])).max() <= 1e-3      # quaternion of a rotation known to 1e-3
        assert np.allclose(np.diag(kf.P), p, atol=1e-12)
    # frame by frame through add_data (sequential get_flower_poses): the identical tracker
    fm2 = FlowerModel(dist_th=50, pose_predictor=pred)
    for img, depth, cam in frames:
        fm2.add_data(img, depth, cam, ignore=True)
    assert len(fm2.kfs) == len(fm.kfs) and np.array_equal(fm2.filtered_state(), fm.filtered_state())
