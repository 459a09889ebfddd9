// conv_stag: 3x3 / stride-1 implicit-GEMM convolution for Cout >= 128 (layer2..4 of the
// ResNet-18 trunk, reference sunflower/models/posenet.py:25), second-generation structure.
//
// Why a second kernel (r01 measurements on conv_mfma, B = 256): with 128-pixel tiles the
// weight stream alone (no MFMA issued) already takes 43 us per layer -- every workgroup re-streams
// the whole [Cout x 9*Cin] weight panel through L2 -> LDS at the ~55-70 GB/s per-CU DMA rate --
// against a 47 us pure-MFMA floor; the two do not overlap well from two independent 4-wave
// workgroups.  Here:
//   * one 8-wave workgroup per CU owns 256 pixels x 128 channels: weight traffic per MAC halves;
//   * a step is one tap x 32 input channels: an 8 KB weight tile (one 1 KiB LDS-DMA per wave) in a
//     6-deep ring, retired by counted s_waitcnt vmcnt -- tiles are in flight ~10 phases (> 1 us);
//   * the input patch is double buffered per 32-channel half-chunk: the next half-chunk's patch is
//     DMA'd while the current one is being consumed, so patch latency is never exposed;
//   * the waves form two groups of four (pixels 0..127 / 128..255) that run HALF A STEP APART:
//     in every phase one group issues its 16 MFMAs while the other issues the ds_read_b128s of its
//     next step, so on each SIMD one wave computes while its partner loads -- deterministic
//     MFMA / LDS overlap instead of hoping two independent workgroups drift out of phase.
// Same operand orientation, register epilogue, pixel permutation and zero-bordered NHWC layout as
// conv_mfma.hip.  LDS rows are 64 B here: weight rows use the 4-slot map g ^ h[(r>>2)&3],
// h = {0,2,3,1}; patch pixels use g ^ ((p>>2)&3).
#include "common.h"

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ int tile_px_s(int c) { return c < 4 ? 2 * c : (c < 12 ? 2 * (c - 4) + 1 : 2 * (c - 8)); }

// PT = LDS-DMA rounds (of 512 x 16 B) per patch burst: a compile-time constant so that every s_waitcnt in
// the main loop is an immediate and the 18-step body (two half-chunks x nine taps) has no runtime control
// flow -- r01's first version of this kernel spent ~0.5 us per phase in scalar bookkeeping (ring slot, tap
// decode, wait-count switch), more than the 16 MFMAs it was wrapped around.
// BN = 128: 256-pixel tiles, each 4-wave group is 2 (px) x 2 (ch) waves.  BN = 64 (layer 1): 512-pixel tiles,
// each group is 4 (px) x 1 (ch) waves and the 4 KB weight tile is moved by waves 0..3 only.
// RES: the layer has a residual input (fixes the number of VM ops of the epilogue at compile time).
//
// PERSISTENT: the grid is at most one workgroup per CU and workgroup b walks tiles b, b+G, b+2G, ...
// (G % ntiles == 0, so its channel tile -- weight panel, bias -- never changes).  The step stream simply
// continues across a tile boundary: the weight ring wraps to the panel's first tiles and the last half-chunk's
// patch burst fetches the NEXT tile's first patch, so no DMA latency is ever re-exposed; only the register
// epilogue + the next tile's address table sit between two tiles (one group at a time, the other keeps going).
// ROWS (layer 1, Wo <= 64, Ho % 8 == 0): a tile is 8 full output rows of ONE image, wave (group, wpx) owns row
// group*4 + wpx and its four pixel tiles are columns 0..63 of that row (columns >= Wo are computed and dropped).
// Every tile then has the same geometry relative to its patch origin: the fragment address table is built once
// per workgroup instead of once per tile (it was ~300 of the ~1000 vector instructions a tile cost outside its
// MFMAs -- with K = 576 those instructions were as expensive as the 288 MFMAs), 7 tiles per image divide the
// 256 x 7 tiles of B = 256 evenly over 256 CUs, and m0 carries the tile's first padded OUTPUT row.
// DSF (BN = 128, no residual input, one tile per workgroup, PT >= 4): the block's 1x1 stride-2 shortcut convolution
// is folded into this kernel as extra K -- before the main loop, for every 64 input channels of the block input x,
// the tile's 256 centre pixels x(2ho, 2wo) are gathered by LDS-DMA into patch buffer 1 (two 32-channel pixel tiles),
// the matching double tile of shortcut weights goes into ring slot 2, and one double step of 32 MFMAs adds
// W_ds . x to the same accumulators (which start at bias2 + bias_ds).  Saves the separate downsample launch, its
// output write and this kernel's residual read; the shortcut sum is never rounded to 16 bits on the way.
template <typename T, int PT, int BN, bool RES, bool ROWS, bool DSF = false>
__global__ __launch_bounds__(512, 2) void conv_stag_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int BM = BN == 128 ? 256 : 512, GP = BM / 2, TILE_B = BN * 64;   // 8 / 4 KB weight tile per (half-chunk, tap)
  constexpr int NBD = 3, DT_B = 2 * TILE_B;                 // ring of 3 double tiles (one double step = two taps)
  constexpr int TG = DT_B / 8192;                           // LDS-DMA ops per wave per double tile (2 / 1)
  constexpr int MT = 4, NT = 4;
  constexpr int PATCH_B = PT * 8192;                              // bytes of one patch buffer
  // WRES (ROWS with an odd PT): the whole 18 x 4 KB weight panel of a 64 -> 64 layer stays in LDS for the life of the
  // workgroup instead of being re-streamed per tile -- the chip-wide L2 -> LDS DMA rate (~6.5-8.7 TB/s measured) is the
  // floor under these kernels, and at K = 576 the weight ring was half of this layer's DMA bytes.
  constexpr bool WRES = ROWS && (PT & 1);
  constexpr int EPI_OPS = MT * 2 + ((RES && !ROWS) ? MT * 2 : 0);  // 16-byte stores (+ residual loads, unless prefetched: ROWS) per lane per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ps = smem;                                   // 2 patch buffers (first: their offsets stay ds_read immediates)
  char* const Bs = smem + 2 * PATCH_B;                     // NBD double-tile weight ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, wl = wave & 3;
  const int wpx = BN == 128 ? (wl & 1) : wl, wch = BN == 128 ? (wl >> 1) : 0;
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px_s(r16);
  const int KSP = (!ROWS && p.ksplit > 1) ? p.ksplit : 1;   // split-K: KSP workgroups per tile, each a share of the K loop
  const int ks = KSP > 1 ? (int)(blockIdx.x % KSP) : 0;
  const int G = gridDim.x / KSP;
  const int lb = xcd_remap(blockIdx.x / KSP, G);
  const int ntile = lb % p.ntiles;                         // constant over this workgroup's tiles
  const int HoWo = p.Ho * p.Wo;
  const int nhc = p.Cin / 32;                              // even: Cin is a multiple of 64
  const int NS = nhc * 9;                                  // (half-chunk, tap) steps per tile; NS/2 double steps
  const size_t pixB = (size_t)p.Cin * 2;
  const size_t rowB = (size_t)p.Wip * pixB;

  // Everything a step needs is precomputed so the loop body is LDS reads, MFMAs, one DMA issue and waits:
  //   psrc[rr]    per-lane 32-bit source offsets of the PT patch rounds (tile independent: the burst always
  //               moves PT*512 pieces; what lies past the tile's rows is never read back -- the activation
  //               buffers carry 1 MB of slack so the over-read stays inside the allocation)
  //   xoff[t][pt] LDS byte offset of this lane's pixel fragment for tap t (buffer / ring slot are immediates)
  // round rr moves pieces rr*512 + wave*64 + lane: pixel rr*128 + wave*16 + (lane>>2), whose swizzle term
  // ((pixel >> 2) & 3) does not depend on rr or wave -> one per-lane offset + a uniform rr stride
  // ROWS on maps wider than 64 columns (p.nseg > 1: layer 1 of 512 x 512 crops is 128 wide): a tile is 8 rows x one
  // 64-column SEGMENT; its patch is 10 rows x 66 columns, stored in LDS at a pitch of kSegPitch pixels, and patch pixel
  // (pr, pc) comes from input pixel (R0 + pr, 64 s0 + pc) -- the per-lane source offsets below carry that mapping, the
  // LDS side does not change.
  // Flat tiles (!ROWS, r03): the LDS image of the patch is NOT a copy of the padded rows.  A 16-pixel MFMA tile of a 28 / 14 /
  // 7-wide map wraps one to three image rows; with the rows at their natural pitch (W + 2) and the slot swizzle taken from the
  // LDS pixel index, the pixels behind a wrap land on bank quads the pixels before it already use (r02: SQ_LDS_BANK_CONFLICT =
  // 29 % of SQ_LDS_IDX_ACTIVE on this kernel, 0 on the row-band and gather kernels).  A ds_read_b128 lane group holds 8 even
  // pixels at k-slot g and 8 odd ones at g ^ 1; it is conflict-free iff the 16 (pixel mod 4, slot) pairs differ.  With
  //     LDS pixel index  P(i, c) = i * (W + 4) + c          (patch row i, column c: the pitch is = W mod 4)
  //     slot swizzle     s(i, c) = ((i * W + c) >> 2) & 3   (v = i * W + c advances by ONE per output pixel across a row wrap)
  // tap (dy, dx) of output pixel k of a tile reads P = P0 + k (mod 4) and s = ((v0 + k) >> 2) & 3 for every k that stays inside one
  // image: sixteen consecutive v give sixteen different (v mod 4, (v >> 2) mod 4) pairs, whatever the row wraps in between.  (A tile
  // that also crosses an IMAGE boundary -- 2 / 8 / 33 % of the pixel tiles at 28 / 14 / 7-wide maps -- skips two border rows and keeps
  // two-way conflicts behind the crossing.)  Both sides are per-lane tables already: DMA source offsets here, xoff[] below.
  constexpr int kSegPitch = 66;
  const bool seg = ROWS && p.nseg > 1;
  const bool skew = !ROWS && p.skew;                       // p.skew = 0: the r02 image (natural pitch, swizzle from the LDS pixel index), A/B only
  const int pitch = seg ? kSegPitch : (skew ? p.Wip + 2 : p.Wip);
  unsigned psrc[PT];
#pragma unroll
  for (int rr = 0; rr < PT; ++rr) {
    const int q = rr * 512 + wave * 64 + lane;
    const int pi = q >> 2;
    int js = (q & 3) ^ ((pi >> 2) & 3);
    int spix = pi;                                         // source pixel, relative to the patch origin
    if (seg) {
      const int pc_ = min(pi, 10 * kSegPitch - 1), pr_ = pc_ / kSegPitch;
      spix = pr_ * p.Wip + (pc_ - pr_ * kSegPitch);
    }
    if (skew) {
      const int pr_ = pi / pitch, pc_ = pi - pr_ * pitch;
      spix = pr_ * p.Wip + min(pc_, p.Wip - 1);            // the two pad pixels of a row re-read its last pixel (never read back)
      js = (q & 3) ^ (((pr_ * p.Wo + pc_) >> 2) & 3);
    }
    psrc[rr] = (unsigned)(spix * (int)pixB + js * 16);
  }
  const int nbody = nhc / 2 / KSP, body0 = ks * nbody;      // bodies (= pairs of half-chunks = 64 input channels) of this workgroup
  const char* const b_base = (const char*)p.w + (size_t)ntile * NS * TILE_B + (size_t)body0 * 9 * (2 * TILE_B) + wave * 1024;   // + g*8192 for op g of a double tile
  const unsigned lane16 = lane * 16;
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wbase = 2 * PATCH_B + (wch * 64 + r16) * 64 + ((g ^ wsw) << 4);
  // this lane's channels: two runs of 8 (host_pack.h stag_row_to_channel): cb .. cb + 7 (accumulator tiles 0, 1) and
  // cb + 32 .. cb + 39 (tiles 2, 3) -- byte offsets 16 g and 64 + 16 g inside the wave's 128-byte channel block
  const int cb = ntile * BN + wch * 64 + g * 8;
  // folded-BN bias of accumulator tile ct: channels cb + (ct >> 1) * 32 + (ct & 1) * 4 .. + 3.  Read again at every tile boundary
  // instead of living in 16 VGPRs for the life of the workgroup: the main loop runs at the 256-register limit (r03)
  const float* const bias_p = p.bias + cb;
#define BIAS4(ct_) (*(const f32x4*)(bias_p + ((ct_) >> 1) * 32 + ((ct_) & 1) * 4))

  // ---- tile geometry: wave-uniform part (first pixel, patch origin) and per-lane part (xoff, output offsets)
  int m0, mend, R0, mc0 = 0;                               // current tile (mc0: first output column, ROWS segments)
  const char* patch_src;
  int n_m0 = 0, n_mend = 0, n_R0 = 0, n_mc0 = 0;           // next tile of this workgroup
  const char* n_patch_src = nullptr;
  bool has_next;
#define TILE_GEOM(tile_, m0_, mend_, R0_, mc_, src_)                                                           \
  do {                                                                                                         \
    mc_ = 0;                                                                                                   \
    if constexpr (ROWS) {                                                                                      \
      const int b0_ = (tile_) / p.tiles_per_image, jj_ = (tile_) - b0_ * p.tiles_per_image;                    \
      const int ns_ = max(p.nseg, 1), j0_ = jj_ / ns_;                                                         \
      mc_ = (jj_ - j0_ * ns_) * 64;                                                                            \
      m0_ = b0_ * p.Hop + j0_ * 8;                                                                             \
      mend_ = m0_ + BM;                                                                                        \
      R0_ = b0_ * p.Hip + j0_ * 8;                                                                             \
    } else {                                                                                                   \
      m0_ = ((tile_) / p.ntiles) * BM;                                                                         \
      mend_ = min(m0_ + BM, p.M);                                                                              \
      const int b0_ = fastdiv(m0_, p.mg_hw, p.sh_hw), ho0_ = fastdiv(m0_ - b0_ * HoWo, p.mg_w, p.sh_w);        \
      R0_ = b0_ * p.Hip + ho0_;                                                                                \
    }                                                                                                          \
    src_ = (const char*)p.in + (size_t)R0_ * rowB + (size_t)(mc_) * pixB;                                      \
  } while (0)
  // ROWS: the four pixel tiles of a wave are 16 pixels apart in one row -> same swizzle term, offsets differ by
  // 16 * 64 B, which folds into the ds_read immediate: 9 address registers instead of 36
  int xoff[9][ROWS ? 1 : MT];
#define XO(t_, pt_) (ROWS ? xoff[t_][0] + (pt_) * 1024 : xoff[t_][ROWS ? 0 : (pt_)])
#define LANE_SETUP()                                                                                           \
  do {                                                                                                         \
    _Pragma("unroll") for (int pt = 0; pt < (ROWS ? 1 : MT); ++pt) {                                           \
      int pi0_, v0_ = 0;                                                                                       \
      if constexpr (ROWS) {                                                                                    \
        pi0_ = (group * 4 + wpx) * pitch + pcol;                                                               \
      } else {                                                                                                 \
        const int mm_ = m0 + group * GP + wpx * 64 + pt * 16 + pcol;                                           \
        const int m_ = min(mm_, mend - 1);                                                                     \
        const int b_ = fastdiv(m_, p.mg_hw, p.sh_hw), r_ = m_ - b_ * HoWo;                                     \
        const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;                                    \
        const int i_ = b_ * p.Hip + ho_ - R0;                                                                  \
        pi0_ = i_ * pitch + wo_;                                                                               \
        v0_ = i_ * p.Wo + wo_;                                                                                 \
      }                                                                                                        \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                          \
        const int pi = pi0_ + (t / 3) * pitch + (t % 3);                                                       \
        const int sv = skew ? v0_ + (t / 3) * p.Wo + (t % 3) : pi;                                             \
        xoff[t][pt] = (pi << 6) + ((g ^ ((sv >> 2) & 3)) << 4);                                                \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)

#define ISSUE_PATCH(src_, buf_)                                                                                \
  do {                                                                                                         \
    _Pragma("unroll") for (int rr = 0; rr < PT; ++rr)                                                          \
      GLDS16((src_) + psrc[rr], Ps + (buf_) * PATCH_B + (rr * 512 + wave * 64) * 16);                          \
  } while (0)
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define BARRIER()                                                                                              \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  frag wf[2][NT], xf[2][MT];
  f32x4 acc[MT][NT];                                       // start at the folded-BN bias: no bias add in the epilogue
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = BIAS4(ct);
  if (KSP > 1) {                                           // partial sums: the finalize kernel adds the bias once
#pragma unroll
    for (int pt = 0; pt < MT; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // fragment reads of one double step: ring slot, patch buffers and taps are literals
#define LOADF(slot_, buf0_, tap0_, buf1_, tap1_)   /* slot_: byte offset of the double tile's first row for this lane */                                                               \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct) {                                                        \
      wf[0][ct] = *(const frag*)(smem + (slot_) + ct * 1024);                                                  \
      wf[1][ct] = *(const frag*)(smem + (slot_) + TILE_B + ct * 1024);                                         \
    }                                                                                                          \
    _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) {                                                        \
      xf[0][pt] = *(const frag*)(smem + XO(tap0_, pt) + (buf0_) * PATCH_B);                                    \
      xf[1][pt] = *(const frag*)(smem + XO(tap1_, pt) + (buf1_) * PATCH_B);                                    \
    }                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
  } while (0)
#define MFMAS()                                                                                                \
  do {                                                                                                         \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                              \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt)                                                        \
        _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                      \
          acc[pt][ct] = Elem<T>::mfma(wf[h][ct], xf[h][pt], acc[pt][ct]);                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define ISSUE_DT(dt_, slot_)                                                                                   \
  do {                                                                                                         \
    _Pragma("unroll") for (int o = 0; o < TG; ++o)                                                             \
      GLDS16(b_base + (size_t)(dt_) * DT_B + o * 8192 + lane16, Bs + (slot_) * DT_B + o * 8192 + wave * 1024); \
  } while (0)

  // byte offset of this lane's 16 channels of pixel tile pt in the padded NHWC output (clamped to a legal pixel)
  auto rows_off = [&](int rowbase, int colbase, int pt) -> size_t {   // ROWS: the tile's first padded output row / column
    const int col = min(colbase + pt * 16 + pcol, p.Wo - 1);
    return ((((size_t)(rowbase + group * 4 + wpx + 1)) * p.Wop + col + 1) * p.Cout + cb) * 2;
  };
  auto out_off = [&](int pt, bool& valid) -> size_t {
    if constexpr (ROWS) {
      valid = mc0 + pt * 16 + pcol < p.Wo;
      return rows_off(m0, mc0, pt);
    } else {
      const int mm = m0 + group * GP + wpx * 64 + pt * 16 + pcol;
      valid = mm < mend;
      const int mc = min(mm, mend - 1);
      const int b_ = fastdiv(mc, p.mg_hw, p.sh_hw), r_ = mc - b_ * HoWo;
      const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;
      return ((((size_t)b_ * p.Hop + ho_ + 1) * p.Wop + wo_ + 1) * p.Cout + cb) * 2;
    }
  };

  // ---- first tile of this workgroup + the one after it
  int tile = lb;
  TILE_GEOM(tile, m0, mend, R0, mc0, patch_src);
  has_next = tile + G < p.total_tiles;
  if (has_next) TILE_GEOM(tile + G, n_m0, n_mend, n_R0, n_mc0, n_patch_src);
  LANE_SETUP();

  // ---- residual of the FIRST tile: loaded before anything else (oldest VM ops, so the wait below covers them) and
  // folded into the accumulators while the first patch / weight tiles are still in flight -- with one workgroup
  // per CU nothing else would hide that latency at the kernel's tail.  Later tiles of a persistent grid load
  // theirs in the epilogue.
  // r03, RES on flat 256 x 128 tiles with one tile per workgroup (p.res_lds): the residual does not go through registers at all.
  // Loaded up front it sat in front of the first patch / weight tiles -- every workgroup of a round asking for its 64 KB at the same
  // moment, 4-6 us per round on the 28 x 28 maps (layer2.1.conv2 116 us against 92 us for the same conv without residual) -- and
  // there are no 32 spare VGPRs to fetch it late.  But the last body of a tile issues 64 KB of LDS-DMA that nobody reads: the patch
  // burst of double step 5 (the NEXT tile's first patch) and the two double tiles that wrap to the start of the weight panel
  // (double steps 7 and 8).  Those exact slots (patch buffer 0: rounds 0..3; ring slots 0 and 1: two rounds each) now receive the
  // tile's residual, lane (wave, l) fetching with round r = 2 pt + c the 16 bytes it will itself add in the epilogue -- no other
  // wave reads them, so the wave's own vmcnt(0) orders the read-back, and every counted wait of the main loop stays as it was.
  const bool res_lds = RES && !ROWS && BN == 128 && PT >= 4 && p.res_lds;
  bool res_pre = RES && !res_lds;
  unsigned roff[(RES && !ROWS) ? MT : 1];                  // res_lds: 32-bit byte offsets of this lane's residual / output pixels
  if constexpr (RES && !ROWS) if (res_lds) {               // (computed once: four registers, not four fastdiv chains inside the loop)
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) { bool v_; roff[pt] = (unsigned)out_off(pt, v_); }
  }
  u32x4 rpre[RES ? MT : 1][2];
  if constexpr (RES) if (!res_lds) {
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      bool v_;
      const char* rp = (const char*)p.res + out_off(pt, v_);
      rpre[pt][0] = *(const u32x4*)rp;
      rpre[pt][1] = *(const u32x4*)(rp + 64);
    }
  }
  // ---- prologue: patch of this workgroup's first half-chunk, its double tiles 0 and 1
  ISSUE_PATCH(patch_src + (size_t)(2 * body0) * 64, 0);
  if constexpr (WRES) {
    const char* wsrc = (const char*)p.w + (size_t)ntile * NS * TILE_B + lane16;
    for (int i = wave; i < 18 * TILE_B / 1024; i += 8) GLDS16(wsrc + i * 1024, Bs + i * 1024);
    WAIT_VM(0);                                // patch 0 and the resident weight panel landed (once per workgroup)
  } else {
    ISSUE_DT(0, 0);
    ISSUE_DT(1, 1);
    WAIT_VM(TG);                               // patch 0 and double tile 0 landed (double tile 1 may fly)
  }
  if constexpr (RES) if (!res_lds) {
#pragma unroll
    for (int pt = 0; pt < MT; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = ct * 4 + q;
          const unsigned w_ = rpre[pt][i >> 3][(i & 7) >> 1];
          acc[pt][ct][q] += (i & 1) ? unpack_hi<T>(w_) : unpack_lo<T>(w_);
        }
  }
  BARRIER();
  if constexpr (DSF) if (ks == 0) {
    static_assert(BN == 128 && !RES && !ROWS && PT >= 4, "folded downsample: 256 x 128 tiles, 32 KB patch buffers");
    // this lane's two DMA source pixels (pieces rr*512 + wave*64 + lane: pixel slot = piece >> 2, 16-byte part = piece & 3)
    const char* dsrc[2];
    const size_t dpix = (size_t)p.ds_Cin * 2;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int q = rr * 512 + wave * 64 + lane, sl = q >> 2;
      const int m = min(m0 + sl, mend - 1);
      const int b_ = fastdiv(m, p.mg_hw, p.sh_hw), r_ = m - b_ * HoWo;
      const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;
      dsrc[rr] = (const char*)p.ds_in + (((size_t)b_ * p.ds_Hip + 2 * ho_ + 1) * p.ds_Wip + 2 * wo_ + 1) * dpix +
                 (((q & 3) ^ ((sl >> 2) & 3)) << 4);
    }
    int xds[MT];                                 // fragment offsets inside a gathered pixel tile (slot-linear, same swizzle)
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int sl = group * GP + wpx * 64 + pt * 16 + pcol;
      xds[pt] = PATCH_B + (sl << 6) + ((g ^ ((sl >> 2) & 3)) << 4);
    }
    const char* const dw_base = (const char*)p.ds_w + (size_t)ntile * (p.ds_Cin / 32) * TILE_B + wave * 1024 + lane16;
    for (int j = 0; j < p.ds_Cin / 64; ++j) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
          GLDS16(dsrc[rr] + (2 * j + t) * 64, Ps + PATCH_B + t * 16384 + (rr * 512 + wave * 64) * 16);
#pragma unroll
      for (int o = 0; o < TG; ++o) GLDS16(dw_base + (size_t)j * DT_B + o * 8192, Bs + 2 * DT_B + o * 8192 + wave * 1024);
      WAIT_VM(0);
      BARRIER();
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        wf[0][ct] = *(const frag*)(smem + wbase + 2 * DT_B + ct * 1024);
        wf[1][ct] = *(const frag*)(smem + wbase + 2 * DT_B + TILE_B + ct * 1024);
      }
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) {
        xf[0][pt] = *(const frag*)(smem + xds[pt]);
        xf[1][pt] = *(const frag*)(smem + xds[pt] + 16384);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MFMAS();
      BARRIER();                                 // buffer 1 / slot 2 are free again (next j, or the main loop's step 0)
    }
  }
  if (group == 1) BARRIER();                   // group B runs one phase behind group A
  // static priority for one half of the workgroup (MI355X_MICROARCH "two waves per SIMD", item 4): the later-dispatched half loses
  // the VALU arbitration on every segment; p.prio 1 raises waves 4..7, 2 raises waves 0..3, 0 leaves both at 0 (A/B option)
  if (p.prio == 1 + (1 - group)) __builtin_amdgcn_s_setprio(1);

  // Every wave executes the SAME stream per double step D (two (half-chunk, tap) pairs u = 2D, 2D+1 of the
  // 18 that make up two half-chunks), no group tests inside the loop:
  //     issue double tile D+2 into the ring slot double step D-1 just released (double tile t lives in slot
  //         t % 3; past the end of the panel it wraps to the panel's start = the next tile's first steps);
  //         at D = 0 the patch of this body's second half-chunk, at D = 5 the patch of the NEXT body's first
  //         half-chunk (or the next tile's) -- each into the buffer whose last reader was one double step ago
  //     LOADF(D): 16 ds_read_b128  ->  wait for this wave's pieces of double tile D+1  ->  barrier
  //     32 MFMAs  ->  barrier
  // and group B is one barrier behind group A, so in every physical phase one group is in its MFMA half
  // while the other is in its load half (A: L(D) at phase 2D-1, M(D) at 2D; B: L(D) at 2D, M(D) at 2D+1).
  // Hazards: slot of D-1 is last read by B's L(D-1) at phase 2D-2, refilled at phase >= 2D-1; double tile D is
  // waited for by every wave right after its own L(D-1), i.e. before the barrier that precedes A's L(D).
  // Patch buffer 0 is last read at D = 4 (tap 8 of the first half-chunk), buffer 1 at D = 8.
  // Younger VM ops than double tile D+1 at the wait: the TG ops of double tile D+2, plus the PT patch rounds
  // when a burst was issued at double step D-1 or D (D in {0,1,5,6}), plus -- at D = 0 right after a tile
  // boundary -- the EPI_OPS stores / residual loads of the epilogue that ran in between.
  // timing experiments only (build with -DFLOPE_STAG_DBG; results are wrong by construction):
  //   1 no weight DMA, 2 no patch refills, 4 no MFMA, 16 no LDS reads, 32 no stores
#ifdef FLOPE_STAG_DBG
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
#ifdef FLOPE_STAG_DBG
  // dbg & 128: four shader-clock stamps per double step (start / before barrier 1 / after barrier 1 / after the MFMAs were issued) of
  // the first kStampD double steps, wave 0 of each group, parked in LDS behind the kernel's own image (the host adds 2 KB of dynamic
  // LDS in this mode) and copied out after the loop: records of 4 x uint64 at p.split_ws + 64 KB + ((block * 2 + group) * kStampD + d) * 32 B
  constexpr int kStampD = 27;
  int st_n = 0;
  bool st_done = false;
  char* const st_lds = smem + p.dbg_lds_off + group * (kStampD * 32);
#define STAMP(k_)                                                                                              \
  do {                                                                                                         \
    if ((dbg & 128) && wl == 0 && st_n < kStampD) {                                                            \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                              \
      if (lane == 0) *(unsigned long long*)(st_lds + st_n * 32 + (k_) * 8) = t_;                               \
    }                                                                                                          \
  } while (0)
#define STAMP_NEXT() do { if (st_n < kStampD) ++st_n; } while (0)
#else
#define STAMP(k_) do {} while (0)
#define STAMP_NEXT() do {} while (0)
#endif
  u32x4 rq[(RES && ROWS) ? MT : 1][2];                      // ROWS + RES: next tile's residual, in flight from double step 6
  int dn = 2;                                               // double tile to issue at the start of the next double step
  int hc = 2 * body0;                                       // first half-chunk of the current body
  const int hc_end = 2 * (body0 + nbody);
  bool after_epi = false;
  const int ND = nbody * 9;                                 // double steps per tile (of this workgroup's K share)
#define DSTEP(D)                                                                                               \
  do {                                                                                                         \
    constexpr int U0_ = 2 * (D), U1_ = 2 * (D) + 1;                                                            \
    constexpr int WN_ = TG + (((D) == 0 || (D) == 1 || (D) == 5 || (D) == 6) ? PT : 0) +                       \
                        ((RES && ROWS && ((D) == 6 || (D) == 7)) ? 2 * MT : 0);                                \
    STAMP(0);                                                                                                  \
    const bool rl_ = RES && !ROWS && BN == 128 && PT >= 4 && res_lds && hc + 2 >= hc_end;   /* last body: residual rides in the idle slots */ \
    if (!WRES && !(dbg & 1)) {                                                                                 \
      if (RES && !ROWS && BN == 128 && PT >= 4 && ((D) == 7 || (D) == 8) && rl_) {                             \
        unsigned ro_ = roff[(RES && !ROWS) ? (D) - 5 : 0];                                                     \
        asm volatile("" : "+v"(ro_));          /* keep the 32-bit offset live, not a hoisted 64-bit address (spills) */ \
        const char* rp_ = (const char*)p.res + ro_;                                                            \
        _Pragma("unroll") for (int o = 0; o < TG; ++o)                                                         \
          GLDS16(rp_ + o * 64, Bs + (((D) + 2) % NBD) * DT_B + o * 8192 + wave * 1024);                        \
      } else {                                                                                                 \
        const int di_ = dn < ND ? dn : dn - ND;                                                                \
        ISSUE_DT(di_, ((D) + 2) % NBD);                                                                        \
      }                                                                                                        \
    }                                                                                                          \
    ++dn;                                                                                                      \
    if ((D) == 0 && !(dbg & 2)) ISSUE_PATCH(patch_src + (hc + 1) * 64, 1);                                     \
    if ((D) == 5 && !(dbg & 2)) {                                                                              \
      if (RES && !ROWS && BN == 128 && PT >= 4 && rl_) {                                                       \
        _Pragma("unroll") for (int pt = 0; pt < 2; ++pt) {                                                     \
          unsigned ro_ = roff[(RES && !ROWS) ? pt : 0];                                                        \
          asm volatile("" : "+v"(ro_));                                                                        \
          const char* rp_ = (const char*)p.res + ro_;                                                          \
          _Pragma("unroll") for (int c = 0; c < 2; ++c) GLDS16(rp_ + c * 64, Ps + ((pt * 2 + c) * 512 + wave * 64) * 16); \
        }                                                                                                      \
        _Pragma("unroll") for (int rr = 4; rr < PT; ++rr)       /* keeps the op count of a patch burst */      \
          GLDS16(patch_src + psrc[rr], Ps + (rr * 512 + wave * 64) * 16);                                      \
      } else {                                                                                                 \
        const char* s_ = hc + 2 < hc_end ? patch_src + (hc + 2) * 64 : (has_next ? n_patch_src : patch_src);   \
        ISSUE_PATCH(s_, 0);                                                                                    \
      }                                                                                                        \
    }                                                                                                          \
    if constexpr (RES && ROWS && (D) == 6) {   /* residual of the NEXT tile (of this one again at the very end) */ \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) {                                                      \
        const char* rp_ = (const char*)p.res + rows_off(has_next ? n_m0 : m0, has_next ? n_mc0 : mc0, pt);     \
        rq[pt][0] = *(const u32x4*)rp_;                                                                        \
        rq[pt][1] = *(const u32x4*)(rp_ + 64);                                                                 \
      }                                                                                                        \
    }                                                                                                          \
    /* ring: slot offsets fold into the ds_read immediates.  WRES: 72 KB of distinct offsets do not fit 16-bit     \
       immediates and the compiler would hoist one base register per step out of the tile loop (+48 VGPRs):       \
       one opaque add per step instead */                                                                      \
    int wof_ = wbase + (WRES ? (D) : (D) % NBD) * DT_B;                                                        \
    if constexpr (WRES) { wof_ = wbase; asm volatile("" : "+v"(wof_)); wof_ += (D) * DT_B; }                   \
    if (!(dbg & 16)) LOADF(wof_, U0_ / 9, U0_ % 9, U1_ / 9, U1_ % 9);                                          \
    if constexpr (WRES) {      /* only the patch bursts of steps 0 / 5 are in flight: landed before steps 4 / 9 */ \
      if ((D) == 3) WAIT_VM(0);                                                                                \
      if ((D) == 8) WAIT_VM(RES ? 2 * MT : 0);     /* the residual prefetch of step 6 is younger */            \
    } else if (dbg & 3) WAIT_VM(0);                                                                            \
    else if ((D) == 0 && after_epi) WAIT_VM(WN_ + EPI_OPS);                                                    \
    else WAIT_VM(WN_);                                                                                         \
    STAMP(1);                                                                                                  \
    BARRIER();                                                                                                 \
    STAMP(2);                                                                                                  \
    if (!(dbg & 4)) MFMAS();                                                                                   \
    STAMP(3);                                                                                                  \
    STAMP_NEXT();                                                                                              \
    BARRIER();                                                                                                 \
  } while (0)

#ifdef FLOPE_STAG_DBG
  // dbg & 64 (diagnostic build only; MI355X_MICROARCH "DVFS give-back" item 6): shader-clock and 100 MHz real-time stamps around the
  // FIRST tile's main loop, one record per (workgroup, wave group) into p.split_ws: {clk0, clk1, rt0, rt1}.  In-kernel clock =
  // (clk1 - clk0) / (rt1 - rt0) * 100 MHz; cycles per double step = (clk1 - clk0) / ND (the MFMA floor is 1024).
  unsigned long long st_c0 = 0, st_r0 = 0;
  bool st_first = true;
  if (dbg & 64) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
  for (;;) {
    for (int hcp = 0; hcp < nbody; ++hcp) {
      DSTEP(0); DSTEP(1); DSTEP(2); DSTEP(3); DSTEP(4); DSTEP(5); DSTEP(6); DSTEP(7); DSTEP(8);
      after_epi = false;
      hc += 2;
    }
#ifdef FLOPE_STAG_DBG
    if ((dbg & 64) && st_first && p.split_ws) {
      const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();
      if (wl == 0 && lane == 0) {
        unsigned long long* d_ = (unsigned long long*)p.split_ws + ((size_t)blockIdx.x * 2 + group) * 4;
        d_[0] = st_c0; d_[1] = c1_; d_[2] = st_r0; d_[3] = r1_;
      }
      st_first = false;
    }
    if ((dbg & 128) && p.split_ws && wl == 0 && lane < 4 && !st_done) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned long long* d_ = (unsigned long long*)((char*)p.split_ws + 65536) + ((size_t)blockIdx.x * 2 + group) * kStampD * 4;
      for (int i = 0; i < st_n; ++i) d_[i * 4 + lane] = *(const unsigned long long*)(st_lds + i * 32 + lane * 8);
    }
    if (dbg & 128) { st_done = true; st_n = kStampD; }      // once per workgroup
#endif
    // ---- tile finished for this wave: + bias (+ residual) (ReLU) -> 16-bit padded NHWC, straight from registers
    const bool full_tile = mend - m0 == BM;
    if (KSP > 1) {                              // split-K: raw fp32 partial sums, [ks][flat pixel][channel]
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) {
        const int mm = m0 + group * GP + wpx * 64 + pt * 16 + pcol;
        if (mm < mend) {
          float* wp = p.split_ws + ((size_t)ks * p.M + mm) * p.Cout + cb;
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) *(f32x4*)(wp + (ct >> 1) * 32 + (ct & 1) * 4) = acc[pt][ct];
        }
      }
    } else {
      size_t ooff[MT];
      bool ok[MT];
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) ooff[pt] = out_off(pt, ok[pt]);   // clamped: loads below are always legal
      // all residual loads first (one latency, not one per pixel tile), then compute + store
      u32x4 rv[RES ? MT : 1][2];
      if constexpr (RES && !ROWS && BN == 128 && PT >= 4) {
        if (res_lds) {                           // this lane's own DMA pieces: patch buffer 0 (pt 0, 1), ring slots 0 and 1 (pt 2, 3)
          WAIT_VM(0);
#pragma unroll
          for (int pt = 0; pt < MT; ++pt)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int off_ = pt < 2 ? ((pt * 2 + c) * 512 + wave * 64 + lane) * 16
                                      : 2 * PATCH_B + (pt - 2) * DT_B + c * 8192 + wave * 1024 + lane * 16;
              rv[pt][c] = *(const u32x4*)(smem + off_);
            }
        }
      }
      if constexpr (RES && !ROWS) {
        if (!res_pre && !res_lds) {
#pragma unroll
          for (int pt = 0; pt < MT; ++pt) {
            const char* rp = (const char*)p.res + ooff[pt];
            rv[pt][0] = *(const u32x4*)rp;
            rv[pt][1] = *(const u32x4*)(rp + 64);
          }
        }
      }
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) {
        float v[NT * 4];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[ct * 4 + q] = acc[pt][ct][q];
        if constexpr (RES && !ROWS) {
          if (!res_pre) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                v[c * 8 + q * 2] += unpack_lo<T>(rv[pt][c][q]);
                v[c * 8 + q * 2 + 1] += unpack_hi<T>(rv[pt][c][q]);
              }
          }
        }
        if (ok[pt] && !(dbg & 32)) {
          char* op = (char*)p.out + ooff[pt];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned w_ = pack2<T>(v[c * 8 + q * 2], v[c * 8 + q * 2 + 1]);
              o[q] = pk_out16<T>(w_, p.relu);                // ReLU (+ float16 saturation) on the packed pair
            }
            *(u32x4*)(op + c * 64) = o;
          }
        }
        if (has_next) {                         // the next tile's accumulators start at the bias again
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = BIAS4(ct);
        }
        if constexpr (RES && ROWS) {              // next tile's residual (prefetched at double step 6) rides in the accumulators
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i = ct * 4 + q;
              const unsigned w_ = rq[pt][i >> 3][(i & 7) >> 1];
              acc[pt][ct][q] += (i & 1) ? unpack_hi<T>(w_) : unpack_lo<T>(w_);
            }
        }
      }
    }
    if (!has_next) break;
    // ---- next tile: geometry, address table; the step stream (ring, patch buffers) just continues
    tile += G;
    m0 = n_m0; mend = n_mend; R0 = n_R0; mc0 = n_mc0; patch_src = n_patch_src;
    has_next = tile + G < p.total_tiles;
    if (has_next) TILE_GEOM(tile + G, n_m0, n_mend, n_R0, n_mc0, n_patch_src);
    if constexpr (!ROWS) LANE_SETUP();          // ROWS: every tile has the first tile's address table
    dn -= ND;
    hc = 2 * body0;
    if (full_tile && (!res_pre || ROWS)) {
      after_epi = true;                        // the epilogue issued exactly EPI_OPS VM ops per lane
    } else {
      WAIT_VM(0);                              // partial / first residual tile: op count differs -> drain once, static counts stay valid
    }
    res_pre = false;
  }
  if (group == 0) BARRIER();                                // every wave executes the same number of barriers
  WAIT_VM(0);                                               // drain the wrapped-around tail DMAs before LDS is released
#undef DSTEP
#undef STAMP
#undef STAMP_NEXT
#undef BIAS4
#undef ISSUE_DT
#undef ISSUE_PATCH
#undef WAIT_VM
#undef BARRIER
#undef LOADF
#undef MFMAS
#undef TILE_GEOM
#undef XO
#undef LANE_SETUP
}

// ---------------------------------------------------------------------------------------------------------------
// conv_gstag: the same 8-wave / two-staggered-groups / 256 px x 128 ch structure for the three 3x3 STRIDE-2 convs.
// A stride-2 window has no patch reuse worth its LDS (the input region of a tile is 4x its output), so each
// (64-channel chunk, tap) double step gets its own gathered pixel tile: 256 pixels x 64 channels = 32 KB by LDS-DMA,
// per-lane source = that lane's input pixel (2ho + ky, 2wo + kx) in the padded tensor, full 128-byte pixel rows (a first
// version gathered 32-channel halves per tap -- 64-byte segments, half a cache line per request -- and lost to the
// 4-wave kernel).  Per double step a wave issues 4 gather pieces + 2 weight pieces (the tap's tiles of the chunk's two
// half-chunks), two double steps ahead, into 3-deep rings (96 KB pixels + 48 KB weights); every wait is the constant
// vmcnt(6).  Against conv_mfma<128x128,gather> the weight stream per MAC halves.  Pixel tiles use conv_mfma's A-tile
// image: 128-byte rows, 16-byte slot j of row r holds chunk j ^ ((r >> 1) & 7).
template <typename T>
__global__ __launch_bounds__(512, 2) void conv_gstag_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int BM = 256, GP = 128, TILE_B = 8192, DT_B = 16384, MT = 4, NT = 4;
  constexpr int XD_B = 32768, X_BYTES = 3 * XD_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;                                   // 3 gathered pixel tiles [256 px][128 B]
  char* const Bs = smem + X_BYTES;                         // 3 double tiles of weights
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, wl = wave & 3, wpx = wl & 1, wch = wl >> 1;
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px_s(r16);
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lb % p.ntiles;
  const int HoWo = p.Ho * p.Wo;
  const int nchunks = p.Cin / 64, ND = nchunks * 9;        // one double step = one tap x one 64-channel chunk
  const size_t pixB = (size_t)p.Cin * 2;
  const int m0 = (lb / p.ntiles) * BM, mend = min(m0 + BM, p.M);

  // this lane's four gather pieces per pixel tile: piece rr*512 + wave*64 + lane -> pixel row (>> 3), 16-byte slot (& 7)
  const char* gsrc[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int q = rr * 512 + wave * 64 + lane, row = q >> 3;
    const int m = min(m0 + row, mend - 1);
    const int b_ = fastdiv(m, p.mg_hw, p.sh_hw), r_ = m - b_ * HoWo;
    const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;
    gsrc[rr] = (const char*)p.in + (((size_t)b_ * p.Hip + 2 * ho_) * p.Wip + 2 * wo_) * pixB + (((q & 7) ^ ((row >> 1) & 7)) << 4);
  }
  int xds[MT];                                             // fragment offset of half-chunk 0; half-chunk 1 is slot ^ 4
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int row = group * GP + wpx * 64 + pt * 16 + pcol;
    xds[pt] = row * 128 + ((g ^ ((row >> 1) & 7)) << 4);
  }
  int toff[9];                                             // byte offset of tap (ky, kx) from the window's top-left pixel
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = (int)(((t / 3) * p.Wip + (t % 3)) * (int)pixB);
  const char* const b_base = (const char*)p.w + (size_t)ntile * (nchunks * 18) * TILE_B + wave * 1024;
  const unsigned lane16 = lane * 16;
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wbase = X_BYTES + (wch * 64 + r16) * 64 + ((g ^ wsw) << 4);
  const int cb = ntile * 128 + wch * 64 + g * 8;           // two runs of 8 channels: cb.. and cb + 32.. (stag_row_to_channel)
  float bias[NT * 4];
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) bias[i] = p.bias[cb + (i >> 3) * 32 + (i & 7)];
  frag wf[2][NT], xf[2][MT];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = f32x4{bias[ct * 4], bias[ct * 4 + 1], bias[ct * 4 + 2], bias[ct * 4 + 3]};

#define G_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define G_BARRIER()                                                                                            \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
  // double step (chunk c_, tap t_): the gathered [256][64 ch] pixel tile and the tap's weight tiles of half-chunks 2c, 2c+1
#define G_ISSUE(c_, t_, slot_)                                                                                 \
  do {                                                                                                         \
    const int o_ = toff[t_] + (c_) * 128;                                                                      \
    _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                                           \
      GLDS16(gsrc[rr] + o_, Xs + (slot_) * XD_B + (rr * 512 + wave * 64) * 16);                                \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                              \
      GLDS16(b_base + (size_t)((2 * (c_) + h) * 9 + (t_)) * TILE_B + lane16, Bs + (slot_) * DT_B + h * TILE_B + wave * 1024); \
  } while (0)

  int ch = 0;                                              // current 64-channel chunk
  G_ISSUE(0, 0, 0);
  G_ISSUE(0, 1, 1);
  G_WAIT(6);                                   // double step 0's pixels and weights landed (step 1's may fly)
  G_BARRIER();
  if (group == 1) G_BARRIER();                 // group B runs one phase behind group A

#define G_DSTEP(D)                                                                                             \
  do {                                                                                                         \
    {   /* two double steps ahead: tap D + 2 of this chunk, or tap D - 7 of the next (past the end: chunk 0 again) */ \
      const int cn_ = ch + 1 < nchunks ? ch + 1 : 0;                                                           \
      if ((D) + 2 < 9) G_ISSUE(ch, (D) + 2, ((D) + 2) % 3); else G_ISSUE(cn_, (D) + 2 - 9, ((D) + 2) % 3);     \
    }                                                                                                          \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct) {                                                        \
      wf[0][ct] = *(const frag*)(smem + wbase + ((D) % 3) * DT_B + ct * 1024);                                 \
      wf[1][ct] = *(const frag*)(smem + wbase + ((D) % 3) * DT_B + TILE_B + ct * 1024);                        \
    }                                                                                                          \
    _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) {                                                        \
      xf[0][pt] = *(const frag*)(smem + ((D) % 3) * XD_B + xds[pt]);                                           \
      xf[1][pt] = *(const frag*)(smem + ((D) % 3) * XD_B + (xds[pt] ^ 64));                                    \
    }                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    G_WAIT(6);                                                                                                 \
    G_BARRIER();                                                                                               \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                              \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt)                                                        \
        _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                      \
          acc[pt][ct] = Elem<T>::mfma(wf[h][ct], xf[h][pt], acc[pt][ct]);                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    G_BARRIER();                                                                                               \
  } while (0)

  for (; ch < nchunks; ++ch) {
    G_DSTEP(0); G_DSTEP(1); G_DSTEP(2); G_DSTEP(3); G_DSTEP(4); G_DSTEP(5); G_DSTEP(6); G_DSTEP(7); G_DSTEP(8);
  }
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = m0 + group * GP + wpx * 64 + pt * 16 + pcol;
    if (m < mend) {                            // accumulators carry the bias; ReLU + pack; two 16-byte stores 64 bytes apart
      const int b_ = fastdiv(m, p.mg_hw, p.sh_hw), r_ = m - b_ * HoWo;
      const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;
      char* op = (char*)p.out + ((((size_t)b_ * p.Hop + ho_ + 1) * p.Wop + wo_ + 1) * p.Cout + cb) * 2;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = c * 8 + q * 2;
          o[q] = pk_out16<T>(pack2<T>(acc[pt][i >> 2][i & 3], acc[pt][(i + 1) >> 2][(i + 1) & 3]), p.relu);
        }
        *(u32x4*)(op + c * 64) = o;
      }
    }
  }
  if (group == 0) G_BARRIER();                 // every wave executes the same number of barriers
  G_WAIT(0);                                   // drain the wrapped-around tail DMAs before LDS is released
#undef G_DSTEP
#undef G_ISSUE
#undef G_BARRIER
#undef G_WAIT
  (void)ND;
}

extern "C" int flope_conv_gstag_init() {
  hipError_t e = hipFuncSetAttribute((const void*)conv_gstag_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_gstag_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return (int)e;
}

// 3x3 stride-2 pad-1 convolution, Cin % 64 == 0, Cout % 128 == 0; p->mtiles = ceil(M / 256), p->ntiles = Cout / 128;
// p->w = the conv_stag weight image ([ntile][hc*9 + tap][128 rows][32 k])
extern "C" int flope_conv_gstag_launch(const ConvP* p, int dtype, void* stream) {
  if (p->stride != 2 || p->ntaps != 9 || p->Cin % 64 || p->Cout % 128 || p->res) return (int)hipErrorInvalidValue;
  const dim3 grid(p->mtiles * p->ntiles), block(512);
  const size_t lds = 3 * 32768 + 3 * 16384;
  if (dtype == 0) hipLaunchKernelGGL(conv_gstag_kernel<bf16_t>, grid, block, lds, (hipStream_t)stream, *p);
  else            hipLaunchKernelGGL(conv_gstag_kernel<f16_t>, grid, block, lds, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

// split-K epilogue: out = act(sum over the K shares of the fp32 partials + bias (+ residual)) -> 16-bit padded NHWC.
// One thread = one output pixel x 8 channels (two 16-byte partial loads per share, one 16-byte store).
template <typename T>
__global__ __launch_bounds__(256) void conv_split_finalize_kernel(const ConvP p) {
  const int c8n = p.Cout >> 3;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.M * c8n) return;
  const int m = (int)(idx / c8n), c0 = (int)(idx - (size_t)m * c8n) * 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = p.bias[c0 + i];
  for (int ks = 0; ks < p.ksplit; ++ks) {
    const float* wp = p.split_ws + ((size_t)ks * p.M + m) * p.Cout + c0;
    const f32x4 a = *(const f32x4*)wp, b = *(const f32x4*)(wp + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] += a[i]; v[4 + i] += b[i]; }
  }
  const int HoWo = p.Ho * p.Wo;
  const int b_ = fastdiv(m, p.mg_hw, p.sh_hw), r_ = m - b_ * HoWo;
  const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;
  const size_t off = ((((size_t)b_ * p.Hop + ho_ + 1) * p.Wop + wo_ + 1) * p.Cout + c0) * 2;
  if (p.res) {
    const u32x4 rv = *(const u32x4*)((const char*)p.res + off);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q * 2] += unpack_lo<T>(rv[q]); v[q * 2 + 1] += unpack_hi<T>(rv[q]); }
  }
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned w_ = pack2<T>(v[q * 2], v[q * 2 + 1]);
    o[q] = pk_out16<T>(w_, p.relu);
  }
  *(u32x4*)((char*)p.out + off) = o;
}

extern "C" int flope_conv_split_finalize_launch(const ConvP* p, int dtype, void* stream) {
  const size_t total = (size_t)p->M * (p->Cout >> 3);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == 0) hipLaunchKernelGGL(conv_split_finalize_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, *p);
  else            hipLaunchKernelGGL(conv_split_finalize_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

template <typename T>
static hipError_t stag_attr() {
  hipError_t e = hipSuccess;
#define A(PT_, BN_) \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_stag_kernel<T, PT_, BN_, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_stag_kernel<T, PT_, BN_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  A(2, 128) A(3, 128) A(4, 128) A(5, 128) A(6, 128) A(8, 128) A(2, 64) A(3, 64) A(4, 64) A(5, 64) A(6, 64) A(8, 64)
#undef A
#define A(PT_) \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_stag_kernel<T, PT_, 128, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  A(4) A(5) A(6) A(8)
#undef A
#define A(PT_) \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_stag_kernel<T, PT_, 64, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_stag_kernel<T, PT_, 64, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  A(3) A(5) A(6) A(8)
#undef A
  return e;
}

extern "C" int flope_conv_stag_init() {
  hipError_t e = stag_attr<bf16_t>();
  if (e == hipSuccess) e = stag_attr<f16_t>();
  return (int)e;
}

template <typename T, int BN, bool RES>
static void stag_launch(const ConvP& p, int pt, int grid_blocks, size_t lds, hipStream_t st) {
  const dim3 grid(grid_blocks), block(512);
  switch (pt) {
    case 2: hipLaunchKernelGGL((conv_stag_kernel<T, 2, BN, RES, false>), grid, block, lds, st, p); break;
    case 3: hipLaunchKernelGGL((conv_stag_kernel<T, 3, BN, RES, false>), grid, block, lds, st, p); break;
    case 4: hipLaunchKernelGGL((conv_stag_kernel<T, 4, BN, RES, false>), grid, block, lds, st, p); break;
    case 5: hipLaunchKernelGGL((conv_stag_kernel<T, 5, BN, RES, false>), grid, block, lds, st, p); break;
    case 6: hipLaunchKernelGGL((conv_stag_kernel<T, 6, BN, RES, false>), grid, block, lds, st, p); break;
    default: hipLaunchKernelGGL((conv_stag_kernel<T, 8, BN, RES, false>), grid, block, lds, st, p); break;
  }
}
template <typename T>
static void stag_dsf_launch(const ConvP& p, int pt, int grid_blocks, size_t lds, hipStream_t st) {
  const dim3 grid(grid_blocks), block(512);
  switch (pt) {
    case 4: hipLaunchKernelGGL((conv_stag_kernel<T, 4, 128, false, false, true>), grid, block, lds, st, p); break;
    case 5: hipLaunchKernelGGL((conv_stag_kernel<T, 5, 128, false, false, true>), grid, block, lds, st, p); break;
    case 6: hipLaunchKernelGGL((conv_stag_kernel<T, 6, 128, false, false, true>), grid, block, lds, st, p); break;
    default: hipLaunchKernelGGL((conv_stag_kernel<T, 8, 128, false, false, true>), grid, block, lds, st, p); break;
  }
}
template <typename T, bool RES>
static void stag_rows_launch(const ConvP& p, int pt, int grid_blocks, size_t lds, hipStream_t st) {
  const dim3 grid(grid_blocks), block(512);
  switch (pt) {
    case 3: hipLaunchKernelGGL((conv_stag_kernel<T, 3, 64, RES, true>), grid, block, lds, st, p); break;   // odd: weights resident
    case 5: hipLaunchKernelGGL((conv_stag_kernel<T, 5, 64, RES, true>), grid, block, lds, st, p); break;
    case 6: hipLaunchKernelGGL((conv_stag_kernel<T, 6, 64, RES, true>), grid, block, lds, st, p); break;
    default: hipLaunchKernelGGL((conv_stag_kernel<T, 8, 64, RES, true>), grid, block, lds, st, p); break;
  }
}

// p->patch_rows_max carries PT (2..6 or 8 DMA rounds per patch buffer); Cout == 64 selects the 512 x 64 tile;
// p->total_tiles = mtiles * ntiles; grid_blocks <= total_tiles and a multiple of ntiles;
// lds = 2*PT*8 KiB + 3 * 2 * (BN*64 B)
extern "C" int flope_conv_stag_launch(const ConvP* p, int dtype, int grid_blocks, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int pt = p->patch_rows_max;
  if (p->ksplit > 1 && (p->per_image == 2 || p->res || !p->split_ws || (p->Cin / 64) % p->ksplit || p->Cout < 128 ||
                        grid_blocks != p->total_tiles * p->ksplit)) return (int)hipErrorInvalidValue;
#define GO(T, BN_) (p->res ? stag_launch<T, BN_, true>(*p, pt, grid_blocks, lds, st) : stag_launch<T, BN_, false>(*p, pt, grid_blocks, lds, st))
  if (p->ds_in) {                   // folded 1x1 stride-2 shortcut: one tile per workgroup, no residual input
    if (p->Cout < 128 || p->res || pt < 4 || pt == 7 || grid_blocks != p->total_tiles * (p->ksplit > 1 ? p->ksplit : 1) || p->ds_Cin % 64 || !p->ds_w) return (int)hipErrorInvalidValue;
    if (dtype == 0) stag_dsf_launch<bf16_t>(*p, pt, grid_blocks, lds, st); else stag_dsf_launch<f16_t>(*p, pt, grid_blocks, lds, st);
  } else if (p->per_image == 2) {          // ROWS geometry: 8-row bands of one image (p->tiles_per_image bands per image)
    if (p->Cout != 64 || (p->Wo > 64 && p->nseg != (p->Wo + 63) / 64) || p->Ho % 8 || (pt != 3 && pt != 5 && pt != 6 && pt != 8) || ((pt & 1) && p->Cin != 64)) return (int)hipErrorInvalidValue;
    if (p->nseg > 1 && (pt < 6 || p->tiles_per_image != (p->Ho / 8) * p->nseg)) return (int)hipErrorInvalidValue;   // 10 x 66 pixels x 4 pieces = 2640 <= 6 x 512
    if (dtype == 0) { if (p->res) stag_rows_launch<bf16_t, true>(*p, pt, grid_blocks, lds, st); else stag_rows_launch<bf16_t, false>(*p, pt, grid_blocks, lds, st); }
    else            { if (p->res) stag_rows_launch<f16_t, true>(*p, pt, grid_blocks, lds, st); else stag_rows_launch<f16_t, false>(*p, pt, grid_blocks, lds, st); }
  } else if (p->Cout == 64) { if (dtype == 0) GO(bf16_t, 64); else GO(f16_t, 64); }
  else               { if (dtype == 0) GO(bf16_t, 128); else GO(f16_t, 128); }
#undef GO
  return (int)hipGetLastError();
}
