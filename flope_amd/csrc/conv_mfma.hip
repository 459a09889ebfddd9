// Implicit-GEMM convolution for gfx950 on v_mfma_f32_16x16x32_{bf16,f16}.
//
// Replaces every Conv2d -> BatchNorm2d -> [ReLU] (+ residual add -> ReLU) group of
// the torchvision ResNet-18 trunk the reference runs at
// sunflower/models/posenet.py:25 (layer1..layer4 BasicBlocks and their 1x1/s2
// downsample convs); eval-mode BN is folded into weight + bias on the host.
//
// GEMM view (operands swapped so the epilogue is channel-contiguous):
//     D[ch][px] = sum_k W[ch][k] * X[px][k],   k = (tap, ci)
//   MFMA A operand = weight rows (output channels), B operand = pixels.
//   With the 16x16x32 C/D map (col = lane&15, row = 4*(lane>>4)+reg) each lane
//   ends up with one pixel and 4*NT *consecutive* channels (the host packs the
//   weight rows in the matching permuted order), i.e. 16-byte NHWC stores and
//   16-byte residual loads straight from the accumulators -- no LDS round trip.
//
// Two pixel-operand feeds, one kernel body:
//   PATCH = true  (3x3 stride 1): the tile's input rows (full padded width, one
//       64-channel chunk) are staged ONCE per chunk in LDS; the nine taps are
//       nine shifted views of that patch.  L2->LDS traffic per MAC drops ~9x
//       versus re-gathering every tap.
//   PATCH = false (stride 2 3x3, 1x1 downsample): classic gather, one
//       [BM][64ch] pixel tile per (chunk, tap) step, double buffered.
//   Weights always stream through a double-buffered [BN][64] LDS tile that the
//   host has already laid out as the (swizzled) LDS image, so the copy is linear.
//
// Staging: global_load_lds_dwordx4 -- 1 KiB per wave instruction lands in LDS at
//   wave-uniform base + lane*16 without touching VGPRs or the ds_write path.  The LDS
//   images are therefore lane-linear and every swizzle is applied to the per-lane SOURCE
//   address.  Tiles live in an NBUF-deep ring: the tiles of step s+NBUF-1 are issued at the
//   top of step s and retired by a COUNTED s_waitcnt vmcnt + raw s_barrier at its bottom
//   (measured issue->landed latency of a weight tile under load is ~0.9 us, i.e. about two
//   steps; a one-step prefetch left the MFMA pipe ~30 % busy).
//
// LDS rows are 128 B (64 x 16-bit); 16-byte slot j of row/pixel r is stored at slot
// j ^ ((r >> 1) & 7).  MFMA column c of a 16-pixel tile is pixel PI(c) of the tile with
// PI = {0,2,4,6, 1,3,5,7,9,11,13,15, 8,10,12,14}: ds_read_b128 is serviced in 16-lane
// groups {0-3,12-15,20-27}/{4-11,16-19,28-31}/..., i.e. 8 lanes at k-slot g and 8 at g^1;
// PI puts same-parity pixels in each half, so the 16 reads of a group hit 16 distinct 16-B
// positions of the 256-B bank row for EVERY tap shift (r01: 26 % conflict cycles without it).
#include "common.h"

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ int tile_px(int c) {   // MFMA column -> pixel offset inside a 16-pixel tile
  return c < 4 ? 2 * c : (c < 12 ? 2 * (c - 4) + 1 : 2 * (c - 8));
}

template <typename T, int BM, int BN, int WPX, int WCH, bool PATCH, int NBUF>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int MT = BM / (WPX * 16);   // pixel tiles per wave
  constexpr int NT = BN / (WCH * 16);   // channel tiles per wave
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int AP = BM / 32, BP = BN / 32;
  constexpr int LOADS_PER_STEP = BP + (PATCH ? 0 : AP);     // glds instructions a wave issues per step
  static_assert(WPX * WCH == 4 && NT % 2 == 0 && (NBUF == 2 || NBUF == 3), "tile config");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Bs = smem;                       // NBUF x B_BYTES weight-tile ring
  char* const As = smem + NBUF * B_BYTES;      // gather: NBUF x A_BYTES ring; patch: rows x Wip x 128 (4 KiB rounded)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px(r16);
  const int wpx = wave % WPX, wch = wave / WPX;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  const int HoWo = p.Ho * p.Wo;
  const int nsteps = p.nchunks * p.ntaps;

  int m0, mend;
  if (p.per_image) {
    const int img = mtile / p.tiles_per_image, t = mtile - img * p.tiles_per_image;
    m0 = img * HoWo + t * BM;
    mend = min(m0 + BM, (img + 1) * HoWo);
  } else {
    m0 = mtile * BM;
    mend = min(m0 + BM, p.M);
  }

  // ---- weight (MFMA A operand) staging: linear copy of the host-built LDS image;
  // wave w moves pieces (i*4 + w)*64 + lane, i < BP
  const char* const b_src = (const char*)p.w + (size_t)ntile * nsteps * B_BYTES + wave * 1024 + lane * 16;
  const int st_dst = wave * 1024;              // wave-uniform LDS-DMA destination inside a tile

  // ---- pixel (MFMA B operand) feed
  const char* a_src[PATCH ? 1 : AP];
  int pi0[MT];              // patch mode: this lane's pixel index inside the patch per pixel tile
  size_t patch_src = 0;     // patch mode: byte offset of patch row 0, column 0
  int patch_pieces = 0;
  if constexpr (PATCH) {
    const int mf = m0, ml = mend - 1;
    const int b0 = mf / HoWo, ho0 = (mf - b0 * HoWo) / p.Wo;
    const int b1 = ml / HoWo, ho1 = (ml - b1 * HoWo) / p.Wo;
    const int R0 = b0 * p.Hip + ho0 * p.stride;
    const int R1 = b1 * p.Hip + ho1 * p.stride + 2;
    patch_src = (size_t)R0 * p.Wip * p.Cin * 2;
    patch_pieces = (R1 - R0 + 1) * p.Wip * 8;
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int m = min(m0 + wpx * MT * 16 + pt * 16 + pcol, mend - 1);
      const int b = fastdiv(m, p.mg_hw, p.sh_hw), r = m - b * HoWo, ho = fastdiv(r, p.mg_w, p.sh_w), wo = r - ho * p.Wo;
      pi0[pt] = (b * p.Hip + ho * p.stride - R0) * p.Wip + wo * p.stride;
    }
  } else {
    // staged row (i*4 + wave)*8 + (lane>>3), physical slot lane&7 <- logical slot (lane&7)^((row>>1)&7)
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int row = (i * 4 + wave) * 8 + (lane >> 3);
      const int m = min(m0 + row, mend - 1);
      const int b = fastdiv(m, p.mg_hw, p.sh_hw), r = m - b * HoWo, ho = fastdiv(r, p.mg_w, p.sh_w), wo = r - ho * p.Wo;
      const size_t pix = ((size_t)b * p.Hip + ho * p.stride) * p.Wip + wo * p.stride +
                         (p.ntaps == 1 ? p.Wip + 1 : 0);
      a_src[i] = (const char*)p.in + pix * p.Cin * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
  }

  // fragment read offsets (bytes)
  const int s0w = (g ^ (r16 >> 1)) << 4;                      // weight rows: row = ..+r16
  const int s0x = (g ^ (pcol >> 1)) << 4;                     // gather pixel rows: row = ..+pcol
  const int wbase = (wch * NT * 16 + r16) * 128;
  const int xbase = (wpx * MT * 16 + pcol) * 128;

  const int cb = ntile * BN + wch * NT * 16 + g * (4 * NT);
  float bias[NT * 4];                          // accumulators start at the folded-BN bias
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) bias[i] = p.bias[cb + i];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = f32x4{bias[ct * 4], bias[ct * 4 + 1], bias[ct * 4 + 2], bias[ct * 4 + 3]};

  // ---- staging helpers (macros keep every array index a compile-time constant)
#define ISSUE_B(step_, slot_)                                                                                  \
  do {                                                                                                         \
    const char* s_ = b_src + (size_t)(step_) * B_BYTES;                                                        \
    _Pragma("unroll") for (int i = 0; i < BP; ++i) GLDS16(s_ + i * 4096, Bs + (slot_) * B_BYTES + st_dst + i * 4096); \
  } while (0)
#define ISSUE_A(step_, slot_)                                                                                  \
  do {                                                                                                         \
    const int c_ = (step_) / p.ntaps, t_ = (step_) - c_ * p.ntaps;                                             \
    const int ky_ = t_ / 3, kx_ = t_ - ky_ * 3;                                                                \
    const size_t off_ = ((size_t)(ky_ * p.Wip + kx_) * p.Cin + c_ * 64) * 2;                                   \
    _Pragma("unroll") for (int i = 0; i < AP; ++i) GLDS16(a_src[i] + off_, As + (slot_) * A_BYTES + st_dst + i * 4096); \
  } while (0)
#define ISSUE_PATCH(chunk_)                                                                                    \
  do {                                                                                                         \
    const char* src_ = (const char*)p.in + patch_src + (chunk_) * 128;                                         \
    const size_t pixB_ = (size_t)p.Cin * 2;                                                                    \
    for (int q0 = 0; q0 < patch_pieces; q0 += 256) {                                                           \
      const int q = min(q0 + wave * 64 + lane, patch_pieces - 1);                                              \
      const int pi = q >> 3, js = (q & 7) ^ ((pi >> 1) & 7);                                                   \
      GLDS16(src_ + (size_t)pi * pixB_ + js * 16, As + (q0 + wave * 64) * 16);                                 \
    }                                                                                                          \
  } while (0)
  // LDS-DMA completion is visible to other waves only after the issuing wave's vmcnt wait AND a
  // barrier; the raw s_barrier (not __syncthreads) keeps younger DMAs in flight across it.
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define BLOCK_BARRIER()                                                                                        \
  do {                                                                                                         \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
  } while (0)
#define LOAD_X(tap_, slot_)                                                                                    \
  do {                                                                                                         \
    int xo_[MT];                                                                                               \
    if constexpr (PATCH) {                                                                                     \
      const int ky_ = (tap_) / 3, kx_ = (tap_) - ky_ * 3;                                                      \
      const int ts_ = ky_ * p.Wip + kx_;                                                                       \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) {                                                      \
        const int pi = pi0[pt] + ts_;                                                                          \
        xo_[pt] = (pi << 7) + ((g ^ ((pi >> 1) & 7)) << 4);                                                    \
      }                                                                                                        \
    } else {                                                                                                   \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) xo_[pt] = (slot_) * A_BYTES + xbase + pt * 2048 + s0x; \
    }                                                                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) xf[kk][pt] = *(const frag*)(As + (xo_[pt] ^ (kk * 64))); \
  } while (0)

  frag wf[2][NT], xf[2][MT];

  // ---- prologue: patch of chunk 0 (oldest in the vmcnt queue), then the first NBUF-1 steps' tiles
  if constexpr (PATCH) ISSUE_PATCH(0);
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j)
    if (j < nsteps) {
      ISSUE_B(j, j);
      if constexpr (!PATCH) ISSUE_A(j, j);
    }
  if constexpr (!PATCH) {
    WAIT_VM((NBUF - 2) * LOADS_PER_STEP);       // step 0 landed (NBUF = 3: step 1 may still fly)
    BLOCK_BARRIER();
  }

  int step = 0, slot = 0;
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    if constexpr (PATCH) {
      // every wave has passed the barrier that ended the previous chunk's last tap: the patch can be
      // overwritten.  Pieces past the end re-read the last valid piece (the LDS region is rounded up to
      // 4 KiB on the host), so no lane is ever masked.
      if (chunk > 0 && !(p.dbg & 2)) ISSUE_PATCH(chunk);
      WAIT_VM(0);
      BLOCK_BARRIER();
      LOAD_X(0, 0);
    }
    for (int tap = 0; tap < p.ntaps; ++tap, ++step) {
      // ---- issue the tiles of step + NBUF - 1 into the slot freed by the previous step
      const bool issued = step + NBUF - 1 < nsteps;
      if (issued && !(p.dbg & 1)) {
        int ns = slot + NBUF - 1; ns = ns >= NBUF ? ns - NBUF : ns;
        ISSUE_B(step + NBUF - 1, ns);
        if constexpr (!PATCH) ISSUE_A(step + NBUF - 1, ns);
      }
      // ---- all fragment reads of the step up front, then 2 x MT x NT MFMAs behind counted lgkmcnt waits
      if (!(p.dbg & 4)) {
        const char* const Bc = Bs + slot * B_BYTES + wbase;
        if (!(p.dbg & 16)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) wf[kk][ct] = *(const frag*)(Bc + ct * 2048 + (s0w ^ (kk * 64)));
        if constexpr (!PATCH) LOAD_X(0, slot);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int pt = 0; pt < MT; ++pt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[kk][ct], xf[kk][pt], acc[pt][ct]);
        __builtin_amdgcn_sched_barrier(0);
        // patch mode: the next tap's pixel fragments are already resident -- fetch them under the barrier
        if constexpr (PATCH) {
          if (tap + 1 < p.ntaps && !(p.dbg & 16)) LOAD_X(tap + 1, 0);
        }
      }
      // ---- next step's tiles landed (this wave) + everyone done with this step's slot
      if (NBUF == 2 || !issued) WAIT_VM(0);
      else WAIT_VM((NBUF - 2) * LOADS_PER_STEP);
      if (!(p.dbg & 8)) BLOCK_BARRIER();
      slot = slot + 1 == NBUF ? 0 : slot + 1;
    }
  }
#undef ISSUE_B
#undef ISSUE_A
#undef ISSUE_PATCH
#undef WAIT_VM
#undef BLOCK_BARRIER
#undef LOAD_X

  // ---- epilogue: + bias (+ residual) (ReLU) -> 16-bit padded NHWC
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = m0 + wpx * MT * 16 + pt * 16 + pcol;
    conv_epilogue_px<T, NT, false>(p, acc[pt], m, m < mend, cb, bias, HoWo);
  }
}

// ---------------------------------------------------------------------------
// host launcher: cfg 0 = BM 128 / BN 64, cfg 1 = BM 128 / BN 128, cfg 2 = BM 256 / BN 64
template <typename T, bool PATCH, int NBUF>
static hipError_t launch_cfg(const ConvP& p, int cfg, size_t lds, hipStream_t st) {
  const dim3 grid(p.mtiles * p.ntiles), block(256);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((conv_mfma_kernel<T, 128, 64, 4, 1, PATCH, NBUF>), grid, block, lds, st, p); break;
    case 1: hipLaunchKernelGGL((conv_mfma_kernel<T, 128, 128, 2, 2, PATCH, NBUF>), grid, block, lds, st, p); break;
    case 2: hipLaunchKernelGGL((conv_mfma_kernel<T, 256, 64, 4, 1, PATCH, NBUF>), grid, block, lds, st, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T, int BM, int BN, int WPX, int WCH, bool PATCH, int NBUF>
static hipError_t set_lds_attr(size_t bytes) {
  return hipFuncSetAttribute((const void*)conv_mfma_kernel<T, BM, BN, WPX, WCH, PATCH, NBUF>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int flope_conv_mfma_init() {
  const size_t big = 160 * 1024;
  hipError_t e = hipSuccess;
#define SET3(T, P, N)                                                        \
  if (e == hipSuccess) e = set_lds_attr<T, 128, 64, 4, 1, P, N>(big);        \
  if (e == hipSuccess) e = set_lds_attr<T, 128, 128, 2, 2, P, N>(big);       \
  if (e == hipSuccess) e = set_lds_attr<T, 256, 64, 4, 1, P, N>(big);
#define SET_ALL(T) SET3(T, true, 2) SET3(T, true, 3) SET3(T, false, 2) SET3(T, false, 3)
  SET_ALL(bf16_t)
  SET_ALL(f16_t)
#undef SET_ALL
#undef SET3
  return (int)e;
}

// dtype: 0 bf16, 1 f16.  patch: 0 gather, 1 patch.  nbuf: tile-ring depth (2 or 3).  Returns hipError_t as int.
extern "C" int flope_conv_mfma_launch(const ConvP* p, int dtype, int cfg, int patch, int nbuf, size_t lds,
                                      void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
#define GO(T)                                                                                        \
  (patch ? (nbuf == 3 ? launch_cfg<T, true, 3>(*p, cfg, lds, st) : launch_cfg<T, true, 2>(*p, cfg, lds, st)) \
         : (nbuf == 3 ? launch_cfg<T, false, 3>(*p, cfg, lds, st) : launch_cfg<T, false, 2>(*p, cfg, lds, st)))
  if (dtype == 0) e = GO(bf16_t);
  else e = GO(f16_t);
#undef GO
  return (int)e;
}
