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
// flow beyond the group test -- r01's first version of this kernel spent ~0.5 us per phase in scalar
// bookkeeping (ring slot, tap decode, wait-count switch), more than the 16 MFMAs it was wrapped around.
template <typename T, int PT>
__global__ __launch_bounds__(512, 2) void conv_stag_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int BM = 256, BN = 128, NB = 6, TILE_B = BN * 64;     // 8 KB weight tile per step
  constexpr int MT = 4, NT = 4;
  constexpr int PATCH_B = PT * 8192;                              // bytes of one patch buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ps = smem;                                   // 2 patch buffers (first: their offsets stay ds_read immediates)
  char* const Bs = smem + 2 * PATCH_B;                     // NB x 8 KB weight ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2, wl = wave & 3, wpx = wl & 1, wch = wl >> 1;
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px_s(r16);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lid % p.ntiles, mtile = lid / p.ntiles;
  const int HoWo = p.Ho * p.Wo;
  const int nhc = p.Cin / 32;                              // even: Cin is a multiple of 64
  const int NS = nhc * 9;
  const int m0 = mtile * BM, mend = min(m0 + BM, p.M);

  // ---- patch geometry (same as conv_mfma patch mode, 64-byte pixels)
  const int b0 = m0 / HoWo, ho0 = (m0 - b0 * HoWo) / p.Wo;
  const int ml = mend - 1, b1 = ml / HoWo, ho1 = (ml - b1 * HoWo) / p.Wo;
  const int R0 = b0 * p.Hip + ho0, R1 = b1 * p.Hip + ho1 + 2;
  const char* const patch_src = (const char*)p.in + (size_t)R0 * p.Wip * p.Cin * 2;
  const int patch_pieces = (R1 - R0 + 1) * p.Wip * 4;      // 16-byte pieces of one half-chunk patch (<= PT*512)
  const size_t pixB = (size_t)p.Cin * 2;
  int pi0[MT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = min(m0 + group * 128 + wpx * 64 + pt * 16 + pcol, mend - 1);
    const int b = m / HoWo, r = m - b * HoWo, ho = r / p.Wo, wo = r - ho * p.Wo;
    pi0[pt] = (b * p.Hip + ho - R0) * p.Wip + wo;
  }
  // Everything a step needs is precomputed so the loop body is LDS reads, MFMAs, one DMA issue and
  // waits only (the r01 counters showed ~2.8 address VALU ops per MFMA otherwise -- the load half of a
  // phase then outlasts the 16-MFMA half it is supposed to hide under):
  //   psrc[rr]   per-lane 32-bit source offsets of the PT patch rounds (constant over half-chunks)
  //   xoff[t][pt] LDS byte offset of this lane's pixel fragment for tap t (buffer / ring slot are immediates)
  unsigned psrc[PT];
#pragma unroll
  for (int rr = 0; rr < PT; ++rr) {
    const int q = min(rr * 512 + wave * 64 + lane, patch_pieces - 1);
    const int pi = q >> 2, js = (q & 3) ^ ((pi >> 2) & 3);
    psrc[rr] = (unsigned)(pi * (int)pixB + js * 16);
  }
  int xoff[9][MT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int pi = pi0[pt] + (t / 3) * p.Wip + (t % 3);
      xoff[t][pt] = (pi << 6) + ((g ^ ((pi >> 2) & 3)) << 4);
    }
  // weight tiles: [ntile][step][128 rows][32 k]; wave w moves 1 KiB piece w of every tile:
  // uniform base (SGPR pair) + per-lane 32-bit offset -> no address VALU at issue time
  const char* const b_base = (const char*)p.w + (size_t)ntile * NS * TILE_B + wave * 1024;
  const unsigned lane16 = lane * 16;
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wbase = 2 * PATCH_B + (wch * 64 + r16) * 64 + ((g ^ wsw) << 4);

#define ISSUE_PATCH(hc_, buf_)                                                                                 \
  do {                                                                                                         \
    const char* src_ = patch_src + (hc_) * 64;          /* wave-uniform */                                      \
    _Pragma("unroll") for (int rr = 0; rr < PT; ++rr)                                                          \
      GLDS16(src_ + psrc[rr], Ps + (buf_) * PATCH_B + (rr * 512 + wave * 64) * 16);                            \
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

  frag wf[NT], xf[MT];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads of one step: ring slot / patch buffer are literals, the tap shift a scalar
#define LOADF(slot_, buf_, tap_)                                                                               \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                          \
      wf[ct] = *(const frag*)(smem + wbase + (slot_) * TILE_B + ct * 1024);                                    \
    _Pragma("unroll") for (int pt = 0; pt < MT; ++pt)                                                          \
      xf[pt] = *(const frag*)(smem + xoff[tap_][pt] + (buf_) * PATCH_B);                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
  } while (0)
#define MFMAS()                                                                                                \
  do {                                                                                                         \
    _Pragma("unroll") for (int pt = 0; pt < MT; ++pt)                                                          \
      _Pragma("unroll") for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], acc[pt][ct]); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  // ---- prologue: patch of half-chunk 0, tiles 0..NB-1 (NS >= 18 > NB always)
  ISSUE_PATCH(0, 0);
#pragma unroll
  for (int t = 0; t < NB - 1; ++t) GLDS16(b_base + (size_t)t * TILE_B + lane16, Bs + t * TILE_B + wave * 1024);
  WAIT_VM(NB - 2);                             // patch 0 and tile 0 landed (tiles 1..NB-2 may fly)
  BARRIER();
  if (group == 1) BARRIER();                   // group B runs one phase behind group A

  // Every wave executes the SAME stream per step j = 18*hcp + U (no group tests inside the loop):
  //     issue tile j+NB-1 into the ring slot step j-1 just released (tile t lives in slot t % NB; clamped to
  //         the last tile: a harmless re-load keeps the op count per step constant); at tap 0 also the next
  //         half-chunk's patch (clamped likewise)
  //     LOADF(j)  ->  wait for this wave's piece of tile j+1  ->  barrier  ->  16 MFMAs  ->  barrier
  // and group B is one barrier behind group A, so in every physical phase one group is in its MFMA half
  // while the other is in its load half.  Hazards (phi = physical phase; A: L(j) at 2j-1, M(j) at 2j;
  // B: L(j) at 2j, M(j) at 2j+1): slot of step j-1 is last read by B's L(j-1) at phi 2j-2, refilled at
  // phi >= 2j-1; tile j is waited for by A after L(j-1) (phi 2j-3) and by B after L(j-1) (phi 2j-2), both
  // before the barrier that precedes A's L(j) at phi 2j-1.
  // Younger VM ops than tile j+1 at the wait: NB-2 tiles, plus the PT patch rounds when this half-chunk's
  // burst (issued at its tap 0) is younger than tile j+1, i.e. for taps 0..NB-2.
  int jn = NB - 1;                                          // tile to issue at the start of the next step
  int hc = 0;
#define STEP(U)                                                                                           \
  do {                                                                                                         \
    constexpr int TAP_ = (U) % 9, BUF_ = (U) / 9, SLOT_ = (U) % NB, PSLOT_ = ((U) + NB - 1) % NB;              \
    constexpr int WN_ = NB - 2 + (TAP_ <= NB - 2 ? PT : 0);                                                    \
    {                                                                                                          \
      const int ti_ = jn < NS - 1 ? jn : NS - 1;                                                               \
      GLDS16(b_base + (size_t)ti_ * TILE_B + lane16, Bs + PSLOT_ * TILE_B + wave * 1024);                      \
      ++jn;                                                                                                    \
    }                                                                                                          \
    if (TAP_ == 0) {                                                                                           \
      const int nh_ = hc + 1 < nhc ? hc + 1 : hc;                                                              \
      ISSUE_PATCH(nh_, BUF_ ^ 1);                                                                              \
    }                                                                                                          \
    if (TAP_ == 8) ++hc;                                                                                       \
    LOADF(SLOT_, BUF_, TAP_);                                                                                  \
    WAIT_VM(WN_);                                                                                              \
    BARRIER();                                                                                                 \
    MFMAS();                                                                                                   \
    BARRIER();                                                                                                 \
  } while (0)
  for (int hcp = 0; hcp < nhc / 2; ++hcp) {
    STEP(0);  STEP(1);  STEP(2);  STEP(3);  STEP(4);  STEP(5);  STEP(6);  STEP(7);  STEP(8);
    STEP(9);  STEP(10); STEP(11); STEP(12); STEP(13); STEP(14); STEP(15); STEP(16); STEP(17);
  }
  if (group == 0) BARRIER();                                // every wave executes the same number of barriers
  WAIT_VM(0);                                               // drain the clamped tail re-loads before LDS is released
#undef STEP
#undef ISSUE_PATCH
#undef WAIT_VM
#undef BARRIER
#undef LOADF
#undef MFMAS

  // ---- epilogue: + bias (+ residual) (ReLU) -> 16-bit padded NHWC
  const int cb = ntile * BN + wch * 64 + g * 16;
  float bias[NT * 4];
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) bias[i] = p.bias[cb + i];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = m0 + group * 128 + wpx * 64 + pt * 16 + pcol;
    conv_epilogue_px<T, NT>(p, acc[pt], m, m < mend, cb, bias, HoWo);
  }
}

template <typename T>
static hipError_t stag_attr() {
  hipError_t e = hipSuccess;
#define A(PT_) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_stag_kernel<T, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  A(2) A(4) A(6) A(8)
#undef A
  return e;
}

extern "C" int flope_conv_stag_init() {
  hipError_t e = stag_attr<bf16_t>();
  if (e == hipSuccess) e = stag_attr<f16_t>();
  return (int)e;
}

template <typename T>
static void stag_launch(const ConvP& p, int pt, size_t lds, hipStream_t st) {
  const dim3 grid(p.mtiles * p.ntiles), block(512);
  switch (pt) {
    case 2: hipLaunchKernelGGL((conv_stag_kernel<T, 2>), grid, block, lds, st, p); break;
    case 4: hipLaunchKernelGGL((conv_stag_kernel<T, 4>), grid, block, lds, st, p); break;
    case 6: hipLaunchKernelGGL((conv_stag_kernel<T, 6>), grid, block, lds, st, p); break;
    default: hipLaunchKernelGGL((conv_stag_kernel<T, 8>), grid, block, lds, st, p); break;
  }
}

// p->patch_rows_max carries PT (2, 4, 6 or 8 DMA rounds per patch buffer); lds = 6*8 KiB + 2*PT*8 KiB
extern "C" int flope_conv_stag_launch(const ConvP* p, int dtype, size_t lds, void* stream) {
  if (dtype == 0) stag_launch<bf16_t>(*p, p->patch_rows_max, lds, (hipStream_t)stream);
  else            stag_launch<f16_t>(*p, p->patch_rows_max, lds, (hipStream_t)stream);
  return (int)hipGetLastError();
}
