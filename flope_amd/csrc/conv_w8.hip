// conv_w8: conv_w4's loop on EIGHT waves -- two per SIMD -- with ONE shared weight ring (r04; 3x3 / stride-1 implicit GEMM on flat
// 256 px x 128 ch tiles, layers 2-4 of the ResNet-18 trunk, reference sunflower/models/posenet.py:25).
//
// VERDICT r3 asked for the configuration in which a second wave per SIMD covers what a wave cannot hide from itself: two co-resident
// 4-wave workgroups on half-height tiles.  Two INDEPENDENT workgroups do not fit (two 20 KB patch buffers + a 3-deep ring each is 88 KB
// against the 80 KB half of a CU, and each would stream its own copy of the weight panel: twice the L2 -> LDS fill per multiply-add of a
// kernel that already fills at ~40 GB/s per CU).  This is the form that does fit: one workgroup of 512 threads, the LDS images, the
// packed weights, the ring, the waits and the one-barrier-per-double-step schedule of conv_w4 unchanged, but a wave owns 64 px x 64 ch
// (4 x 4 MFMA tiles, 64 accumulator registers, <= 256 registers per wave), so every SIMD holds two waves that run the SAME stream on
// different pixels -- free-running between barriers (unlike conv_stag's two groups, which hand the matrix pipe over behind two
// barriers per double step): when one wave of a SIMD blocks on an LDS-DMA issue behind a cold patch burst, waits for
// its fragments or sits at the barrier, the other one issues MFMAs.  Costs: 8 fragment reads per 16 MFMAs instead of 12 per 32 (+33 %
// LDS read bytes), eight waves per barrier.  Per wave and double step: 2 pieces of the double tile (a quarter of a KB each ... 1 KB
// per piece, 16 pieces per double tile), PT pieces per patch burst; the wait counts come from the same queue model (w4_sched.h, tgw = 2).
// One tile per workgroup; MFMA-order stores.  Every accumulator sees the same operands in the same order as in conv_w4 / conv_stag:
// bit-identical (tests/test_gpu_parity.py).
#include "common.h"
#include "w4_sched.h"
#ifndef FLOPE_W4_SPREAD
#define FLOPE_W4_SPREAD 2
#endif

#define GLDS16(gptr, lptr)                                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                      \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ int tile_px_w8(int c) { return c < 4 ? 2 * c : (c < 12 ? 2 * (c - 4) + 1 : 2 * (c - 8)); }

constexpr int kW8Scratch = 64 * 208 + 256;               // address-table exchange: [4 wpx][16 r16] rows of 208 bytes

// PT: 8 KB DMA rounds per patch buffer (4, 5 or 6).  RES: residual input.  DSF: folded 1x1 stride-2 shortcut (no residual).
template <typename T, int PT, bool RES, bool DSF>
__global__ __launch_bounds__(512, 2) void conv_w8_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int MT = 4, BM = 256, WPXB = 64, TILE_B = 128 * 64, DT_B = 2 * TILE_B, NT = 4;
  constexpr int NBD = w4_ring(PT, false);
  static_assert(NBD >= 3 && !(RES && DSF), "conv_w8 variants");
  constexpr int PATCH_B = PT * 8192;
  constexpr int PD = NBD - 1;
  constexpr int TGW = 2;                                   // LDS-DMA pieces per wave per double tile (16 KB / 8 waves / 1 KB)
  constexpr int PW = PT;                                   // ... per patch burst
  constexpr int SPREAD = FLOPE_W4_SPREAD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ps = smem;                                   // 2 patch buffers
  char* const Bs = smem + 2 * PATCH_B;                     // NBD double tiles of weights
  constexpr int DSW_B = (NBD - 1) * DT_B;                  // DSF: the shortcut's double tile in the ring's last look-ahead slot
  constexpr int SCR_B = 2 * PATCH_B + NBD * DT_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave & 3, wch = wave >> 2;
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px_w8(r16);
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lb % p.ntiles;
  const int HoWo = p.Ho * p.Wo;
  const int nhc = p.Cin / 32;
  const int NS = nhc * 9;
  const int nbody = nhc / 2, ND = nbody * 9;
  const size_t pixB = (size_t)p.Cin * 2;
  const size_t rowB = (size_t)p.Wip * pixB;
  const int pitch = p.Wip + 2;

  const int mt = lb / p.ntiles;
  const int m0 = mt * BM, mend = min(m0 + BM, p.M);
  int R0;
  const char* patch_src;
  {
    const int b0_ = fastdiv(m0, p.mg_hw, p.sh_hw), ho0_ = fastdiv(m0 - b0_ * HoWo, p.mg_w, p.sh_w);
    R0 = b0_ * p.Hip + ho0_;
    patch_src = (const char*)p.in + (size_t)R0 * rowB;
  }
  const char* const b_base = (const char*)p.w + (size_t)ntile * NS * TILE_B + wave * 1024 + lane * 16;
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wbase = 2 * PATCH_B + (wch * 64 + r16) * 64 + ((g ^ wsw) << 4);
  const int cb = ntile * 128 + wch * 64 + g * 8;
  const float* const bias_p = p.bias + cb;

#define W8_WAIT_VM(n_)                                                                                         \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory");                                                  \
  } while (0)
#define W8_BARRIER()                                                                                           \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define W8_ISSUE_DT(dt_, slotb_)                                                                               \
  do {                                                                                                         \
    _Pragma("unroll") for (int o = 0; o < TGW; ++o)                                                            \
      GLDS16(b_base + (size_t)(dt_) * DT_B + o * 8192, Bs + (slotb_) + o * 8192 + wave * 1024);                \
  } while (0)

  f32x4 b4[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) b4[ct] = *(const f32x4*)(bias_p + (ct >> 1) * 32 + (ct & 1) * 4);

  // ---- prologue (conv_w4.hip): weights first, the shortcut's first gather, the first patch, the tables while all of that flies
#pragma unroll
  for (int d = 0; d < PD; ++d) W8_ISSUE_DT(d, d * DT_B);

  const char* dsrc[DSF ? 2 : 1];
  const char* dw_base = nullptr;
#define W8_ISSUE_DS(j_)                                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                              \
      _Pragma("unroll") for (int rr = 0; rr < 2; ++rr)                                                         \
        GLDS16(dsrc[rr] + (2 * (j_) + t) * 64, Ps + PATCH_B + t * 16384 + (rr * 512 + wave * 64) * 16);        \
    _Pragma("unroll") for (int o = 0; o < TGW; ++o)                                                            \
      GLDS16(dw_base + (size_t)(j_) * DT_B + o * 8192, Bs + DSW_B + o * 8192 + wave * 1024);                   \
  } while (0)
  if constexpr (DSF) {
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
    dw_base = (const char*)p.ds_w + (size_t)ntile * (p.ds_Cin / 32) * TILE_B + wave * 1024 + lane * 16;
    W8_ISSUE_DS(0);
  }

  // per-lane DMA source offsets of a patch burst: piece j of this wave = 16-byte piece j * 512 + wave * 64 + lane of the image
  unsigned psrc[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int q = j * 512 + wave * 64 + lane;
    const int pi = q >> 2;
    const int pr_ = fastdiv(pi, p.mg_pitch, p.sh_pitch), pc_ = pi - __mul24(pr_, pitch);
    const int js = (q & 3) ^ (((__mul24(pr_, p.Wo) + pc_) >> 2) & 3);
    psrc[j] = (unsigned)(__mul24(__mul24(pr_, p.Wip) + min(pc_, p.Wip - 1), (int)pixB) + js * 16);
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) GLDS16(patch_src + psrc[j], Ps + (j * 512 + wave * 64) * 16);

  // fragment address table (conv_w4.hip): the lanes g = 0..3 of a pixel column (wpx, r16) compute one pixel tile each (both channel
  // halves compute the same values; the wch = 0 waves publish them)
  int xoff[9][MT];
  unsigned ooff[MT];
  bool ok[MT];
  {
    char* const scr = smem + SCR_B + (wpx * 16 + r16) * 208;
    const int mm = m0 + wpx * WPXB + g * 16 + pcol;
    const int m_ = min(mm, mend - 1);
    const int b_ = fastdiv(m_, p.mg_hw, p.sh_hw), r_ = m_ - __mul24(b_, HoWo);
    const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - __mul24(ho_, p.Wo);
    const int i_ = __mul24(b_, p.Hip) + ho_ - R0;
    const int pb = (__mul24(i_, pitch) + wo_) << 6, vb = (__mul24(i_, p.Wo) + wo_) << 2;
    u32x4 e0, e1, e2;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const unsigned v = (unsigned)(pb + (((t / 3) * pitch + (t % 3)) << 6) + ((vb + (((t / 3) * p.Wo + (t % 3)) << 2)) & 0x30));
      if (t < 4) e0[t] = v; else if (t < 8) e1[t - 4] = v; else e2[0] = v;
    }
    e2[1] = (unsigned)(__mul24(__mul24(__mul24(b_, p.Hop) + ho_ + 1, p.Wop) + wo_ + 1, p.Cout) * 2);
    e2[2] = 0u; e2[3] = 0u;
    if (wch == 0) { *(u32x4*)(scr + g * 48) = e0; *(u32x4*)(scr + g * 48 + 16) = e1; *(u32x4*)(scr + g * 48 + 32) = e2; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W8_BARRIER();
    const unsigned g4 = (unsigned)(g << 4);
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const u32x4 a0 = *(const u32x4*)(scr + pt * 48), a1 = *(const u32x4*)(scr + pt * 48 + 16);
      const u32x2 a2 = *(const u32x2*)(scr + pt * 48 + 32);
#pragma unroll
      for (int t = 0; t < 4; ++t) { xoff[t][pt] = (int)(a0[t] ^ g4); xoff[4 + t][pt] = (int)(a1[t] ^ g4); }
      xoff[8][pt] = (int)(a2[0] ^ g4);
      ooff[pt] = a2[1] + (unsigned)(cb * 2);
      ok[pt] = m0 + wpx * WPXB + pt * 16 + pcol < mend;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) asm volatile("" ::"v"(xoff[t][pt]));
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) acc[pt][ct] = b4[ct];
  frag wf[2][NT], xf[2][MT];

  W8_WAIT_VM(0);
  W8_BARRIER();

  if constexpr (DSF) {
    int xds[MT];
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int sl = wpx * WPXB + pt * 16 + pcol;
      xds[pt] = PATCH_B + (sl << 6) + ((g ^ ((sl >> 2) & 3)) << 4);
    }
    for (int j = 0; j < p.ds_Cin / 64; ++j) {
      if (j > 0) {
        W8_ISSUE_DS(j);
        W8_WAIT_VM(0);
        W8_BARRIER();
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) wf[h][ct] = *(const frag*)(smem + wbase + DSW_B + h * TILE_B + ct * 1024);
#pragma unroll
        for (int pt = 0; pt < MT; ++pt) xf[h][pt] = *(const frag*)(smem + xds[pt] + h * 16384);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pt = 0; pt < MT; ++pt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[h][ct], xf[h][pt], acc[pt][ct]);
      W8_BARRIER();
    }
  }

#pragma unroll
  for (int ct = 0; ct < NT; ++ct) wf[0][ct] = *(const frag*)(smem + wbase + ct * 1024);
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) xf[0][pt] = *(const frag*)(smem + xoff[0][pt]);

  // One sub-step = 16 MFMAs of fragment set C_ in four groups of 4; the 8 fragment reads of the next sub-step (set N_) ride in the
  // groups' gaps (weights in groups 0, 1; pixels in 2, 3), the LDS-DMA pieces of a second sub-step KV_ per group.
#define W8_VMEM_GROUP(P_, NV_, KV_)                                                                            \
  do {                                                                                                         \
    if constexpr ((KV_) > 0 && (P_) * (KV_) < (NV_))                                                           \
      __builtin_amdgcn_sched_group_barrier(0x020, (((P_) + 1) * (KV_) <= (NV_) ? (KV_) : (NV_) - (P_) * (KV_)), 0); \
  } while (0)
#define W8_GRP(P_, C_, N_, wo_, nb_, nt_, NV_, KV_)                                                            \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                          \
      acc[P_][ct] = Elem<T>::mfma(wf[C_][ct], xf[C_][P_], acc[P_][ct]);                                        \
    if constexpr ((P_) < 2) {                                                                                  \
      wf[N_][2 * (P_)] = *(const frag*)(smem + (wo_) + (2 * (P_)) * 1024);                                     \
      wf[N_][2 * (P_) + 1] = *(const frag*)(smem + (wo_) + (2 * (P_) + 1) * 1024);                             \
    } else {                                                                                                   \
      xf[N_][2 * ((P_) - 2)] = *(const frag*)(smem + xoff[nt_][2 * ((P_) - 2)] + (nb_) * PATCH_B);             \
      xf[N_][2 * ((P_) - 2) + 1] = *(const frag*)(smem + xoff[nt_][2 * ((P_) - 2) + 1] + (nb_) * PATCH_B);     \
    }                                                                                                          \
    if constexpr ((KV_) > 0 && (P_) * (KV_) + 0 < (NV_)) W8_DMA_PIECE(((P_) * (KV_) + 0));                     \
    if constexpr ((KV_) > 1 && (P_) * (KV_) + 1 < (NV_)) W8_DMA_PIECE(((P_) * (KV_) + 1));                     \
    if constexpr ((KV_) > 2 && (P_) * (KV_) + 2 < (NV_)) W8_DMA_PIECE(((P_) * (KV_) + 2));                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                         \
    W8_VMEM_GROUP(P_, NV_, KV_);                                                                               \
  } while (0)
#define W8_SUB(C_, N_, wo_, nb_, nt_, NV_, KV_)                                                                \
  do {                                                                                                         \
    W8_GRP(0, C_, N_, wo_, nb_, nt_, NV_, KV_); W8_GRP(1, C_, N_, wo_, nb_, nt_, NV_, KV_);                    \
    W8_GRP(2, C_, N_, wo_, nb_, nt_, NV_, KV_); W8_GRP(3, C_, N_, wo_, nb_, nt_, NV_, KV_);                    \
  } while (0)

  int dn = PD;
  int slot_b = 0;
  int hc = 0;
  int res_slot[2] = {0, 0};
  const char* const res_b = (const char*)p.res;

  // DMA piece i_ of the second sub-step of double step D_: pieces 0 .. TGW - 1 = double tile D + PD; then the patch pieces of that
  // double step (w4_sched.h).  RES, last body: the residual (8 pieces of 16 bytes per lane: pixel tile r >> 1, half r & 1) rides in
  // the slots the look-ahead leaves unused -- r = 0..3 in the first four pieces of the next tile's patch burst (buffer 0), r = 4, 5 in
  // the double tile of D = 7, r = 6, 7 in that of D = 8.
#define W8_DMA_PIECE(i_)                                                                                       \
  do {                                                                                                         \
    if constexpr ((i_) < TGW) {                                                                                \
      if (RES && lastb_ && (D_ == 7 || D_ == 8)) {                                                             \
        constexpr int rr_ = (D_ == 7 ? 4 : 6) + (i_);                                                          \
        unsigned ro_ = ooff[rr_ >> 1]; asm volatile("" : "+v"(ro_));                                           \
        GLDS16(res_b + ro_ + (rr_ & 1) * 64, Bs + iss_b_ + (i_) * 8192 + wave * 1024);                         \
      } else {                                                                                                 \
        GLDS16(b_base + (size_t)di_ * DT_B + (i_) * 8192, Bs + iss_b_ + (i_) * 8192 + wave * 1024);            \
      }                                                                                                        \
    } else {                                                                                                   \
      constexpr int jr_ = (i_) - TGW + w4_patch_first(D_, PW, SPREAD);                                         \
      constexpr int j_ = jr_ < 0 ? 0 : (jr_ >= PW ? PW - 1 : jr_);                                             \
      if (D_ == 0) {                                                                                           \
        GLDS16(patch_src + (hc + 1) * 64 + psrc[j_], Ps + PATCH_B + (j_ * 512 + wave * 64) * 16);              \
      } else if (RES && lastb_ && j_ < 4) {                                                                    \
        unsigned ro_ = ooff[j_ >> 1]; asm volatile("" : "+v"(ro_));                                            \
        GLDS16(res_b + ro_ + (j_ & 1) * 64, Ps + (j_ * 512 + wave * 64) * 16);                                 \
      } else {                                                                                                 \
        GLDS16(patch_src + (hc + 2 < nhc ? (hc + 2) * 64 : 0) + psrc[j_], Ps + (j_ * 512 + wave * 64) * 16);   \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
#define W8_DSTEP(D)                                                                                            \
  do {                                                                                                         \
    constexpr int D_ = (D), U1_ = 2 * (D) + 1, U2_ = (2 * (D) + 2) % 18, RS_ = (D) == 8 ? 1 : 0;               \
    constexpr int WN_ = w4_wait_n(D_, PD, PW, SPREAD, 0, 0, TGW);                                              \
    constexpr int NV_ = w4_pieces(D_, PW, SPREAD, TGW), KV_ = (NV_ + MT - 1) / MT;                             \
    const bool lastb_ = hc + 2 >= nhc;                                                                         \
    const int next_b_ = slot_b + DT_B >= NBD * DT_B ? 0 : slot_b + DT_B;                                       \
    int wof_ = wbase + slot_b, wofn_ = wbase + next_b_;                                                        \
    asm volatile("" : "+v"(wof_), "+v"(wofn_));                                                                \
    int iss_b_ = slot_b + PD * DT_B; if (iss_b_ >= NBD * DT_B) iss_b_ -= NBD * DT_B;                           \
    const int di_ = dn < ND ? dn : dn - ND;                                                                    \
    W8_SUB(0, 1, wof_ + TILE_B, U1_ / 9, U1_ % 9, 0, 0);                                                       \
    W8_WAIT_VM(WN_);                                                                                           \
    W8_BARRIER();                                                                                              \
    if (RES && lastb_ && ((D) == 7 || (D) == 8)) res_slot[RS_] = iss_b_;                                       \
    ++dn;                                                                                                      \
    W8_SUB(1, 0, wofn_, U2_ / 9, U2_ % 9, NV_, KV_);                                                           \
    slot_b = next_b_;                                                                                          \
  } while (0)

  for (int hcp = 0; hcp < nbody; ++hcp) {
    W8_DSTEP(0); W8_DSTEP(1); W8_DSTEP(2); W8_DSTEP(3); W8_DSTEP(4); W8_DSTEP(5); W8_DSTEP(6); W8_DSTEP(7); W8_DSTEP(8);
    hc += 2;
  }

  // ---- epilogue: (+ residual) (ReLU) -> 16-bit padded NHWC straight from the accumulators
  W8_WAIT_VM(0);
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    float v[NT * 4];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[ct * 4 + q] = acc[pt][ct][q];
    if constexpr (RES) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int r = pt * 2 + c;                          // this lane's own DMA pieces (no other wave reads them)
        const int off_ = r < 4 ? (r * 512 + wave * 64 + lane) * 16
                               : 2 * PATCH_B + res_slot[(r - 4) >> 1] + ((r - 4) & 1) * 8192 + wave * 1024 + lane * 16;
        const u32x4 rv = *(const u32x4*)(smem + off_);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[c * 8 + q * 2] += unpack_lo<T>(rv[q]);
          v[c * 8 + q * 2 + 1] += unpack_hi<T>(rv[q]);
        }
      }
    }
    if (ok[pt]) {
      char* op = (char*)p.out + ooff[pt];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pk_out16<T>(pack2<T>(v[c * 8 + q * 2], v[c * 8 + q * 2 + 1]), p.relu);
        *(u32x4*)(op + c * 64) = o;
      }
    }
  }
#undef W8_DSTEP
#undef W8_DMA_PIECE
#undef W8_SUB
#undef W8_GRP
#undef W8_VMEM_GROUP
#undef W8_ISSUE_DS
#undef W8_ISSUE_DT
#undef W8_BARRIER
#undef W8_WAIT_VM
}

static constexpr size_t w8_lds_bytes(int pt) { return (size_t)2 * pt * 8192 + (size_t)w4_ring(pt, false) * 16384 + kW8Scratch; }

#define W8_FOR_VARIANTS(X, T)                                                                                  \
  X(T, 4, false, false) X(T, 4, true, false) X(T, 4, false, true)                                              \
  X(T, 5, false, false) X(T, 5, true, false) X(T, 5, false, true)                                              \
  X(T, 6, false, false) X(T, 6, true, false) X(T, 6, false, true)

extern "C" int flope_conv_w8_init() {
  hipError_t e = hipSuccess;
#define A(T, PT_, RES_, DSF_)                                                                                  \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_w8_kernel<T, PT_, RES_, DSF_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  W8_FOR_VARIANTS(A, bf16_t)
  W8_FOR_VARIANTS(A, f16_t)
#undef A
  return (int)e;
}

extern "C" size_t flope_conv_w8_lds(int pt) { return (pt < 4 || pt > 6) ? 0 : w8_lds_bytes(pt); }

// as flope_conv_w4_launch with mt = 8 (256-pixel tiles), one tile per workgroup, 512 threads
extern "C" int flope_conv_w8_launch(const ConvP* p, int dtype, void* stream) {
  const int pt = p->patch_rows_max;
  if (p->mtiles != (p->M + 255) / 256 || p->stride != 1 || p->ntaps != 9 || p->Cin % 64 || p->Cout % 128 || !p->skew || !p->mg_pitch ||
      p->ksplit > 1 || (p->res && p->ds_in) || (p->ds_in && (p->ds_Cin % 64 || !p->ds_w)) || p->total_tiles != p->mtiles * p->ntiles)
    return (int)hipErrorInvalidValue;
  const size_t lds = flope_conv_w8_lds(pt);
  if (lds == 0 || lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(p->total_tiles), block(512);
  const int res = p->res ? 1 : 0, dsf = p->ds_in ? 1 : 0;
  bool done = false;
#define L(T, PT_, RES_, DSF_)                                                                                  \
  if (!done && pt == PT_ && res == (RES_ ? 1 : 0) && dsf == (DSF_ ? 1 : 0)) {                                  \
    hipLaunchKernelGGL((conv_w8_kernel<T, PT_, RES_, DSF_>), grid, block, lds, st, *p); done = true;           \
  }
  if (dtype == 0) { W8_FOR_VARIANTS(L, bf16_t) } else { W8_FOR_VARIANTS(L, f16_t) }
#undef L
  if (!done) return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
