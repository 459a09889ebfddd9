// conv_r4: layer 1 of the ResNet-18 trunk (3x3 stride-1 64 -> 64 on the 56 x 56 map; reference sunflower/models/posenet.py:25) as
// persistent 8-row bands on FOUR waves, one per SIMD -- the row-band counterpart of conv_w4.hip (r03).
//
// conv_stag's row-band mode (8 waves, two staggered groups) spends 2,100-2,400 cycles per double step at 2.3 GHz against a
// 1,024-cycle MFMA floor (profiles/r03_inkernel_clock_*.txt, r03_phase_stamps_conv_stag.txt: two barriers per double step, a load
// half as long as the MFMA half, a stall at every patch hand-over), and computes 64 columns of a 56-column map (12.5 % of its MFMAs
// are dropped).  Here:
//   * a wave owns TWO full output rows = 112 pixels = 7 MFMA pixel tiles x 64 channels (112 accumulator registers): no padded
//     columns, 11 fragment reads per 28 MFMAs;
//   * the wave issues the ds_read_b128s of sub-step u + 1 and its LDS-DMA pieces in the gaps of the 28 MFMAs of sub-step u
//     (second fragment register set);
//   * the 72 KB weight panel is resident in LDS; the two patch buffers (one per 32-channel half-chunk, 10 input rows each, the
//     conflict-free image of conv_stag r03: row pitch W + 4) are refilled NINE sub-steps before their first reader:
//       sub-step 0  : issue this tile's second half-chunk            -> buffer 1   (free since the previous tile's sub-step 17)
//       before 8    : wait vmcnt(0) [RES: vmcnt(2 MT)], barrier       (buffer 1 visible; every wave is done with buffer 0)
//       sub-step 8  : issue the NEXT tile's first half-chunk          -> buffer 0
//       before 17   : wait vmcnt(0), barrier                          (buffer 0 visible; every wave is done with buffer 1)
//     -- two barriers per tile (504 MFMAs per wave), every DMA / store / residual load has >= 8 sub-steps (~2 us) to complete,
//     the waits are plain vmcnt(0) except that the residual loads (RES: 2 per pixel tile, issued in sub-steps 1..7 straight into
//     registers, added in the epilogue) stay in flight across the first barrier;
// Every tile has the same geometry relative to its patch origin: all per-lane tables are built once per workgroup.
#include "common.h"

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ int tile_px_r4(int c) { return c < 4 ? 2 * c : (c < 12 ? 2 * (c - 4) + 1 : 2 * (c - 8)); }

// MT = 2 * Wo / 16 pixel tiles per wave (7 for the 56-wide map).  PT = 8 KB DMA rounds per patch buffer.
template <typename T, int MT, int PT, bool RES>
__global__ __launch_bounds__(256, 1) void conv_r4_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int NT = 4, TILE_B = 64 * 64, NSTEP = 18;
  constexpr int PATCH_B = PT * 8192, PW = 2 * PT;
  constexpr int NR = 4 + MT;                               // fragment reads per sub-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ps = smem;                                   // 2 patch buffers
  char* const Ws = smem + 2 * PATCH_B;                     // resident weights: 18 x [64 rows][32 k]
  // r04: a wave-private 2 KB LINE IMAGE per wave behind the weights.  A lane's accumulators are one pixel x two runs of 8 channels
  // (MFMA order), so a wave-store / residual load touched 16 pixels x 64 bytes; through the image (XOR-swizzled, conflict-free both
  // ways, LDS operations of a wave execute in order: no barrier) the 16-bit outputs leave as 8 whole
  // 128-byte lines per instruction (conv_w4.hip; tools/probes/store_probe.hip: 4.2 k -> 2.9 k cycles per 64 KB).  Same bytes.

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px_r4(r16);
  const int Wo = p.Wo, pitch = p.Wip + 2;
  const size_t pixB = (size_t)p.Cin * 2;                   // 128
  const size_t rowB = (size_t)p.Wip * pixB;
  const int cb = g * 8;                                    // this lane's channels: cb .. cb + 7 and cb + 32 .. cb + 39
  char* const lscr = smem + 2 * PATCH_B + NSTEP * TILE_B + wave * 2048;   // this wave's line image

#define R4_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define R4_BARRIER()                                                                                           \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  // ---- once per workgroup: the weight panel, and the per-lane tables
  {
    const char* wsrc = (const char*)p.w + wave * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) GLDS16(wsrc + i * 4096, Ws + i * 4096 + wave * 1024);
  }
  unsigned psrc[PW];                                       // op j of a patch burst moves pieces j * 256 + wave * 64 + lane
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int q = j * 256 + wave * 64 + lane;
    const int pi = min(q >> 2, 10 * pitch - 1);            // pieces past the 10 patch rows re-read the last pixel (never read back)
    const int pr_ = fastdiv(pi, p.mg_pitch, p.sh_pitch), pc_ = pi - __mul24(pr_, pitch);
    const int js = (q & 3) ^ (((__mul24(pr_, Wo) + pc_) >> 2) & 3);
    psrc[j] = (unsigned)(__mul24(__mul24(pr_, p.Wip) + min(pc_, p.Wip - 1), (int)pixB) + js * 16);
  }
  // LDS byte offset of this lane's pixel fragment (pixel tile pt, tap t).  Pixel tiles that lie in one row are 16 pixels = 1,024
  // bytes apart with the same swizzle term (16 pixels move v by 64, outside bits 4..5 of v << 2), so with 56-pixel rows only tiles
  // 0, 3 (the one that wraps into the second row) and 4 need registers: tiles 1, 2 = tile 0 + 1 / 2 KB, tiles 5, 6 = tile 4 + 1 / 2 KB
  // as ds_read immediates -- 27 address registers instead of 63 (which had pushed the residual registers into AGPR copies)
  static_assert(MT == 7, "conv_r4: the pixel-tile -> address-register map below is the one of 56-pixel rows");
  constexpr int kXB[7] = {0, 0, 0, 1, 2, 2, 2}, kXA[7] = {0, 1024, 2048, 0, 0, 1024, 2048}, kXP[3] = {0, 3, 4};
  int xoff[9][3];
  // byte offsets relative to the band's first padded output row, in LINE order: row pp = c * 8 + (lane >> 3) of pixel tile pt's line
  // image is written by the lanes with r16 = pp, i.e. holds pixel pt * 16 + tile_px_r4(pp); this lane moves its 16-byte chunk lane & 7
  unsigned osto[MT][2];
  unsigned oconst[RES ? MT : 1];                           // RES: this lane's own pixel of tile pt (MFMA order): where its residual lives
  const int g4 = g << 4;
  if constexpr (RES) {
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int k = pt * 16 + pcol;
      const int row = k >= Wo ? 1 : 0, col = k - row * Wo;
      oconst[pt] = (unsigned)((((2 * wave + row + 1) * p.Wop + col + 1) * p.Cout + cb) * 2);
    }
  }
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = pt * 16 + tile_px_r4(c * 8 + (lane >> 3));   // pixel index inside the wave's two rows
      const int row = k >= Wo ? 1 : 0, col = k - row * Wo;
      const int i_ = 2 * wave + row;                       // output row inside the band = patch row of tap dy = 0
      osto[pt][c] = (unsigned)((((i_ + 1) * p.Wop + col + 1) * p.Cout) * 2 + (lane & 7) * 16);
    }
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int k = kXP[b] * 16 + pcol;
    const int row = k >= Wo ? 1 : 0, col = k - row * Wo;
    const int i_ = 2 * wave + row;
    const int pb = (i_ * pitch + col) << 6, vb = (i_ * Wo + col) << 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      xoff[t][b] = pb + ((((t / 3) * pitch + (t % 3)) << 6)) + (((vb + (((t / 3) * Wo + (t % 3)) << 2)) ^ g4) & 0x30);
  }
#define R4_XO(t_, pt_) (xoff[t_][kXB[pt_]] + kXA[pt_])
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wbase = 2 * PATCH_B + r16 * 64 + ((g ^ wsw) << 4);
  f32x4 b4[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) b4[ct] = *(const f32x4*)(p.bias + cb + (ct >> 1) * 32 + (ct & 1) * 4);

  // ---- tile walk: tile = image * tiles_per_image + band
  // r04: workgroup b runs on XCD b % 8 (round-robin dispatch).  Each XCD owns one CONTIGUOUS eighth of the tiles and its G / 8
  // workgroups walk it side by side (j, j + G / 8, ...), so the bands of an image -- which share two of their ten patch rows with
  // each neighbour -- meet in one L2 (with tile = blockIdx.x + k G neighbours sat on different XCDs and every halo row came from the
  // fabric twice: 1.31 - 1.33 x the algorithmic bytes, r03 counters), and the last, partly filled round is spread over all eight
  // XCDs (xcd_remap's order left it on four of them: -1.3 % on the two-slice step).  Time: unchanged within noise (the launch is
  // bound by the bytes it has to move), fabric traffic: down.
#ifdef FLOPE_STAG_DBG
  const unsigned long long rt_entry_ = __builtin_amdgcn_s_memrealtime();      // r05: finish-time spread of the workgroups of a launch
#endif
  const int G = gridDim.x;
  int tile = blockIdx.x, tstep = G, tend = p.total_tiles;
  if ((G & 7) == 0) {
    const int x = blockIdx.x & 7;
    tile = (int)(((long)p.total_tiles * x) >> 3) + (blockIdx.x >> 3);
    tend = (int)(((long)p.total_tiles * (x + 1)) >> 3);
    tstep = G >> 3;
  }
  auto tile_rows = [&](int tl, int& R0, int& m0) {
    const int b0 = tl / p.tiles_per_image, j0 = tl - b0 * p.tiles_per_image;
    R0 = b0 * p.Hip + j0 * 8;                              // first padded input row of the patch
    m0 = b0 * p.Hop + j0 * 8;                              // padded output row above the band's first row
  };
  int R0, m0;
  tile_rows(tile, R0, m0);
  const char* patch_src = (const char*)p.in + (size_t)R0 * rowB;
#define R4_ISSUE_PATCH(src_, hc_, buf_)                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < PW; ++j)                                                             \
      GLDS16((src_) + (hc_) * 64 + psrc[j], Ps + (buf_) * PATCH_B + (j * 256 + wave * 64) * 16);               \
  } while (0)
  R4_ISSUE_PATCH(patch_src, 0, 0);

  // Residual of the current tile: 2 x MT 16-byte loads per lane, two per sub-step in sub-steps 1..7 by
  // inline asm and first touched after the barrier in front of sub-step 17.  Hidden from the compiler on purpose: with ordinary
  // loads it placed `s_waitcnt vmcnt(0)` at the top of the tile loop, right behind their issue (the whole HBM latency of the
  // residual exposed per tile: conv2 89 us against 68 us for conv1).  Their completion is implied by this wave's own vmcnt(0) in
  // front of sub-step 8 (they are older than everything issued later); the "+v" statements behind the second barrier keep every
  // compiler-generated read of these registers behind that point.  Defined in ONE place per iteration: no loop-carried copy.
  u32x4 rq[RES ? MT : 1][2];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = b4[ct];
  frag wf[2][NT], xf[2][MT];
  R4_WAIT_VM0();
  R4_BARRIER();
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) wf[0][ct] = *(const frag*)(smem + wbase + ct * 1024);
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) xf[0][pt] = *(const frag*)(smem + R4_XO(0, pt));

  // sub-step U_ (tap U_ % 9 of half-chunk U_ / 9): its MT x 4 MFMAs on fragment set U_ & 1, with the NR fragment reads of sub-step
  // U_ + 1 (U_ = 17: the next tile's sub-step 0) and, at U_ = 0 / 8, the PW LDS-DMA pieces of a patch burst in the gaps
#define R4_GRP(P_, U_)                                                                                         \
  do {                                                                                                         \
    constexpr int C_ = (U_) & 1, N_ = C_ ^ 1, UN_ = ((U_) + 1) % NSTEP, NB_ = UN_ / 9, NTP_ = UN_ % 9;         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                          \
      acc[P_][ct] = Elem<T>::mfma(wf[C_][ct], xf[C_][P_], acc[P_][ct]);                                        \
    if constexpr ((P_) < 2) {                                                                                  \
      wf[N_][2 * (P_)] = *(const frag*)(smem + wofn_ + (2 * (P_)) * 1024);                                     \
      wf[N_][2 * (P_) + 1] = *(const frag*)(smem + wofn_ + (2 * (P_) + 1) * 1024);                             \
    } else {                                                                                                   \
      if constexpr (2 * ((P_) - 2) < MT) xf[N_][2 * ((P_) - 2)] = *(const frag*)(smem + R4_XO(NTP_, 2 * ((P_) - 2)) + NB_ * PATCH_B); \
      if constexpr (2 * ((P_) - 2) + 1 < MT) xf[N_][2 * ((P_) - 2) + 1] = *(const frag*)(smem + R4_XO(NTP_, (2 * ((P_) - 2) + 1 < MT ? 2 * ((P_) - 2) + 1 : 0)) + NB_ * PATCH_B); \
    }                                                                                                          \
    if constexpr (((U_) == 0 || (U_) == 8) && 2 * (P_) < PW) {                                                 \
      GLDS16(dsrc_ + psrc[2 * (P_)], Ps + ((U_) == 0 ? PATCH_B : 0) + ((2 * (P_)) * 256 + wave * 64) * 16);    \
      if constexpr (2 * (P_) + 1 < PW)                                                                         \
        GLDS16(dsrc_ + psrc[2 * (P_) + 1], Ps + ((U_) == 0 ? PATCH_B : 0) + ((2 * (P_) + 1) * 256 + wave * 64) * 16); \
    }                                                                                                          \
    if constexpr (RES && (U_) >= 1 && (U_) <= MT && (P_) < 2) {   /* residual of pixel tile U_ - 1, half P_: one load per group */ \
      /* (MFMA order.  r04 tried line order + a hop through the line image: the launch got 3 us SLOWER -- 82.5 vs 79.5 us same-run; \
         profiles/r04_conv_r4_line_order_ab.txt -- while line-order STORES gain 2 us on the launches without a residual) */ \
      const char* rp_ = rb + oconst[(RES && (U_) - 1 < MT) ? (U_) - 1 : 0];                                    \
      if constexpr ((P_) == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[(U_) - 1 < MT ? (U_) - 1 : 0][0]) : "v"(rp_) : "memory"); \
      else asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(rq[(U_) - 1 < MT ? (U_) - 1 : 0][1]) : "v"(rp_) : "memory"); \
    }                                                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                         \
    if constexpr ((P_) < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                 \
    else if constexpr (2 * ((P_) - 2) + 1 < MT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);             \
    else if constexpr (2 * ((P_) - 2) < MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 \
    if constexpr (((U_) == 0 || (U_) == 8) && 2 * (P_) < PW)                                                   \
      __builtin_amdgcn_sched_group_barrier(0x020, (2 * (P_) + 1 < PW ? 2 : 1), 0);                             \
  } while (0)
#define R4_SUB(U_)                                                                                             \
  do {                                                                                                         \
    int wofn_ = wbase + ((((U_) + 1) % NSTEP) * TILE_B);                                                       \
    asm volatile("" : "+v"(wofn_));                                                                            \
    /* U_ = 0: this tile's second half-chunk; 8: the next tile's first (the last tile re-reads its own: no branch in the stream) */ \
    const char* const dsrc_ = (U_) == 0 ? patch_src + 64 : n_patch_src;                                        \
    R4_GRP(0, U_); R4_GRP(1, U_); R4_GRP(2, U_); R4_GRP(3, U_); R4_GRP(4, U_); R4_GRP(5, U_); R4_GRP(6, U_);   \
    if constexpr (MT > 7) R4_GRP(7, U_);                                                                       \
  } while (0)

#ifdef FLOPE_STAG_DBG
  // diagnostic build, dbg & 64: shader-clock stamps of this workgroup's THIRD tile (steady state), wave 0:
  // {tile start, before wait 1, after barrier 1, before wait 2, after barrier 2, loop end, epilogue end} + real time {start, end}
  unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_r[2] = {0, 0};   // 9: behind sub-step 8, 10: behind 12, 11: behind 0
  int st_it = 0;
#define R4_STAMP(i_) do { if ((p.dbg & 64) && st_it == 2) { __builtin_amdgcn_sched_barrier(0); st[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define R4_STAMP(i_) do {} while (0)
#endif
  for (;;) {
    const bool has_next = tile + tstep < tend;
    int nR0 = R0, nm0 = m0;
    if (has_next) tile_rows(tile + tstep, nR0, nm0);
#ifdef FLOPE_STAG_DBG      /* ablation (results wrong by construction), dbg & 1: the "next tile" burst re-reads this tile's own (L2-warm) patch */
    const char* const n_patch_src = (p.dbg & 1) ? patch_src : (const char*)p.in + (size_t)nR0 * rowB;
#else
    const char* const n_patch_src = (const char*)p.in + (size_t)nR0 * rowB;
#endif

    R4_STAMP(0);
#ifdef FLOPE_STAG_DBG
    if ((p.dbg & 64) && st_it == 2) st_r[0] = __builtin_amdgcn_s_memrealtime();
#endif
    const char* const rb = RES ? (const char*)p.res + (size_t)m0 * p.Wop * p.Cout * 2 : nullptr;
    R4_SUB(0); R4_STAMP(11);
    R4_SUB(1); R4_SUB(2); R4_SUB(3); R4_SUB(4); R4_SUB(5); R4_SUB(6); R4_SUB(7);
    R4_STAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's fragments of sub-step 8 (the last reads of buffer 0) are in registers
    // its pieces of the second half-chunk (and the last tile's stores) are done.  RES (r03b): the 2 MT residual loads of sub-steps
    // 1..7 are YOUNGER than that burst and are not needed before the epilogue -- they stay in flight (waiting for them here, with
    // a plain vmcnt(0), was 2.7-3.0 k cycles of every residual tile: profiles/r03_conv_r4_residual_wait.txt); the vmcnt(0) in
    // front of sub-step 17 still covers them
    if constexpr (RES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT) : "memory"); else R4_WAIT_VM0();
    R4_BARRIER();
    R4_STAMP(2);
    R4_SUB(8); R4_STAMP(9); R4_SUB(9); R4_SUB(10); R4_SUB(11); R4_SUB(12); R4_STAMP(10); R4_SUB(13); R4_SUB(14); R4_SUB(15); R4_SUB(16);
    R4_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... of sub-step 17 (the last reads of buffer 1)
    R4_WAIT_VM0();                                           // its pieces of the next tile's first half-chunk have landed
    R4_BARRIER();
    R4_STAMP(4);
    if constexpr (RES) {
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) asm volatile("" : "+v"(rq[pt][0]), "+v"(rq[pt][1]));
    }
    R4_SUB(17);
    R4_STAMP(5);

    // ---- epilogue: (+ residual) (ReLU) -> 16-bit padded NHWC through the wave's line image; accumulators back to the bias
    char* const ob = (char*)p.out + (size_t)m0 * p.Wop * p.Cout * 2;
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      float v[NT * 4];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[ct * 4 + q] = acc[pt][ct][q];
      const int pp0 = lane >> 3, pp1 = 8 + (lane >> 3);
      char* const li0 = lscr + pp0 * 128 + (((lane & 7) ^ (pp0 & 7)) << 4);    // this lane's chunk of the image, line order (rows pp0, pp1)
      char* const li1 = lscr + pp1 * 128 + (((lane & 7) ^ (pp1 & 7)) << 4);
      if constexpr (RES) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[c * 8 + q * 2] += unpack_lo<T>(rq[pt][c][q]);
            v[c * 8 + q * 2 + 1] += unpack_hi<T>(rq[pt][c][q]);
          }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pk_out16<T>(pack2<T>(v[c * 8 + q * 2], v[c * 8 + q * 2 + 1]), p.relu);
        if constexpr (RES) *(u32x4*)(ob + oconst[pt] + c * 64) = o;      // (with a residual the launch is HBM-bound and the hop through the image only adds latency: +2..5 us same-run)
        else *(u32x4*)(lscr + r16 * 128 + (((c * 4 + g) ^ (r16 & 7)) << 4)) = o;
      }
      if constexpr (!RES) {
        *(u32x4*)(ob + osto[pt][0]) = *(const u32x4*)li0;
        *(u32x4*)(ob + osto[pt][1]) = *(const u32x4*)li1;
      }
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = b4[ct];
    }
    R4_STAMP(6);
#ifdef FLOPE_STAG_DBG
    if ((p.dbg & 64) && st_it == 2) {
      st_r[1] = __builtin_amdgcn_s_memrealtime();
      if (p.split_ws && wave == 0 && lane == 0) {
        unsigned long long* d_ = (unsigned long long*)p.split_ws + (size_t)blockIdx.x * 16;
        for (int i = 0; i < 7; ++i) d_[i] = st[i];
        d_[7] = st_r[0]; d_[8] = st_r[1]; d_[9] = st[9]; d_[10] = st[10]; d_[11] = st[11];
      }
    }
    ++st_it;
#endif
    if (!has_next) break;
    tile += tstep;
    R0 = nR0; m0 = nm0; patch_src = n_patch_src;
  }
  R4_WAIT_VM0();
#ifdef FLOPE_STAG_DBG
  if ((p.dbg & 64) && p.split_ws && wave == 0 && lane == 0) {
    unsigned long long* d_ = (unsigned long long*)((char*)p.split_ws + 65536) + (size_t)blockIdx.x * 4;
    d_[0] = rt_entry_; d_[1] = __builtin_amdgcn_s_memrealtime(); d_[2] = (unsigned long long)st_it; d_[3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  }
#endif
#undef R4_STAMP
#undef R4_SUB
#undef R4_GRP
#undef R4_XO
#undef R4_ISSUE_PATCH
#undef R4_BARRIER
#undef R4_WAIT_VM0
}

extern "C" int flope_conv_r4_init() {
  hipError_t e = hipSuccess;
#define A(T, RES) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_r4_kernel<T, 7, 5, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  A(bf16_t, false) A(bf16_t, true) A(f16_t, false) A(f16_t, true)
#undef A
  return (int)e;
}

// can this conv run on conv_r4?  64 -> 64 channels, two output rows = 7 MFMA pixel tiles (Wo = 56), 8-row bands
extern "C" int flope_conv_r4_ok(const ConvP* p) {
  return p->stride == 1 && p->ntaps == 9 && p->Cin == 64 && p->Cout == 64 && p->Wo == 56 && p->Ho % 8 == 0 &&
         10 * (p->Wip + 2) * 4 <= 5 * 512 && p->ksplit <= 1 && !p->ds_in;
}

// p->tiles_per_image = Ho / 8, p->total_tiles = B * tiles_per_image, p->w = the conv_stag weight image of a 64 -> 64 layer
// ([18 steps][64 rows][32 k]), p->mg_pitch / sh_pitch = fastdiv magic of Wip + 2; grid_blocks <= total_tiles (persistent)
extern "C" int flope_conv_r4_launch(const ConvP* p, int dtype, int grid_blocks, void* stream) {
  if (!flope_conv_r4_ok(p) || !p->mg_pitch || grid_blocks < 1 || grid_blocks > p->total_tiles || p->tiles_per_image != p->Ho / 8)
    return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)2 * 5 * 8192 + 18 * 4096 + 4 * 2048;      // patch buffers, resident weights, line images
  const dim3 grid(grid_blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) { if (p->res) hipLaunchKernelGGL((conv_r4_kernel<bf16_t, 7, 5, true>), grid, block, lds, st, *p); else hipLaunchKernelGGL((conv_r4_kernel<bf16_t, 7, 5, false>), grid, block, lds, st, *p); }
  else            { if (p->res) hipLaunchKernelGGL((conv_r4_kernel<f16_t, 7, 5, true>), grid, block, lds, st, *p); else hipLaunchKernelGGL((conv_r4_kernel<f16_t, 7, 5, false>), grid, block, lds, st, *p); }
  return (int)hipGetLastError();
}
