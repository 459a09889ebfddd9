// conv_w4: 3x3 / stride-1 implicit-GEMM convolution on flat 256 px x 128 ch tiles (layer2..4 of the ResNet-18 trunk,
// reference sunflower/models/posenet.py:25) -- third-generation structure (r03): FOUR waves, one per SIMD.
//
// Why (r03 in-kernel stamps of conv_stag, profiles/r03_phase_stamps_conv_stag.txt + r03_inkernel_clock_*.txt): the chip holds
// 2.1-2.4 GHz inside these loops -- not the ~1.55 GHz r02 had inferred from an MFMA-only ablation -- so the matrix pipe was not
// clock-bound, it was waiting.  conv_stag's two staggered 4-wave groups alternate a load half (DMA issue + 16 ds_read_b128 +
// waits, ~600 cycles) and an MFMA half (32 MFMAs, ~640 cycles with the partner's issue traffic) separated by TWO workgroup
// barriers per double step (~170 cycles each from last arrival to release): ~1600 cycles per double step against a 1024-cycle
// MFMA floor.  Here:
//   * a wave owns 128 px x 64 ch (8 x 4 MFMA tiles, 128 accumulator registers -- the 512-register budget of one wave per
//     SIMD): 12 fragment reads per 32 MFMAs instead of 16 (LDS bytes per MFMA -25 %);
//   * software pipelining INSIDE the wave: while the 32 MFMAs of sub-step u issue, the wave's own 12 ds_read_b128 of
//     sub-step u + 1 and its LDS-DMA pieces go out in the gaps (an MFMA holds the vector issue for 8 of its 16 cycles), into
//     the second fragment register set -- no partner wave, no load half;
//   * ONE barrier per double step (64 MFMAs per wave), in the middle of it: the wave waits for its DMA pieces of the next
//     double tile (counted vmcnt), then the barrier publishes them; the reads of the following sub-step come after it.
// Same LDS images (conflict-free patch image of conv_stag r03, weight ring of double tiles), same packed weights, same folded 1x1
// stride-2 shortcut, residual by LDS-DMA in the slots the look-ahead of the last body leaves unused.  The tile HEIGHT is a template
// argument (MT = 8..4 pixel tiles of 16 per wave = 256..128-pixel workgroup tiles): the engine picks it per launch by whole rounds
// of the chip -- every height runs the same MFMA sequence per output, so all of them are bit-identical.
//
// r04:
//   * PERS = a CLASS WALK.  A workgroup walks M tiles mt, mt + G, mt + 2 G, ... where G tiles are a whole number of images
//     (p.cw_imgs), so every tile of its walk has the same geometry relative to its patch origin: the fragment address table is
//     built ONCE, a tile boundary is {epilogue, uniform pointer bumps, accumulators back to the bias}.  The step stream does not
//     stop at the boundary: the weight ring wraps to the start of the panel (= the next tile's first double tiles), the patch
//     burst of double step 5 of a tile's last body fetches the NEXT tile's first half-chunk, and the fragments the last
//     sub-step prefetches ARE the next tile's first (same table).  r03's persistent form recomputed the table at every
//     boundary (3.7-5.3 k cycles of a 10.8 k boundary against 15 k for a fresh workgroup: measured slower).
//     Residual inputs of persistent tiles come by register loads during the last body (their LDS-DMA slots carry the next
//     tile's data).
//   * LINE-ORDER stores.  The MFMA accumulator layout gives a lane one pixel x two runs of 8 channels, so a wave-store touched 16
//     pixels x 64 bytes; the store path takes ~66 cycles per such instruction (64 per workgroup tile = 4.2 k cycles, measured:
//     tools/probes/store_probe.hip, profiles/r04_store_probe.txt).  The packed outputs now pass through a wave-private 2 KB LDS
//     image (XOR-swizzled, conflict-free both ways) and leave as 8 whole 128-byte lines per instruction: 2.9 k cycles.  Only the
//     order of the stores changes -- same bytes, same addresses.
#include "common.h"
#include "w4_sched.h"
#ifndef FLOPE_W4_SPREAD
#define FLOPE_W4_SPREAD 2
#endif

// LDS-DMA in the compiler-visible form.  While hipcc's wait-count pass knows of one outstanding flat-encoded access that may land in
// LDS (global_load_lds counts as one) it forces every wait it inserts to 0: lgkmcnt(0) in front of each sub-step's first MFMA and
// vmcnt(0) at the first use of any ordinary load -- so this kernel counts its DMAs itself (W4_WAIT_VM) and keeps ordinary loads out
// of the loop (r03 also built the DMAs as inline assembly hidden from that pass: exact lgkmcnt waits, 1.5-2.5 % slower per conv
// for the s_nop + s_mov m0 per piece; removed in r04).
#define GLDS16(gptr, lptr)                                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                      \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ int tile_px_w4(int c) { return c < 4 ? 2 * c : (c < 12 ? 2 * (c - 4) + 1 : 2 * (c - 8)); }

// Address table of a lane: xoff[t][pt] = LDS byte offset of its pixel fragment of pixel tile pt for tap t
//   = (pi << 6) + ((g ^ ((sv >> 2) & 3)) << 4),  pi = pi0 + dy * pitch + dx,  sv = v0 + dy * W + dx   (conv_stag.hip, r03 image)
//   and ((g ^ ((sv >> 2) & 3)) << 4) = ((sv << 2) ^ (g << 4)) & 0x30 = ((sv << 2) & 0x30) ^ (g << 4): everything but the last
//   XOR is the same for the four k-slots g of a pixel column (and for both channel-half waves).
// (r03: loading the table from memory instead -- precomputed per tile geometry class, 49 classes x 20 KiB per conv, 20 coalesced
// 16-byte loads per lane -- was measured and dropped: the loads took 10-12 k cycles against 5 k for computing it, the tables do not
// stay in L2 between tiles; step 1.116 vs 1.094 ms in same-run A/B.)
// PT: 8 KB DMA rounds per patch buffer (4, 5 or 6); the weight ring holds NBD = w4_ring(PT, ..) double tiles (the DMA runs NBD - 1
// double steps ahead of its consumer).  MT: pixel tiles per wave (8..4; below 7 only PT = 4).  RES: residual input.  DSF: folded
// 1x1 stride-2 shortcut (no residual).  PERS: class walk (header; MT = 7 only).
template <typename T, int PT, bool RES, bool DSF, bool PERS, int MT>
__global__ __launch_bounds__(256, 1) void conv_w4_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  // MT = pixel tiles of 16 per wave: 8 -> 256-pixel workgroup tiles; 7 -> 224 (r03c: 224 divides the 28 x 28, 14 x 14 and 7 x 7
  // maps of a 224 x 224 crop batch -- no ragged last tile -- and 224-pixel tiles land closer under a whole number of rounds);
  // 6, 5, 4 -> 192, 160, 128: launches of fewer than ~200 tiles (layer 4, batch slices) fill more of the chip with smaller ones
  constexpr int BM = 2 * MT * 16, WPXB = MT * 16, TILE_B = 128 * 64, DT_B = 2 * TILE_B, NT = 4;
  static_assert(MT >= 5 && MT <= 8, "conv_w4: 5..8 pixel tiles per wave");
  constexpr int NBD = w4_ring(PT, DSF && PERS);
  static_assert(NBD >= 3, "conv_w4: this variant does not fit the CU's LDS");
  constexpr int PATCH_B = PT * 8192;
  constexpr int PD = NBD - 1;                              // double tiles in flight ahead of the one being consumed
  constexpr int TGW = W4_TGW;                              // LDS-DMA ops per wave per double tile (4)
  constexpr int PW = 2 * PT;                               // ... per patch burst
  constexpr int SPREAD = FLOPE_W4_SPREAD;                  // parts the cold patch burst is issued in (w4_sched.h)
  static_assert(!(RES && DSF) && PT >= 4 && !(PERS && MT != 7), "conv_w4 variants");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ps = smem;                                   // 2 patch buffers
  char* const Bs = smem + 2 * PATCH_B;                     // NBD double tiles of weights
  // DSF: the shortcut's double tile of weights.  One tile per workgroup: the ring's last look-ahead slot, idle until the first
  // body double step issues into it.  PERS: at a tile boundary every ring slot is in flight -> a slot of its own behind the ring.
  constexpr int DSW_B = (PERS ? NBD : NBD - 1) * DT_B;
  constexpr int SCR_B = 2 * PATCH_B + NBD * DT_B + (DSF && PERS ? DT_B : 0);   // 12.5 KB: address-table exchange, then the epilogue's line image

#ifdef FLOPE_STAG_DBG
  // diagnostic build, dbg & 64: shader-clock stamps {entry, loop start, loop end, exit} + 100 MHz real time {loop start, loop end}
  // of wave 0, one record of 6 x uint64 per workgroup at p.split_ws (tools/clock_probe.py)
  const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
  unsigned long long st_p[4] = {0, 0, 0, 0}, st_b[6] = {0, 0, 0, 0, 0, 0};
  int st_tile = 0;
#define W4_BSTAMP(i_) do { if (st_tile == 0) { __builtin_amdgcn_sched_barrier(0); st_b[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define W4_PSTAMP(i_) do { __builtin_amdgcn_sched_barrier(0); st_p[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // per double step of the LAST body: lane 2 D = clock in front of the DMA wait, lane 2 D + 1 = behind the barrier (one VGPR)
  unsigned st_v = 0;
  // (dbg & 256: of the FIRST body of the workgroup's second tile instead -- class walk -- or, one tile per workgroup, of its first body)
#define W4_DSTAMP(i_) do { __builtin_amdgcn_sched_barrier(0); if ((p.dbg & 256) ? (hc == 0 && st_tile == (PERS ? 1 : 0)) : lastb_) { const unsigned lo_ = (unsigned)__builtin_amdgcn_s_memtime(); asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(st_v) : "s"(lo_), "n"(i_)); } __builtin_amdgcn_sched_barrier(0); } while (0)
  // dbg & 512: inside double step 6 / 7 of the same body, a stamp behind every MFMA group of the second sub-step of D = 6 (lanes 0..7)
  // and of the first sub-step of D = 7 (lanes 8..15)
  unsigned st_w = 0;
#define W4_GSTAMP(i_) do { if (p.dbg & 512) { __builtin_amdgcn_sched_barrier(0); if ((p.dbg & 256) ? (hc == 0 && st_tile == (PERS ? 1 : 0)) : lastb_) { const unsigned lo_ = (unsigned)__builtin_amdgcn_s_memtime(); asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(st_w) : "s"(lo_), "n"(i_)); } __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define W4_PSTAMP(i_) do {} while (0)
#define W4_BSTAMP(i_) do {} while (0)
#define W4_DSTAMP(i_) do {} while (0)
#define W4_GSTAMP(i_) do {} while (0)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave & 1, wch = wave >> 1;
  const int g = lane >> 4, r16 = lane & 15;
  const int pcol = tile_px_w4(r16);
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = lb % p.ntiles;
  const int HoWo = p.Ho * p.Wo;
  const int nhc = p.Cin / 32;
  const int NS = nhc * 9;
  const int nbody = nhc / 2, ND = nbody * 9;               // double steps of this tile
  const size_t pixB = (size_t)p.Cin * 2;
  const size_t rowB = (size_t)p.Wip * pixB;
  const int pitch = p.Wip + 2;                             // conflict-free patch image (conv_stag.hip, r03)

  int mt = lb / p.ntiles;                                  // this workgroup's current M tile
  const int G_mt = gridDim.x / p.ntiles;                   // PERS: stride of its walk (the grid is a multiple of ntiles; G_mt tiles = p.cw_imgs images)
  const int m0 = mt * BM, mend = min(m0 + BM, p.M);        // the FIRST tile (PERS: every tile is whole and has its geometry)
  int R0;
  const char* patch_src;
  {
    const int b0_ = fastdiv(m0, p.mg_hw, p.sh_hw), ho0_ = fastdiv(m0 - b0_ * HoWo, p.mg_w, p.sh_w);
    R0 = b0_ * p.Hip + ho0_;
    patch_src = (const char*)p.in + (size_t)R0 * rowB;
  }
  // PERS: what one step of the walk (p.cw_imgs images) adds to the pointers
  const size_t d_patch = PERS ? (size_t)p.cw_imgs * p.Hip * rowB : 0;
  const unsigned d_out = PERS ? (unsigned)((size_t)p.cw_imgs * p.Hop * p.Wop * p.Cout * 2) : 0u;

  const char* const b_base = (const char*)p.w + (size_t)ntile * NS * TILE_B + wave * 1024 + lane * 16;
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wbase = 2 * PATCH_B + (wch * 64 + r16) * 64 + ((g ^ wsw) << 4);
  const int cb = ntile * 128 + wch * 64 + g * 8;
  const float* const bias_p = p.bias + cb;

#define W4_WAIT_PIN() __builtin_amdgcn_sched_barrier(0)
#define W4_WAIT_VM(n_)                                                                                         \
  do {                                                                                                         \
    W4_WAIT_PIN();                                         /* behind the sub-step's last MFMAs, not in front of them */ \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory");                                                  \
  } while (0)
#define W4_BARRIER()                                                                                           \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                              \
    asm volatile("" ::: "memory");                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define W4_ISSUE_PATCH(src_, buf_)                                                                             \
  do {                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < PW; ++j)                                                             \
      GLDS16((src_) + psrc[j], Ps + (buf_) * PATCH_B + (j * 256 + wave * 64) * 16);                            \
  } while (0)
#define W4_ISSUE_DT(dt_, slotb_)   /* slotb_: byte offset of the ring slot */                                   \
  do {                                                                                                         \
    _Pragma("unroll") for (int o = 0; o < TGW; ++o)                                                            \
      GLDS16(b_base + (size_t)(dt_) * DT_B + o * 4096, Bs + (slotb_) + o * 4096 + wave * 1024);                \
  } while (0)

  // the bias first: an ordinary global load -- hipcc waits vmcnt(0) at its first use, which would drain every LDS-DMA issued before
  // that point, so the use (accumulator init) is pinned behind the address tables, where this wave waits for its DMAs anyway
  f32x4 b4[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) b4[ct] = *(const f32x4*)(bias_p + (ct >> 1) * 32 + (ct & 1) * 4);

  // ---- prologue, ordered by latency (r03 stamps: the address tables below are ~1,500 vector instructions = 3 us of a wave that
  // has its SIMD to itself; issued behind them, the first DMAs added their own 1.5 us of flight time to every tile): the first PD
  // double tiles of weights need nothing but the tile's channel block -> out first; then the shortcut's first gather (DSF), the
  // first patch; the tables are computed while all of that is in flight.
  W4_PSTAMP(0);
#pragma unroll
  for (int d = 0; d < PD; ++d) W4_ISSUE_DT(d, d * DT_B);

  // DSF: folded 1x1 stride-2 shortcut (conv_stag.hip): per 64 channels of the block input x, the tile's 256 centre pixels
  // x(2 ho, 2 wo) as two 32-channel pixel tiles in patch buffer 1, the matching double tile of shortcut weights in the LAST ring
  // slot, one double step of MFMAs into the same accumulators (which start at bias2 + bias_ds)
  const char* dsrc[DSF ? 4 : 1];
  const char* dw_base = nullptr;
#define W4_ISSUE_DS(j_)                                                                                        \
  do {                                                                                                         \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                              \
      _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                                         \
        GLDS16(dsrc[rr] + (2 * (j_) + t) * 64, Ps + PATCH_B + t * 16384 + (rr * 256 + wave * 64) * 16);        \
    _Pragma("unroll") for (int o = 0; o < TGW; ++o)                                                            \
      GLDS16(dw_base + (size_t)(j_) * DT_B + o * 4096, Bs + DSW_B + o * 4096 + wave * 1024);                   \
  } while (0)
  if constexpr (DSF) {
    const size_t dpix = (size_t)p.ds_Cin * 2;              // this lane's four gather sources of the first tile
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int q = rr * 256 + wave * 64 + lane, sl = q >> 2;
      const int m = min(m0 + sl, mend - 1);
      const int b_ = fastdiv(m, p.mg_hw, p.sh_hw), r_ = m - b_ * HoWo;
      const int ho_ = fastdiv(r_, p.mg_w, p.sh_w), wo_ = r_ - ho_ * p.Wo;
      dsrc[rr] = (const char*)p.ds_in + (((size_t)b_ * p.ds_Hip + 2 * ho_ + 1) * p.ds_Wip + 2 * wo_ + 1) * dpix +
                 (((q & 3) ^ ((sl >> 2) & 3)) << 4);
    }
    dw_base = (const char*)p.ds_w + (size_t)ntile * (p.ds_Cin / 32) * TILE_B + wave * 1024 + lane * 16;
    W4_ISSUE_DS(0);
  }
  const size_t d_ds = (PERS && DSF) ? (size_t)p.cw_imgs * p.ds_Hip * p.ds_Wip * p.ds_Cin * 2 : 0;

  // per-lane DMA source offsets of a patch burst: op j = 2 rr + h moves pieces rr * 512 + h * 256 + wave * 64 + lane
  unsigned psrc[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int q = j * 256 + wave * 64 + lane;
    const int pi = q >> 2;
    const int pr_ = fastdiv(pi, p.mg_pitch, p.sh_pitch), pc_ = pi - __mul24(pr_, pitch);
    const int js = (q & 3) ^ (((__mul24(pr_, p.Wo) + pc_) >> 2) & 3);
    psrc[j] = (unsigned)(__mul24(__mul24(pr_, p.Wip) + min(pc_, p.Wip - 1), (int)pixB) + js * 16);
  }
  W4_ISSUE_PATCH(patch_src, 0);
  W4_PSTAMP(1);

  // Fragment address table xoff[t][pt] and output offsets.  Eight lanes of the workgroup -- the four k-slots g of a pixel
  // column, in both channel-half waves -- need the same 8 x 9 offsets up to a final `^ (g << 4)` / `+ channel block`: each of the
  // eight computes ONE pixel tile (a fastdiv chain + 9 entries, ~60 vector instructions instead of ~700) and they trade through 12 KB
  // of LDS behind the weight ring (r03 stamps: the table was 3-5 k cycles of a 9-10 k-cycle prologue in which the matrix pipe idles).
  // Output offsets come in two orders: ooff[pt] = this lane's own pixel (MFMA order; RES only: where its residual lives) and
  // osto[pt][c] = pixel c * 8 + (lane >> 3) of the tile's LDS line image, 16-byte chunk lane & 7 of the wave's 128 bytes (line order).
  int xoff[9][MT];
  unsigned ooff[RES ? MT : 1];
  unsigned osto[MT][2];
  bool okl[MT][2];
  {
    /* [wpx][r16][pt][12 dwords], 400 B (25 bank quads, odd) per r16: the 16 pixel columns of a ds_read_b128 lane group land on
       16 different quads (at 384 B they fell on two: the reads below were 8-way conflicts, r03a counters) */
    char* const scr = smem + SCR_B + (wpx * 16 + r16) * 400;
    const int ptm = wch * 4 + g;                                      /* the pixel tile this lane computes */
    const int mm = m0 + wpx * WPXB + ptm * 16 + pcol;
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
    *(u32x4*)(scr + ptm * 48) = e0; *(u32x4*)(scr + ptm * 48 + 16) = e1; *(u32x4*)(scr + ptm * 48 + 32) = e2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the LDS writes are done before the barrier lets the readers go (no vmcnt: the DMAs stay in flight) */
    W4_BARRIER();
    const unsigned g4 = (unsigned)(g << 4);
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const u32x4 a0 = *(const u32x4*)(scr + pt * 48), a1 = *(const u32x4*)(scr + pt * 48 + 16);
      const u32x2 a2 = *(const u32x2*)(scr + pt * 48 + 32);
#pragma unroll
      for (int t = 0; t < 4; ++t) { xoff[t][pt] = (int)(a0[t] ^ g4); xoff[4 + t][pt] = (int)(a1[t] ^ g4); }
      xoff[8][pt] = (int)(a2[0] ^ g4);
      if constexpr (RES) ooff[pt] = a2[1] + (unsigned)(cb * 2);
      // line order: row pp = c * 8 + (lane >> 3) of the line image is written by the lanes with r16 = pp, i.e. it holds the pixel
      // whose table entry sits at (wpx, pp)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int pp = c * 8 + (lane >> 3);
        osto[pt][c] = *(const unsigned*)(smem + SCR_B + (wpx * 16 + pp) * 400 + pt * 48 + 36) +
                      (unsigned)((ntile * 128 + wch * 64) * 2 + (lane & 7) * 16);
        okl[pt][c] = PERS || (m0 + wpx * WPXB + pt * 16 + tile_px_w4(pp) < mend);
      }
    }
  }

  // pin the tables in FRONT of the DMA wait (left alone, the compiler sinks these ~800 pure vector instructions behind the wait
  // and the barrier, next to their first use -- the DMA flight time and the table time then add up instead of overlapping)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) asm volatile("" ::"v"(xoff[t][pt]));
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) asm volatile("" ::"v"(osto[pt][0]), "v"(osto[pt][1]));
  __builtin_amdgcn_sched_barrier(0);
#ifdef FLOPE_W4_MFMA32_TIMING
  // TIMING-ONLY build (r05, results wrong by construction): every group of four v_mfma_f32_16x16x32 becomes two
  // v_mfma_f32_32x32x16 on the same fragment registers and the same 16 accumulator registers -- the same FLOPs, LDS reads, DMA
  // pieces, barriers and stores as a real 32x32 kernel would issue, so that the shape's effect on cycles AND on the clock the chip
  // holds is measured on the real step before the kernel is rebuilt around it (profiles/r05_conv_w4_mfma32_timing.txt)
  f32x16 acc32[MT];
#define W4_ACCQ(pt_, ct_, q_) acc32[pt_][(ct_) * 4 + (q_)]
#define W4_MFMA_GRP(C_, P_)                                                                                    \
  do {                                                                                                         \
    acc32[P_] = Elem<T>::mfma32(wf[C_][(2 * (P_)) & 3], xf[C_][P_], acc32[P_]);                                \
    acc32[P_] = Elem<T>::mfma32(wf[C_][(2 * (P_) + 1) & 3], xf[C_][P_], acc32[P_]);                            \
  } while (0)
#define W4_MFMAS_PER_GRP 2
#define W4_ACC_INIT()                                                                                          \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                          \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt)                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) acc32[pt][ct * 4 + q] = b4[ct][q];                       \
  } while (0)
  W4_ACC_INIT();
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) asm volatile("" : "+a"(acc32[pt]));
#else
  f32x4 acc[MT][NT];
#define W4_ACCQ(pt_, ct_, q_) acc[pt_][ct_][q_]
#define W4_MFMA_GRP(C_, P_)                                                                                    \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                          \
      acc[P_][ct] = Elem<T>::mfma(wf[C_][ct], xf[C_][P_], acc[P_][ct]);                                        \
  } while (0)
#define W4_MFMAS_PER_GRP 4
#define W4_ACC_INIT()                                                                                          \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                          \
      _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) acc[pt][ct] = b4[ct];                                  \
  } while (0)
  W4_ACC_INIT();
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)                          // ... and the accumulator init too (128 register moves)
    asm volatile("" : "+a"(acc[pt][0]), "+a"(acc[pt][1]), "+a"(acc[pt][2]), "+a"(acc[pt][3]));
#endif
  frag wf[2][NT], xf[2][MT];                               // two fragment sets: sub-step u uses set u & 1

  W4_PSTAMP(2);
  W4_WAIT_VM(0);                                            // patch 0, the first double tiles (and the shortcut's first gather) landed
  W4_BARRIER();
  W4_PSTAMP(3);

  int xds[DSF ? MT : 1];
  if constexpr (DSF) {
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int sl = wpx * WPXB + pt * 16 + pcol;
      xds[pt] = PATCH_B + (sl << 6) + ((g ^ ((sl >> 2) & 3)) << 4);
    }
  }
  // the shortcut's double steps of the current tile; FIRST_: its first gather is already in LDS (issued with the prologue's DMAs)
#define W4_DS_CHAIN(FIRST_)                                                                                    \
  do {                                                                                                         \
    for (int j = 0; j < p.ds_Cin / 64; ++j) {                                                                  \
      if (!(FIRST_) || j > 0) {                                                                                \
        W4_ISSUE_DS(j);                                                                                        \
        W4_WAIT_VM(0);                                                                                         \
        W4_BARRIER();                                                                                          \
      }                                                                                                        \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                          \
        _Pragma("unroll") for (int ct = 0; ct < NT; ++ct) wf[h][ct] = *(const frag*)(smem + wbase + DSW_B + h * TILE_B + ct * 1024); \
        _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) xf[h][pt] = *(const frag*)(smem + xds[pt] + h * 16384); \
      }                                                                                                        \
      _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                            \
        _Pragma("unroll") for (int pt = 0; pt < MT; ++pt) W4_MFMA_GRP(h, pt);                                  \
      W4_BARRIER();                                        /* buffer 1 / the shortcut's weight slot are free again */ \
    }                                                                                                          \
  } while (0)
  if constexpr (DSF) W4_DS_CHAIN(true);

  // fragments of sub-step 0: double tile 0 (slot 0), half 0; patch buffer 0, tap 0
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) wf[0][ct] = *(const frag*)(smem + wbase + ct * 1024);
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) xf[0][pt] = *(const frag*)(smem + xoff[0][pt]);

  // PERS + RES: the tile's residual, 2 MT 16-byte loads per lane into ACCUMULATOR registers (the loop has no arch VGPR to spare: 256
  // is the encoding's limit) in MFMA order (the lane's own pixel), issued by inline
  // assembly in the first sub-steps of double steps 7 and 8 of the last body (hidden from hipcc's wait-count pass, which would answer
  // an ordinary load among LDS-DMAs with vmcnt(0) at its first use): counted in W4_DSTEP's waits (w4_sched.h), complete behind
  // the epilogue's own vmcnt(TGW)
  u32x4 rq[(RES && PERS) ? MT : 1][2];
  const char* const res_b = (const char*)p.res;

  // One sub-step: the 32 MFMAs of fragment set C_ with the 12 fragment reads of the NEXT sub-step (set N_: weights at LDS
  // byte offset wo_, pixel fragments of patch buffer nb_ / tap nt_) issued in their gaps, 4 weight fragments first, and -- in
  // the second sub-step of a double step -- this wave's LDS-DMA pieces (DMA_(i): piece i of the double step's NV_ pieces, KV_ per
  // group of 4 MFMAs; an LDS-DMA issue costs ~60 cycles of the wave's issue time among MFMAs: spread out, never in front of
  // them).  Program order IS the wanted order (the compiler keeps LDS-DMA and ds_read in program order: both touch LDS);
  // sched_group_barrier pins the {4 MFMA, 2 reads, KV_ DMA} x 8 interleave.  VG_ = 0: the pieces are inline-assembly loads (the
  // residual of a persistent tile), which the scheduler does not class as VMEM: no VMEM group.
#define W4_VMEM_GROUP(P_, NV_, KV_)                                                                            \
  do {                                                                                                         \
    if constexpr ((KV_) > 0 && (P_) * (KV_) < (NV_))                                                           \
      __builtin_amdgcn_sched_group_barrier(0x020, (((P_) + 1) * (KV_) <= (NV_) ? (KV_) : (NV_) - (P_) * (KV_)), 0); \
  } while (0)
#define W4_GRP(P_, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_)                                                 \
  do {                                                                                                         \
    if constexpr ((P_) < MT) {                             /* MT = 7: the eighth group does not exist */          \
      W4_MFMA_GRP(C_, P_);                                                                                     \
      constexpr int X0_ = 2 * ((P_) - 2), X1_ = 2 * ((P_) - 2) + 1;   /* this group's two pixel fragments of the next set */ \
      if constexpr ((P_) < 2) {                                                                                \
        wf[N_][2 * (P_)] = *(const frag*)(smem + (wo_) + (2 * (P_)) * 1024);                                   \
        wf[N_][2 * (P_) + 1] = *(const frag*)(smem + (wo_) + (2 * (P_) + 1) * 1024);                           \
      } else if constexpr ((P_) < 6) {                                                                         \
        if constexpr (X0_ < MT) xf[N_][X0_ < MT ? X0_ : 0] = *(const frag*)(smem + xoff[nt_][X0_ < MT ? X0_ : 0] + (nb_) * PATCH_B); \
        if constexpr (X1_ < MT) xf[N_][X1_ < MT ? X1_ : 0] = *(const frag*)(smem + xoff[nt_][X1_ < MT ? X1_ : 0] + (nb_) * PATCH_B); \
      }                                                                                                        \
      if constexpr ((KV_) > 0 && (P_) * (KV_) + 0 < (NV_)) DMA_(((P_) * (KV_) + 0));                           \
      if constexpr ((KV_) > 1 && (P_) * (KV_) + 1 < (NV_)) DMA_(((P_) * (KV_) + 1));                           \
      if constexpr ((KV_) > 2 && (P_) * (KV_) + 2 < (NV_)) DMA_(((P_) * (KV_) + 2));                           \
      __builtin_amdgcn_sched_group_barrier(0x008, W4_MFMAS_PER_GRP, 0);                                        \
      constexpr int NR_ = (P_) < 2 ? 2 : ((P_) < 6 ? (X0_ < MT ? 1 : 0) + (X1_ < MT ? 1 : 0) : 0);   /* reads of this group */ \
      if constexpr (NR_ > 0) __builtin_amdgcn_sched_group_barrier(0x100, NR_, 0);                              \
      if constexpr (VG_) W4_VMEM_GROUP(P_, NV_, KV_);                                                          \
      if constexpr (((C_) == 1 && D_ == 6) || ((C_) == 0 && D_ == 7)) W4_GSTAMP(((C_) == 1 ? (P_) : 8 + (P_))); \
    }                                                                                                          \
  } while (0)
#define W4_SUB(C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_)                                                     \
  do {                                                                                                         \
    W4_GRP(0, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); W4_GRP(1, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); \
    W4_GRP(2, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); W4_GRP(3, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); \
    W4_GRP(4, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); W4_GRP(5, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); \
    W4_GRP(6, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); W4_GRP(7, C_, N_, wo_, nb_, nt_, NV_, KV_, DMA_, VG_); \
  } while (0)

  int dn = PD;                                             // next double tile to issue
  int slot_b = 0;                                          // byte offset of the ring slot of the double tile being consumed
  int hc = 0;                                              // first half-chunk of the current body
  int res_slot[2] = {0, 0};                                // ring slots (byte offsets) that received residual pieces of pixel tiles 4, 5 / 6, 7
  bool after_epi = false;                                  // PERS: the epilogue stores of the previous tile are still in the VM queue
  const char* n_patch_src = patch_src;                     // PERS: first half-chunk of the next tile's patch (else: unused prefetch)

  // double step D of a body: sub-steps u0 = 2 D, u1 = 2 D + 1 (tap u % 9 of half-chunk u / 9); the sub-step after u1 is 2 D + 2
  // (D = 8: the next body's first).  DMA pieces of the double step: 0 .. TGW - 1 = double tile D + PD into the slot PD ahead;
  // then, at D = 0 and D = 5, the PW pieces of a patch burst.  RES, one tile per workgroup, last body: the burst of D = 5 (the next
  // tile's patch: unused) and the double tiles of D = 7, 8 (they wrap to the start of the panel: unused) carry the tile's residual
  // instead -- 8 + 4 + 4 pieces per lane, the 16 bytes each lane adds itself in the epilogue.  Every wait count below comes from
  // w4_sched.h, where the host-side schedule model (tests/host_harness, test_host.py) replays them.
#ifdef FLOPE_STAG_DBG      /* ablation (results wrong by construction): dbg & 1 = no weight DMA in the loop, dbg & 2 = no patch DMA */
#define W4_ABL_W if (!(p.dbg & 1))
#define W4_ABL_P if (!(p.dbg & 2))
#else
#define W4_ABL_W
#define W4_ABL_P
#endif
#define W4_DMA_PIECE(i_)                                                                                       \
  do {                                                                                                         \
    if constexpr ((i_) < TGW) {                                                                                \
      constexpr int rpt_ = 4 + 2 * RS_ + (((i_) >> 1) & 1);   /* the pixel tile whose residual rides here (MT = 7: none for 7) */ \
      if (RES && !PERS && lastb_ && (D_ == 7 || D_ == 8) && rpt_ < MT) {                                       \
        unsigned ro_ = ooff[(RES && rpt_ < MT) ? rpt_ : 0]; asm volatile("" : "+v"(ro_));                      \
        GLDS16(res_b + ro_ + ((i_) & 1) * 64, Bs + iss_b_ + (i_) * 4096 + wave * 1024);                        \
      } else {                                                                                                 \
        W4_ABL_W GLDS16(b_base + (size_t)di_ * DT_B + (i_) * 4096, Bs + iss_b_ + (i_) * 4096 + wave * 1024);   \
      }                                                                                                        \
    } else {                                                                                                   \
      constexpr int jr_ = (i_) - TGW + w4_patch_first(D_, PW, SPREAD);      /* index of the piece in its burst */  \
      constexpr int j_ = jr_ < 0 ? 0 : (jr_ >= PW ? PW - 1 : jr_);                                             \
      if (D_ == 0) {                                                                                           \
        W4_ABL_P GLDS16(patch_src + (hc + 1) * 64 + psrc[j_], Ps + PATCH_B + (j_ * 256 + wave * 64) * 16);     \
      } else if (RES && !PERS && lastb_ && j_ < 8) {                                                           \
        unsigned ro_ = ooff[RES ? (j_ >> 1) : 0]; asm volatile("" : "+v"(ro_));                                \
        GLDS16(res_b + ro_ + (j_ & 1) * 64, Ps + (j_ * 256 + wave * 64) * 16);                                 \
      } else {                                                                                                 \
        W4_ABL_P GLDS16((hc + 2 < nhc ? patch_src + (hc + 2) * 64 : n_patch_src) + psrc[j_], Ps + (j_ * 256 + wave * 64) * 16); \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
  // PERS + RES, last body: residual load i_ of the first sub-step of double step 7 (loads 0 .. 7) / 8 (8 .. 2 MT - 1)
#define W4_RES_PIECE(i_)                                                                                       \
  do {                                                                                                         \
    constexpr int ri_ = (D_ == 7 ? 0 : w4_res_first(MT)) + (i_), rp_ = (ri_ >> 1) < MT ? (ri_ >> 1) : 0;      \
    if (lastb_) {                                                                                              \
      const char* ra_ = res_b + ooff[RES ? rp_ : 0] + (ri_ & 1) * 64;                                          \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(rq[(RES && PERS) ? rp_ : 0][ri_ & 1]) : "v"(ra_) : "memory"); \
    }                                                                                                          \
  } while (0)
#define W4_DSTEP(D)                                                                                            \
  do {                                                                                                         \
    constexpr int D_ = (D), U1_ = 2 * (D) + 1, U2_ = (2 * (D) + 2) % 18, RS_ = (D) == 8 ? 1 : 0;               \
    /* vmcnt in front of the barrier, from the queue model (w4_sched.h): inside a tile; in the first body behind a class walk's tile \
       boundary (E: the 2 MT epilogue stores are in the queue); in a class walk's last body with a residual input (R: its register loads) */ \
    constexpr int WN_ = w4_wait_n(D_, PD, PW, SPREAD, 0, 0);                                                   \
    constexpr int WNE_ = PERS ? w4_wait_n(D_, PD, PW, SPREAD, 2 * MT, 0) : WN_;                                \
    constexpr int WNR_ = (RES && PERS) ? w4_wait_n(D_, PD, PW, SPREAD, 0, MT) : WN_;                           \
    constexpr int WNER_ = (RES && PERS) ? w4_wait_n(D_, PD, PW, SPREAD, 2 * MT, MT) : WNE_;                    \
    constexpr int NV_ = w4_pieces(D_, PW, SPREAD), KV_ = (NV_ + MT - 1) / MT;                                  \
    constexpr int NRL_ = (RES && PERS) ? w4_res_loads(D_, MT) : 0, KRL_ = (NRL_ + MT - 1) / MT;                \
    const bool lastb_ = hc + 2 >= nhc;                                                                         \
    const int next_b_ = slot_b + DT_B >= NBD * DT_B ? 0 : slot_b + DT_B;                                       \
    int wof_ = wbase + slot_b, wofn_ = wbase + next_b_;                                                        \
    asm volatile("" : "+v"(wof_), "+v"(wofn_));                                                                \
    W4_SUB(0, 1, wof_ + TILE_B, U1_ / 9, U1_ % 9, NRL_, KRL_, W4_RES_PIECE, 0);                                \
    W4_DSTAMP(2 * (D));                                                                                        \
    if (WNE_ != WN_ && after_epi) { if (WNER_ != WNE_ && lastb_) W4_WAIT_VM(WNER_); else W4_WAIT_VM(WNE_); }   \
    else if (WNR_ != WN_ && lastb_) W4_WAIT_VM(WNR_);                                                          \
    else W4_WAIT_VM(WN_);                                                                                      \
    W4_BARRIER();                                                                                              \
    W4_DSTAMP(2 * (D) + 1);                                                                                    \
    int iss_b_ = slot_b + PD * DT_B; if (iss_b_ >= NBD * DT_B) iss_b_ -= NBD * DT_B;                           \
    const int di_ = dn < ND ? dn : dn - ND;                                                                    \
    if (RES && !PERS && lastb_ && ((D) == 7 || (D) == 8)) res_slot[RS_] = iss_b_;                              \
    ++dn;                                                                                                      \
    W4_SUB(1, 0, wofn_, U2_ / 9, U2_ % 9, NV_, KV_, W4_DMA_PIECE, 1);                                          \
    slot_b = next_b_;                                                                                          \
  } while (0)

#ifdef FLOPE_STAG_DBG
  const unsigned long long st_l0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_l1 = 0, st_r1 = 0;
#endif
  char* const lscr = smem + SCR_B + wave * 3200;           // the epilogue's line image of one pixel tile (2 KB, wave-private)
  for (;;) {
  const bool has_next = PERS && mt + G_mt < p.mtiles;
  if constexpr (PERS) n_patch_src = has_next ? patch_src + d_patch : patch_src;
  for (int hcp = 0; hcp < nbody; ++hcp) {
    W4_DSTEP(0); W4_DSTEP(1); W4_DSTEP(2); W4_DSTEP(3); W4_DSTEP(4); W4_DSTEP(5); W4_DSTEP(6); W4_DSTEP(7); W4_DSTEP(8);
    hc += 2;
    after_epi = false;
  }

#ifdef FLOPE_STAG_DBG
  if (st_l1 == 0) { st_l1 = __builtin_amdgcn_s_memtime(); st_r1 = __builtin_amdgcn_s_memrealtime(); }
  else if (st_b[5] == 0) st_b[5] = __builtin_amdgcn_s_memtime();      // end of the SECOND tile's loop
#endif
  W4_BSTAMP(0);
  // ---- epilogue: (+ residual) (ReLU) -> 16-bit padded NHWC.  A lane's accumulators are one pixel x two runs of 8 channels (MFMA
  // order); the packed words go through the wave's 2 KB line image and leave as whole 128-byte lines (header).  The image is
  // wave-private and the LDS operations of a wave execute in order: no barrier.
  if constexpr (!PERS) W4_WAIT_VM(0);                      // look-ahead DMAs (and the residual rounds) have landed
  else if constexpr (RES) W4_WAIT_VM(w4_pieces(8, PW, SPREAD));   // the residual loads have; the pieces of double step 8 stay in flight
  if constexpr (RES && PERS) {
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) asm volatile("" : "+a"(rq[pt][0]), "+a"(rq[pt][1]));   // no read of them is scheduled above the wait
  }
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    float v[NT * 4];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[ct * 4 + q] = W4_ACCQ(pt, ct, q);
    if constexpr (RES) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 rv;
        if constexpr (PERS) rv = rq[PERS ? pt : 0][c];
        else {                                             // this lane's own DMA pieces (no other wave reads them)
          const int off_ = pt < 4 ? ((pt * 2 + c) * 256 + wave * 64 + lane) * 16
                                  : 2 * PATCH_B + res_slot[(pt - 4) >> 1] + (((pt - 4) & 1) * 2 + c) * 4096 + wave * 1024 + lane * 16;
          rv = *(const u32x4*)(smem + off_);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[c * 8 + q * 2] += unpack_lo<T>(rv[q]);
          v[c * 8 + q * 2 + 1] += unpack_hi<T>(rv[q]);
        }
      }
    }
    u32x4 o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) o[c][q] = pk_out16<T>(pack2<T>(v[c * 8 + q * 2], v[c * 8 + q * 2 + 1]), p.relu);
#pragma unroll
    for (int c = 0; c < 2; ++c) *(u32x4*)(lscr + r16 * 128 + (((c * 4 + g) ^ (r16 & 7)) << 4)) = o[c];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int pp = c * 8 + (lane >> 3);
      o[c] = *(const u32x4*)(lscr + pp * 128 + (((lane & 7) ^ (pp & 7)) << 4));
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (okl[pt][c]) *(u32x4*)((char*)p.out + osto[pt][c]) = o[c];
  }
  W4_BSTAMP(1);
  if (!has_next) break;
  // ---- PERS: next tile of this workgroup's class walk.  Its first half-chunk (burst of double step 5 of the last body), its first
  // double tiles and -- same table -- the fragments of its first sub-step are in LDS, in flight or in registers; the boundary is
  // pointer bumps and the accumulators.  What is younger than that burst in this wave's queue: the double tiles of double steps
  // 6, 7, 8 and the 2 MT stores just issued (w4_after_epi_extra in the first waits of the next body).
  {
    mt += G_mt; patch_src = n_patch_src;
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) { osto[pt][0] += d_out; osto[pt][1] += d_out; }
    if constexpr (RES) {
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) ooff[pt] += d_out;
    }
    W4_BSTAMP(2);
    W4_ACC_INIT();
    W4_BSTAMP(3);
    after_epi = true;
    dn -= ND;
    hc = 0;
    if constexpr (DSF) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dsrc[rr] += d_ds;
      W4_BARRIER();                                        // every wave is done with patch buffer 1 (the last body's second half-chunk)
      W4_DS_CHAIN(false);
      after_epi = false;                                   // the chain's vmcnt(0) drained the queue
    }
    if constexpr (DSF || RES) {                            // the chain used the fragment registers; RES: not carried across the epilogue (registers)
      int wof0_ = wbase + slot_b;
      asm volatile("" : "+v"(wof0_));
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) wf[0][ct] = *(const frag*)(smem + wof0_ + ct * 1024);
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) xf[0][pt] = *(const frag*)(smem + xoff[0][pt]);
    }
    W4_BSTAMP(4);
#ifdef FLOPE_STAG_DBG
    ++st_tile;
#endif
  }
  }   // tile walk
  if constexpr (PERS) W4_WAIT_VM(0);                       // the wrapped-around look-ahead DMAs land before LDS is released
#ifdef FLOPE_STAG_DBG
  if ((p.dbg & 64) && p.split_ws && wave == 0 && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* d_ = (unsigned long long*)p.split_ws + (size_t)blockIdx.x * 8;
    d_[0] = st_e0; d_[1] = st_l0; d_[2] = st_l1; d_[3] = __builtin_amdgcn_s_memtime(); d_[4] = st_r0; d_[5] = st_r1;
    unsigned long long* q_ = (unsigned long long*)((char*)p.split_ws + 65536) + (size_t)blockIdx.x * 4;
    q_[0] = st_p[0]; q_[1] = st_p[1]; q_[2] = st_p[2]; q_[3] = st_p[3];
    unsigned long long* b_ = (unsigned long long*)((char*)p.split_ws + 131072) + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 6; ++i) b_[i] = st_b[i];
    b_[6] = st_l1;
  }
  if ((p.dbg & 64) && p.split_ws && wave == 0 && lane < 18) ((unsigned*)((char*)p.split_ws + 196608))[(size_t)blockIdx.x * 32 + lane] = st_v;
  if ((p.dbg & 512) && p.split_ws && wave == 0 && lane < 16) ((unsigned*)((char*)p.split_ws + 327680))[(size_t)blockIdx.x * 16 + lane] = st_w;
#endif
#undef W4_PSTAMP
#undef W4_BSTAMP
#undef W4_DSTAMP
#undef W4_GSTAMP
#undef W4_ACC_INIT
#undef W4_ACCQ
#undef W4_MFMA_GRP
#undef W4_MFMAS_PER_GRP
#undef W4_DS_CHAIN
#undef W4_DSTEP
#undef W4_ISSUE_DS
#undef W4_DMA_PIECE
#undef W4_RES_PIECE
#undef W4_ABL_W
#undef W4_ABL_P
#undef W4_VMEM_GROUP
#undef W4_SUB
#undef W4_GRP
#undef W4_ISSUE_DT
#undef W4_ISSUE_PATCH
#undef W4_BARRIER
#undef W4_WAIT_VM
#undef W4_WAIT_PIN
}

static constexpr size_t w4_lds_bytes(int pt, bool own_ds_slot) {
  return (size_t)2 * pt * 8192 + (size_t)(w4_ring(pt, own_ds_slot) + (own_ds_slot ? 1 : 0)) * 16384 + 12800;
}

// instantiated variants: one tile per workgroup at MT = 8, 7 (PT = 4, 5, 6) and MT = 6, 5 (PT = 4; r05: 4 dropped, the planner's floor is 5); the class walk at MT = 7
// (PT = 4, 5; with a folded shortcut the 6-round patch leaves no room for the shortcut's own weight slot)
#define W4_FOR_VARIANTS(X, T)                                                                                  \
  X(T, 4, false, false, false, 8) X(T, 4, true, false, false, 8) X(T, 4, false, true, false, 8)                \
  X(T, 5, false, false, false, 8) X(T, 5, true, false, false, 8) X(T, 5, false, true, false, 8)                \
  X(T, 6, false, false, false, 8) X(T, 6, true, false, false, 8) X(T, 6, false, true, false, 8)                \
  X(T, 4, false, false, false, 7) X(T, 4, true, false, false, 7) X(T, 4, false, true, false, 7)                \
  X(T, 5, false, false, false, 7) X(T, 5, true, false, false, 7) X(T, 5, false, true, false, 7)                \
  X(T, 6, false, false, false, 7) X(T, 6, true, false, false, 7) X(T, 6, false, true, false, 7)                \
  X(T, 4, false, false, false, 6) X(T, 4, true, false, false, 6) X(T, 4, false, true, false, 6)                \
  X(T, 4, false, false, false, 5) X(T, 4, true, false, false, 5) X(T, 4, false, true, false, 5)                \
  X(T, 4, false, false, true, 7) X(T, 4, true, false, true, 7) X(T, 4, false, true, true, 7)                   \
  X(T, 5, false, false, true, 7) X(T, 5, true, false, true, 7) X(T, 5, false, true, true, 7)

extern "C" int flope_conv_w4_init() {
  hipError_t e = hipSuccess;
#define A(T, PT_, RES_, DSF_, PERS_, MT_)                                                                      \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_w4_kernel<T, PT_, RES_, DSF_, PERS_, MT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  W4_FOR_VARIANTS(A, bf16_t)
  W4_FOR_VARIANTS(A, f16_t)
#undef A
  return (int)e;
}

// lds bytes of a variant: 2 patch buffers, the weight ring (+ 1 double tile for a persistent workgroup's folded shortcut), 12.5 KB in
// which the lanes trade their shares of the address table and the epilogue builds its line image; 0 = not instantiated
extern "C" size_t flope_conv_w4_lds(int pt, int mt, int dsf, int pers) {
  if (pt < 4 || pt > 6 || mt < 5 || mt > 8 || (mt < 7 && pt != 4) || (pers && (mt != 7 || pt > 5))) return 0;
  return w4_lds_bytes(pt, dsf && pers);
}

// 3x3 stride-1 pad-1, Cin % 64 == 0, Cout % 128 == 0, the skewed patch image (p->skew, p->mg_pitch / sh_pitch):
// mt = 8 .. 4 pixel tiles per wave = 256 .. 128-pixel workgroup tiles (below 7: pt = 4 only); p->patch_rows_max = PT (4, 5 or 6: the
// patch of such a tile), p->mtiles = ceil(M / (32 mt)), p->total_tiles = mtiles * Cout / 128, p->w the conv_stag weight image.
// grid_blocks = total_tiles: one tile per workgroup.  Fewer = the class walk (mt = 7, pt <= 5): grid_blocks = G * Cout / 128 where G
// divides mtiles, M % 224 == 0 and G tiles are p->cw_imgs whole images (G * 224 == cw_imgs * Ho * Wo).
extern "C" int flope_conv_w4_launch(const ConvP* p, int dtype, int grid_blocks, int mt, void* stream) {
  const int pt = p->patch_rows_max;
  const bool pers = grid_blocks < p->total_tiles;
  if (mt < 5 || mt > 8 || p->mtiles != (p->M + 32 * mt - 1) / (32 * mt) || p->stride != 1 || p->ntaps != 9 || p->Cin % 64 || p->Cout % 128 ||
      !p->skew || !p->mg_pitch || p->ksplit > 1 || (p->res && p->ds_in) || (p->ds_in && (p->ds_Cin % 64 || !p->ds_w)) ||
      grid_blocks < p->ntiles || grid_blocks > p->total_tiles || grid_blocks % p->ntiles || p->total_tiles != p->mtiles * p->ntiles)
    return (int)hipErrorInvalidValue;
  if (pers) {
    const int G = grid_blocks / p->ntiles;
    if (mt != 7 || p->M % 224 || p->mtiles % G || p->cw_imgs < 1 || (long)G * 224 != (long)p->cw_imgs * p->Ho * p->Wo) return (int)hipErrorInvalidValue;
  }
  const size_t lds = flope_conv_w4_lds(pt, mt, p->ds_in != nullptr, pers);
  if (lds == 0 || lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_blocks), block(256);
  const int res = p->res ? 1 : 0, dsf = p->ds_in ? 1 : 0;
  bool done = false;
#define L(T, PT_, RES_, DSF_, PERS_, MT_)                                                                      \
  if (!done && pt == PT_ && res == (RES_ ? 1 : 0) && dsf == (DSF_ ? 1 : 0) && pers == PERS_ && mt == MT_) {     \
    hipLaunchKernelGGL((conv_w4_kernel<T, PT_, RES_, DSF_, PERS_, MT_>), grid, block, lds, st, *p); done = true; \
  }
  if (dtype == 0) { W4_FOR_VARIANTS(L, bf16_t) } else { W4_FOR_VARIANTS(L, f16_t) }
#undef L
  if (!done) return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}
