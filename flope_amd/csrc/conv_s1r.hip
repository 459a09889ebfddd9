// conv_s1r: the 3x3 stride-1 128 -> 128 convolutions of layer 2 on the 28 x 28 map (layer2.1.conv1 / conv2) with ALL weights in
// registers and K split over wave pairs (r05).
//
// Reference: torchvision ResNet-18 BasicBlock of layer2[1] (conv + bn (+ identity shortcut) + ReLU), as instantiated by
// /root/reference/sunflower/models/posenet.py:26-31; BN folded at load time (engine.hip load_weights).
//
// conv_s2r (layer2.0.conv1) showed what a step stream with two waves per SIMD and no weight traffic reaches.  K = 9 x 128 = 1152
// here: 32 output channels x 1152 would be 288 weight registers per wave.  So a wave owns 32 output channels (cg = w & 3) and ONE
// HALF of the input channels (kh = w >> 2: channels 64 kh .. 64 kh + 63): 18 steps x 2 channel tiles = 144 VGPRs, exactly conv_s2r's
// budget.  Both waves of a pair (cg, 0) / (cg, 1) run the pixel tiles of the workgroup's band (4 output rows x 28 columns of one
// image; four pixel tiles, then three) over their half of K; then they swap partial sums through LDS -- each wave finishes a share of
// the pixel tiles: the result is (bias + partial of kh 0) + partial of kh 1 for every output, in that order, whichever wave
// finishes it (wave kh = 0 starts every accumulator from the bias, wave kh = 1 from zero).
// LDS: the tile's 6 x 30-pixel input patch as four images [kh][32-channel half hc] of 64-byte pixels (slot swizzle by the row;
// the lane -> pixel map puts even columns on lanes r16 in {0-3, 12-15} and odd ones on {4-11}: conflict-free ds_read_b128 for every
// tap), two tile buffers of 48 KB filled by LDS-DMA one tile ahead (pieces in the first three steps), + 56 KB for the swap.
// One barrier per sub-tile (two per band): swap data visible; the second one also publishes the next patch and releases this one.
// K order: input-channel half, 32-channel half-chunk, tap, channel: results equal conv_w4's within accumulation-order rounding.
#include "common.h"
#include <type_traits>

namespace {

// LDS-DMA piece by inline assembly: 16 bytes per lane from the lane's own global address to LDS address m0 + 16 * lane.
// Not the builtin: while hipcc's wait-count pass knows of an outstanding global_load_lds it turns EVERY wait it inserts into a
// wait for zero -- `s_waitcnt lgkmcnt(0)` in front of each step's first MFMA, i.e. behind the fragment reads issued a few cycles
// before (r05: one exposed LDS round trip per step).  Hidden from it, the fragment waits are counted (`lgkmcnt(N)`); the pieces'
// own completion is this kernel's business either way (manual `s_waitcnt vmcnt` + barrier, "memory" clobbers on both).
__device__ __forceinline__ void glds16(const char* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}
#define GLDS16(gptr, lptr) glds16((gptr), (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) char*)(lptr)))

// FOLD (layer2.0.conv2): the block's 1x1 stride-2 shortcut conv as one more step -- out += W_ds . x(2 r, 2 c) over the 64 channels of the
// block input x, wave kh taking channels 32 kh .. 32 kh + 31 (one more A fragment pair); its pixel fragments come straight from global
// memory (16 bytes per lane, issued five steps before the extra step: x is 14 KB per band and there is no LDS left for it).
template <typename T, bool RES, bool FOLD>
__global__ __launch_bounds__(512, 1) void conv_s1r_kernel(const ConvP p, const u32x4* __restrict__ wpk) {
  static_assert(!(RES && FOLD), "the folded shortcut replaces the residual");
  typedef typename Elem<T>::frag frag;
  constexpr int NPT = 7, WO = 28;
  constexpr int PROW_B = 32 * 64;          // image row pitch: 32 pixels (30 used) of 64 bytes
  constexpr int IMG_B = 6 * PROW_B;        // 12288: one (kh, hc) image of a tile: 6 patch rows
  constexpr int BUF_B = 4 * IMG_B;         // 49152
  constexpr int SWAP_OFF = 2 * BUF_B;      // swap area: 4 pairs x 56 accumulator registers x 256 B
  constexpr int NSTEP = 18;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int cg = wave & 3, kh = wave >> 2;
  const int G = gridDim.x, total = p.B * (p.Ho >> 2), rgs = p.Ho >> 2;

  // ---- LDS-DMA map.  A piece = 16 pixels x 64 B of one image row.  Wave w moves image (kh' = w >> 2, hc' = (w >> 1) & 1), pixel half
  // h = w & 1 of every patch row; lane -> pixel 16 h + (lane >> 2), LDS slot lane & 3 <- source slot (lane & 3) ^ (row & 3).
  int dso[4];
  {
    const int px = 16 * (wave & 1) + (lane >> 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) dso[k] = min(px, WO + 1) * 256 + (wave >> 1) * 64 + (((lane & 3) ^ k) << 4);
  }
  const int wrow = p.Wip * 256;
  char* const dbase = smem + (wave >> 1) * IMG_B + (wave & 1) * 1024;
#define S1R_PIECE(src_, buf_, k_) GLDS16((src_) + (k_) * wrow + dso[(k_) & 3], dbase + (buf_) * BUF_B + (k_) * PROW_B)

  // ---- fragment read addresses (see conv_s2r.hip for the lane groups of ds_read_b128)
  const int i8 = r16 < 4 ? r16 : (r16 < 12 ? r16 - 4 : r16 - 8);
  const int rr = i8 >> 1, cc = 2 * (i8 & 1) + ((r16 >= 4 && r16 < 12) ? 1 : 0);
  int rd[3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) rd[dy] = kh * 2 * IMG_B + (rr + dy) * PROW_B + cc * 64 + ((g ^ ((rr + dy) & 3)) << 4);
  const int ooff = ((rr * p.Wop + cc) * p.Cout + 32 * cg + 8 * g) * 2;

#ifdef FLOPE_STAG_DBG
  const unsigned long long t_entry_ = __builtin_amdgcn_s_memtime(), rt_entry_ = __builtin_amdgcn_s_memrealtime();
#endif
  int tile = blockIdx.x;
  if (tile >= total) return;
  auto band = [&](int tl) -> const char* {          // the tile's input band: padded rows 4 rg .. 4 rg + 5 of its image
    tl = min(tl, total - 1);
    const int img = tl / rgs, rg = tl - img * rgs;
    return (const char*)p.in + ((size_t)img * p.Hip + 4 * rg) * p.Wip * 256;
  };
#pragma unroll
  for (int k = 0; k < 6; ++k) S1R_PIECE(band(tile), 0, k);

  // ---- this wave's weights (36 A fragments) and bias: plain loads, settled before the loop by the empty asm "uses" (conv_s2r.hip)
  const u32x4* const wl = wpk + (size_t)(cg * 2 + kh) * NSTEP * 2 * 64 + lane;
  frag wres[NSTEP][2];
  f32x4 b4[2];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wres[s][ct] = __builtin_bit_cast(frag, wl[(s * 2 + ct) * 64]);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) b4[ct] = *(const f32x4*)(p.bias + 32 * cg + 8 * g + 4 * ct);
  frag wds[2];
  if constexpr (FOLD) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wds[ct] = __builtin_bit_cast(frag, ((const u32x4*)p.ds_w)[((cg * 2 + kh) * 2 + ct) * 64 + lane]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(wds[ct]));
  }
  // shortcut pixel of output (rr, cc) of pixel tile 0: padded x(2 rr + 1, 2 cc + 1), this wave's 32 channels, this lane's 8
  const int xsoff = FOLD ? ((2 * rr + 1) * p.ds_Wip + 2 * cc + 1) * 128 + kh * 64 + g * 16 : 0;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(wres[s][ct]));
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(b4[ct]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#ifdef FLOPE_STAG_DBG
  const unsigned long long t_loop_ = __builtin_amdgcn_s_memtime();
#endif
  // swap area of this wave's pair: 14 slots of [64 lanes] float4: slot j of lane l at (j * 64 + l) * 16 bytes
  char* const swp = smem + SWAP_OFF + cg * (14 * 1024) + lane * 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  const char* nsrc = nullptr;
  const char* xsrc = nullptr;                      // FOLD: the band's rows of the block input (+ this lane's offset)
  size_t opix = 0;

  // One SUB-TILE = NP of the band's seven pixel tiles (4, then 3: seven at once need 56 accumulator + 28 fragment registers beside the
  // 144 of the weights -- 46 spilled), both waves of a pair over their half of K; then the swap: the wave finishes NF of the NP and
  // sends the others' partial sums to its partner.  One barrier per sub-tile: the two sub-tiles use different slots of the swap
  // area, so a slot is rewritten only behind the OTHER sub-tile's barrier, which the partner passes after it has read the slot.
#ifdef FLOPE_STAG_DBG
  // diagnostic build, dbg & 64: shader-clock stamps of this workgroup's SECOND band, wave 0 (tools/clock_probe_s1r.py)
  unsigned long long stp[12] = {0};
  int st_it = 0;
#define S1R_STAMP(i_) do { if ((p.dbg & 64) && st_it == 1) { __builtin_amdgcn_sched_barrier(0); stp[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define S1R_STAMP(i_) do {} while (0)
#endif
  auto sub = [&](auto kh_, auto p0_, auto np_) {
    constexpr int KH = decltype(kh_)::value, P0 = decltype(p0_)::value, NP = decltype(np_)::value;
    constexpr bool FIRST = P0 == 0;
    constexpr int NF = KH ? NP - 2 : 2, F0 = KH ? 2 : 0;     // local pixel tiles this wave finishes: kh 0 the first two, kh 1 the rest
    constexpr int NS = NP - NF, S0 = KH ? 0 : 2;             // ... and sends
    constexpr int WSLOT = (FIRST ? 0 : 8) + (KH ? (FIRST ? 4 : 2) : 0);      // slots this wave writes; the partner's: RSLOT
    constexpr int RSLOT = (FIRST ? 0 : 8) + (KH ? 0 : (FIRST ? 4 : 2));
    char* const obase = (char*)p.out + opix;
    const char* const rbase = RES ? (const char*)p.res + opix : nullptr;
    char* const xbuf = smem + cur * BUF_B + P0 * 256;
    const int nbuf = cur ^ 1;
    f32x4 acc[NP][2];
    frag xf[2][NP];
    frag xs[NP];
    S1R_STAMP(FIRST ? 0 : 5);
    auto xaddr = [&](int s) -> const char* {   // pixel fragments of step s (half-chunk s / 9, tap s % 9), the sub-tile's first pixel tile
      const int hc = s / 9, tap = s - 9 * hc, ky = tap / 3, kx = tap - 3 * ky;
      return xbuf + hc * IMG_B + rd[ky] + kx * 64;
    };
#pragma unroll
    for (int pt = 0; pt < NP; ++pt) xf[0][pt] = *(const frag*)(xaddr(0) + pt * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[pt][ct] = Elem<T>::mfma(wres[s][ct], xf[s & 1][pt], s == 0 ? (KH ? zero4 : b4[ct]) : acc[pt][ct]);
        if (s + 1 < NSTEP) xf[(s + 1) & 1][pt] = *(const frag*)(xaddr(s + 1) + pt * 256);
        if (FIRST && s < 3 && pt < 2) S1R_PIECE(nsrc, nbuf, 2 * s + pt);       // the next band's six pieces
        if (FOLD && s == 12) xs[pt] = *(const frag*)(xsrc + (P0 + pt) * 1024);   // 4 columns = 8 pixels of x = 1024 bytes per pixel tile
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (s + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if ((FIRST && s < 3 && pt < 2) || (FOLD && s == 12)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    if constexpr (FOLD) {
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = Elem<T>::mfma(wds[ct], xs[pt], acc[pt][ct]);
    }
    S1R_STAMP(FIRST ? 1 : 6);
    // the partner's share of the partial sums -> swap area; the residual of the own share comes in meanwhile
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) *(f32x4*)(swp + ((WSLOT + i * 2 + ct) * 64) * 16) = acc[S0 + i][ct];
    u32x4 rq[NF];
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < NF; ++i) rq[i] = *(const u32x4*)(rbase + (P0 + F0 + i) * 4 * p.Cout * 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // second sub-tile: this wave's pieces of the next band (issued a whole sub-tile ago) must have landed before the barrier that
    // publishes them; loads return in order and only the residual loads are younger
    if constexpr (!FIRST) { if constexpr (RES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    S1R_STAMP(FIRST ? 2 : 7);
#ifdef FLOPE_STAG_DBG
    if ((p.dbg & 64) && st_it == 1 && p.split_ws && lane == 0)      // every wave's arrival at the barrier
      ((unsigned long long*)p.split_ws)[8192 + (size_t)blockIdx.x * 16 + (FIRST ? 0 : 8) + wave] = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();                    // swap data visible (second sub-tile: + next patch visible, this patch released)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    S1R_STAMP(FIRST ? 3 : 8);
    // own share: (bias + partial of kh 0) + partial of kh 1
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      u32x4 o;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 pa = *(const f32x4*)(swp + ((RSLOT + i * 2 + ct) * 64) * 16);
        f32x4 v = KH ? pa + acc[F0 + i][ct] : acc[F0 + i][ct] + pa;
        if constexpr (RES) {
          v[0] += unpack_lo<T>(rq[i][2 * ct]); v[1] += unpack_hi<T>(rq[i][2 * ct]);
          v[2] += unpack_lo<T>(rq[i][2 * ct + 1]); v[3] += unpack_hi<T>(rq[i][2 * ct + 1]);
        }
        o[2 * ct] = pk_out16<T>(pack2<T>(v[0], v[1]), p.relu);
        o[2 * ct + 1] = pk_out16<T>(pack2<T>(v[2], v[3]), p.relu);
      }
      *(u32x4*)(obase + (P0 + F0 + i) * 4 * p.Cout * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
    S1R_STAMP(FIRST ? 4 : 9);
  };
  auto run = [&](auto kh_) {
    for (; tile < total; tile += G) {
      nsrc = band(tile + G);
      const int img = tile / rgs, rg = tile - img * rgs;
      if constexpr (FOLD) xsrc = (const char*)p.ds_in + ((size_t)img * p.ds_Hip + 8 * rg) * p.ds_Wip * 128 + xsoff;
      opix = (((size_t)img * p.Hop + 4 * rg + 1) * p.Wop + 1) * p.Cout * 2 + ooff;
      sub(kh_, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
      sub(kh_, std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
#ifdef FLOPE_STAG_DBG
      if ((p.dbg & 64) && st_it == 1 && p.split_ws && tid == 0) {
        unsigned long long* d_ = (unsigned long long*)p.split_ws + (size_t)blockIdx.x * 16;
        for (int i = 0; i < 10; ++i) d_[i] = stp[i];
      }
      ++st_it;
#endif
      cur ^= 1;
    }
  };
  if (kh == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
#ifdef FLOPE_STAG_DBG
  if ((p.dbg & 64) && p.split_ws && tid == 0) {
    unsigned long long* d_ = (unsigned long long*)p.split_ws + (size_t)blockIdx.x * 16;
    d_[10] = t_entry_; d_[11] = t_loop_; d_[12] = __builtin_amdgcn_s_memtime(); d_[13] = rt_entry_; d_[14] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead pieces of the tile past the end land before the LDS is released
#undef S1R_PIECE
#undef S1R_STAMP
}

}  // namespace

// layer shapes this kernel takes: 3x3 stride 1, 128 -> 128 channels, 28-wide map with a multiple of 4 rows; a folded 1x1 stride-2
// shortcut over 64 input channels (ds_w: pack_s1r_ds image) instead of a residual
extern "C" int flope_conv_s1r_ok(const ConvP* p) {
  if (p->ds_in && !(p->ds_Cin == 64 && p->ds_Hip == 2 * p->Ho + 2 && p->ds_Wip == 2 * p->Wo + 2 && p->ds_w && !p->res)) return 0;
  return p->stride == 1 && p->ntaps == 9 && p->Cin == 128 && p->Cout == 128 && p->Wo == 28 && (p->Ho & 3) == 0 &&
         p->ksplit <= 1 && p->Wip == p->Wo + 2 && p->Hip == p->Ho + 2;
}

extern "C" int flope_conv_s1r_lds() { return 2 * 4 * 6 * 32 * 64 + 4 * 14 * 1024; }

extern "C" int flope_conv_s1r_init() {
  hipError_t e = hipSuccess;
#define A(T, R, F) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_s1r_kernel<T, R, F>, hipFuncAttributeMaxDynamicSharedMemorySize, flope_conv_s1r_lds());
  A(bf16_t, false, false) A(bf16_t, true, false) A(bf16_t, false, true) A(f16_t, false, false) A(f16_t, true, false) A(f16_t, false, true)
#undef A
  return (int)e;
}

// w: pack_s1r image.  grid: workgroups (one per CU: 152 KB of LDS); each walks tiles blockIdx.x + k * grid of batch * Ho / 4.
extern "C" int flope_conv_s1r_launch(const ConvP* p, const void* w, int dtype, int grid, void* stream) {
  if (!flope_conv_s1r_ok(p) || !w) return (int)hipErrorInvalidValue;
  const int total = p->B * (p->Ho >> 2);
  if (grid > total) grid = total;
  if (grid < 1) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)flope_conv_s1r_lds();
  hipStream_t st = (hipStream_t)stream;
#define GO(T) do { if (p->ds_in) hipLaunchKernelGGL((conv_s1r_kernel<T, false, true>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w);  \
                   else if (p->res) hipLaunchKernelGGL((conv_s1r_kernel<T, true, false>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w);  \
                   else hipLaunchKernelGGL((conv_s1r_kernel<T, false, false>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w); } while (0)
  if (dtype == 0) GO(bf16_t); else GO(f16_t);
#undef GO
  return (int)hipGetLastError();
}
