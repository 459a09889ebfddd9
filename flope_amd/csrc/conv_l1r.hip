// conv_l1r: the four 3x3 stride-1 64 -> 64 convolutions of layer 1 on the 56 x 56 map with ALL weights in registers (r05).
//
// Reference: torchvision ResNet-18 BasicBlocks of layer1 (conv + bn (+ identity shortcut) + ReLU), as instantiated by
// /root/reference/sunflower/models/posenet.py:26-31; BN folded at load time (engine.hip load_weights).
//
// conv_s2r / conv_s1r (layer 2) showed what a step stream with two waves per SIMD and no weight traffic reaches.  Here K = 9 x 64 = 576
// and N = 64: a wave owns 32 output channels (cg = w & 1) over the WHOLE K -- 18 steps x 2 channel tiles = 144 VGPRs, the same budget --
// and a quarter of the band's pixel tiles (pg = w >> 1).  A band is 8 output rows x 56 columns of one image = 28 pixel tiles of 4 x 4;
// wave (cg, pg) runs the seven tiles of row group pg >> 1, column half pg & 1 as two sub-tiles (4 + 3: seven at once do not fit the
// register file beside the weights).  No partial sums to swap: one barrier per band.
// LDS: the band's 10 x 58-pixel input patch as two images [32-channel half hc] of 64-byte pixels at a pitch of 64 (slot swizzle by the
// row; even / odd columns of a pixel tile on the two lane sets of a ds_read_b128 group -- conv_s2r.hip), two band buffers of 80 KB
// (all of the CU's LDS) filled by LDS-DMA one band ahead; conv2 adds the residual in the epilogue.
// K order: 32-channel half-chunk, tap, channel -- the order of conv_r4 / conv_stag; the accumulators start from the bias and the
// residual is added last, as there: results are bit-identical to conv_r4's (tests/test_gpu_parity.py).
#include "common.h"
#include <type_traits>

namespace {

// LDS-DMA piece by inline assembly (see conv_s2r.hip: hidden from hipcc's wait-count pass, so that the fragment waits stay counted)
__device__ __forceinline__ void glds16(const char* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}
#define GLDS16(gptr, lptr) glds16((gptr), (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) char*)(lptr)))

template <typename T, bool RES>
__global__ __launch_bounds__(512, 1) void conv_l1r_kernel(const ConvP p, const u32x4* __restrict__ wpk) {
  typedef typename Elem<T>::frag frag;
  constexpr int WO = 56;
  constexpr int PROW_B = 64 * 64;          // image row pitch: 64 pixels (58 used) of 64 bytes
  constexpr int IMG_B = 10 * PROW_B;       // 40960: one 32-channel image of a band: 10 patch rows
  constexpr int BUF_B = 2 * IMG_B;         // 81920
  constexpr int NSTEP = 18;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int cg = wave & 1, pg = wave >> 1, rgp = pg >> 1, chf = pg & 1;
  const int G = gridDim.x, bands = p.Ho >> 3, total = p.B * bands;

  // ---- LDS-DMA map.  A piece = 16 pixels x 64 B of one image row.  Wave w moves image hc' = w >> 2, pixel quarter q = w & 3 of every
  // patch row; lane -> pixel 16 q + (lane >> 2), LDS slot lane & 3 <- source slot (lane & 3) ^ (row & 3).
  int dso[4];
  {
    const int px = 16 * (wave & 3) + (lane >> 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) dso[k] = min(px, WO + 1) * 128 + (wave >> 2) * 64 + (((lane & 3) ^ k) << 4);
  }
  const int wrow = p.Wip * 128;
  char* const dbase = smem + (wave >> 2) * IMG_B + (wave & 3) * 1024;
#define L1R_PIECE(src_, buf_, k_) GLDS16((src_) + (k_) * wrow + dso[(k_) & 3], dbase + (buf_) * BUF_B + (k_) * PROW_B)

  // ---- fragment read addresses (lane -> pixel of a 4 x 4 tile: conv_s2r.hip)
  const int i8 = r16 < 4 ? r16 : (r16 < 12 ? r16 - 4 : r16 - 8);
  const int rr = i8 >> 1, cc = 2 * (i8 & 1) + ((r16 >= 4 && r16 < 12) ? 1 : 0);
  int rd[3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) rd[dy] = (4 * rgp + rr + dy) * PROW_B + (28 * chf + cc) * 64 + ((g ^ ((rr + dy) & 3)) << 4);
  const int ooff = (((4 * rgp + rr) * p.Wop + 28 * chf + cc) * p.Cout + 32 * cg + 8 * g) * 2;

  int tile = blockIdx.x;
  if (tile >= total) return;
  auto band = [&](int tl) -> const char* {          // the band's input rows: padded rows 8 j .. 8 j + 9 of its image
    tl = min(tl, total - 1);
    const int img = tl / bands, j = tl - img * bands;
    return (const char*)p.in + ((size_t)img * p.Hip + 8 * j) * p.Wip * 128;
  };
#pragma unroll
  for (int k = 0; k < 10; ++k) L1R_PIECE(band(tile), 0, k);

  // ---- this wave's weights (36 A fragments) and bias: plain loads, settled before the loop by the empty asm "uses" (conv_s2r.hip)
  const u32x4* const wl = wpk + (size_t)cg * NSTEP * 2 * 64 + lane;
  frag wres[NSTEP][2];
  f32x4 b4[2];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wres[s][ct] = __builtin_bit_cast(frag, wl[(s * 2 + ct) * 64]);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) b4[ct] = *(const f32x4*)(p.bias + 32 * cg + 8 * g + 4 * ct);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(wres[s][ct]));
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(b4[ct]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int cur = 0;
  const char* nsrc = nullptr;
  size_t opix = 0;
  // One SUB-TILE = NP of the wave's seven pixel tiles: 18 steps of 2 NP MFMAs with the next step's NP fragment reads; the first
  // sub-tile also issues this wave's ten pieces of the NEXT band's patch (into the other buffer, free since the last barrier).
  auto sub = [&](auto p0_, auto np_) {
    constexpr int P0 = decltype(p0_)::value, NP = decltype(np_)::value;
    constexpr bool FIRST = P0 == 0;
    char* const obase = (char*)p.out + opix;
    const char* const rbase = RES ? (const char*)p.res + opix : nullptr;
    char* const xbuf = smem + cur * BUF_B + P0 * 256;
    const int nbuf = cur ^ 1;
    f32x4 acc[NP][2];
    frag xf[2][NP];
    u32x4 rq[NP];
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
        for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = Elem<T>::mfma(wres[s][ct], xf[s & 1][pt], s == 0 ? b4[ct] : acc[pt][ct]);
        if (s + 1 < NSTEP) xf[(s + 1) & 1][pt] = *(const frag*)(xaddr(s + 1) + pt * 256);
        if (FIRST && s < 5 && pt < 2) L1R_PIECE(nsrc, nbuf, 2 * s + pt);       // the next band's ten pieces
        if (RES && s == 14) rq[pt] = *(const u32x4*)(rbase + (P0 + pt) * 4 * p.Cout * 2);   // the residual: three steps before its use
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (s + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if ((FIRST && s < 5 && pt < 2) || (RES && s == 14)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    // second sub-tile: this wave's pieces of the next band (issued a whole sub-tile ago) must have landed before the barrier that
    // publishes them (they are older than everything else in flight: vmcnt(0) also covers the residual loads, needed now anyway)
    if constexpr (!FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      u32x4 o;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 v = acc[i][ct];
        if constexpr (RES) {
          v[0] += unpack_lo<T>(rq[i][2 * ct]); v[1] += unpack_hi<T>(rq[i][2 * ct]);
          v[2] += unpack_lo<T>(rq[i][2 * ct + 1]); v[3] += unpack_hi<T>(rq[i][2 * ct + 1]);
        }
        o[2 * ct] = pk_out16<T>(pack2<T>(v[0], v[1]), p.relu);
        o[2 * ct + 1] = pk_out16<T>(pack2<T>(v[2], v[3]), p.relu);
      }
      *(u32x4*)(obase + (P0 + i) * 4 * p.Cout * 2) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (; tile < total; tile += G) {
    nsrc = band(tile + G);
    const int img = tile / bands, j = tile - img * bands;
    opix = (((size_t)img * p.Hop + 8 * j + 1) * p.Wop + 1) * p.Cout * 2 + ooff;
    sub(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
    sub(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // everyone's pieces of the next band are in LDS; everyone has left this band's buffer
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead pieces of the band past the end land before the LDS is released
#undef L1R_PIECE
}

}  // namespace

// layer shapes this kernel takes: 3x3 stride 1, 64 -> 64 channels, 56-wide map with a multiple of 8 rows
extern "C" int flope_conv_l1r_ok(const ConvP* p) {
  return p->stride == 1 && p->ntaps == 9 && p->Cin == 64 && p->Cout == 64 && p->Wo == 56 && (p->Ho & 7) == 0 && !p->ds_in &&
         p->ksplit <= 1 && p->Wip == p->Wo + 2 && p->Hip == p->Ho + 2;
}

extern "C" int flope_conv_l1r_lds() { return 2 * 2 * 10 * 64 * 64; }

extern "C" int flope_conv_l1r_init() {
  hipError_t e = hipSuccess;
#define A(T, R) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_l1r_kernel<T, R>, hipFuncAttributeMaxDynamicSharedMemorySize, flope_conv_l1r_lds());
  A(bf16_t, false) A(bf16_t, true) A(f16_t, false) A(f16_t, true)
#undef A
  return (int)e;
}

// w: pack_l1r image.  grid: workgroups (one per CU: all 160 KB of LDS); each walks bands blockIdx.x + k * grid of batch * Ho / 8.
extern "C" int flope_conv_l1r_launch(const ConvP* p, const void* w, int dtype, int grid, void* stream) {
  if (!flope_conv_l1r_ok(p) || !w) return (int)hipErrorInvalidValue;
  const int total = p->B * (p->Ho >> 3);
  if (grid > total) grid = total;
  if (grid < 1) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)flope_conv_l1r_lds();
  hipStream_t st = (hipStream_t)stream;
#define GO(T) do { if (p->res) hipLaunchKernelGGL((conv_l1r_kernel<T, true>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w);  \
                   else hipLaunchKernelGGL((conv_l1r_kernel<T, false>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w); } while (0)
  if (dtype == 0) GO(bf16_t); else GO(f16_t);
#undef GO
  return (int)hipGetLastError();
}
