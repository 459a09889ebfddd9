// conv_s2r: the first stride-2 convolution of the trunk (layer2.0.conv1: 3x3, stride 2, 64 -> 128 channels, 56 x 56 -> 28 x 28) as
// a row-band kernel with the input patch in LDS and the weights resident in registers (r05).
//
// Reference: torchvision ResNet-18 BasicBlock conv1 + bn1 + ReLU of layer2[0], as instantiated by
// /root/reference/sunflower/models/posenet.py:26-31 (resnet18 trunk); BN folded at load time (engine.hip load_weights).
//
// Why its own kernel.  K = 9 x 64 = 576 is short and N = 128 is one channel tile, so the gathered-tile kernel (conv_mfma<gather>)
// spends its time on LDS-DMA issue: per 128-pixel tile it streams the whole 144 KB weight panel AND gathers every input pixel
// 2.25 times (9 taps / 4 parities) -- 288 KB through the DMA queue for 2.3 us of MFMAs (21 % of the MFMA rate).
// Here one workgroup of 8 waves per CU walks tiles of 4 output rows of one image (112 pixels):
//   * the tile's 9 x 57-pixel input patch goes to LDS once, whole 128-byte pixels (cache lines), de-interleaved into the four
//     (row, column) parity planes so that a tap's 16 pixels are unit-stride again (plane (ky & 1, kx & 1), shifted by
//     (ky >> 1, kx >> 1)); two tile buffers of 72 KB: the next tile's patch arrives by LDS-DMA while this one is multiplied;
//   * wave w owns output channels 32 (w & 3) .. + 31 and the pixel tiles of half w >> 2 (4 + 3 of the 7), and keeps ALL its A
//     fragments (18 steps x 2 channel tiles = 144 VGPRs) in registers for the whole launch: no weight traffic in the loop at all;
//   * a pixel tile is 4 rows x 4 columns and the 16-byte slots of a pixel are XOR-swizzled by (row, column) -- conflict-free
//     ds_read_b128 for every tap (see the image description in the kernel).
// The step stream has two waves per SIMD (225 VGPRs), so one wave's reads, DMA pieces and epilogue run under the other's MFMAs:
// the 18 steps of a tile take 4.4 - 4.7 k cycles against the 4.0 k of their MFMAs alone.  What bounds the launch is the patch stream
// (110 MB of input for 14 us of MFMAs; 171 MB of HBM traffic per launch = 55 - 60 % of the HBM peak): the next tile's pieces must go out
// in the first steps of a tile -- issued one per step they arrived 5.6 k cycles late
// (profiles/r05_conv_s2r.txt has the stamps and the five designs that came before this one).
// K order: half-chunk, tap, channel (conv_mfma's is tap, channel): results equal conv_mfma's within accumulation-order rounding,
// not bit for bit (tests/test_gpu_parity.py::test_stride2_patch_kernel_against_the_gathered_tile_kernel).
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

template <typename T, int NPT>
__global__ __launch_bounds__(512, 1) void conv_s2r_kernel(const ConvP p, const u32x4* __restrict__ wpk) {
  typedef typename Elem<T>::frag frag;
  constexpr int WO = 4 * NPT;              // output columns of a tile (= the map's width)
  constexpr int PROW_B = 32 * 128;         // plane row pitch: 32 pixels (WO + 1 used) of 128 bytes (all 64 channels: whole cache lines)
  constexpr int BUF_B = 18 * PROW_B;       // 73728: planes (0,0), (0,1) with 5 rows, (1,0), (1,1) with 4
  constexpr int NSTEP = 18;                // (half-chunk, tap) steps of a tile
  extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 tile buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int cg = wave & 3, ph = wave >> 2;  // this wave: output channels 32 cg .. 32 cg + 31, pixel tiles 4 ph .. (ph ? 6 : 3)
  const int G = gridDim.x, total = p.B * (p.Ho >> 2), rgs = p.Ho >> 2;

  // ---- LDS image of a tile's patch.  Pixel (row, col) of plane (pa, pb) = input pixel (2 row + pa, 2 col + pb) of the band, 128
  // bytes = eight 16-byte slots; slot q (channels 8 q .. 8 q + 7) sits at position q ^ sw(row, col),
  //     sw = ((col >> 1) & 1) << 2 | (row & 3).
  // A fragment read's 16-lane group covers 4 rows x 4 consecutive columns of one slot: its bank quad (address / 16) % 16 =
  // 8 (col & 1) + position, and (col & 1, (col >> 1) & 1, row & 3) takes all 16 values -> conflict-free for every tap shift.
  //
  // ---- LDS-DMA map (tile independent).  A piece = one wave-instruction = 8 pixels x 128 B = eight WHOLE cache lines (r05: with
  // 64-byte half-lines -- one 32-channel half-chunk per buffer -- every line was fetched twice and the launch ran at the rate
  // the L1's outstanding misses allow).  Wave w moves, of every plane row (pa, row), column parity pb = w >> 2 and column octet
  // w & 3; lane -> column 8 octet + (lane >> 3), LDS position lane & 7.  The DMA writes a wave's 1 KB linearly, so the swizzle is
  // applied on the SOURCE side: position q receives source slot q ^ sw.  Unused columns repeat a valid pixel.
  const int dpb = wave >> 2;
  int dso[4];
  {
    const int col = 8 * (wave & 3) + (lane >> 3);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dso[k] = (2 * min(col, dpb ? WO - 1 : WO) + dpb) * 128 + (((lane & 7) ^ ((((col >> 1) & 1) << 2) | k)) << 4);
  }
  const int wrow = p.Wip * 128;
  char* const dbase = smem + dpb * 5 * PROW_B + (wave & 3) * 1024;
  // piece k = 0 .. 8 of a tile: plane row (pa = k / 5, row = k % 5)   (plane offsets: (0,0) 0, (0,1) 5 rows, (1,0) 10, (1,1) 14)
#define S2R_PIECE(src_, buf_, k_)                                                                              \
  GLDS16((src_) + (2 * ((k_) % 5) + (k_) / 5) * wrow + dso[((k_) % 5) & 3],                                    \
         dbase + (buf_) * BUF_B + ((k_) / 5) * (10 * PROW_B - dpb * PROW_B) + ((k_) % 5) * PROW_B)

  // ---- fragment read addresses: pixel tile pt = columns 4 pt .. 4 pt + 3 of the tile's 4 rows; lane -> (row r16 >> 2, column r16 & 3)
  // (a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: 8 lanes of slot g with r16 in
  // {0-3, 12-15} and 8 lanes of slot g ^ 1 with r16 in {4-11}.  The first set takes the even columns, the second the odd ones:
  // bit 3 of the bank quad separates them, (row, column >> 1) the eight pixels inside each.)
  const int i8 = r16 < 4 ? r16 : (r16 < 12 ? r16 - 4 : r16 - 8);
  const int rr = i8 >> 1, cc = 2 * (i8 & 1) + ((r16 >= 4 && r16 < 12) ? 1 : 0);
  int rd[2][2][2];                          // [half-chunk][row shift ky >> 1][column shift kx >> 1]
#pragma unroll
  for (int hc = 0; hc < 2; ++hc)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
        rd[hc][dy][dx] = (rr + dy) * PROW_B + (cc + dx) * 128 + (((4 * hc + g) ^ (((((cc + dx) >> 1) & 1) << 2) | ((rr + dy) & 3))) << 4);
  const int ooff = ((rr * p.Wop + cc) * p.Cout + 32 * cg + 8 * g) * 2;

  int tile = blockIdx.x;
  if (tile >= total) return;
  auto band = [&](int tl) -> const char* {          // the tile's input band, row 2 * ho0 of its image (a tile past the end: the last one)
    tl = min(tl, total - 1);
    const int img = tl / rgs, rg = tl - img * rgs;
    return (const char*)p.in + ((size_t)img * p.Hip + 8 * rg) * p.Wip * 128;
  };
  // the first tile's patch goes out first: it is in flight while the weights arrive
#pragma unroll
  for (int k = 0; k < 9; ++k) S2R_PIECE(band(tile), 0, k);

  // ---- this wave's weights: all 36 A fragments (18 steps x 2 channel tiles) and the bias stay in registers for the whole launch.
  // Registers that a GLOBAL load wrote in front of the loop and that are live into it make hipcc's wait-count pass put a
  // `s_waitcnt vmcnt(0)` in front of the loop's first MFMA on every iteration (r05 stem) -- unless their arrival is
  // settled before the loop: the empty asm statements below "use" every fragment once, so the compiler waits for them HERE.
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

#ifdef FLOPE_STAG_DBG
  // diagnostic build, dbg & 64: shader-clock stamps of this workgroup's SECOND tile, wave 0 (tools/clock_probe_s2r.py)
  unsigned long long stp[20] = {0};
  int st_it = 0;
#define S2R_STAMP(i_) do { if ((p.dbg & 64) && st_it == 1) { __builtin_amdgcn_sched_barrier(0); stp[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define S2R_STAMP(i_) do {} while (0)
#endif

  // One tile (two waves per SIMD: the other wave's MFMAs cover this one's reads, pieces and epilogue): 18 steps of 2 x NP MFMAs with
  // the next step's NP fragment reads and, in steps 0 .. 8, this wave's 9 pieces of the NEXT tile's patch (into the other buffer,
  // free since the barrier); then the wait for those pieces (vmcnt(0): they went out nine steps ago; the stores below stay in
  // flight through the next tile), the epilogue, the barrier.
  auto run = [&](auto ph_) {
    constexpr int PH = decltype(ph_)::value, P0 = 4 * PH, NP = PH ? NPT - 4 : 4;
    int cur = 0;
    for (; tile < total; tile += G) {
      const char* const nsrc = band(tile + G);
      const int img = tile / rgs, rg = tile - img * rgs;
      char* const obase = (char*)p.out + (((size_t)img * p.Hop + 4 * rg + 1) * p.Wop + 1) * p.Cout * 2 + ooff;
      char* const xbuf = smem + cur * BUF_B + P0 * 512;
      const int nbuf = cur ^ 1;
      f32x4 acc[NP][2];
      frag xf[2][NP];
      auto xaddr = [&](int s) -> const char* {   // pixel fragments of step s (half-chunk s / 9, tap s % 9), this wave's first pixel tile
        const int hc = s / 9, tap = s - 9 * hc, ky = tap / 3, kx = tap - 3 * ky;
        return xbuf + ((ky & 1) * 10 + (kx & 1) * (5 - (ky & 1))) * PROW_B + rd[hc][ky >> 1][kx >> 1];
      };
      S2R_STAMP(0);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) xf[0][pt] = *(const frag*)(xaddr(0) + pt * 512);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s == 9) S2R_STAMP(1);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = Elem<T>::mfma(wres[s][ct], xf[s & 1][pt], s == 0 ? b4[ct] : acc[pt][ct]);
          if (s + 1 < NSTEP) xf[(s + 1) & 1][pt] = *(const frag*)(xaddr(s + 1) + pt * 512);
          if (s < 3 && pt < 3) S2R_PIECE(nsrc, nbuf, 3 * s + pt);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (s + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (s < 3 && pt < 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      S2R_STAMP(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      S2R_STAMP(3);
      // epilogue: ReLU (+ float16 clamp), 8 consecutive channels = one 16-byte store per pixel and lane
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) {
        u32x4 o;
        o[0] = pk_out16<T>(pack2<T>(acc[pt][0][0], acc[pt][0][1]), p.relu);
        o[1] = pk_out16<T>(pack2<T>(acc[pt][0][2], acc[pt][0][3]), p.relu);
        o[2] = pk_out16<T>(pack2<T>(acc[pt][1][0], acc[pt][1][1]), p.relu);
        o[3] = pk_out16<T>(pack2<T>(acc[pt][1][2], acc[pt][1][3]), p.relu);
        *(u32x4*)(obase + (P0 + pt) * 4 * p.Cout * 2) = o;
      }
      S2R_STAMP(4);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // everyone's pieces of the next tile are in LDS; everyone has left this tile's buffer
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      S2R_STAMP(5);
#ifdef FLOPE_STAG_DBG
      if ((p.dbg & 64) && st_it == 1 && p.split_ws && tid == 0) {
        unsigned long long* d_ = (unsigned long long*)p.split_ws + (size_t)blockIdx.x * 16;
        for (int i = 0; i < 6; ++i) d_[i] = stp[i];
      }
      ++st_it;
#endif
      cur ^= 1;
    }
  };
  if (ph == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead pieces of the tile past the end land before the LDS is released
#undef S2R_STAMP
#undef S2R_PIECE
}

}  // namespace

// layer shapes this kernel takes: 3x3 stride 2, 64 -> 128 channels, output map 28 wide and a multiple of 4 rows, no residual
extern "C" int flope_conv_s2r_ok(const ConvP* p) {
  return p->stride == 2 && p->ntaps == 9 && p->Cin == 64 && p->Cout == 128 && p->Wo == 28 && (p->Ho & 3) == 0 && !p->res && !p->ds_in &&
         p->ksplit <= 1 && p->Wip == 2 * p->Wo + 2 && p->Hip == 2 * p->Ho + 2;
}

extern "C" int flope_conv_s2r_lds() { return 2 * 18 * 32 * 128; }

extern "C" int flope_conv_s2r_init() {
  hipError_t e = hipFuncSetAttribute((const void*)conv_s2r_kernel<bf16_t, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, flope_conv_s2r_lds());
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_s2r_kernel<f16_t, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, flope_conv_s2r_lds());
  return (int)e;
}

// w: pack_s2r image.  grid: workgroups (one per CU: 144 KB of LDS); each walks tiles blockIdx.x + k * grid of batch * Ho / 4.
extern "C" int flope_conv_s2r_launch(const ConvP* p, const void* w, int dtype, int grid, void* stream) {
  if (!flope_conv_s2r_ok(p) || !w) return (int)hipErrorInvalidValue;
  const int total = p->B * (p->Ho >> 2);
  if (grid > total) grid = total;
  if (grid < 1) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)flope_conv_s2r_lds();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL((conv_s2r_kernel<bf16_t, 7>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w);
  else hipLaunchKernelGGL((conv_s2r_kernel<f16_t, 7>), dim3(grid), dim3(512), lds, st, *p, (const u32x4*)w);
  return (int)hipGetLastError();
}
