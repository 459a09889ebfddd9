// conv_s2r: the first stride-2 convolution of the trunk (layer2.0.conv1: 3x3, stride 2, 64 -> 128 channels, 56 x 56 -> 28 x 28) as
// a row-band kernel with the input patch in LDS and the weights streamed through registers (r05).
//
// Reference: torchvision ResNet-18 BasicBlock conv1 + bn1 + ReLU of layer2[0], as instantiated by
// /root/reference/sunflower/models/posenet.py:26-31 (resnet18 trunk); BN folded at load time (engine.hip load_weights).
//
// Why its own kernel.  K = 9 x 64 = 576 is short and N = 128 is one channel tile, so the gathered-tile kernel (conv_mfma<gather>)
// spends its time on LDS-DMA issue: per 128-pixel tile it streams the whole 144 KB weight panel AND gathers every input pixel
// 2.25 times (9 taps / 4 parities) -- 288 KB through the DMA queue for 2.3 us of MFMAs (21 % of the MFMA rate).
// Here a workgroup owns 4 output rows of one image (112 pixels):
//   * the 9 x 57-pixel input patch goes to LDS ONCE per 32-channel half-chunk, de-interleaved into the four (row, column) parity
//     planes so that a tap's 16 pixels are unit-stride again (plane (ky & 1, kx & 1), shifted by (ky >> 1, kx >> 1));
//   * each of the four waves owns 32 output channels and reads its A fragments (weights) straight from L2 into registers, three
//     steps ahead of their MFMAs (the whole panel is 144 KB and every workgroup reads the same one); the first three steps'
//     fragments stay resident;
//   * a pixel tile is 4 rows x 4 columns, the 64-byte pixel rows are XOR-swizzled by the plane row: every ds_read_b128 of a
//     16-lane group hits 16 different bank quads for every tap (4 columns x (slot ^ row)).
// LDS: 2 buffers (one per half-chunk) x 4 planes x 5 rows x 32 pixels x 64 B = 80 KB -> two workgroups per CU; while a half-chunk
// is being multiplied the other buffer receives the next tile's (global loads -> registers -> ds_write, two barriers per tile).
// K order: half-chunk, tap, channel (conv_mfma's is tap, channel): results equal conv_mfma's within accumulation-order rounding,
// not bit for bit (tests/test_gpu_parity.py).
#include "common.h"
#include <type_traits>

namespace {

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <typename T, int NPT>
__global__ __launch_bounds__(256, 1) void conv_s2r_kernel(const ConvP p, const u32x4* __restrict__ wpk) {
  typedef typename Elem<T>::frag frag;
  constexpr int WO = 4 * NPT;              // output columns of a tile (= the map's width)
  constexpr int PROW_B = 32 * 128;         // plane row pitch: 32 pixels (WO + 1 used) of 128 bytes (all 64 channels: whole cache lines)
  constexpr int BUF_B = 18 * PROW_B;       // 73728: planes (0,0), (0,1) with 5 rows, (1,0), (1,1) with 4
  constexpr int NSTEP = 18;                // (half-chunk, tap) steps of a tile
  extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 tile buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
  const int G = gridDim.x, total = p.B * (p.Ho >> 2), rgs = p.Ho >> 2;

  // ---- LDS image of a tile's patch.  Pixel (row, col) of plane (pa, pb) = input pixel (2 row + pa, 2 col + pb) of the band, 128
  // bytes = eight 16-byte slots; slot q (channels 8 q .. 8 q + 7) sits at position q ^ sw(row, col),
  //     sw = ((col >> 1) & 1) << 2 | (row & 3).
  // A fragment read's 16-lane group covers 4 rows x 4 consecutive columns of one slot: its bank quad (address / 16) % 16 =
  // 8 (col & 1) + position, and (col & 1, (col >> 1) & 1, row & 3) takes all 16 values -> conflict-free for every tap shift.
  //
  // ---- LDS-DMA map (tile independent).  A piece = one wave-instruction = 8 pixels x 128 B = eight WHOLE cache lines (r05: with
  // 64-byte half-lines -- one 32-channel half-chunk per buffer -- every line was fetched twice and the launch ran at the rate
  // the L1's outstanding misses allow).  Wave w moves, of every plane row (pa, row), column parity pb = w >> 1 and the two column
  // octets 2 (w & 1) + oo; lane -> column 8 octet + (lane >> 3), LDS position lane & 7.  The DMA writes a wave's 1 KB linearly, so
  // the swizzle is applied on the SOURCE side: position q receives source slot q ^ sw.  Unused columns repeat a valid pixel.
  const int dpb = wave >> 1;
  int dso[2][4];
#pragma unroll
  for (int oo = 0; oo < 2; ++oo) {
    const int col = 8 * (2 * (wave & 1) + oo) + (lane >> 3);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dso[oo][k] = (2 * min(col, dpb ? WO - 1 : WO) + dpb) * 128 + (((lane & 7) ^ ((((col >> 1) & 1) << 2) | k)) << 4);
  }
  const int wrow = p.Wip * 128;
  char* const dbase = smem + dpb * 5 * PROW_B + (wave & 1) * 2048;
  // piece k = 0 .. 17 of a tile: plane row (pa = (k >> 1) / 5, row = (k >> 1) % 5), octet oo = k & 1
#define S2R_PA(k_) (((k_) >> 1) / 5)
#define S2R_ROW(k_) (((k_) >> 1) % 5)
#define S2R_PIECE(src_, buf_, k_)                                                                              \
  GLDS16((src_) + (2 * S2R_ROW(k_) + S2R_PA(k_)) * wrow + dso[(k_) & 1][S2R_ROW(k_) & 3],                      \
         dbase + (buf_) * BUF_B + S2R_PA(k_) * (10 * PROW_B - dpb * PROW_B) + S2R_ROW(k_) * PROW_B + ((k_) & 1) * 1024)
  // (plane offsets: (0,0) 0, (0,1) 5 rows, (1,0) 10 rows, (1,1) 14 rows: pa adds 10 rows - pb)

  // ---- fragment read addresses: pixel tile pt = columns 4 pt .. 4 pt + 3 of the tile's 4 rows; lane -> (row r16 >> 2, column r16 & 3)
  const int rr = r16 >> 2, cc = r16 & 3;
  int rd[2][2][2];                          // [half-chunk][row shift ky >> 1][column shift kx >> 1]
#pragma unroll
  for (int hc = 0; hc < 2; ++hc)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
        rd[hc][dy][dx] = (rr + dy) * PROW_B + (cc + dx) * 128 + (((4 * hc + g) ^ (((((cc + dx) >> 1) & 1) << 2) | ((rr + dy) & 3))) << 4);
  const int ooff = ((rr * p.Wop + cc) * p.Cout + 32 * wave + 8 * g) * 2;

  // ---- this wave's weights: all 36 A fragments (18 steps x 2 channel tiles) and the bias stay in registers for the whole launch.
  // They take the detour over LDS: registers a GLOBAL load wrote in front of the loop and that are live into it make hipcc wait
  // vmcnt(0) in front of the loop's first MFMA on every iteration (r05 stem).
  const u32x4* const wl = wpk + (size_t)wave * NSTEP * 2 * 64 + lane;
  frag wres[NSTEP][2];
  f32x4 b4[2];
#pragma unroll
  for (int s0 = 0; s0 < NSTEP; s0 += 6) {           // six steps (12 fragments, 48 KB of LDS) per round
    u32x4* const stage = (u32x4*)smem + tid * 12;
#pragma unroll
    for (int s = s0; s < s0 + 6; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) stage[(s - s0) * 2 + ct] = wl[(s * 2 + ct) * 64];
#pragma unroll
    for (int s = s0; s < s0 + 6; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) wres[s][ct] = __builtin_bit_cast(frag, stage[(s - s0) * 2 + ct]);
    // the values must be IN the registers before the area is written again (hipcc moves these thread-private reads behind a later
    // barrier otherwise: a random tile per launch came out as garbage)
#pragma unroll
    for (int s = s0; s < s0 + 6; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(wres[s][ct]));
  }
  {
    u32x4* const stage = (u32x4*)smem + tid * 2;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) stage[ct] = *(const u32x4*)(p.bias + 32 * wave + 8 * g + 4 * ct);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) b4[ct] = __builtin_bit_cast(f32x4, stage[ct]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) asm volatile("" : "+v"(b4[ct]));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                                  // the staging area is the patch image from here on

  auto band = [&](int tile) -> const char* {        // the tile's input band, row 2 * ho0 of its image (a tile past the end: the last one)
    tile = min(tile, total - 1);
    const int img = tile / rgs, rg = tile - img * rgs;
    return (const char*)p.in + ((size_t)img * p.Hip + 8 * rg) * p.Wip * 128;
  };
  auto outp = [&](int tile) -> char* {
    const int img = tile / rgs, rg = tile - img * rgs;
    return (char*)p.out + (((size_t)img * p.Hop + 4 * rg + 1) * p.Wop + 1) * p.Cout * 2 + ooff;
  };

  int tile = blockIdx.x;
  if (tile >= total) return;
#pragma unroll
  for (int k = 0; k < 18; ++k) S2R_PIECE(band(tile), 0, k);

#ifdef FLOPE_STAG_DBG
  // diagnostic build, dbg & 64: shader-clock stamps of this workgroup's SECOND tile, wave 0 (tools/clock_probe_s2r.py)
  unsigned long long stp[20] = {0};
  int st_it = 0;
#define S2R_STAMP(i_) do { if ((p.dbg & 64) && st_it == 1) { __builtin_amdgcn_sched_barrier(0); stp[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define S2R_STAMP(i_) do {} while (0)
#endif

  // One tile: barrier; 18 steps of 14 MFMAs.  Beside the MFMAs: the next step's 7 fragment reads, in steps 0 .. 8 the 18 pieces of the
  // NEXT tile's patch (into the other buffer, which the barrier has just freed), in steps 9 .. 15 the PREVIOUS tile's epilogue
  // (pack, ReLU, one 16-byte store per pixel tile).  The wait in front of the barrier is a plain vmcnt(0): this wave's pieces of
  // this tile went out at least nine steps ago, and the youngest stores a whole tile ago.
  f32x4 acc[2][NPT][2];
  // (the first tile has no previous one: its epilogue slot stores the other accumulator set's junk to this tile's OWN outputs, which
  // the same lanes overwrite with the results one tile later -- no branch in the step stream)
  char* oprev = outp(tile);
  auto body = [&](auto cur_) {
    constexpr int CUR = decltype(cur_)::value;
    const char* const nsrc = band(tile + G);
    char* const xbuf = smem + CUR * BUF_B;
    frag xf[2][NPT];
    auto xaddr = [&](int s) -> const char* {   // pixel fragments of step s (half-chunk s / 9, tap s % 9), pixel tile 0
      const int hc = s / 9, tap = s - 9 * hc, ky = tap / 3, kx = tap - 3 * ky;
      return xbuf + ((ky & 1) * 10 + (kx & 1) * (5 - (ky & 1))) * PROW_B + rd[hc][ky >> 1][kx >> 1];
    };
    S2R_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    S2R_STAMP(1);
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) xf[0][pt] = *(const frag*)(xaddr(0) + pt * 512);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s == 9) S2R_STAMP(2);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[CUR][pt][ct] = Elem<T>::mfma(wres[s][ct], xf[s & 1][pt], s == 0 ? b4[ct] : acc[CUR][pt][ct]);
        if (s + 1 < NSTEP) xf[(s + 1) & 1][pt] = *(const frag*)(xaddr(s + 1) + pt * 512);
        if (s < 9 && pt < 2) S2R_PIECE(nsrc, CUR ^ 1, 2 * s + pt);
        if (s >= 9 && s - 9 == pt) {
          u32x4 o;
          o[0] = pk_out16<T>(pack2<T>(acc[CUR ^ 1][pt][0][0], acc[CUR ^ 1][pt][0][1]), p.relu);
          o[1] = pk_out16<T>(pack2<T>(acc[CUR ^ 1][pt][0][2], acc[CUR ^ 1][pt][0][3]), p.relu);
          o[2] = pk_out16<T>(pack2<T>(acc[CUR ^ 1][pt][1][0], acc[CUR ^ 1][pt][1][1]), p.relu);
          o[3] = pk_out16<T>(pack2<T>(acc[CUR ^ 1][pt][1][2], acc[CUR ^ 1][pt][1][3]), p.relu);
          *(u32x4*)(oprev + pt * 4 * p.Cout * 2) = o;
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (s + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (s < 9 && pt < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    S2R_STAMP(3);
#ifdef FLOPE_STAG_DBG
    if ((p.dbg & 64) && st_it == 1 && p.split_ws && tid == 0) {
      unsigned long long* d_ = (unsigned long long*)p.split_ws + (size_t)blockIdx.x * 16;
      for (int i = 0; i < 4; ++i) d_[i] = stp[i];
    }
    ++st_it;
#endif
    oprev = outp(tile);
    tile += G;
  };
  for (;;) {
    body(std::integral_constant<int, 0>{});
    if (tile >= total) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        u32x4 o;
        o[0] = pk_out16<T>(pack2<T>(acc[0][pt][0][0], acc[0][pt][0][1]), p.relu);
        o[1] = pk_out16<T>(pack2<T>(acc[0][pt][0][2], acc[0][pt][0][3]), p.relu);
        o[2] = pk_out16<T>(pack2<T>(acc[0][pt][1][0], acc[0][pt][1][1]), p.relu);
        o[3] = pk_out16<T>(pack2<T>(acc[0][pt][1][2], acc[0][pt][1][3]), p.relu);
        *(u32x4*)(oprev + pt * 4 * p.Cout * 2) = o;
      }
      break;
    }
    body(std::integral_constant<int, 1>{});
    if (tile >= total) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        u32x4 o;
        o[0] = pk_out16<T>(pack2<T>(acc[1][pt][0][0], acc[1][pt][0][1]), p.relu);
        o[1] = pk_out16<T>(pack2<T>(acc[1][pt][0][2], acc[1][pt][0][3]), p.relu);
        o[2] = pk_out16<T>(pack2<T>(acc[1][pt][1][0], acc[1][pt][1][1]), p.relu);
        o[3] = pk_out16<T>(pack2<T>(acc[1][pt][1][2], acc[1][pt][1][3]), p.relu);
        *(u32x4*)(oprev + pt * 4 * p.Cout * 2) = o;
      }
      break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead pieces of the tile past the end land before the LDS is released
#undef S2R_STAMP
#undef S2R_PIECE
#undef S2R_PA
#undef S2R_ROW
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

// w: pack_s2r image.  grid: workgroups (two per CU); each walks tiles blockIdx.x + k * grid of batch * Ho / 4.
extern "C" int flope_conv_s2r_launch(const ConvP* p, const void* w, int dtype, int grid, void* stream) {
  if (!flope_conv_s2r_ok(p) || !w) return (int)hipErrorInvalidValue;
  const int total = p->B * (p->Ho >> 2);
  if (grid > total) grid = total;
  if (grid < 1) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)flope_conv_s2r_lds();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL((conv_s2r_kernel<bf16_t, 7>), dim3(grid), dim3(256), lds, st, *p, (const u32x4*)w);
  else hipLaunchKernelGGL((conv_s2r_kernel<f16_t, 7>), dim3(grid), dim3(256), lds, st, *p, (const u32x4*)w);
  return (int)hipGetLastError();
}
