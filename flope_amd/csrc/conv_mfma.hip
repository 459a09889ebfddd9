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
// Two A-operand (pixel) feeds, one kernel body:
//   PATCH = true  (3x3 stride 1): the tile's input rows (full padded width, one
//       64-channel chunk) are staged ONCE per chunk in LDS; the nine taps are
//       nine shifted views of that patch.  L2->LDS traffic per MAC drops ~9x
//       versus re-gathering every tap, which is what keeps Cout=64/128 layers
//       off the L2-gather ceiling (~70 GB/s per CU).
//   PATCH = false (stride 2 3x3, 1x1 downsample): classic gather, one
//       [BM][64ch] A tile per (chunk, tap) step, double buffered.
//   Weights always stream through a double-buffered [BN][64] LDS tile that the
//   host has already laid out as the (swizzled) LDS image, so the copy is linear.
//
// LDS rows are 128 B (64 x 16-bit); 16-byte slot j of row r is stored at slot
// j ^ ((r >> 1) & 7): 16 consecutive rows read at one logical slot by
// ds_read_b128 land on 16 distinct 16-B positions of the 256-B bank row.
#include "common.h"

template <typename T, int BM, int BN, int WPX, int WCH, bool PATCH>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int MT = BM / (WPX * 16);   // pixel tiles per wave
  constexpr int NT = BN / (WCH * 16);   // channel tiles per wave
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert(WPX * WCH == 4 && NT % 2 == 0, "tile config");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Bs = smem;                       // 2 x B_BYTES weight tiles
  char* const As = smem + 2 * B_BYTES;         // gather: 2 x A_BYTES; patch: rows x Wip x 128

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
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

  // ---- weight (MFMA A operand) staging: linear copy of the host-built LDS image
  const char* const b_src = (const char*)p.w + (size_t)ntile * nsteps * B_BYTES + tid * 16;
  u32x4 rb[BP];

  // ---- pixel (MFMA B operand) feed
  const char* a_src[PATCH ? 1 : AP];
  u32x4 ra[PATCH ? 1 : AP];
  int a_dst = 0;
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
      const int m = min(m0 + wpx * MT * 16 + pt * 16 + r16, mend - 1);
      const int b = m / HoWo, r = m - b * HoWo, ho = r / p.Wo, wo = r - ho * p.Wo;
      pi0[pt] = (b * p.Hip + ho * p.stride - R0) * p.Wip + wo * p.stride;
    }
  } else {
    const int slot = tid & 7;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = min(m0 + (tid >> 3) + 32 * i, mend - 1);
      const int b = m / HoWo, r = m - b * HoWo, ho = r / p.Wo, wo = r - ho * p.Wo;
      const size_t pix = ((size_t)b * p.Hip + ho * p.stride) * p.Wip + wo * p.stride +
                         (p.ntaps == 1 ? p.Wip + 1 : 0);
      a_src[i] = (const char*)p.in + pix * p.Cin * 2 + slot * 16;
    }
    a_dst = (tid >> 3) * 128 + ((slot ^ ((tid >> 4) & 7)) << 4);
  }

  // fragment read offsets (bytes)
  const int s0 = (g ^ (r16 >> 1)) << 4;                       // logical slot g of k-step 0
  const int wbase = (wch * NT * 16 + r16) * 128;
  const int xbase = (wpx * MT * 16 + r16) * 128;              // gather mode

  f32x4 acc[MT][NT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: step 0 operands
#pragma unroll
  for (int i = 0; i < BP; ++i) rb[i] = *(const u32x4*)(b_src + i * 4096);
  if constexpr (!PATCH) {
    const size_t off0 = 0;
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = *(const u32x4*)(a_src[i] + off0);
#pragma unroll
    for (int i = 0; i < AP; ++i) *(u32x4*)(As + a_dst + i * 4096) = ra[i];
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) *(u32x4*)(Bs + tid * 16 + i * 4096) = rb[i];
  if constexpr (!PATCH) __syncthreads();

  int step = 0;
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    if constexpr (PATCH) {
      // every wave has passed the barrier that ended the previous chunk's last tap:
      // the patch can be overwritten.
      const char* src = (const char*)p.in + patch_src + chunk * 128;
      const size_t pixB = (size_t)p.Cin * 2;
      for (int q0 = 0; q0 < patch_pieces; q0 += 256 * 4) {
        u32x4 rp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * 256 + tid;
          if (q < patch_pieces) {
            const int pi = q >> 3, js = (q & 7) ^ ((pi >> 1) & 7);
            rp[u] = *(const u32x4*)(src + (size_t)pi * pixB + js * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * 256 + tid;
          if (q < patch_pieces) *(u32x4*)(As + (size_t)q * 16) = rp[u];
        }
      }
      __syncthreads();
    }
    for (int tap = 0; tap < p.ntaps; ++tap, ++step) {
      const int cur = step & 1;
      const bool more = step + 1 < nsteps;
      // ---- issue next step's global loads
      if (more) {
#pragma unroll
        for (int i = 0; i < BP; ++i) rb[i] = *(const u32x4*)(b_src + (size_t)(step + 1) * B_BYTES + i * 4096);
        if constexpr (!PATCH) {
          int nchunk = chunk, ntap = tap + 1;
          if (ntap == p.ntaps) { ntap = 0; ++nchunk; }
          const int ky = ntap / 3, kx = ntap - ky * 3;
          const size_t off = ((size_t)(ky * p.Wip + kx) * p.Cin + nchunk * 64) * 2;
#pragma unroll
          for (int i = 0; i < AP; ++i) ra[i] = *(const u32x4*)(a_src[i] + off);
        }
      }
      // ---- MFMA on the current tiles
      const char* const Bc = Bs + cur * B_BYTES + wbase;
      if constexpr (PATCH) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int tshift = ky * p.Wip + kx;
        int xo[MT];
#pragma unroll
        for (int pt = 0; pt < MT; ++pt) {
          const int pi = pi0[pt] + tshift;
          xo[pt] = (pi << 7) + ((g ^ ((pi >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          frag wf[NT], xf[MT];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(Bc + ct * 2048 + (s0 ^ (kk * 64)));
#pragma unroll
          for (int pt = 0; pt < MT; ++pt) xf[pt] = *(const frag*)(As + (xo[pt] ^ (kk * 64)));
#pragma unroll
          for (int pt = 0; pt < MT; ++pt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], acc[pt][ct]);
        }
      } else {
        const char* const Ac = As + cur * A_BYTES + xbase;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          frag wf[NT], xf[MT];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(Bc + ct * 2048 + (s0 ^ (kk * 64)));
#pragma unroll
          for (int pt = 0; pt < MT; ++pt) xf[pt] = *(const frag*)(Ac + pt * 2048 + (s0 ^ (kk * 64)));
#pragma unroll
          for (int pt = 0; pt < MT; ++pt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], acc[pt][ct]);
        }
      }
      // ---- commit next step's tiles to the other buffer
      if (more) {
#pragma unroll
        for (int i = 0; i < BP; ++i) *(u32x4*)(Bs + (cur ^ 1) * B_BYTES + tid * 16 + i * 4096) = rb[i];
        if constexpr (!PATCH) {
#pragma unroll
          for (int i = 0; i < AP; ++i) *(u32x4*)(As + (cur ^ 1) * A_BYTES + a_dst + i * 4096) = ra[i];
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: + bias (+ residual) (ReLU) -> 16-bit padded NHWC
  const int cb = ntile * BN + wch * NT * 16 + g * (4 * NT);
  float bias[NT * 4];
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) bias[i] = p.bias[cb + i];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = m0 + wpx * MT * 16 + pt * 16 + r16;
    conv_epilogue_px<T, NT>(p, acc[pt], m, m < mend, cb, bias, HoWo);
  }
}

// ---------------------------------------------------------------------------
// host launcher: cfg 0 = BN 64 / BM 128, cfg 1 = BN 128 / BM 128, cfg 2 = BN 64 / BM 256
template <typename T, bool PATCH>
static hipError_t launch_cfg(const ConvP& p, int cfg, size_t lds, hipStream_t st) {
  const dim3 grid(p.mtiles * p.ntiles), block(256);
  switch (cfg) {
    case 0: hipLaunchKernelGGL((conv_mfma_kernel<T, 128, 64, 4, 1, PATCH>), grid, block, lds, st, p); break;
    case 1: hipLaunchKernelGGL((conv_mfma_kernel<T, 128, 128, 2, 2, PATCH>), grid, block, lds, st, p); break;
    case 2: hipLaunchKernelGGL((conv_mfma_kernel<T, 256, 64, 4, 1, PATCH>), grid, block, lds, st, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T, int BM, int BN, int WPX, int WCH, bool PATCH>
static hipError_t set_lds_attr(size_t bytes) {
  return hipFuncSetAttribute((const void*)conv_mfma_kernel<T, BM, BN, WPX, WCH, PATCH>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int flope_conv_mfma_init() {
  const size_t big = 160 * 1024;
  hipError_t e = hipSuccess;
#define SET_ALL(T)                                                       \
  if (e == hipSuccess) e = set_lds_attr<T, 128, 64, 4, 1, true>(big);   \
  if (e == hipSuccess) e = set_lds_attr<T, 128, 64, 4, 1, false>(big);  \
  if (e == hipSuccess) e = set_lds_attr<T, 128, 128, 2, 2, true>(big);  \
  if (e == hipSuccess) e = set_lds_attr<T, 128, 128, 2, 2, false>(big); \
  if (e == hipSuccess) e = set_lds_attr<T, 256, 64, 4, 1, true>(big);   \
  if (e == hipSuccess) e = set_lds_attr<T, 256, 64, 4, 1, false>(big);
  SET_ALL(bf16_t)
  SET_ALL(f16_t)
#undef SET_ALL
  return (int)e;
}

// dtype: 0 bf16, 1 f16.  patch: 0 gather, 1 patch.  Returns hipError_t as int.
extern "C" int flope_conv_mfma_launch(const ConvP* p, int dtype, int cfg, int patch, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (dtype == 0) e = patch ? launch_cfg<bf16_t, true>(*p, cfg, lds, st) : launch_cfg<bf16_t, false>(*p, cfg, lds, st);
  else            e = patch ? launch_cfg<f16_t, true>(*p, cfg, lds, st) : launch_cfg<f16_t, false>(*p, cfg, lds, st);
  return (int)e;
}
