// TransformerEncoder of the reference (scripts/tf_encoder.py:5-27) on gfx950, behind the C-ABI
// flope_tf_* of include/flope_amd.h:
//     embedding Linear(in, d)  ->  L x post-norm TransformerEncoderLayer(d, heads, ff, ReLU, batch_first)
//     ->  out_layer Linear(d, out);   eval-mode semantics (dropout = identity), no mask, no positions.
// Two arithmetic modes share one launch sequence:
//   FLOPE_DT_F32          every op in fp32 on the vector ALU (any dimensions; pins the reference's toy fixture)
//   FLOPE_DT_F16 / BF16   activations in HBM as 16-bit row-major [tokens][features];
//                         linears whose (N % 128 == 0, K % 64 == 0) run on v_mfma_f32_16x16x32 (tf_gemm_mfma),
//                         attention with head_dim 64 runs on MFMA with the softmax in registers (tf_attn_mfma);
//                         everything else (embedding, out_layer, odd shapes) falls to the generic kernels.
// Bias, residual add and ReLU live in the linear kernels' epilogues; LayerNorm is one wave per token row.
#include "../../include/flope_amd.h"
#include "common.h"
#include "host_pack.h"

#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <type_traits>
#include <vector>

using namespace flope_host;

namespace {

std::string g_tf_error;

#define GLDS16(gptr, lptr)                                                                             \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),              \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#define WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define BLOCK_BARRIER()                  \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_s_barrier();        \
    asm volatile("" ::: "memory");       \
  } while (0)

template <typename T> __device__ __forceinline__ float ld_any(const void* p, size_t i, int f32) {
  return f32 ? ((const float*)p)[i] : to_f32<T>(((const T*)p)[i]);
}
template <typename T> __device__ __forceinline__ void st_any(void* p, size_t i, int f32, float v) {
  if (f32) ((float*)p)[i] = v; else ((T*)p)[i] = from_f32<T>(v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- generic kernels (any shape; fp32 accumulate) ------------------------------------------------------------
// Y[m][n] = act(sum_k X[m][k] * W[n][k] + b[n] (+ R[m][n]))        W fp32 [N][K] as stored in the checkpoint
template <typename T>
__global__ void tf_linear_generic(const void* X, int x_f32, const float* W, const float* b, const void* R, void* Y,
                                  int y_f32, int M, int K, int N, int relu) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx - (size_t)m * N);
  const float* w = W + (size_t)n * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(ld_any<T>(X, (size_t)m * K + k, x_f32), w[k], acc);
  acc += b[n];
  if (R) acc += ld_any<T>(R, idx, 0);
  if (relu) acc = fmaxf(acc, 0.f);
  st_any<T>(Y, idx, y_f32, acc);
}

// Narrow outputs (N <= 16, e.g. out_layer): one wave per token row, lanes stride K (coalesced X and W reads), one
// wave reduction per output feature.
template <typename T>
__global__ void tf_linear_rowwave(const void* X, int x_f32, const float* W, const float* b, void* Y, int y_f32, int M,
                                  int K, int N, int relu) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int m = blockIdx.x * nw + (threadIdx.x >> 6); m < M; m += gridDim.x * nw) {
    float acc[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) acc[n] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float x = ld_any<T>(X, (size_t)m * K + k, x_f32);
#pragma unroll
      for (int n = 0; n < 16; ++n)
        if (n < N) acc[n] = fmaf(x, W[(size_t)n * K + k], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < 16; ++n)
      if (n < N) {
        float v = wave_sum(acc[n]) + b[n];
        if (relu) v = fmaxf(v, 0.f);
        if (lane == 0) st_any<T>(Y, (size_t)m * N + n, y_f32, v);
      }
  }
}

// The same for 16-bit activations with K % 8 == 0 (r03: the kernel above took 184 us for the out_layer of the throughput shape --
// 2-byte loads, one row per wave at a time -- against ~10 us of HBM time for its 50 MB): W is staged once per workgroup in LDS
// (N x K floats), a lane reads 16 bytes of its row per step and keeps two rows in flight.
template <typename T>
__global__ __launch_bounds__(256) void tf_linear_rowwave_vec(const T* __restrict__ X, const float* __restrict__ W,
                                                             const float* __restrict__ b, void* Y, int y_f32, int M, int K, int N, int relu) {
  extern __shared__ __attribute__((aligned(16))) float sw[];          // [N][K]
  for (int i = threadIdx.x; i < N * K; i += 256) sw[i] = W[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, nw = 4;
  const int k8 = K >> 3;
  for (int m0 = (blockIdx.x * nw + (threadIdx.x >> 6)) * 2; m0 < M; m0 += gridDim.x * nw * 2) {
    float acc[2][16];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int n = 0; n < 16; ++n) acc[r][n] = 0.f;
    for (int c = lane; c < k8; c += 64) {
      u32x4 xv[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) xv[r] = *(const u32x4*)(X + (size_t)min(m0 + r, M - 1) * K + c * 8);
#pragma unroll
      for (int n = 0; n < 16; ++n)
        if (n < N) {
          const f32x4 w0 = *(const f32x4*)(sw + n * K + c * 8), w1 = *(const f32x4*)(sw + n * K + c * 8 + 4);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            float a = acc[r][n];
            a = fmaf(unpack_lo<T>(xv[r][0]), w0[0], a); a = fmaf(unpack_hi<T>(xv[r][0]), w0[1], a);
            a = fmaf(unpack_lo<T>(xv[r][1]), w0[2], a); a = fmaf(unpack_hi<T>(xv[r][1]), w0[3], a);
            a = fmaf(unpack_lo<T>(xv[r][2]), w1[0], a); a = fmaf(unpack_hi<T>(xv[r][2]), w1[1], a);
            a = fmaf(unpack_lo<T>(xv[r][3]), w1[2], a); a = fmaf(unpack_hi<T>(xv[r][3]), w1[3], a);
            acc[r][n] = a;
          }
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int n = 0; n < 16; ++n)
        if (n < N) {
          float v = wave_sum(acc[r][n]) + b[n];
          if (relu) v = fmaxf(v, 0.f);
          if (lane == 0 && m0 + r < M) st_any<T>(Y, (size_t)(m0 + r) * N + n, y_f32, v);
        }
  }
}

// float32 [M][K] -> 16-bit [M][Kp], columns K..Kp-1 zero (feeds the MFMA linear when K is not a multiple of 64)
template <typename T>
__global__ void tf_cast_pad(const float* X, T* Y, int M, int K, int Kp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * Kp) return;
  const int m = (int)(idx / Kp), k = (int)(idx - (size_t)m * Kp);
  Y[idx] = from_f32<T>(k < K ? X[(size_t)m * K + k] : 0.f);
}

// LayerNorm over the last dimension, one wave per row (eps 1e-5, biased variance)
template <typename T>
__global__ void tf_layernorm(const T* in, T* out, const float* w, const float* b, int M, int d) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const T* x = in + (size_t)row * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += to_f32<T>(x[c]);
  const float mean = wave_sum(s) / d;
  float v = 0.f;
  for (int c = lane; c < d; c += 64) { const float t = to_f32<T>(x[c]) - mean; v = fmaf(t, t, v); }
  const float rstd = 1.f / sqrtf(wave_sum(v) / d + 1e-5f);
  T* y = out + (size_t)row * d;
  for (int c = lane; c < d; c += 64) y[c] = from_f32<T>((to_f32<T>(x[c]) - mean) * rstd * w[c] + b[c]);
}

// 16-bit rows with d % 8 == 0 and d <= 2048: each lane keeps its 16-byte vectors in registers (one HBM read)
template <typename T>
__global__ void tf_layernorm_vec(const T* in, T* out, const float* w, const float* b, int M, int d) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = d >> 3;
  const u32x4* x = (const u32x4*)(in + (size_t)row * d);
  u32x4 r[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      r[i] = x[c];
#pragma unroll
      for (int q = 0; q < 4; ++q) s += unpack_lo<T>(r[i][q]) + unpack_hi<T>(r[i][q]);
    }
  }
  const float mean = wave_sum(s) / d;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = unpack_lo<T>(r[i][q]) - mean, c = unpack_hi<T>(r[i][q]) - mean;
        v = fmaf(a, a, fmaf(c, c, v));
      }
    }
  const float rstd = 1.f / sqrtf(wave_sum(v) / d + 1e-5f);
  u32x4* y = (u32x4*)(out + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = c * 8 + q * 2;
        o[q] = pack2<T>((unpack_lo<T>(r[i][q]) - mean) * rstd * w[ch] + b[ch],
                        (unpack_hi<T>(r[i][q]) - mean) * rstd * w[ch + 1] + b[ch + 1]);
      }
      y[c] = o;
    }
  }
}

// softmax(q k^T / sqrt(dh)) v for one (batch, head) per blockIdx.x; one wave per query row.  Any L / dh.
template <typename T>
__global__ void tf_attn_generic(const T* qkv, T* out, int L, int d, int H) {
  extern __shared__ float sc[];                 // [waves][L]
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, dh = d / H;
  const float scale = 1.f / sqrtf((float)dh);
  float* s = sc + (size_t)wave * L;
  const T* base = qkv + (size_t)b * L * 3 * d + h * dh;
  for (int i = blockIdx.y * nw + wave; i < L; i += gridDim.y * nw) {
    const T* q = base + (size_t)i * 3 * d;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
      const T* k = base + (size_t)j * 3 * d + d;
      float a = 0.f;
      for (int c = 0; c < dh; ++c) a = fmaf(to_f32<T>(q[c]), to_f32<T>(k[c]), a);
      a *= scale;
      s[j] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) { const float p = expf(s[j] - mx); s[j] = p; sum += p; }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.f / sum;
    for (int c = lane; c < dh; c += 64) {
      float o = 0.f;
      for (int j = 0; j < L; ++j) o = fmaf(s[j], to_f32<T>(base[(size_t)j * 3 * d + 2 * d + c]), o);
      out[((size_t)b * L + i) * d + h * dh + c] = from_f32<T>(o * inv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- MFMA linear: Y[128-token tile][128-feature tile], K walked in 64-wide chunks ----------------------------
// Weights are the MFMA A operand, tokens the B operand (same roles as the conv kernels): a lane ends with one token x
// 16 consecutive features, so bias / residual / ReLU / 16-bit pack happen in registers and leave as 32-byte stores.
// Both operand tiles reach LDS by LDS-DMA (16 B per lane, 1 KiB per wave instruction) into a 2-deep ring:
//   weights: host-packed as the swizzled LDS image ([ntile][chunk][128 rows][128 B]) -> linear copy
//   tokens : swizzle applied on the per-lane source address
// LDS image: row r (128 B = 64 k), 16-byte slot j holds k-chunk j ^ ((r >> 1) & 7)  -> ds_read_b128 conflict-free.
#ifdef FLOPE_STAG_DBG
// diagnostic build: shader-clock stamps of workgroup 0, wave 0 of every tf_gemm_mfma launch {entry, first DMA issued, first chunk landed, K loop done,
// epilogue issued, realtime entry, realtime exit, K | N << 32}; flope_tfdbg_read
__device__ unsigned long long g_tfdbg[8 * 512];
__device__ unsigned g_tfdbg_n;
#define TFDBG_STAMP(i_) do { __builtin_amdgcn_sched_barrier(0); tst[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TFDBG_STAMP(i_) do {} while (0)
#endif

template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(256, 2) void tf_gemm_mfma(const T* __restrict__ X, const void* __restrict__ Wp,
                                                       const float* __restrict__ bias, const T* __restrict__ res,
                                                       T* __restrict__ Y, int K, int N) {
  typedef typename Elem<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][ W 16 KiB | X 16 KiB ]
#ifdef FLOPE_STAG_DBG
  unsigned long long tst[5] = {0, 0, 0, 0, 0};
  const unsigned long long trt0 = __builtin_amdgcn_s_memrealtime();
  TFDBG_STAMP(0);
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ntiles = N >> 7, mt = id / ntiles, nt = id - mt * ntiles, nch = K >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const char* wsrc = (const char*)Wp + (size_t)nt * nch * 16384 + wave * 4096 + lane * 16;
  const char* xsrc = (const char*)X + (size_t)mt * 128 * K * 2;
  int xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3), slot = lane & 7;
    xoff[i] = row * K * 2 + ((slot ^ ((row >> 1) & 7)) << 4);
  }
  const int sw = (lane >> 1) & 7;
  const int rdW = (wn * 64 + (lane & 15)) * 128, rdX = 16384 + (wm * 64 + (lane & 15)) * 128;

  const int n0 = nt * 128 + wn * 64 + g * 16;
  f32x4 acc[4][4];
  {
    float bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bv[i] = bias[n0 + i];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[pt][ct] = f32x4{bv[ct * 4], bv[ct * 4 + 1], bv[ct * 4 + 2], bv[ct * 4 + 3]};
  }

#define TF_ISSUE(c_, b_)                                                                        \
  do {                                                                                          \
    char* d_ = smem + (b_) * 32768 + wave * 4096;                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(wsrc + (size_t)(c_) * 16384 + i * 1024, d_ + i * 1024); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(xsrc + (c_) * 128 + xoff[i], d_ + 16384 + i * 1024);    \
  } while (0)

  TF_ISSUE(0, 0);
  TFDBG_STAMP(1);
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) { TF_ISSUE(c + 1, (c + 1) & 1); WAIT_VM(8); } else { WAIT_VM(0); }
    BLOCK_BARRIER();
#ifdef FLOPE_STAG_DBG
    if (c == 0) TFDBG_STAMP(2);
#endif
    const char* bs = smem + (c & 1) * 32768;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((ks * 4 + g) ^ sw) << 4;
      frag wf[4], xf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wf[t] = *(const frag*)(bs + rdW + t * 2048 + so);
        xf[t] = *(const frag*)(bs + rdX + t * 2048 + so);
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], acc[pt][ct]);
    }
    BLOCK_BARRIER();
  }
#undef TF_ISSUE
#ifdef FLOPE_STAG_DBG
  { float keep_ = acc[0][0][0]; asm volatile("" : "+v"(keep_)); }
  TFDBG_STAMP(3);
#endif

#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const size_t m = (size_t)mt * 128 + wm * 64 + pt * 16 + (lane & 15);
    const size_t off = m * N + n0;
    float v[16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[ct * 4 + q] = acc[pt][ct][q];
    if constexpr (RES) {
      const u32x4 r0 = *(const u32x4*)(res + off), r1 = *(const u32x4*)(res + off + 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q * 2] += unpack_lo<T>(r0[q]); v[q * 2 + 1] += unpack_hi<T>(r0[q]);
        v[8 + q * 2] += unpack_lo<T>(r1[q]); v[8 + q * 2 + 1] += unpack_hi<T>(r1[q]);
      }
    }
    u32x4 o0, o1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o0[q] = pack2<T>(v[q * 2], v[q * 2 + 1]);
      o1[q] = pack2<T>(v[8 + q * 2], v[8 + q * 2 + 1]);
      if constexpr (RELU) { o0[q] = pk_relu16(o0[q]); o1[q] = pk_relu16(o1[q]); }
    }
    *(u32x4*)(Y + off) = o0;
    *(u32x4*)(Y + off + 8) = o1;
  }
#ifdef FLOPE_STAG_DBG
  TFDBG_STAMP(4);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned slot = atomicAdd(&g_tfdbg_n, 1u) & 511u;
    for (int i = 0; i < 5; ++i) g_tfdbg[slot * 8 + i] = tst[i];
    g_tfdbg[slot * 8 + 5] = trt0; g_tfdbg[slot * 8 + 6] = __builtin_amdgcn_s_memrealtime(); g_tfdbg[slot * 8 + 7] = (unsigned long long)K | ((unsigned long long)N << 32);
  }
#endif
}

// ---- MFMA attention, head_dim 64 ------------------------------------------------------------------------------
// One workgroup per (batch, head); K and V of that head are staged once into LDS (row-major [key][64], 128-byte
// rows); each wave owns 32 queries and walks the keys 32 at a time:
//   S^T[key][query] = K . Q^T        (A = K rows by ds_read_b128, B = Q fragments kept in registers)
//   online softmax over keys         (in-lane over 8 values, then lanes +16 / +32 that share the query column)
//   O^T[dh][query] += V^T . P^T      (B = P^T straight from the S^T accumulators: k-slot j of lane group g is key
//                                     4g+j (j<4) or 16+4g+(j-4); A = V^T read with ds_read_b64_tr_b16 in the SAME
//                                     permuted key order: two 4-row x 16-column transposed blocks per fragment)
// K image swizzle: slot ^ ((row>>1)&7) (row reads); V image swizzle: slot ^ (((row>>1)&3)<<1) (keeps the 32-byte
// column pairs of the transposed reads adjacent and spreads the 8 rows of a 32-lane half over all banks).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename T>
__global__ __launch_bounds__(1024) void tf_attn_mfma(const T* __restrict__ qkv, T* __restrict__ out, int L, int d,
                                                     int H, int Lp, float scale_log2e) {
  typedef typename Elem<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // K image [Lp][128 B] | V image [Lp][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const T* base = qkv + (size_t)b * L * 3 * d + h * 64;
  char* Ki = smem;
  char* Vi = smem + (size_t)Lp * 128;
  for (int idx = tid; idx < Lp * 8; idx += blockDim.x) {
    const int row = idx >> 3, ch = idx & 7;
    u32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
    if (row < L) {
      const T* src = base + (size_t)row * 3 * d + ch * 8;
      kv = *(const u32x4*)(src + d);
      vv = *(const u32x4*)(src + 2 * d);
    }
    *(u32x4*)(Ki + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kv;
    *(u32x4*)(Vi + row * 128 + ((ch ^ (((row >> 1) & 3) << 1)) << 4)) = vv;
  }
  __syncthreads();

  const int q0 = wave * 32;
  frag qf[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int qi = q0 + qt * 16 + li;
    qi = qi < L ? qi : L - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *(const frag*)(base + (size_t)qi * 3 * d + (ks * 4 + g) * 8);
  }
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
  f32x4 o[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ksw = (li >> 1) & 7;                 // K-image swizzle of this lane's key row (row = 16-aligned + li)
  const int trq = li >> 2, trp = li & 3;         // transposed-read role of this lane inside its 16-lane group

  for (int kb = 0; kb < Lp; kb += 32) {
    f32x4 s[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) s[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const frag kf = *(const frag*)(Ki + (kb + t * 16 + li) * 128 + (((ks * 4 + g) ^ ksw) << 4));
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) s[t][qt] = Elem<T>::mfma(kf, qf[qt][ks], s[t][qt]);
      }
    frag pf[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float v[8];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int key = kb + t * 16 + g * 4 + q;
          const float x = key < L ? s[t][qt][q] * scale_log2e : -INFINITY;
          v[t * 4 + q] = x;
          mx = fmaxf(mx, x);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(mrun[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i] - mnew); ps += v[i]; }
      lrun[qt] = lrun[qt] * alpha + ps;
      mrun[qt] = mnew;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[qt][dt] *= alpha;
      u32x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) pk[i] = pack2<T>(v[i * 2], v[i * 2 + 1]);
      pf[qt] = __builtin_bit_cast(frag, pk);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      s16x4 lo, hi;
      {
        const int r0 = kb + g * 4 + trq, r1 = r0 + 16;
        const int a0 = r0 * 128 + (((dt * 2 + (trp >> 1)) ^ (((r0 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
        const int a1 = r1 * 128 + (((dt * 2 + (trp >> 1)) ^ (((r1 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vi + a0));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vi + a1));
      }
      const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const frag vf = __builtin_bit_cast(frag, v8);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) o[qt][dt] = Elem<T>::mfma(vf, pf[qt], o[qt][dt]);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lt = lrun[qt];
    lt += __shfl_xor(lt, 16);
    lt += __shfl_xor(lt, 32);
    const float inv = 1.f / lt;
    const int qi = q0 + qt * 16 + li;
    if (qi < L) {
      T* dst = out + ((size_t)b * L + qi) * d + h * 64 + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        u32x2 w2;
        w2[0] = pack2<T>(o[qt][dt][0] * inv, o[qt][dt][1] * inv);
        w2[1] = pack2<T>(o[qt][dt][2] * inv, o[qt][dt][3] * inv);
        *(u32x2*)(dst + dt * 16) = w2;
      }
    }
  }
}

}  // namespace

// ---- handle ----------------------------------------------------------------------------------------------------
struct TfLinear {
  float* w = nullptr; float* b = nullptr;   // fp32 [N][K], [N]
  void* packed = nullptr;                    // MFMA image (16-bit) when eligible
  int N = 0, K = 0, Kp = 0;                  // Kp: K rounded up to 64 (the packed image's K)
};
struct TfLayer { TfLinear in_proj, out_proj, lin1, lin2; float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr; };

struct flope_tf_encoder {
  int device = 0, in_dim = 0, d = 0, out_dim = 0, H = 0, nl = 0, ff = 0, max_tokens = 0, Mpad = 0, dtype = 0, esz = 2;
  int opt_generic = 0;                       // 1: force the generic kernels (A/B checks)
  bool loaded = false;
  TfLinear emb, outl;
  std::vector<TfLayer> layers;
  void *h = nullptr, *h2 = nullptr, *qkv = nullptr, *att = nullptr, *ffb = nullptr;
  void* xin = nullptr;                       // 16-bit zero-padded copy of the fp32 input [Mpad][roundup(in_dim, 64)]
  std::vector<void*> allocs;
  std::string err;
};

namespace {

int tf_fail(flope_tf_encoder* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  g_tf_error = msg;
  return code;
}
#define TF_HIP(e, call)                                                                          \
  do {                                                                                           \
    hipError_t _s = (call);                                                                      \
    if (_s != hipSuccess) return tf_fail(e, FLOPE_EHIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
  } while (0)

// [N][K] fp32 -> [N/128][K/64][128 rows][8 slots x 8 k] 16-bit, rows permuted so that MFMA row 4g+q of channel tile
// ct is feature 16g + 4ct + q of the wave's 64-feature half, slots pre-swizzled (slot j holds chunk j ^ ((r>>1)&7)).
std::vector<uint16_t> pack_linear(const float* W, int N, int K, int dtype) {
  const int nch = (K + 63) / 64;                       // K zero-padded to a multiple of 64
  std::vector<uint16_t> out((size_t)N * nch * 64);
  for (int nt = 0; nt < N / 128; ++nt)
    for (int c = 0; c < nch; ++c)
      for (int r = 0; r < 128; ++r) {
        const int half = r >> 6, rr = r & 63;
        const int n = nt * 128 + half * 64 + lds_row_to_channel(rr);
        for (int j = 0; j < 8; ++j) {
          const int chunk = j ^ ((r >> 1) & 7);
          uint16_t* dst = &out[(((size_t)nt * nch + c) * 128 + r) * 64 + j * 8];
          const float* src = W + (size_t)n * K + c * 64 + chunk * 8;
          for (int k = 0; k < 8; ++k) dst[k] = cvt16(c * 64 + chunk * 8 + k < K ? src[k] : 0.f, dtype);
        }
      }
  return out;
}

template <typename V> int tf_upload(flope_tf_encoder* e, const V* src, size_t count, void** dst) {
  void* p = nullptr;
  TF_HIP(e, hipMalloc(&p, count * sizeof(V)));
  e->allocs.push_back(p);
  TF_HIP(e, hipMemcpy(p, src, count * sizeof(V), hipMemcpyHostToDevice));
  *dst = p;
  return 0;
}

template <typename T>
int launch_linear(flope_tf_encoder* e, const TfLinear& l, const void* X, int x_f32, const void* R, void* Y, int y_f32,
                  int M, int relu, hipStream_t st) {
  if (l.packed && !y_f32 && !e->opt_generic) {
    if constexpr (!std::is_same<T, float>::value) {
      if (x_f32) {                                   // network input: fp32 [M][K] -> 16-bit [M][Kp]
        const size_t tot = (size_t)M * l.Kp;
        hipLaunchKernelGGL((tf_cast_pad<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)X, (T*)e->xin, M, l.K, l.Kp);
        X = e->xin;
      }
      const int grid = ((M + 127) / 128) * (l.N / 128);
      const size_t lds = 65536;
#define TF_GO(RELU_, RES_)                                                                                       \
  hipLaunchKernelGGL((tf_gemm_mfma<T, RELU_, RES_>), dim3(grid), dim3(256), lds, st, (const T*)X, l.packed, l.b, \
                     (const T*)R, (T*)Y, l.Kp, l.N)
      if (relu && R) TF_GO(true, true); else if (relu) TF_GO(true, false); else if (R) TF_GO(false, true); else TF_GO(false, false);
#undef TF_GO
      TF_HIP(e, hipGetLastError());
      return 0;
    }
  }
  if constexpr (!std::is_same<T, float>::value) {
    if (l.N <= 16 && !R && !x_f32 && l.K % 8 == 0 && (size_t)l.N * l.K * 4 <= 48 * 1024) {
      const int blocks = (M + 7) / 8 < 2048 ? (M + 7) / 8 : 2048;
      hipLaunchKernelGGL((tf_linear_rowwave_vec<T>), dim3(blocks), dim3(256), (size_t)l.N * l.K * 4, st, (const T*)X, l.w, l.b, Y, y_f32, M, l.K, l.N, relu);
      TF_HIP(e, hipGetLastError());
      return 0;
    }
  }
  if (l.N <= 16 && !R) {
    const int blocks = (M + 3) / 4 < 8192 ? (M + 3) / 4 : 8192;
    hipLaunchKernelGGL((tf_linear_rowwave<T>), dim3(blocks), dim3(256), 0, st, X, x_f32, l.w, l.b, Y, y_f32, M, l.K, l.N, relu);
    TF_HIP(e, hipGetLastError());
    return 0;
  }
  const size_t total = (size_t)M * l.N;
  hipLaunchKernelGGL((tf_linear_generic<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, x_f32, l.w, l.b,
                     R, Y, y_f32, M, l.K, l.N, relu);
  TF_HIP(e, hipGetLastError());
  return 0;
}

template <typename T>
int run_forward(flope_tf_encoder* e, const float* x, int B, int L, float* y, hipStream_t st) {
  const int M = B * L, d = e->d;
  int rc;
  if ((rc = launch_linear<T>(e, e->emb, x, 1, nullptr, e->h, 0, M, 0, st))) return rc;
  for (TfLayer& ly : e->layers) {
    if ((rc = launch_linear<T>(e, ly.in_proj, e->h, 0, nullptr, e->qkv, 0, M, 0, st))) return rc;
    const int Lp = (L + 31) / 32 * 32;
    bool fast_attn = false;
    if constexpr (!std::is_same<T, float>::value) {
      if (!e->opt_generic && d / e->H == 64 && Lp <= 512) {
        fast_attn = true;
        hipLaunchKernelGGL((tf_attn_mfma<T>), dim3(B * e->H), dim3(Lp / 32 * 64), (size_t)Lp * 256, st, (const T*)e->qkv,
                           (T*)e->att, L, d, e->H, Lp, 1.4426950408889634f / sqrtf(64.f));
      }
    }
    if (!fast_attn) {
      const int nw = 4, gy = (L + nw - 1) / nw < 64 ? (L + nw - 1) / nw : 64;
      hipLaunchKernelGGL((tf_attn_generic<T>), dim3(B * e->H, gy), dim3(nw * 64), (size_t)nw * L * 4, st, (const T*)e->qkv,
                         (T*)e->att, L, d, e->H);
    }
    TF_HIP(e, hipGetLastError());
    if ((rc = launch_linear<T>(e, ly.out_proj, e->att, 0, e->h, e->h2, 0, M, 0, st))) return rc;
    auto ln = [&](const void* in, void* out, const float* w, const float* b) {
      if constexpr (!std::is_same<T, float>::value) {
        if (d % 8 == 0 && d <= 2048) {
          hipLaunchKernelGGL((tf_layernorm_vec<T>), dim3((M + 3) / 4), dim3(256), 0, st, (const T*)in, (T*)out, w, b, M, d);
          return;
        }
      }
      hipLaunchKernelGGL((tf_layernorm<T>), dim3((M + 3) / 4), dim3(256), 0, st, (const T*)in, (T*)out, w, b, M, d);
    };
    ln(e->h2, e->h, ly.n1w, ly.n1b);
    if ((rc = launch_linear<T>(e, ly.lin1, e->h, 0, nullptr, e->ffb, 0, M, 1, st))) return rc;
    if ((rc = launch_linear<T>(e, ly.lin2, e->ffb, 0, e->h, e->h2, 0, M, 0, st))) return rc;
    ln(e->h2, e->h, ly.n2w, ly.n2b);
    TF_HIP(e, hipGetLastError());
  }
  return launch_linear<T>(e, e->outl, e->h, 0, nullptr, y, 1, M, 0, st);
}

}  // namespace

extern "C" const char* flope_tf_last_error(flope_tf_handle h) { return h ? h->err.c_str() : g_tf_error.c_str(); }

extern "C" int flope_tf_create(int device_id, int input_dim, int model_dim, int out_dim, int num_heads, int num_layers,
                               int ff_dim, int max_tokens, int dtype, flope_tf_handle* out) {
  if (!out) return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_create: NULL out");
  *out = nullptr;
  if (input_dim < 1 || model_dim < 1 || out_dim < 1 || num_heads < 1 || num_layers < 0 || ff_dim < 1 || max_tokens < 1)
    return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_create: non-positive dimension");
  if (model_dim % num_heads)
    return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_create: model_dim must be divisible by num_heads (torch.nn.MultiheadAttention)");
  if (dtype != FLOPE_DT_BF16 && dtype != FLOPE_DT_F16 && dtype != FLOPE_DT_F32)
    return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_create: dtype must be FLOPE_DT_BF16 / F16 / F32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return tf_fail(nullptr, FLOPE_EHIP, "flope_tf_create: no HIP device (there is no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_create: bad device id");
  flope_tf_encoder* e = new flope_tf_encoder();
  e->device = device_id; e->in_dim = input_dim; e->d = model_dim; e->out_dim = out_dim; e->H = num_heads;
  e->nl = num_layers; e->ff = ff_dim; e->max_tokens = max_tokens; e->dtype = dtype; e->esz = dtype == FLOPE_DT_F32 ? 4 : 2;
  e->Mpad = (max_tokens + 127) / 128 * 128;
  e->layers.resize(num_layers);
  auto fin = [&](int rc) { flope_tf_destroy(e); return rc; };
  if (hipSetDevice(device_id) != hipSuccess) return fin(tf_fail(nullptr, FLOPE_EHIP, "hipSetDevice failed"));
  const size_t in_pad = (size_t)(input_dim + 63) / 64 * 64;
  struct { void** p; size_t cols; } bufs[] = {{&e->xin, in_pad}, {&e->h, (size_t)model_dim}, {&e->h2, (size_t)model_dim}, {&e->qkv, (size_t)3 * model_dim},
                                              {&e->att, (size_t)model_dim}, {&e->ffb, (size_t)ff_dim}};
  for (auto& bf : bufs) {
    const size_t bytes = (size_t)e->Mpad * bf.cols * e->esz;
    if (hipMalloc(bf.p, bytes) != hipSuccess) return fin(tf_fail(nullptr, FLOPE_EHIP, "flope_tf_create: hipMalloc failed"));
    e->allocs.push_back(*bf.p);
    if (hipMemset(*bf.p, 0, bytes) != hipSuccess) return fin(tf_fail(nullptr, FLOPE_EHIP, "flope_tf_create: hipMemset failed"));
  }
  {                                                  // per create: the attribute is per device (a process-wide flag would leave a second GPU without it)
    hipFuncSetAttribute((const void*)tf_attn_mfma<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)tf_attn_mfma<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
#define TF_ATTR(T_, A_, B_) hipFuncSetAttribute((const void*)tf_gemm_mfma<T_, A_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)
    TF_ATTR(f16_t, false, false); TF_ATTR(f16_t, false, true); TF_ATTR(f16_t, true, false); TF_ATTR(f16_t, true, true);
    TF_ATTR(bf16_t, false, false); TF_ATTR(bf16_t, false, true); TF_ATTR(bf16_t, true, false); TF_ATTR(bf16_t, true, true);
#undef TF_ATTR
  }
  *out = e;
  return FLOPE_OK;
}

extern "C" int flope_tf_destroy(flope_tf_handle e) {
  if (!e) return FLOPE_OK;
  hipSetDevice(e->device);
  for (void* p : e->allocs) hipFree(p);
  delete e;
  return FLOPE_OK;
}

extern "C" int flope_tf_set_option(flope_tf_handle e, const char* name, int value) {
  if (!e || !name) return FLOPE_EINVAL;
  if (!strcmp(name, "generic")) { const int old = e->opt_generic; e->opt_generic = value ? 1 : 0; return old; }
  return tf_fail(e, FLOPE_EINVAL, std::string("flope_tf_set_option: unknown option ") + name);
}

extern "C" int flope_tf_load_weights(flope_tf_handle e, int n, const char* const* names, const float* const* host_ptrs,
                                     const int* ndims, const int64_t* const* shapes) {
  if (!e) return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_load_weights: NULL handle");
  if (n < 0 || (n > 0 && (!names || !host_ptrs || !ndims || !shapes)))
    return tf_fail(e, FLOPE_EINVAL, "flope_tf_load_weights: NULL argument");
  TF_HIP(e, hipSetDevice(e->device));
  e->loaded = false;
  std::map<std::string, std::pair<const float*, std::vector<int64_t>>> ts;
  for (int i = 0; i < n; ++i) {
    if (!names[i] || !host_ptrs[i] || ndims[i] < 0 || ndims[i] > 8 || (ndims[i] > 0 && !shapes[i]))
      return tf_fail(e, FLOPE_EINVAL, "flope_tf_load_weights: malformed entry " + std::to_string(i));
    ts[names[i]] = std::make_pair(host_ptrs[i], std::vector<int64_t>(shapes[i], shapes[i] + ndims[i]));
  }
  auto get = [&](const std::string& name, std::vector<int64_t> want, const float** p) -> int {
    auto it = ts.find(name);
    if (it == ts.end()) return tf_fail(e, FLOPE_EWEIGHTS, "state_dict entry missing: " + name);
    if (it->second.second != want) return tf_fail(e, FLOPE_EWEIGHTS, "state_dict entry has the wrong shape: " + name);
    size_t cnt = 1;
    for (int64_t s : want) cnt *= (size_t)s;
    for (size_t i = 0; i < cnt; ++i)
      if (!std::isfinite(it->second.first[i])) return tf_fail(e, FLOPE_EWEIGHTS, "non-finite value in " + name);
    *p = it->second.first;
    return 0;
  };
  auto linear = [&](const std::string& wn, const std::string& bn, int N, int K, TfLinear* l) -> int {
    const float *w, *b;
    int rc;
    if ((rc = get(wn, {N, K}, &w)) || (rc = get(bn, {N}, &b))) return rc;
    l->N = N; l->K = K;
    if ((rc = tf_upload(e, w, (size_t)N * K, (void**)&l->w)) || (rc = tf_upload(e, b, (size_t)N, (void**)&l->b))) return rc;
    l->Kp = (K + 63) / 64 * 64;
    // 16-bit activations have row stride K, so only the fp32 network input (re-laid out by tf_cast_pad) may need padding
    if (e->dtype != FLOPE_DT_F32 && N % 128 == 0 && (K % 64 == 0 || l == &e->emb)) {
      const std::vector<uint16_t> pk = pack_linear(w, N, K, e->dtype);
      if ((rc = tf_upload(e, pk.data(), pk.size(), &l->packed))) return rc;
    }
    return 0;
  };
  auto vec = [&](const std::string& name, int N, float** dst) -> int {
    const float* p;
    int rc;
    if ((rc = get(name, {N}, &p))) return rc;
    return tf_upload(e, p, (size_t)N, (void**)dst);
  };
  int rc;
  if ((rc = linear("embedding.weight", "embedding.bias", e->d, e->in_dim, &e->emb))) return rc;
  for (int i = 0; i < e->nl; ++i) {
    const std::string p = "transformer_encoder.layers." + std::to_string(i) + ".";
    TfLayer& ly = e->layers[i];
    if ((rc = linear(p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", 3 * e->d, e->d, &ly.in_proj)) ||
        (rc = linear(p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", e->d, e->d, &ly.out_proj)) ||
        (rc = linear(p + "linear1.weight", p + "linear1.bias", e->ff, e->d, &ly.lin1)) ||
        (rc = linear(p + "linear2.weight", p + "linear2.bias", e->d, e->ff, &ly.lin2)) ||
        (rc = vec(p + "norm1.weight", e->d, &ly.n1w)) || (rc = vec(p + "norm1.bias", e->d, &ly.n1b)) ||
        (rc = vec(p + "norm2.weight", e->d, &ly.n2w)) || (rc = vec(p + "norm2.bias", e->d, &ly.n2b)))
      return rc;
  }
  if ((rc = linear("out_layer.weight", "out_layer.bias", e->out_dim, e->d, &e->outl))) return rc;
  TF_HIP(e, hipDeviceSynchronize());
  e->loaded = true;
  return FLOPE_OK;
}

extern "C" int flope_tf_forward(flope_tf_handle e, const float* x_dev, int batch, int seq_len, float* y_dev, void* stream) {
  if (!e) return tf_fail(nullptr, FLOPE_EINVAL, "flope_tf_forward: NULL handle");
  if (!e->loaded) return tf_fail(e, FLOPE_ESTATE, "flope_tf_forward: weights not loaded");
  if (batch < 0 || seq_len < 0) return tf_fail(e, FLOPE_EINVAL, "flope_tf_forward: negative size");
  if (batch == 0 || seq_len == 0) return FLOPE_OK;                 // empty batch: nothing to do, buffers may be NULL
  if (!x_dev || !y_dev) return tf_fail(e, FLOPE_EINVAL, "flope_tf_forward: NULL buffer");
  if ((long long)batch * seq_len > e->max_tokens)
    return tf_fail(e, FLOPE_EINVAL, "flope_tf_forward: batch*seq_len exceeds max_tokens given to flope_tf_create");
  TF_HIP(e, hipSetDevice(e->device));
  hipStream_t st = (hipStream_t)stream;
  if (e->dtype == FLOPE_DT_F32) return run_forward<float>(e, x_dev, batch, seq_len, y_dev, st);
  if (e->dtype == FLOPE_DT_F16) return run_forward<f16_t>(e, x_dev, batch, seq_len, y_dev, st);
  return run_forward<bf16_t>(e, x_dev, batch, seq_len, y_dev, st);
}

// algorithmic FLOPs of one forward (2*MAC: linears + QK^T + PV)
extern "C" double flope_tf_forward_flops(flope_tf_handle e, int batch, int seq_len) {
  if (!e) return 0.0;
  const double M = (double)batch * seq_len, d = e->d;
  double mac = M * e->in_dim * d + M * d * e->out_dim;
  mac += e->nl * (M * d * 3 * d + M * d * d + 2.0 * M * d * e->ff + 2.0 * M * seq_len * d);
  return 2.0 * mac;
}

#ifdef FLOPE_STAG_DBG
extern "C" int flope_tfdbg_read(unsigned long long* dst_host, int cap_records) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tfdbg_n), sizeof(n)) != hipSuccess) return -1;
  const int m0 = (int)(n < 512u ? n : 512u), m = m0 < cap_records ? m0 : cap_records;
  if (m > 0 && hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_tfdbg), (size_t)m * 64) != hipSuccess) return -1;
  n = 0;
  hipMemcpyToSymbol(HIP_SYMBOL(g_tfdbg_n), &n, sizeof(n));
  return m;
}
#endif
