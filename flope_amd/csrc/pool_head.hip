// Max-pool, global average pool and the fp32 regression head + special Procrustes.
//
// Reference call sites (all eval-mode semantics):
//   maxpool      torchvision ResNet._forward_impl (MaxPool2d k3 s2 p1) via posenet.py:25
//   avgpool      base.avgpool = AdaptiveAvgPool2d(1)                  posenet.py:12
//   fc.0 + ReLU  base.fc = Sequential(Linear(512,2048), ReLU), F.relu posenet.py:13-16,26
//   fc_rot       Linear(2048, 9)                                      posenet.py:19,33
//   Procrustes   roma.special_procrustes(x.reshape(-1,3,3))           utils/conversion.py:54-58
//   yaw-null     nullify_yaw_batch                                    utils/mvg.py:240-251
//   Rt compose   Rt[:,:3,:3]=R; Rt[:,:3,3]=xyz                        fast_pose_predictor.py:142-144
// The whole head is fp32 (fp64 inside the 4x4 eigen-solve): it is 0.06 % of the
// FLOPs and carries the 1e-3 rotation tolerance.
#include "common.h"
#include "pose_math.h"

// ---------------------------------------------------------------------------
// 3x3 / s2 / p1 max-pool on zero-bordered NHWC.  Inputs are post-ReLU (>= 0), so
// the zero border is equivalent to the reference's -inf padding.
template <typename T> struct Vec8;   // 8 consecutive channels
template <> struct Vec8<bf16_t> { typedef u32x4 type; };
template <> struct Vec8<f16_t> { typedef u32x4 type; };

template <typename T>
__device__ __forceinline__ void max8(u32x4& a, const u32x4 b) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float lo = fmaxf(unpack_lo<T>(a[q]), unpack_lo<T>(b[q]));
    const float hi = fmaxf(unpack_hi<T>(a[q]), unpack_hi<T>(b[q]));
    a[q] = pack2<T>(lo, hi);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const PoolP p) {
  const int cg = p.C / 8;                                  // 16-byte channel groups
  const size_t total = (size_t)p.B * p.Ho * p.Wo * cg;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % cg);
    size_t r = i / cg;
    const int wo = (int)(r % p.Wo); r /= p.Wo;
    const int ho = (int)(r % p.Ho);
    const int b = (int)(r / p.Ho);
    const char* src = (const char*)p.in + ((((size_t)b * p.Hip + 2 * ho) * p.Wip + 2 * wo) * p.C + c8 * 8) * 2;
    const size_t rowB = (size_t)p.Wip * p.C * 2, pixB = (size_t)p.C * 2;
    u32x4 m = *(const u32x4*)src;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        if (dy | dx) max8<T>(m, *(const u32x4*)(src + dy * rowB + dx * pixB));
    char* dst = (char*)p.out + ((((size_t)b * (p.Ho + 2) + ho + 1) * (p.Wo + 2) + wo + 1) * p.C + c8 * 8) * 2;
    *(u32x4*)dst = m;
  }
}

__global__ __launch_bounds__(256) void maxpool_f32_kernel(const PoolP p) {
  const size_t total = (size_t)p.B * p.Ho * p.Wo * p.C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % p.C);
    size_t r = i / p.C;
    const int wo = (int)(r % p.Wo); r /= p.Wo;
    const int ho = (int)(r % p.Ho);
    const int b = (int)(r / p.Ho);
    const float* src = (const float*)p.in + (((size_t)b * p.Hip + 2 * ho) * p.Wip + 2 * wo) * p.C + c;
    float m = 0.f;
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, src[((size_t)dy * p.Wip + dx) * p.C]);
    ((float*)p.out)[(((size_t)b * (p.Ho + 2) + ho + 1) * (p.Wo + 2) + wo + 1) * p.C + c] = m;
  }
}

extern "C" int flope_maxpool_launch(const PoolP* p, int dtype, void* stream) {
  const size_t work = (size_t)p->B * p->Ho * p->Wo * (dtype == 2 ? p->C : p->C / 8);
  const int grid = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, *p);
  else if (dtype == 1) hipLaunchKernelGGL(maxpool_kernel<f16_t>, dim3(grid), dim3(256), 0, st, *p);
  else hipLaunchKernelGGL(maxpool_f32_kernel, dim3(grid), dim3(256), 0, st, *p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Global average pool: padded NHWC [B][h+2][w+2][C] -> float [B][C].
// One block per (image, 128 channels); thread = (pixel group pg of 4, channel pair of 64): the h*w pixels are strided
// over the 4 groups so four independent 4-byte loads per channel pair are in flight, and the 128-channel slabs of one
// image go to different workgroups (one workgroup per image took 41 us on 16 x 16 maps).
template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const void* in, float* out, int h, int w, int C) {
  __shared__ float red[4][128];
  const int b = blockIdx.x, c0 = blockIdx.y * 64;          // c0 in channel pairs
  const int Wp = w + 2, npx = h * w;
  const float inv = 1.f / (float)npx;
  const int pg = threadIdx.x >> 6, cl = threadIdx.x & 63;
  const int c2 = c0 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c2 < C / 2) {
    const unsigned* base = (const unsigned*)((const char*)in + (size_t)b * (h + 2) * Wp * C * 2) + c2;
    if (npx > 64) {
#pragma unroll 16
      for (int i = pg; i < npx; i += 4) {        // large maps (512 x 512 crops: 16 x 16): 16 independent loads in flight
        const int y = i / w, x = i - y * w;
        const unsigned v = base[(size_t)((y + 1) * Wp + x + 1) * (C / 2)];
        s0 += unpack_lo<T>(v);
        s1 += unpack_hi<T>(v);
      }
    } else {
      // r04 (the head is the step's exposed tail): the <= 16 pixels of this thread's group are loaded FIRST, their offsets stepped
      // without a division, then added in the same order as before (bit-identical)
      constexpr int NJ = 16;
      unsigned v[NJ];
      int x = pg % w, y = pg / w;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool in_ = pg + 4 * j < npx;
        v[j] = in_ ? base[(size_t)((y + 1) * Wp + x + 1) * (C / 2)] : 0u;
        x += 4;
        while (x >= w) { x -= w; ++y; }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (pg + 4 * j < npx) { s0 += unpack_lo<T>(v[j]); s1 += unpack_hi<T>(v[j]); }
    }
  }
  red[pg][cl * 2] = s0; red[pg][cl * 2 + 1] = s1;
  __syncthreads();
  if (threadIdx.x < 128 && c0 * 2 + (int)threadIdx.x < C) {
    const int t = threadIdx.x;
    out[(size_t)b * C + c0 * 2 + t] = (red[0][t] + red[1][t] + red[2][t] + red[3][t]) * inv;
  }
}

__global__ __launch_bounds__(256) void avgpool_f32_kernel(const float* in, float* out, int h, int w, int C) {
  const int b = blockIdx.x;
  const int Wp = w + 2;
  const float inv = 1.f / (float)(h * w);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) s += in[(((size_t)b * (h + 2) + y + 1) * Wp + x + 1) * C + c];
    out[(size_t)b * C + c] = s * inv;
  }
}

extern "C" int flope_avgpool_launch(const void* in, float* out, int B, int h, int w, int C, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(B, (C / 2 + 63) / 64);
  if (dtype == 0) hipLaunchKernelGGL(avgpool_kernel<bf16_t>, grid, dim3(256), 0, st, in, out, h, w, C);
  else if (dtype == 1) hipLaunchKernelGGL(avgpool_kernel<f16_t>, grid, dim3(256), 0, st, in, out, h, w, C);
  else hipLaunchKernelGGL(avgpool_f32_kernel, dim3(B), dim3(256), 0, st, (const float*)in, out, h, w, C);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// fc.0 + ReLU: hidden[b][n] = relu(sum_k feat[b][k] * W1[n][k] + b1[n]) on the
// exact-fp32 MFMA v_mfma_f32_16x16x4_f32 (A = W1 rows, B = images).
// One wave owns 16 outputs x 64 images.  Each lane loads 16 contiguous bytes of
// its W1 row / feature row per 16-deep K block and feeds element s to MFMA step s
// on BOTH operands (lane (r, g) supplies k = k0 + 4g + s), so the K order inside a
// block is permuted identically for A and B and the sum is unchanged.
__global__ __launch_bounds__(256) void fc1_kernel(const float* __restrict__ feat, const float* __restrict__ W1,
                                                  const float* __restrict__ b1, float* __restrict__ hidden,
                                                  int B, int K, int N) {
  // one wave = 16 outputs x 32 images (two MFMA column tiles); N/16 * B/32 waves keep every CU busy
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int ntile = blockIdx.x * 4 + wave;
  const int n0 = ntile * 16;
  const int img0 = blockIdx.y * 32;
  if (n0 >= N) return;
  const float* wrow = W1 + (size_t)(n0 + r16) * K + 4 * g;
  const float* frow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) frow[t] = feat + (size_t)min(img0 + t * 16 + r16, B - 1) * K + 4 * g;
  f32x4 acc[2][2];                                          // [image tile][K parity]: 4 independent chains
#pragma unroll
  for (int t = 0; t < 2; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
#pragma unroll 2
  for (int k0 = 0; k0 < K; k0 += 32) {
    const f32x4 a0 = *(const f32x4*)(wrow + k0), a1 = *(const f32x4*)(wrow + k0 + 16);
    f32x4 f0[2], f1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { f0[t] = *(const f32x4*)(frow[t] + k0); f1[t] = *(const f32x4*)(frow[t] + k0 + 16); }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], f0[t][s], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], f1[t][s], acc[t][1], 0, 0, 0);
      }
  }
  // D: col = lane&15 = image, row = 4*(lane>>4)+reg = output n
  const f32x4 bias = *(const f32x4*)(b1 + n0 + 4 * g);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int img = img0 + t * 16 + r16;
    if (img < B) {
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = fmaxf(acc[t][0][q] + acc[t][1][q] + bias[q], 0.f);
      *(f32x4*)(hidden + (size_t)img * N + n0 + 4 * g) = o;
    }
  }
}

// The same arithmetic (same K order per output: bit-identical) with coalesced operands.  In fc1_kernel every wave-load puts
// its 64 lanes on 16 rows = 64 different cache lines (weights 2 KB apart, feature rows too) and the L1's tag rate, not bytes,
// sets its 19 us.  Here W1 is pre-packed in A-fragment order [n tile][K / 32][2][lane][4] (a wave-load = one contiguous KiB)
// and the workgroup's 32 feature rows are staged once into LDS (row pitch K + 4 floats: conflict-free ds_read_b128).
// TI = image tiles of 16 per wave.  r04: 1 (was 2) -- a wave's 128 instead of 256 dependent-chain MFMAs (32 cycles of issue each) are
// the kernel's latency, and twice the workgroups fill more of the chip; the K order per output is unchanged (bit-identical).
template <int TI>
__global__ __launch_bounds__(256) void fc1_packed_kernel(const float* __restrict__ feat, const float* __restrict__ W1p,
                                                         const float* __restrict__ b1, float* __restrict__ hidden,
                                                         int B, int K, int N) {
  extern __shared__ __attribute__((aligned(16))) float sfeat[];           // [16 TI][K + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int g = lane >> 4, r16 = lane & 15;
  const int ntile = blockIdx.x * (nthr >> 6) + wave;        // 4 waves per workgroup, 1 for small batches (4x the workgroups)
  const int n0 = ntile * 16;
  const int img0 = blockIdx.y * (16 * TI);
  const int KB = K >> 5, pitch = K + 4, k4 = K >> 2;
  const f32x4* wp = (const f32x4*)W1p + ((size_t)min(ntile, N / 16 - 1) * KB * 2) * 64 + lane;
#ifndef FLOPE_FC1_WPD
#define FLOPE_FC1_WPD 4
#endif
  constexpr int WPD = FLOPE_FC1_WPD;                        // weight K blocks in flight (issued before the feature staging)
  f32x4 wa[WPD][2];
#pragma unroll
  for (int d = 0; d < WPD; ++d) { const int kb = min(d, KB - 1); wa[d][0] = wp[(size_t)kb * 128]; wa[d][1] = wp[(size_t)kb * 128 + 64]; }
  for (int i0 = tid; i0 < 16 * TI * k4; i0 += 8 * nthr) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u * nthr, 16 * TI * k4 - 1), row = i / k4, c = i - row * k4;
      v[u] = *(const f32x4*)(feat + (size_t)min(img0 + row, B - 1) * K + 4 * c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nthr, row = i / k4, c = i - row * k4;
      if (i < 16 * TI * k4) *(f32x4*)(sfeat + row * pitch + 4 * c) = v[u];
    }
  }
  __syncthreads();
  if (n0 >= N) return;
  f32x4 acc[TI][2];                                         // [image tile][K parity]: independent chains
#pragma unroll
  for (int t = 0; t < TI; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
  const float* f0p = sfeat + r16 * pitch + 4 * g;
  for (int kb0 = 0; kb0 < KB; kb0 += WPD) {
#pragma unroll
    for (int d = 0; d < WPD; ++d) {
      const int kb = kb0 + d;
      if (kb >= KB) break;
      const f32x4 a0 = wa[d][0], a1 = wa[d][1];
      const int nb = min(kb + WPD, KB - 1);
      wa[d][0] = wp[(size_t)nb * 128]; wa[d][1] = wp[(size_t)nb * 128 + 64];
      f32x4 f0[TI], f1[TI];
#pragma unroll
      for (int t = 0; t < TI; ++t) { f0[t] = *(const f32x4*)(f0p + t * 16 * pitch + kb * 32); f1[t] = *(const f32x4*)(f0p + t * 16 * pitch + kb * 32 + 16); }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < TI; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], f0[t][s], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], f1[t][s], acc[t][1], 0, 0, 0);
        }
    }
  }
  const f32x4 bias = *(const f32x4*)(b1 + n0 + 4 * g);
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    const int img = img0 + t * 16 + r16;
    if (img < B) {
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = fmaxf(acc[t][0][q] + acc[t][1][q] + bias[q], 0.f);
      *(f32x4*)(hidden + (size_t)img * N + n0 + 4 * g) = o;
    }
  }
}

// (r03 built avgpool + fc.0 as ONE launch -- 8 images x 256 outputs per workgroup, bit-identical to the two launches -- and measured
// 38.9 us against 8.5 + 16.0: the fusion multiplies either the pooling reads or the weight stream.  Removed in r04)

// generic fallback when K % 16 != 0 or N % 16 != 0 (non-default backbone_out_dim)
__global__ __launch_bounds__(256) void fc1_simple_kernel(const float* feat, const float* W1, const float* b1,
                                                         float* hidden, int B, int K, int N) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * N) return;
  const int n = (int)(i % N), b = (int)(i / N);
  float s = b1[n];
  for (int k = 0; k < K; ++k) s = fmaf(feat[(size_t)b * K + k], W1[(size_t)n * K + k], s);
  hidden[i] = fmaxf(s, 0.f);
}

// W1p: W1 in fragment order (host_pack.h pack_fc1) or NULL
extern "C" int flope_fc1_launch(const float* feat, const float* W1, const float* W1p, const float* b1, float* hidden, int B, int K,
                                int N, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)16 * (K + 4) * sizeof(float);  // 33 KB at K = 512: inside the default dynamic-LDS limit (no per-device attribute)
  if (W1p && K % 32 == 0 && N % 16 == 0 && lds <= 64 * 1024) {
    const int waves = 4;                                    // (one wave per workgroup for small batches measured slower: 20.9 vs 19.0 us)
    const dim3 grid((N / 16 + waves - 1) / waves, (B + 15) / 16);
    hipLaunchKernelGGL(fc1_packed_kernel<1>, grid, dim3(64 * waves), lds, st, feat, W1p, b1, hidden, B, K, N);
  } else if (K % 32 == 0 && N % 16 == 0) {
    const dim3 grid((N / 16 + 3) / 4, (B + 31) / 32);
    hipLaunchKernelGGL(fc1_kernel, grid, dim3(256), 0, st, feat, W1, b1, hidden, B, K, N);
  } else {
    const size_t total = (size_t)B * N;
    hipLaunchKernelGGL(fc1_simple_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feat, W1, b1,
                       hidden, B, K, N);
  }
  return (int)hipGetLastError();
}

// fc_rot + Procrustes: one wave per image.
// Optional pose assembly in the same launch (fast_pose_predictor.py:131-144): Rt_out [B,16] = [[R', xyz],[0,0,0,1]] with
// R' = yaw-nullified R when `nullify`; xyz [B,3] may be NULL (zeros).
__global__ __launch_bounds__(256) void fc2_procrustes_kernel(const float* __restrict__ hidden,
                                                             const float* __restrict__ W2,
                                                             const float* __restrict__ b2, float* r9_out,
                                                             float* R_out, int B, int K, const float* xyz,
                                                             int nullify, float* Rt_out) {
  const int lane = threadIdx.x & 63;
  const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (img >= B) return;
  float acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.f;
  const float* h = hidden + (size_t)img * K;
  if ((K & 3) == 0) {
#pragma unroll 2
    for (int k = lane * 4; k < K; k += 256) {
      const f32x4 hv = *(const f32x4*)(h + k);
      f32x4 wv[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) wv[j] = *(const f32x4*)(W2 + (size_t)j * K + k);
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[j] += hv[0] * wv[j][0] + hv[1] * wv[j][1] + hv[2] * wv[j][2] + hv[3] * wv[j][3];
    }
  } else {
    for (int k = lane; k < K; k += 64)
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[j] = fmaf(h[k], W2[(size_t)j * K + k], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
  }
  if (lane == 0) {
    float M[9], R[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) M[j] = acc[j] + b2[j];
    if (r9_out)
      for (int j = 0; j < 9; ++j) r9_out[(size_t)img * 9 + j] = M[j];
    if (R_out || Rt_out) {
      procrustes3x3(M, R);
      if (R_out)
        for (int j = 0; j < 9; ++j) R_out[(size_t)img * 9 + j] = R[j];
      if (Rt_out) {
        float o[9];
        if (nullify) nullify_yaw3x3(R, o);
        else for (int j = 0; j < 9; ++j) o[j] = R[j];
        float* t = Rt_out + (size_t)img * 16;
        for (int a = 0; a < 3; ++a) {
          for (int c = 0; c < 3; ++c) t[a * 4 + c] = o[a * 3 + c];
          t[a * 4 + 3] = xyz ? xyz[(size_t)img * 3 + a] : 0.f;
        }
        t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 1.f;
      }
    }
  }
}

// The same head with the K range of one image split over the four waves of a workgroup (r03): one workgroup per image, every
// load of a wave in flight at once (K = 2048: 2 x (1 + 9) 16-byte loads per lane), wave sums combined in a fixed order through
// LDS -- deterministic, but not the summation order of fc2_procrustes_kernel (which stays for K % 1024 != 0).
__global__ __launch_bounds__(256) void fc2_procrustes_k4_kernel(const float* __restrict__ hidden, const float* __restrict__ W2,
                                                                const float* __restrict__ b2, float* r9_out, float* R_out,
                                                                int B, int K, const float* xyz, int nullify, float* Rt_out) {
  __shared__ float part[4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int img = blockIdx.x;
  const int kw = K >> 2;                                    // this wave's K range: [wave * kw, (wave + 1) * kw), kw % 256 == 0
  float acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.f;
  const float* hp = hidden + (size_t)img * K + wave * kw + lane * 4;
  const float* wq = W2 + wave * kw + lane * 4;
#pragma unroll 2
  for (int k = 0; k < kw; k += 256) {
    const f32x4 hv = *(const f32x4*)(hp + k);
    f32x4 wv[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wv[j] = *(const f32x4*)(wq + (size_t)j * K + k);
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] += hv[0] * wv[j][0] + hv[1] * wv[j][1] + hv[2] * wv[j][2] + hv[3] * wv[j][3];
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 9; ++j) part[wave][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M[9], R[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) M[j] = ((part[0][j] + part[1][j]) + (part[2][j] + part[3][j])) + b2[j];
    if (r9_out)
      for (int j = 0; j < 9; ++j) r9_out[(size_t)img * 9 + j] = M[j];
    if (R_out || Rt_out) {
      procrustes3x3(M, R);
      if (R_out)
        for (int j = 0; j < 9; ++j) R_out[(size_t)img * 9 + j] = R[j];
      if (Rt_out) {
        float o[9];
        if (nullify) nullify_yaw3x3(R, o);
        else for (int j = 0; j < 9; ++j) o[j] = R[j];
        float* t = Rt_out + (size_t)img * 16;
        for (int a = 0; a < 3; ++a) {
          for (int c = 0; c < 3; ++c) t[a * 4 + c] = o[a * 3 + c];
          t[a * 4 + 3] = xyz ? xyz[(size_t)img * 3 + a] : 0.f;
        }
        t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 1.f;
      }
    }
  }
}

extern "C" int flope_fc2_procrustes_k4_launch(const float* hidden, const float* W2, const float* b2, float* r9, float* R, int B,
                                              int K, const float* xyz, int nullify, float* Rt, void* stream) {
  if (K % 1024 || B < 1) return 0;
  hipLaunchKernelGGL(fc2_procrustes_k4_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, hidden, W2, b2, r9, R, B, K, xyz, nullify, Rt);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}

extern "C" int flope_fc2_procrustes_launch(const float* hidden, const float* W2, const float* b2, float* r9,
                                           float* R, int B, int K, const float* xyz, int nullify, float* Rt,
                                           void* stream) {
  hipLaunchKernelGGL(fc2_procrustes_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, hidden, W2, b2, r9,
                     R, B, K, xyz, nullify, Rt);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(64) void procrustes_kernel(const float* M, float* R, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float m[9], r[9];
  for (int j = 0; j < 9; ++j) m[j] = M[(size_t)i * 9 + j];
  procrustes3x3(m, r);
  for (int j = 0; j < 9; ++j) R[(size_t)i * 9 + j] = r[j];
}

__global__ __launch_bounds__(64) void nullify_yaw_kernel(const float* R, float* O, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float r[9], o[9];
  for (int j = 0; j < 9; ++j) r[j] = R[(size_t)i * 9 + j];
  nullify_yaw3x3(r, o);
  for (int j = 0; j < 9; ++j) O[(size_t)i * 9 + j] = o[j];
}

__global__ __launch_bounds__(64) void compose_pose_kernel(const float* R, const float* xyz, int n, int nullify,
                                                          float* Rt) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float r[9], o[9];
  for (int j = 0; j < 9; ++j) r[j] = R[(size_t)i * 9 + j];
  if (nullify) nullify_yaw3x3(r, o);
  else for (int j = 0; j < 9; ++j) o[j] = r[j];
  float* t = Rt + (size_t)i * 16;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) t[a * 4 + b] = o[a * 3 + b];
    t[a * 4 + 3] = xyz ? xyz[(size_t)i * 3 + a] : 0.f;
  }
  t[12] = 0.f; t[13] = 0.f; t[14] = 0.f; t[15] = 1.f;
}

extern "C" int flope_procrustes(const float* M_dev, float* R_dev, int n, void* stream) {
  if (n < 0 || (n > 0 && (!M_dev || !R_dev))) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(procrustes_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, M_dev, R_dev, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int flope_nullify_yaw(const float* R_dev, float* out_dev, int n, void* stream) {
  if (n < 0 || (n > 0 && (!R_dev || !out_dev))) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(nullify_yaw_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, R_dev, out_dev, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int flope_compose_pose(const float* R_dev, const float* xyz_dev, int n, int nullify_yaw, float* Rt_dev,
                                  void* stream) {
  if (n < 0 || (n > 0 && (!R_dev || !Rt_dev))) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(compose_pose_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, R_dev, xyz_dev, n,
                     nullify_yaw, Rt_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
