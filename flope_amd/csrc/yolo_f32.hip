// Strict float32 mode of the YOLO11-seg detector (flope_yolo_create(..., FLOPE_DT_F32)): the counterpart of the pose
// engine's naive.hip.  The reference runs ultralytics in float32 and returns INTEGER boxes (int16) and a uint8 mask
// (sunflower/predictor/fast_pose_predictor.py:49-56); with 16-bit maps those integers can only be shown close to the
// float32 arithmetic's, never equal.  Here every map is float32 NHWC (same views, same graph, same schedule in program
// order), every convolution is a plain fused-multiply-add chain over k = (tap, channel) in float32 -- no MFMA, no
// 16-bit rounding, accurate expf -- so that bbox / mask / poses can be compared with the all-float32 oracle pipeline
// (tests/test_gpu_yolo.py).  Parity / debug only: ~50x slower than the 16-bit path.
//
// Same launch-parameter structs as yolo.hip (yolo.h), read with float32 semantics:
//   YConvP.w    float32 [rows][k*k*Cin] (row = output channel, k index = tap * Cin + ci; rows of the transposed conv are
//               (dy*2+dx)*dc + co), bias float32 [rows] in channel order; ld* in float elements
//   YDwP / YPoolP / YUpP / YAttnP / YLetterP / YMaskP: as documented there, maps float32
#include "common.h"
#include "yolo.h"

namespace {

__device__ __forceinline__ float silu32(float v) { return __fdiv_rn(v, 1.f + expf(-v)); }

// one thread = one output pixel x CO consecutive output rows (uniform per workgroup: weights are scalar loads)
template <int CO>
__global__ __launch_bounds__(256) void y32_conv_kernel(const YConvP p) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * CO;
  const int rows = p.out_mode == 2 ? 4 * p.dc : p.Cout;
  if (m >= p.M) return;
  const int oy = m / p.Wo, ox = m - oy * p.Wo;
  const int pad = p.k == 3 ? 1 : 0, K = p.k * p.k * p.Cin;
  const float* const in = (const float*)p.in;
  const float* const w = (const float*)p.w;
  float acc[CO];
#pragma unroll
  for (int i = 0; i < CO; ++i) acc[i] = r0 + i < rows ? p.bias[r0 + i] : 0.f;
  for (int ky = 0; ky < p.k; ++ky) {
    const int iy = oy * p.stride - pad + ky;
    if ((unsigned)iy >= (unsigned)p.Hi) continue;
    for (int kx = 0; kx < p.k; ++kx) {
      const int ix = ox * p.stride - pad + kx;
      if ((unsigned)ix >= (unsigned)p.Wi) continue;
      const float* ip = in + (size_t)(iy * p.Wi + ix) * p.ldi;
      const float* wp = w + (size_t)r0 * K + (ky * p.k + kx) * p.Cin;
      for (int c = 0; c < p.Cin; c += 4) {
        const f32x4 x = *(const f32x4*)(ip + c);
#pragma unroll
        for (int i = 0; i < CO; ++i) {
          if (r0 + i >= rows) break;                      // uniform
          const f32x4 wv = *(const f32x4*)(wp + (size_t)i * K + c);
          acc[i] = fmaf(x[0], wv[0], acc[i]);
          acc[i] = fmaf(x[1], wv[1], acc[i]);
          acc[i] = fmaf(x[2], wv[2], acc[i]);
          acc[i] = fmaf(x[3], wv[3], acc[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CO; ++i) {
    const int r = r0 + i;
    if (r >= rows) break;
    float v = p.act ? silu32(acc[i]) : acc[i];
    if (p.out_mode == 2) {                                // ConvTranspose2d 2x2 s2: row -> (dy, dx) quadrant, channel
      const int quad = r / p.dc, co = r - quad * p.dc;
      const size_t opix = (size_t)(2 * oy + (quad >> 1)) * (2 * p.Wo) + 2 * ox + (quad & 1);
      ((float*)p.out)[opix * p.ldo + co] = v;
    } else {
      if (p.res) v += ((const float*)p.res)[(size_t)m * p.ldr + r];
      ((float*)p.out)[(size_t)m * p.ldo + r] = v;         // out_mode 0: map view; 1: prediction rows (both float32 here)
    }
  }
}

__global__ __launch_bounds__(256) void y32_dw_kernel(const YDwP p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.H * p.W * p.C) return;
  const int pix = idx / p.C, c = idx - pix * p.C;
  const int y = pix / p.W, x = pix - y * p.W;
  const int ci = p.blk ? (c / p.blk) * p.blk_stride + p.blk_off + c % p.blk : c;
  float a = p.bias[c];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
    a = fmaf(((const float*)p.in)[(size_t)(iy * p.W + ix) * p.ldi + ci], p.w[t * p.C + c], a);
  }
  if (p.act) a = silu32(a);
  if (p.add) a += ((const float*)p.add)[(size_t)pix * p.lda + c];
  ((float*)p.out)[(size_t)pix * p.ldo + c] = a;
}

// n cascaded 5x5 s1 p2 max-pools = ring-wise maxima of one (4n+1)^2 sweep (yolo.hip: ypool_kernel)
__global__ __launch_bounds__(256) void y32_pool_kernel(const YPoolP p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.H * p.W * p.C) return;
  const int pix = idx / p.C, c = idx - pix * p.C;
  const int y = pix / p.W, x = pix - y * p.W;
  float a[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  const int R = 2 * p.n;
  for (int dy = -R; dy <= R; ++dy) {
    const int iy = y + dy;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    const int ry = dy < 0 ? -dy : dy;
    for (int dx = -R; dx <= R; ++dx) {
      const int ix = x + dx;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      const int rx = dx < 0 ? -dx : dx, ring = ry > rx ? ry : rx;
      const float f = ((const float*)p.in)[(size_t)(iy * p.W + ix) * p.ldi + c];
      a[2] = fmaxf(a[2], f);
      if (ring <= 4) a[1] = fmaxf(a[1], f);
      if (ring <= 2) a[0] = fmaxf(a[0], f);
    }
  }
  for (int r = 0; r < p.n; ++r)
    ((float*)p.out)[(size_t)pix * p.ldo + r * p.C + c] = (r == p.n - 1) ? a[2] : a[r];
}

__global__ __launch_bounds__(256) void y32_up_kernel(const YUpP p) {
  const int W2 = 2 * p.W;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4 * p.H * p.W * p.C) return;
  const int pix = idx / p.C, c = idx - pix * p.C;
  const int y = pix / W2, x = pix - y * W2;
  ((float*)p.out)[(size_t)pix * p.ldo + c] = ((const float*)p.in)[(size_t)((y >> 1) * p.W + (x >> 1)) * p.ldi + c];
}

// ultralytics Attention, float32: attn = softmax((q^T k) * scale), out = v attn^T.  One workgroup = 16 queries of one head;
// 16 lanes share a query (keys j = lane mod 16 in the score passes, output dims 4 lane .. 4 lane + 3 in the value pass).
__global__ __launch_bounds__(256) void y32_attn_kernel(const YAttnP p) {
  extern __shared__ float S32[];                       // [16][N]
  const int tid = threadIdx.x, ql = tid >> 4, kl = tid & 15;
  const int h = blockIdx.y;
  const int qi = min(blockIdx.x * 16 + ql, p.N - 1);
  const float* base = (const float*)p.qkv + (size_t)h * 128;
  float q[32];
  {
    const float* qp = base + (size_t)qi * p.ld;
#pragma unroll
    for (int c = 0; c < 32; ++c) q[c] = qp[c];
  }
  float* Sq = S32 + (size_t)ql * p.N;
  float mx = -3.0e38f;
  for (int j = kl; j < p.N; j += 16) {
    const float* kp = base + (size_t)j * p.ld + 32;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) s = fmaf(q[c], kp[c], s);
    s *= p.scale;
    Sq[j] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
  float l = 0.f;
  for (int j = kl; j < p.N; j += 16) {
    const float e = expf(Sq[j] - mx);
    Sq[j] = e;
    l += e;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 16);
  __syncthreads();
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float inv = 1.f / l;
  const float* vp = base + 64 + kl * 4;
  for (int j = 0; j < p.N; ++j) {
    const float e = Sq[j] * inv;                        // softmax first, as the reference does, then the weighted sum
    const f32x4 v = *(const f32x4*)(vp + (size_t)j * p.ld);
    a0 = fmaf(e, v[0], a0); a1 = fmaf(e, v[1], a1); a2 = fmaf(e, v[2], a2); a3 = fmaf(e, v[3], a3);
  }
  if (blockIdx.x * 16 + ql < p.N)
    *(f32x4*)((float*)p.out + (size_t)qi * p.ldo + h * 64 + kl * 4) = f32x4{a0, a1, a2, a3};
}

// cv2 INTER_LINEAR tap of an 8-bit image (same arithmetic as yolo.hip's lin_tap / prep.hip's resize_linear_u8_kernel)
__device__ __forceinline__ void lin_tap32(int d, int src, double scale, int* s0, int* s1, int* a0, int* a1) {
  const float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);
  int s = (int)floorf(f);
  float fr = __fsub_rn(f, (float)s);
  if (s < 0) { fr = 0.f; s = 0; }
  if (s >= src - 1) { fr = 0.f; s = src - 1; }
  *a1 = (int)rintf(__fmul_rn(fr, 2048.f));
  *a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, fr), 2048.f));
  *s0 = s;
  *s1 = min(s + 1, src - 1);
}

__global__ __launch_bounds__(256) void y32_letter_kernel(const YLetterP p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.h * p.w) return;
  const int y = idx / p.w, x = idx - y * p.w;
  int bgr[3] = {114, 114, 114};
  const int cy = y - p.top, cx = x - p.left;
  if ((unsigned)cy < (unsigned)p.nh && (unsigned)cx < (unsigned)p.nw) {
    if (p.nh == p.H && p.nw == p.W) {
      const uint8_t* s = p.frame + ((size_t)cy * p.W + cx) * 3;
      bgr[0] = s[0]; bgr[1] = s[1]; bgr[2] = s[2];
    } else {
      int x0, x1, a0, a1, y0, y1, b0, b1;
      lin_tap32(cx, p.W, p.sx, &x0, &x1, &a0, &a1);
      lin_tap32(cy, p.H, p.sy, &y0, &y1, &b0, &b1);
      const uint8_t* r0 = p.frame + (size_t)y0 * p.W * 3;
      const uint8_t* r1 = p.frame + (size_t)y1 * p.W * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
        const int S1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
        bgr[c] = ((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2) & 255;
      }
    }
  }
  float* o = (float*)p.out + (size_t)idx * 8;
  *(f32x4*)o = f32x4{__fdiv_rn((float)bgr[2], 255.f), __fdiv_rn((float)bgr[1], 255.f), __fdiv_rn((float)bgr[0], 255.f), 0.f};
  *(f32x4*)(o + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ops.process_mask step 1 with a float32 proto map (yolo.hip: ymask_low_kernel)
__global__ __launch_bounds__(256) void y32_mask_low_kernel(const YMaskP p) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.mh * p.mw) return;
  const int n = min(*p.det_count, p.max_det);
  const int y = pix / p.mw, x = pix - y * p.mw;
  const float wr = (float)((double)p.mw / (double)p.iw), hr = (float)((double)p.mh / (double)p.ih);
  float pr[32];
  const float* ps = (const float*)p.proto + (size_t)pix * 32;
#pragma unroll
  for (int k = 0; k < 32; ++k) pr[k] = ps[k];
  for (int i = 0; i < n; ++i) {
    const float* bl = p.det_lb + i * 4;
    const float x1 = __fmul_rn(bl[0], wr), y1 = __fmul_rn(bl[1], hr), x2 = __fmul_rn(bl[2], wr), y2 = __fmul_rn(bl[3], hr);
    float m = 0.f;
    if ((float)x >= x1 && (float)x < x2 && (float)y >= y1 && (float)y < y2) {
      const float* coef = p.pred + (size_t)p.det_anchor[i] * p.no + 64 + p.nc;
#pragma unroll
      for (int k = 0; k < 32; ++k) m = fmaf(coef[k], pr[k], m);
    }
    p.low[(size_t)i * p.mh * p.mw + pix] = m;
  }
}

}  // namespace

extern "C" int flope_y32_conv_launch(const YConvP* p, void* stream) {
  if ((p->k != 1 && p->k != 3) || p->Cin % 4 || p->M < 1) return (int)hipErrorInvalidValue;
  const int rows = p->out_mode == 2 ? 4 * p->dc : p->Cout;
  hipLaunchKernelGGL(y32_conv_kernel<8>, dim3((p->M + 255) / 256, (rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int flope_y32_dw_launch(const YDwP* p, void* stream) {
  hipLaunchKernelGGL(y32_dw_kernel, dim3((p->H * p->W * p->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
// r04: the same maxima, separably -- the window of the r-th cascaded pool is (4 r + 5)^2, its maximum the column maximum of row
// maxima: a workgroup of 1024 threads takes 4 channels of the whole map into LDS, forms the row maxima of half-widths 2, 4, 6 there and the column
// maxima from those (13 + 27 LDS reads per output instead of 169 global ones; max is exact in any order: bit-identical to the sweep
// above, which stays for maps beyond LDS).
template <int N>
__global__ __launch_bounds__(1024) void y32_pool_lds_kernel(const YPoolP p) {
  extern __shared__ float PL[];                          // [4][H * W][4]: the slab (4 channels), then its row maxima for half-widths 2, 4, 6
  const int HW = p.H * p.W, c0 = blockIdx.x * 4;
  for (int pix = threadIdx.x; pix < HW; pix += 1024)
    *(f32x4*)(PL + pix * 4) = *(const f32x4*)((const float*)p.in + (size_t)pix * p.ldi + c0);
  __syncthreads();
  for (int i = threadIdx.x; i < HW * 4; i += 1024) {
    const int pix = i >> 2, c = i & 3, y = pix / p.W, x = pix - y * p.W;
    float m[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int dx = -2 * N; dx <= 2 * N; ++dx) {
      const int ix = x + dx;
      const float f = (unsigned)ix < (unsigned)p.W ? PL[(y * p.W + ix) * 4 + c] : -3.0e38f;
      const int a = dx < 0 ? -dx : dx;
      m[2] = fmaxf(m[2], f);
      if (a <= 4) m[1] = fmaxf(m[1], f);
      if (a <= 2) m[0] = fmaxf(m[0], f);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) PL[(size_t)(r + 1) * HW * 4 + i] = m[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW * 4; i += 1024) {
    const int pix = i >> 2, c = i & 3, y = pix / p.W, x = pix - y * p.W;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const int hw = 2 * (r + 1);                         // half-width of pool r's window; its row maxima: array r, or (the last pool) array 2
      const float* rm = PL + (size_t)((r == N - 1 ? 2 : r) + 1) * HW * 4;
      float a = -3.0e38f;
#pragma unroll
      for (int dy = -2 * N; dy <= 2 * N; ++dy) {
        if (dy < -hw || dy > hw) continue;
        const int iy = y + dy;
        a = fmaxf(a, (unsigned)iy < (unsigned)p.H ? rm[(iy * p.W + x) * 4 + c] : -3.0e38f);
      }
      ((float*)p.out)[(size_t)pix * p.ldo + r * p.C + c0 + c] = a;
    }
  }
}

extern "C" int flope_y32_pool_init() {
  hipError_t e = hipFuncSetAttribute((const void*)y32_pool_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)y32_pool_lds_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)y32_pool_lds_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return (int)e;
}
extern "C" int flope_y32_pool_launch(const YPoolP* p, void* stream) {
  if (p->n < 1 || p->n > 3) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)4 * p->H * p->W * 4 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (lds <= 160 * 1024 && p->C % 4 == 0 && p->ldi % 4 == 0 && !((uintptr_t)p->in & 15)) {   // (the LDS form reads float32x4 pixels)
    if (p->n == 1) hipLaunchKernelGGL(y32_pool_lds_kernel<1>, dim3(p->C / 4), dim3(1024), lds, st, *p);
    else if (p->n == 2) hipLaunchKernelGGL(y32_pool_lds_kernel<2>, dim3(p->C / 4), dim3(1024), lds, st, *p);
    else hipLaunchKernelGGL(y32_pool_lds_kernel<3>, dim3(p->C / 4), dim3(1024), lds, st, *p);
  } else
    hipLaunchKernelGGL(y32_pool_kernel, dim3((p->H * p->W * p->C + 255) / 256), dim3(256), 0, st, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_y32_up_launch(const YUpP* p, void* stream) {
  hipLaunchKernelGGL(y32_up_kernel, dim3((4 * p->H * p->W * p->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

// r04: the same attention on the exact-fp32 matrix instruction.  One workgroup = 16 queries of one head, four waves:
//   1. scores  S^T[key][query] = K[key][:] . Q[query][:] (A = 16 key rows, B = the 16 queries, 8 MFMAs per key tile of 16; a wave takes
//      every fourth key tile), x scale, into LDS as S[query][key] (row pitch N + 4 floats: the value pass reads it conflict-free);
//   2. softmax over the keys, 16 lanes per query, exactly as above (max, expf, sum, p = e / l: softmax first, then the weighted sum);
//   3. values  O^T[dim][query] = V^T[dim][key] . P^T[key][query] (A = 16 dims of V for 4 keys, B = P from LDS; a wave takes every
//      fourth group of 4 keys; partial sums added in wave order through LDS).
// 126 -> ~10 us on 920 tokens x 2 heads.  Other summation order than y32_attn_kernel (which stays as the checker: f32mfma = 0).
__global__ __launch_bounds__(256) void y32m_attn_kernel(const YAttnP p) {
  extern __shared__ float S32m[];                        // [16][N + 4] scores / probabilities, then [3][4][64][4] partial outputs
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, c16 = lane & 15;
  const int h = blockIdx.y, q0 = blockIdx.x * 16;
  const int pitch = ((p.N + 15) & ~15) + 4;
  const float* const base = (const float*)p.qkv + (size_t)h * 128;
  float* const red = S32m + 16 * pitch;
  // 1. scores
  f32x4 qf[2];
  {
    const float* qp = base + (size_t)min(q0 + c16, p.N - 1) * p.ld + kq * 4;
    qf[0] = *(const f32x4*)qp; qf[1] = *(const f32x4*)(qp + 16);
  }
  const int ntile = (p.N + 15) >> 4;
  for (int kt = wave; kt < ntile; kt += 4) {
    const float* kp = base + (size_t)min(kt * 16 + c16, p.N - 1) * p.ld + 32 + kq * 4;
    const f32x4 k0 = *(const f32x4*)kp, k1 = *(const f32x4*)(kp + 16);
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[s], qf[0][s], d, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[s], qf[1][s], d, 0, 0, 0);
    // lane (kq, c16): keys kt * 16 + 4 kq + r of query c16
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = kt * 16 + 4 * kq + r < p.N ? d[r] * p.scale : -3.0e38f;
    *(f32x4*)(S32m + c16 * pitch + kt * 16 + 4 * kq) = o;
  }
  __syncthreads();
  // 2. softmax (thread = query tid >> 4, keys tid & 15, + 16, ...)
  {
    const int ql = tid >> 4, kl = tid & 15;
    float* Sq = S32m + ql * pitch;
    float mx = -3.0e38f;
    for (int j = kl; j < p.N; j += 16) mx = fmaxf(mx, Sq[j]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    float l = 0.f;
    for (int j = kl; j < p.N; j += 16) { const float e = expf(Sq[j] - mx); Sq[j] = e; l += e; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 16);
    const float inv = 1.f / l;
    for (int j = kl; j < ((p.N + 15) & ~15); j += 16) Sq[j] = j < p.N ? Sq[j] * inv : 0.f;
  }
  __syncthreads();
  // 3. values: k4 step g covers keys 4 g .. 4 g + 3; lane (kq, c16): A = V[4 g + kq][ct * 16 + c16], B = P[query c16][4 g + kq]
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ng = ((p.N + 15) & ~15) >> 2;
  const float* const vb = base + 64 + c16;
  for (int g0 = wave; g0 < ng; g0 += 16) {               // four groups of this wave per trip: 16 loads in flight
    float av[4][4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = g0 + 4 * u, key = min(4 * g + kq, p.N - 1);
      bv[u] = g < ng ? S32m[c16 * pitch + 4 * g + kq] : 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) av[u][ct] = vb[(size_t)key * p.ld + ct * 16];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][ct], bv[u], acc[ct], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) *(f32x4*)(red + (((wave - 1) * 4 + ct) * 64 + lane) * 4) = acc[ct];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const f32x4 r = *(const f32x4*)(red + ((w * 4 + ct) * 64 + lane) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[ct][q] += r[q];
    }
  // lane (kq, c16): query c16, dims ct * 16 + 4 kq + r
  if (q0 + c16 < p.N) {
    float* o = (float*)p.out + (size_t)(q0 + c16) * p.ldo + h * 64 + 4 * kq;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) *(f32x4*)(o + ct * 16) = acc[ct];
  }
}

extern "C" int flope_y32_attn_init() {
  hipError_t e = hipFuncSetAttribute((const void*)y32_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)y32m_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return (int)e;
}
extern "C" int flope_y32_attn_launch(const YAttnP* p, void* stream) {
  const size_t lds = (size_t)16 * p->N * sizeof(float);
  if (p->N < 1 || lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(y32_attn_kernel, dim3((p->N + 15) / 16, p->heads), dim3(256), lds, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int flope_y32m_attn_launch(const YAttnP* p, void* stream) {
  const size_t lds = ((size_t)16 * (((p->N + 15) & ~15) + 4) + 3 * 4 * 64 * 4) * sizeof(float);
  if (p->N < 1 || lds > 160 * 1024 || p->ld % 4 || p->ldo % 4) return flope_y32_attn_launch(p, stream);
  hipLaunchKernelGGL(y32m_attn_kernel, dim3((p->N + 15) / 16, p->heads), dim3(256), lds, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int flope_y32_letter_launch(const YLetterP* p, void* stream) {
  hipLaunchKernelGGL(y32_letter_kernel, dim3((p->h * p->w + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int flope_y32_mask_low_launch(const YMaskP* p, void* stream) {
  hipLaunchKernelGGL(y32_mask_low_kernel, dim3((p->mh * p->mw + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// r04: the float32 detector on the matrix cores.  v_mfma_f32_16x16x4_f32 takes float32 operands and accumulates in float32 --
// each product-sum is an fmaf chain, bit for bit, at 64 FLOP / clock / SIMD (157 TFLOP/s on the chip: the rate of the vector
// FMA the kernels above use, but issued by one instruction per 1024 multiply-adds instead of sixteen, and with the weights as a
// shared operand instead of scalar loads per thread).  VERDICT r3 item 2a: the mode whose int16 boxes / uint8 mask EQUAL the
// reference's float32 arithmetic (fast_pose_predictor.py:49-56) should not cost 12 x the 16-bit detector.
//   D[channel][pixel] += W[channel][k] X[k][pixel], weights = A operand (16 rows x 4 k per instruction), 16 output pixels = B.
//   A lane loads 16 bytes = 4 consecutive k of its row / its pixel and feeds element s to MFMA s of a 16-deep step on BOTH
//   operands (the k order inside a step is permuted identically for A and B: fc1_kernel's trick, pool_head.hip).
//   Wave tile: MP x 16 pixels x 16 NT channels, operands straight from L2 (every map of the detector fits the Infinity Cache, the
//   weights L2), next step's loads in flight under the current step's MFMAs.  SPLITK (maps of a few thousand pixels with deep K):
//   the four waves of a workgroup share one pixel tile and take every fourth k16 step; partial sums are added in wave order
//   through LDS -- deterministic.  Same summation order per output in both forms only up to that split: float32 rounding noise of
//   ~1e-7 relative against the plain kernel above (which stays as the checker: option f32mfma = 0), not bit equality.
typedef float f32x4m __attribute__((ext_vector_type(4)));

// (m0: first pixel of this wave's tile, blk: its channel block -- y32m_conv_body derives them from the workgroup index; the chain kernel
// hands every wave its own channel block of one shared pixel tile)
template <int NT, int MP, bool SPLITK>
__device__ __forceinline__ void y32m_conv_body_at(const YConvP& p, const int m0, const int blk, float* red) {   // red (SPLITK): 3 x 64 x NT x MP x 4 floats of LDS
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kq = lane >> 4, c16 = lane & 15;
  constexpr int CB = 16 * NT;
  const int rows = p.out_mode == 2 ? 4 * p.dc : p.Cout;
  const int pad = p.k == 3 ? 1 : 0, kk2 = p.k * p.k, cin = p.Cin;
  const float* const in = (const float*)p.in;
  const f32x4m* const wb = (const f32x4m*)p.w32m + (size_t)blk * p.k16steps * NT * 64 + lane;
  // this lane's pixel of each of the MP pixel tiles
  int oy[MP], ox[MP];
  bool pv[MP];
#pragma unroll
  for (int t = 0; t < MP; ++t) {
    const int m = m0 + t * 16 + c16;
    pv[t] = m < p.M;
    const int mc = pv[t] ? m : p.M - 1;
    oy[t] = fastdiv(mc, p.wo_mg, p.wo_sh); ox[t] = mc - oy[t] * p.Wo;
  }
  f32x4m acc[MP][NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const f32x4m b = (!SPLITK || wave == 0) ? *(const f32x4m*)(p.bias32m + blk * CB + ct * 16 + 4 * kq) : f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MP; ++t) acc[t][ct] = b;
  }
  // operand loads of k16 step ks: this lane's 4 k values are ks * 16 + kq * 4 .. + 3 = (tap, ci .. ci + 3)  (Cin % 4 == 0)
  auto load_x = [&](int ks, f32x4m (&x)[MP]) {
    const int kk = ks * 16 + kq * 4;
    int tap = fastdiv(kk >> 3, p.cg_mg, p.cg_sh);                               // kk / Cin = (kk / 8) / (Cin / 8)
    const int ci = kk - tap * cin;
    const bool tv = tap < kk2;                                                  // (the zero-padded tail of K)
    tap = tv ? tap : 0;
    const int ky = p.k == 3 ? (tap * 11) >> 5 : 0, kx = tap - ky * 3;           // tap / 3 for 0..8
#pragma unroll
    for (int t = 0; t < MP; ++t) {
      const int iy = oy[t] * p.stride - pad + ky, ix = ox[t] * p.stride - pad + kx;
      const bool ok = tv && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      const int iyc = min(max(iy, 0), p.Hi - 1), ixc = min(max(ix, 0), p.Wi - 1);
      const f32x4m v = *(const f32x4m*)(in + (size_t)(iyc * p.Wi + ixc) * p.ldi + ci);
      x[t] = ok ? v : f32x4m{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto load_w = [&](int ks, f32x4m (&w)[NT]) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) w[ct] = wb[(size_t)(ks * NT + ct) * 64];
  };
  // K loop in chunks of U k16 steps: the loads of chunk c + 1 are issued before the MFMAs of chunk c (operands come straight from
  // L2: ~500-800 cycles per round trip, a step's MFMAs are 128-512 -- one step of look-ahead left the loop latency-bound)
  const int kstep0 = SPLITK ? wave : 0, kinc = SPLITK ? 4 : 1;
  constexpr int U = 2;                                    // (4: no gain for the deep launches, fewer resident waves for the wide ones: 1.26 vs 1.04 ms per frame)
  f32x4m xa[U][MP], wa[U][NT], xb[U][MP], wbf[U][NT];
  const int nsteps = p.k16steps;
  auto load_chunk = [&](int ks0, f32x4m (&x)[U][MP], f32x4m (&w)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ks_ = min(ks0 + u * kinc, nsteps - 1);      // (steps past the end re-load the last one; their MFMAs are skipped)
      load_x(ks_, x[u]); load_w(ks_, w[u]);
    }
  };
  auto mfma_chunk = [&](int ks0, const f32x4m (&x)[U][MP], const f32x4m (&w)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ks0 + u * kinc >= nsteps) break;                  // uniform
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < MP; ++t)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][ct][s], x[u][t][s], acc[t][ct], 0, 0, 0);
    }
  };
  int ks = kstep0;
  if (ks < nsteps) load_chunk(ks, xa, wa);
  for (; ks < nsteps; ks += 2 * U * kinc) {
    const bool n1 = ks + U * kinc < nsteps;
    if (n1) load_chunk(ks + U * kinc, xb, wbf);
    mfma_chunk(ks, xa, wa);
    if (!n1) break;
    if (ks + 2 * U * kinc < nsteps) load_chunk(ks + 2 * U * kinc, xa, wa);
    mfma_chunk(ks + U * kinc, xb, wbf);
  }
  // epilogue of one (pixel tile t, channel tile ct) unit: lane (kq, c16) holds pixel c16 x rows blk * CB + kq * 4 NT + ct * 4 + q
  const int r0 = blk * CB + kq * 4 * NT;
  auto epilogue_unit = [&](int t, int ct, const f32x4m a) {
    const int m = m0 + t * 16 + c16;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = p.act ? silu32(a[q]) : a[q];
    if (p.out_mode == 2) {                                 // ConvTranspose2d 2x2 s2: rows r0 .. lie in one (dy, dx) quadrant (dc % 16 == 0)
      if (r0 >= rows) return;
      const int quad = r0 / p.dc, co = r0 - quad * p.dc;
      const size_t opix = (size_t)(2 * oy[t] + (quad >> 1)) * (2 * p.Wo) + 2 * ox[t] + (quad & 1);
      float* o = (float*)p.out + opix * p.ldo + co;
      *(f32x4m*)(o + ct * 4) = f32x4m{v[0], v[1], v[2], v[3]};
    } else {
      if (p.res) {
        const float* rp = (const float*)p.res + (size_t)m * p.ldr + r0 + ct * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (r0 + ct * 4 + q < rows) v[q] += rp[q];
      }
      float* o = (float*)p.out + (size_t)m * p.ldo + r0 + ct * 4;
      if (p.out_mode == 0 && r0 + 4 * NT <= rows) {        // map views: 16-byte aligned runs
        *(f32x4m*)o = f32x4m{v[0], v[1], v[2], v[3]};
      } else {                                             // prediction rows (odd pitch) and ragged channel counts
#pragma unroll
        for (int q = 0; q < 4; ++q) if (r0 + ct * 4 + q < rows) o[q] = v[q];
      }
    }
  };
  if constexpr (SPLITK) {
    // r04b: unit u = t * NT + ct belongs to wave u & 3; a wave parks the units it does not own ([source index among the three
    // others][unit][lane]), the owner adds the four partial sums in WAVE order -- ((P0 + P1) + P2) + P3, the order of the one-wave
    // combine this replaces: bit-identical -- and runs the unit's epilogue (activation in exact float32: ~40 instructions per value,
    // which one wave of a one-wave-per-SIMD kernel used to do for the whole workgroup)
#pragma unroll
    for (int t = 0; t < MP; ++t)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const int own = (t * NT + ct) & 3;
        if (own != wave) {
          const int si = wave < own ? wave : wave - 1;
          *(f32x4m*)(red + (((si * MP + t) * NT + ct) * 64 + lane) * 4) = acc[t][ct];
        }
      }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MP; ++t)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        if (((t * NT + ct) & 3) != wave || !pv[t]) continue;
        f32x4m tot;
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
          f32x4m part;
          if (sw == wave) part = acc[t][ct];
          else part = *(const f32x4m*)(red + ((((sw < wave ? sw : sw - 1) * MP + t) * NT + ct) * 64 + lane) * 4);
          if (sw == 0) tot = part;
          else { tot[0] += part[0]; tot[1] += part[1]; tot[2] += part[2]; tot[3] += part[3]; }
        }
        epilogue_unit(t, ct, tot);
      }
  } else {
#pragma unroll
    for (int t = 0; t < MP; ++t) {
      if (!pv[t]) continue;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) epilogue_unit(t, ct, acc[t][ct]);
    }
  }
}

template <int NT, int MP, bool SPLITK>
__device__ __forceinline__ void y32m_conv_body(const YConvP& p, int bx, int by, float* red) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  y32m_conv_body_at<NT, MP, SPLITK>(p, (SPLITK ? bx : bx * 4 + wave) * (16 * MP), by, red);
}

// A chain of 1x1 convs on a small map in one launch (yolo.h YChainP; the float32 counterpart of yolo.hip's ychain_kernel): workgroup =
// one 16-pixel tile, its four waves take the channel blocks of each conv side by side (the full K loop each), a workgroup barrier
// between the convs.
__global__ __launch_bounds__(256) void y32m_chain_kernel(const YChainP* __restrict__ Pd) {
  const YChainP& P = *Pd;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * 16;
  for (int i = 0; i < P.n; ++i) {
    const YConvP& p = P.op[i];
    const int nt = p.nt32m;
    const int nby = (p.Cout + 16 * nt - 1) / (16 * nt);
    for (int b = wave; b < nby; b += 4) {
      if (nt == 1) y32m_conv_body_at<1, 1, false>(p, m0, b, nullptr);
      else if (nt == 2) y32m_conv_body_at<2, 1, false>(p, m0, b, nullptr);
      else y32m_conv_body_at<4, 1, false>(p, m0, b, nullptr);
    }
    __syncthreads();
  }
}

template <int NT, int MP, bool SPLITK>
__global__ __launch_bounds__(256) void y32m_conv_kernel(const YConvP p) {
  __shared__ __attribute__((aligned(16))) float red[SPLITK ? 3 * 64 * NT * MP * 4 : 4];
  y32m_conv_body<NT, MP, SPLITK>(p, blockIdx.x, blockIdx.y, red);
}

// the form a launch takes: split-K over the workgroup's waves where pixels are few and K is deep; else one 16-pixel tile per wave, two
// where that still leaves two workgroups per CU.  -> grid (nbx, nby)
static bool y32m_geometry(const YConvP* p, bool* splitk, int* mp, int* nbx, int* nby, int mp2_from = 32768) {
  if ((p->k != 1 && p->k != 3) || p->Cin % 8 || p->M < 1 || !p->w32m || !p->bias32m || (p->nt32m != 1 && p->nt32m != 2 && p->nt32m != 4)) return false;
  // what the epilogue assumes: float32x4 stores -- 16-byte aligned rows -- and, for the transposed-conv scatter, whole
  // units of 4 NT channels per (dy, dx) block
  // (prediction rows, out_mode 1, have an odd pitch and are stored element by element)
  if ((p->out_mode != 1 && (((uintptr_t)p->out & 15) || (p->ldo & 3))) || (p->out_mode == 2 && p->dc % (4 * p->nt32m))) return false;
  const int rows = p->out_mode == 2 ? 4 * p->dc : p->Cout;
  *splitk = p->M <= 4096 && p->k16steps >= 8;
  *mp = (!*splitk && p->M >= mp2_from) ? 2 : 1;          // (two pixel tiles per wave halve the weight loads per MFMA; inside a shared grid the other ops keep the CUs filled)
  const int tiles = (p->M + 16 * *mp - 1) / (16 * *mp);
  *nbx = *splitk ? tiles : (tiles + 3) / 4;
  *nby = (rows + 16 * p->nt32m - 1) / (16 * p->nt32m);
  return true;
}

// does the MFMA kernel take this op?  (no: the engine launches the plain fused-multiply-add kernel instead, and keeps the op out of
// the shared grids -- yolo_engine.hip launch_op / build_schedules)
extern "C" int flope_y32m_conv_ok(const YConvP* p) {
  bool splitk; int mp, nbx, nby;
  return y32m_geometry(p, &splitk, &mp, &nbx, &nby) ? 1 : 0;
}

// float32 detector convolution on the matrix cores; p->w32m / bias32m / k16steps / nt32m from the builder (yolo_engine.hip pack)
extern "C" int flope_y32m_conv_launch(const YConvP* p, void* stream) {
  bool splitk; int mp, nbx, nby;
  if (!y32m_geometry(p, &splitk, &mp, &nbx, &nby)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
#define GO(NT_, MP_, SK_) hipLaunchKernelGGL((y32m_conv_kernel<NT_, MP_, SK_>), dim3(nbx, nby), dim3(256), 0, st, *p)
#define GN(NT_) do { if (splitk) GO(NT_, 1, true); else if (mp == 2) GO(NT_, 2, false); else GO(NT_, 1, false); } while (0)
  if (p->nt32m == 1) GN(1); else if (p->nt32m == 2) GN(2); else GN(4);
#undef GN
#undef GO
  return (int)hipGetLastError();
}

// ---- several INDEPENDENT float32 ops of one dependency level in one grid (the counterpart of yolo.hip's ymulti_kernel: the Segment
// head's branches, the Proto block beside them, the two 1x1 convs of a C3k ... are a few dozen workgroups each and cost a whole
// dependent launch when queued alone).  Same table (YMultiP in device memory, uniform reads), op codes of their own:
//   conv: (nt index 0 / 1 / 2) * 4 + (MP == 2 ? 2 : 0) + (split-K ? 1 : 0);   12: depthwise
__device__ __forceinline__ void y32_dw_body(const YDwP& p, int lb) {
  const int idx = lb * 256 + threadIdx.x;
  if (idx >= p.H * p.W * p.C) return;
  const int pix = idx / p.C, c = idx - pix * p.C;
  const int y = pix / p.W, x = pix - y * p.W;
  const int ci = p.blk ? (c / p.blk) * p.blk_stride + p.blk_off + c % p.blk : c;
  float a = p.bias[c];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
    a = fmaf(((const float*)p.in)[(size_t)(iy * p.W + ix) * p.ldi + ci], p.w[t * p.C + c], a);
  }
  if (p.act) a = silu32(a);
  if (p.add) a += ((const float*)p.add)[(size_t)pix * p.lda + c];
  ((float*)p.out)[(size_t)pix * p.ldo + c] = a;
}

__global__ __launch_bounds__(256) void y32m_multi_kernel(const YMultiP* __restrict__ Pd) {
  __shared__ __attribute__((aligned(16))) float red[3 * 64 * 4 * 4];
  const YMultiP& P = *Pd;
  const int b = blockIdx.x;
  int s = 0;
  for (int i = 1; i < P.n; ++i) s = b >= P.op[i].start ? i : s;
  const YMultiOp& o = P.op[s];
  const int lb = b - o.start;
  if (lb >= o.nblocks) return;
  if (o.code == 12) { y32_dw_body(o.u.d, lb); return; }
  const int by = lb / o.nbx, bx = lb - by * o.nbx;
  switch (o.code) {
    case 0: y32m_conv_body<1, 1, false>(o.u.c, bx, by, red); break;
    case 1: y32m_conv_body<1, 1, true>(o.u.c, bx, by, red); break;
    case 2: y32m_conv_body<1, 2, false>(o.u.c, bx, by, red); break;
    case 4: y32m_conv_body<2, 1, false>(o.u.c, bx, by, red); break;
    case 5: y32m_conv_body<2, 1, true>(o.u.c, bx, by, red); break;
    case 6: y32m_conv_body<2, 2, false>(o.u.c, bx, by, red); break;
    case 8: y32m_conv_body<4, 1, false>(o.u.c, bx, by, red); break;
    case 9: y32m_conv_body<4, 1, true>(o.u.c, bx, by, red); break;
    default: y32m_conv_body<4, 2, false>(o.u.c, bx, by, red); break;
  }
}

extern "C" int flope_y32m_multi_add_conv(YMultiP* m, const YConvP* p) {
  bool splitk; int mp, nbx, nby;
  if (m->n >= kYMultiMax || !y32m_geometry(p, &splitk, &mp, &nbx, &nby)) return (int)hipErrorInvalidValue;
  YMultiOp& o = m->op[m->n++];
  o.code = (p->nt32m == 1 ? 0 : p->nt32m == 2 ? 4 : 8) + (mp == 2 ? 2 : 0) + (splitk ? 1 : 0);
  o.nbx = nbx; o.start = m->total; o.nblocks = nbx * nby; o.u.c = *p;
  m->total += (o.nblocks + 7) / 8 * 8;
  return 0;
}
extern "C" int flope_y32m_chain_ok(const YConvP* p) {
  bool splitk; int mp, nbx, nby;
  return p->k == 1 && p->stride == 1 && p->out_mode == 0 && p->M >= 1 && p->M <= 4096 && p->Hi == p->Ho && p->Wi == p->Wo &&
         y32m_geometry(p, &splitk, &mp, &nbx, &nby);
}
extern "C" int flope_y32m_chain_launch(const YChainP* c, const YChainP* c_dev, void* stream) {
  if (!c_dev || c->n < 2 || c->n > kYChainMax || c->tiles < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(y32m_chain_kernel, dim3(c->tiles), dim3(256), 0, (hipStream_t)stream, c_dev);
  return (int)hipGetLastError();
}

extern "C" int flope_y32m_multi_add_dw(YMultiP* m, const YDwP* p) {
  if (m->n >= kYMultiMax) return (int)hipErrorInvalidValue;
  YMultiOp& o = m->op[m->n++];
  o.code = 12; o.nbx = 1; o.start = m->total; o.nblocks = (p->H * p->W * p->C + 255) / 256; o.u.d = *p;
  m->total += (o.nblocks + 7) / 8 * 8;
  return 0;
}
extern "C" int flope_y32m_multi_launch(const YMultiP* m, const YMultiP* m_dev, void* stream) {
  if (!m_dev || m->n < 1 || m->n > kYMultiMax || m->total < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(y32m_multi_kernel, dim3(m->total), dim3(256), 0, (hipStream_t)stream, m_dev);
  return (int)hipGetLastError();
}
