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
extern "C" int flope_y32_pool_launch(const YPoolP* p, void* stream) {
  if (p->n < 1 || p->n > 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(y32_pool_kernel, dim3((p->H * p->W * p->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int flope_y32_up_launch(const YUpP* p, void* stream) {
  hipLaunchKernelGGL(y32_up_kernel, dim3((4 * p->H * p->W * p->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
extern "C" int flope_y32_attn_init() {
  return (int)hipFuncSetAttribute((const void*)y32_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
extern "C" int flope_y32_attn_launch(const YAttnP* p, void* stream) {
  const size_t lds = (size_t)16 * p->N * sizeof(float);
  if (p->N < 1 || lds > 160 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(y32_attn_kernel, dim3((p->N + 15) / 16, p->heads), dim3(256), lds, (hipStream_t)stream, *p);
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
