// Data-format kernels either side of the trunk (all HBM-bandwidth work):
//   * crop batch -> stem input (4-channel, 3-pixel-bordered NHWC in the trunk dtype)
//   * internal activation -> float32 NCHW (parity tests, flope_read_stage)
//   * crop + Lanczos-4 resize + mask multiply (reference fast_pose_predictor.py:108-123,
//     pose_predictor.py:138-153, scripts/test_posenet.py:124-140,
//     scripts/generate_metrics_utils.py:17-35)
//   * depth statistics + back-projection (image_manipulation.py:21-96, mvg.py:387-408)
#include "common.h"

// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void store4(void* dst, float a, float b, float c);
template <> __device__ __forceinline__ void store4<bf16_t>(void* dst, float a, float b, float c) {
  *(u32x2*)dst = u32x2{pack2<bf16_t>(a, b), pack2<bf16_t>(c, 0.f)};
}
template <> __device__ __forceinline__ void store4<f16_t>(void* dst, float a, float b, float c) {
  *(u32x2*)dst = u32x2{pack2<f16_t>(a, b), pack2<f16_t>(c, 0.f)};
}
template <> __device__ __forceinline__ void store4<float>(void* dst, float a, float b, float c) {
  *(f32x4*)dst = f32x4{a, b, c, 0.f};
}

// in_format: 0 f32 NCHW, 1 bf16 NHWC, 2 f16 NHWC, 3 u8 NHWC (include/flope_amd.h)
template <typename T>
__global__ __launch_bounds__(256) void prep_input_kernel(const void* x, int in_format, int B, int H, int W,
                                                         void* out, int Hip, int Wip) {
  const size_t total = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xx = (int)(i % W);
    const size_t r = i / W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    float c0, c1, c2;
    if (in_format == 0) {
      const float* p = (const float*)x + ((size_t)b * 3 * H + yy) * W + xx;
      c0 = p[0]; c1 = p[(size_t)H * W]; c2 = p[(size_t)2 * H * W];
    } else if (in_format == 1) {
      const unsigned short* p = (const unsigned short*)x + i * 3;
      c0 = to_f32(__builtin_bit_cast(bf16_t, p[0])); c1 = to_f32(__builtin_bit_cast(bf16_t, p[1]));
      c2 = to_f32(__builtin_bit_cast(bf16_t, p[2]));
    } else if (in_format == 2) {
      const unsigned short* p = (const unsigned short*)x + i * 3;
      c0 = to_f32(__builtin_bit_cast(f16_t, p[0])); c1 = to_f32(__builtin_bit_cast(f16_t, p[1]));
      c2 = to_f32(__builtin_bit_cast(f16_t, p[2]));
    } else {
      const unsigned char* p = (const unsigned char*)x + i * 3;
      c0 = (float)p[0] / 255.0f; c1 = (float)p[1] / 255.0f; c2 = (float)p[2] / 255.0f;
    }
    char* dst = (char*)out + (((size_t)b * Hip + yy + 3) * Wip + xx + 3) * (4 * sizeof(T));
    store4<T>(dst, c0, c1, c2);
  }
}

extern "C" int flope_prep_input_launch(const void* x, int in_format, int B, int H, int W, void* out, int Hip,
                                       int Wip, int dtype, void* stream) {
  const size_t total = (size_t)B * H * W;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(prep_input_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, in_format, B, H, W, out, Hip, Wip);
  else if (dtype == 1) hipLaunchKernelGGL(prep_input_kernel<f16_t>, dim3(grid), dim3(256), 0, st, x, in_format, B, H, W, out, Hip, Wip);
  else hipLaunchKernelGGL(prep_input_kernel<float>, dim3(grid), dim3(256), 0, st, x, in_format, B, H, W, out, Hip, Wip);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// padded NHWC (border 1) -> float32 NCHW
template <typename T>
__global__ __launch_bounds__(256) void read_stage_kernel(const void* in, float* out, int B, int C, int h, int w) {
  const size_t total = (size_t)B * C * h * w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % w);
    size_t r = i / w;
    const int y = (int)(r % h); r /= h;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const size_t src = (((size_t)b * (h + 2) + y + 1) * (w + 2) + x + 1) * C + c;
    out[i] = to_f32(((const T*)in)[src]);
  }
}

extern "C" int flope_read_stage_launch(const void* in, float* out, int B, int C, int h, int w, int dtype,
                                       void* stream) {
  const size_t total = (size_t)B * C * h * w;
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(read_stage_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, in, out, B, C, h, w);
  else if (dtype == 1) hipLaunchKernelGGL(read_stage_kernel<f16_t>, dim3(grid), dim3(256), 0, st, in, out, B, C, h, w);
  else hipLaunchKernelGGL(read_stage_kernel<float>, dim3(grid), dim3(256), 0, st, in, out, B, C, h, w);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Crop + Lanczos-4 resize + mask multiply.
//
// cv2.resize(src, (S,S), interpolation=INTER_LANCZOS4) on uint8 (OpenCV 4.10, not
// vendored by the reference; restated from the published algorithm): for output index
// d the source coordinate is f = (d + 0.5) * (n_src / S) - 0.5, s = floor(f), t = f - s;
// the eight taps s-3 .. s+4 (clamped to the crop: edge replication) get weights
// w_i = sinc-Lanczos(a=4) evaluated through the 45-degree rotation recurrence, normalised
// to sum 1 in float, then quantised to int16 fixed point (x 2048, round-half-even,
// saturated).  Horizontal then vertical passes accumulate in int32 without intermediate
// rounding; the result is (acc + 2^21) >> 22 saturated to uint8.  Because no rounding
// happens between the passes a direct 8x8 sum is bit-identical to the two-pass form.  Then (fast_pose_predictor.py:118,121): out = img * (mask / 255.0) / 255.0.
// Every floating-point step below is the one OpenCV's C++ performs (resize.cpp, interpolateLanczos4 + the 8-bit
// coefficient quantisation), spelled with explicit round-to-nearest intrinsics so that the device compiler can neither
// fuse a multiply-add nor re-associate: (x + 3) and (x + 3 - i) are FLOAT operations there, the angle products are
// double, left to right.
__device__ __forceinline__ void lanczos4_coeffs(float x, short* c16) {
  const double s45 = 0.70710678118654752440084436210485;
  const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  const double PI = 3.1415926535897932384626433832795;
  float coeffs[8];
  float sum = 0.f;
  const float xp3 = __fadd_rn(x, 3.f);
  const double y0 = __dmul_rn(__dmul_rn(-(double)xp3, PI), 0.25), s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; ++i) {
    const float y0_ = __fsub_rn(xp3, (float)i);
    if (fabsf(y0_) >= 1e-6f) {
      const double y = __dmul_rn(__dmul_rn(-(double)y0_, PI), 0.25);
      coeffs[i] = (float)__ddiv_rn(__dadd_rn(__dmul_rn(cs[i][0], s0), __dmul_rn(cs[i][1], c0)), __dmul_rn(y, y));
    } else {
      coeffs[i] = 1e30f;
    }
    sum = __fadd_rn(sum, coeffs[i]);
  }
  sum = __fdiv_rn(1.f, sum);
  for (int i = 0; i < 8; ++i) {
    const float v = __fmul_rn(__fmul_rn(coeffs[i], sum), 2048.f);
    int q = (int)rintf(v);                       // cvRound: round half to even
    q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
    c16[i] = (short)q;
  }
}

// destination index d of an axis resized n_src -> n_dst: first tap position and the eight int16 weights.
// cv::resize: inv_scale = (double)n_dst / n_src, scale = 1. / inv_scale; fx = (float)((d + 0.5) * scale - 0.5);
// sx = cvFloor(fx); fx -= sx
__device__ __forceinline__ int lanczos4_axis(int d, int n_src, int n_dst, short* c16) {
  const double scale = __ddiv_rn(1.0, __ddiv_rn((double)n_dst, (double)n_src));
  float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);
  const int s0 = (int)floorf(f);
  f = __fsub_rn(f, (float)s0);
  lanczos4_coeffs(f, c16);
  return s0;
}

// test hook: the coefficient tables of one axis exactly as the crop kernel evaluates them
__global__ void lanczos4_table_kernel(int n_src, int n_dst, int* s0_out, short* coef_out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_dst) return;
  short c[8];
  s0_out[d] = lanczos4_axis(d, n_src, n_dst, c);
  for (int k = 0; k < 8; ++k) coef_out[d * 8 + k] = c[k];
}

extern "C" int flope_lanczos4_table(int n_src, int n_dst, int32_t* s0_dev, int16_t* coef_dev, void* stream) {
  if (n_src < 1 || n_dst < 1 || !s0_dev || !coef_dev) return -1;
  hipLaunchKernelGGL(lanczos4_table_kernel, dim3((n_dst + 63) / 64), dim3(64), 0, (hipStream_t)stream, n_src, n_dst, s0_dev,
                     coef_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// out = img * (mask / 255.0) / 255.0 in double like the reference's numpy expression (fast_pose_predictor.py:118,121),
// rounded once to float32 by the torch conversion (:122)
__device__ __forceinline__ float masked_unit(int img, int m) {
  return (float)__ddiv_rn(__dmul_rn((double)img, __ddiv_rn((double)m, 255.0)), 255.0);
}

// Tiled, separable form (bit-identical to the direct 8x8 sum: sum_ky (sum_kx src*ax) * ay with exact integer partial
// sums, no rounding between the passes).  One workgroup = one 16 x 16 output tile of one crop:
//   1. 32 threads evaluate the 16 + 16 Lanczos coefficient sets the tile needs (the direct kernel evaluated two sets --
//      double-precision sin/cos -- per output pixel: that, not the taps, was its cost: 225 us for 16 crops of 512 x 512);
//   2. horizontal pass: for every source row the tile touches, the 16 output columns x (3 channels + mask) -> LDS int32;
//   3. vertical pass from LDS, rounding, mask multiply, store.
// Tiles that would need more than kCropRows source rows (down-scaling by more than ~3.5x) use the direct kernel.
constexpr int kCropRows = 64;
__global__ __launch_bounds__(256) void crop_resize_mask_tiled_kernel(const unsigned char* __restrict__ frame,
                                                                     const unsigned char* __restrict__ mask, int FH,
                                                                     int FW, const int* __restrict__ boxes, int S,
                                                                     int tiles_x, int out_format, void* out) {
  __shared__ short cx[16][8], cy[16][8];
  __shared__ int sxs[16], sys_[16];
  __shared__ int hor[kCropRows][16][4];
  const int b = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int xmin = boxes[b * 4], ymin = boxes[b * 4 + 1], xmax = boxes[b * 4 + 2], ymax = boxes[b * 4 + 3];
  const int cw = xmax - xmin, ch = ymax - ymin;
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const int dx = tx * 16 + lx, dy = ty * 16 + ly;
  const bool live = cw > 0 && ch > 0;
  if (live && tid < 32) {
    const bool isy = tid >= 16;
    const int d = isy ? ty * 16 + (tid - 16) : tx * 16 + tid;
    short c[8];
    const int s0 = lanczos4_axis(d, isy ? ch : cw, S, c);
    for (int k = 0; k < 8; ++k) (isy ? cy[tid - 16][k] : cx[tid][k]) = c[k];
    (isy ? sys_[tid - 16] : sxs[tid]) = s0;
  }
  __syncthreads();
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  if (live) {
    const int r0 = sys_[0] - 3;                                     // first (unclamped) source row of the tile
    const int R = sys_[15] + 4 - r0 + 1;                            // rows touched (uniform over the workgroup)
    if (R <= kCropRows) {
      for (int it = tid; it < R * 16; it += 256) {
        const int rr = it >> 4, c = it & 15;
        int yy = r0 + rr;
        yy = yy < 0 ? 0 : (yy >= ch ? ch - 1 : yy);
        const unsigned char* frow = frame + ((size_t)(ymin + yy) * FW + xmin) * 3;
        const unsigned char* mrow = mask + (size_t)(ymin + yy) * FW + xmin;
        const int sx = sxs[c];
        int h0 = 0, h1 = 0, h2 = 0, hm = 0;
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) {
          int xx = sx - 3 + kx;
          xx = xx < 0 ? 0 : (xx >= cw ? cw - 1 : xx);
          const int w = cx[c][kx];
          h0 += frow[xx * 3] * w; h1 += frow[xx * 3 + 1] * w; h2 += frow[xx * 3 + 2] * w;
          hm += mrow[xx] * w;
        }
        hor[rr][c][0] = h0; hor[rr][c][1] = h1; hor[rr][c][2] = h2; hor[rr][c][3] = hm;
      }
      __syncthreads();
      int a0 = 0, a1 = 0, a2 = 0, am = 0;
      const int rb = sys_[ly] - 3 - r0;
#pragma unroll
      for (int ky = 0; ky < 8; ++ky) {
        const int wv = cy[ly][ky];
        const int* hp = hor[rb + ky][lx];
        a0 += hp[0] * wv; a1 += hp[1] * wv; a2 += hp[2] * wv; am += hp[3] * wv;
      }
      auto fin = [](int v) { v = (v + (1 << 21)) >> 22; return v < 0 ? 0 : (v > 255 ? 255 : v); };
      const int mk = fin(am);
      o0 = masked_unit(fin(a0), mk); o1 = masked_unit(fin(a1), mk); o2 = masked_unit(fin(a2), mk);
    } else {                                                        // very strong down-scaling: direct 8 x 8 sum
      int a0 = 0, a1 = 0, a2 = 0, am = 0;
      if (dx < S && dy < S) {
        for (int ky = 0; ky < 8; ++ky) {
          int yy = sys_[ly] - 3 + ky;
          yy = yy < 0 ? 0 : (yy >= ch ? ch - 1 : yy);
          const unsigned char* frow = frame + ((size_t)(ymin + yy) * FW + xmin) * 3;
          const unsigned char* mrow = mask + (size_t)(ymin + yy) * FW + xmin;
          int h0 = 0, h1 = 0, h2 = 0, hm = 0;
          for (int kx = 0; kx < 8; ++kx) {
            int xx = sxs[lx] - 3 + kx;
            xx = xx < 0 ? 0 : (xx >= cw ? cw - 1 : xx);
            const int w = cx[lx][kx];
            h0 += frow[xx * 3] * w; h1 += frow[xx * 3 + 1] * w; h2 += frow[xx * 3 + 2] * w;
            hm += mrow[xx] * w;
          }
          const int wv = cy[ly][ky];
          a0 += h0 * wv; a1 += h1 * wv; a2 += h2 * wv; am += hm * wv;
        }
      }
      auto fin = [](int v) { v = (v + (1 << 21)) >> 22; return v < 0 ? 0 : (v > 255 ? 255 : v); };
      const int mk = fin(am);
      o0 = masked_unit(fin(a0), mk); o1 = masked_unit(fin(a1), mk); o2 = masked_unit(fin(a2), mk);
    }
  }
  if (dx >= S || dy >= S) return;
  const size_t i = ((size_t)b * S + dy) * S + dx;
  if (out_format == 0) {
    float* o = (float*)out + ((size_t)b * 3 * S + dy) * S + dx;
    o[0] = o0; o[(size_t)S * S] = o1; o[(size_t)2 * S * S] = o2;
  } else if (out_format == 1) {
    bf16_t* o = (bf16_t*)out + i * 3;
    o[0] = from_f32<bf16_t>(o0); o[1] = from_f32<bf16_t>(o1); o[2] = from_f32<bf16_t>(o2);
  } else {
    f16_t* o = (f16_t*)out + i * 3;
    o[0] = from_f32<f16_t>(o0); o[1] = from_f32<f16_t>(o1); o[2] = from_f32<f16_t>(o2);
  }
}

extern "C" int flope_crop_resize_mask(const uint8_t* frame_dev, const uint8_t* mask_dev, int frame_h, int frame_w,
                                      const int32_t* boxes_dev, int n, int size, int out_format, void* out_dev,
                                      void* stream) {
  if (n < 0 || size <= 0 || frame_h <= 0 || frame_w <= 0 || out_format < 0 || out_format > 2) return -1;
  if (n == 0) return 0;
  if (!frame_dev || !mask_dev || !boxes_dev || !out_dev) return -1;
  const int tiles = (size + 15) / 16;
  hipLaunchKernelGGL(crop_resize_mask_tiled_kernel, dim3(tiles * tiles, n), dim3(256), 0, (hipStream_t)stream, frame_dev,
                     mask_dev, frame_h, frame_w, boxes_dev, size, tiles, out_format, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------
// Depth: valid = (near < d < far) & (mask > 128); eroded by cv2's 10x10 MORPH_ELLIPSE
// (anchor (5,5), out-of-frame taps ignored = erode's default +inf constant border).
// Element rows (OpenCV getStructuringElement: dx = round(c * sqrt(1 - dy^2/r^2)), r=c=5):
//   i: 0 -> col 5 | 1,9 -> 2..8 | 2,8 -> 1..9 | 3..7 -> 0..9
__constant__ signed char kEllipseLo[10] = {5, 2, 1, 0, 0, 0, 0, 0, 1, 2};
__constant__ signed char kEllipseHi[10] = {5, 8, 9, 9, 9, 9, 9, 9, 9, 8};

// Masked mean of depth over eroded-valid pixels.  A box is cut into kDepthStrips horizontal strips, one workgroup
// each (one workgroup per box kept 16 CUs busy for 1.9 ms on a 1080p frame with 16 flowers); the strip partials
// (double sum, count) are combined in strip order by depth_box_final_kernel, so the result does not depend on timing.
constexpr int kDepthStrips = 32;
struct DepthPartial { double sum; int cnt; int pad; };

// valid = (near < d < far) & (mask > 128) straight from the inputs (out-of-frame = "valid": erode's default border)
template <typename DT>
__device__ __forceinline__ unsigned char depth_is_valid(const DT* depth, const unsigned char* mask, int FH, int FW, int y,
                                                        int x, float div, float nearp, float farp) {
  if (y < 0 || y >= FH || x < 0 || x >= FW) return 1;
  const size_t i = (size_t)y * FW + x;
  const float d = (float)depth[i] / div;
  return (d > nearp && d < farp && mask[i] > 128) ? 1 : 0;
}

constexpr int kDepthLds = 48 * 1024;            // validity tile of one strip (+5 halo) when it fits

template <typename DT>
__global__ __launch_bounds__(256) void depth_box_kernel(const DT* depth, const unsigned char* mask, int FH, int FW,
                                                        float div, float nearp, float farp, const int* boxes,
                                                        DepthPartial* part) {
  __shared__ unsigned char vt[kDepthLds];
  const int b = blockIdx.x, strip = blockIdx.y;
  const int x0 = boxes[b * 4], y0 = boxes[b * 4 + 1], x1 = boxes[b * 4 + 2], y1 = boxes[b * 4 + 3];
  // numpy slicing semantics of depth[hmin:hmax, wmin:wmax] for in-frame, non-negative boxes
  const int xs = max(x0, 0), ys = max(y0, 0), xe = min(x1, FW), ye = min(y1, FH);
  const int bw = max(xe - xs, 0), bh = max(ye - ys, 0);
  const int r0 = (int)((long)bh * strip / kDepthStrips), r1 = (int)((long)bh * (strip + 1) / kDepthStrips);
  const int nr = r1 - r0;
  // validity of the strip and its 10 x 10 neighbourhood: rows ys+r0-5 .. ys+r1+3, columns xs-5 .. xe+3
  const int tw = bw + 9, th = nr + 9;
  const bool tiled = nr > 0 && bw > 0 && (long)tw * th <= kDepthLds;
  if (tiled) {
    for (int i = threadIdx.x; i < tw * th; i += 256) {
      const int ty = i / tw, tx = i - ty * tw;
      vt[i] = depth_is_valid(depth, mask, FH, FW, ys + r0 - 5 + ty, xs - 5 + tx, div, nearp, farp);
    }
  }
  __syncthreads();
  double sum = 0.0;
  int cnt = 0;
  for (int i = threadIdx.x; i < bw * nr; i += 256) {
    const int px = i % bw, py = i / bw;
    const int x = xs + px, y = ys + r0 + py;
    bool ok = true;
    if (tiled) {
      // all 76 taps unconditionally (fully unrolled, no early exit): independent LDS reads pipeline, whereas a
      // break-on-first-hole loop serialises one LDS latency per tap (60 us of an 80 us launch)
      constexpr int LO[10] = {5, 2, 1, 0, 0, 0, 0, 0, 1, 2}, HI[10] = {5, 8, 9, 9, 9, 9, 9, 9, 9, 8};
      unsigned allv = 1;
      const unsigned char* t0 = vt + py * tw + px;
#pragma unroll
      for (int ey = 0; ey < 10; ++ey)
#pragma unroll
        for (int ex = LO[ey]; ex <= HI[ey]; ++ex) allv &= t0[ey * tw + ex];
      ok = allv != 0;
    } else {
      for (int ey = 0; ey < 10 && ok; ++ey)
        for (int ex = kEllipseLo[ey]; ex <= kEllipseHi[ey]; ++ex)
          if (!depth_is_valid(depth, mask, FH, FW, y + ey - 5, x + ex - 5, div, nearp, farp)) { ok = false; break; }
    }
    if (ok) {
      const float dm = (float)depth[(size_t)y * FW + x] / div;   // metres, float32 like the reference
      sum += (double)(dm * 1000.0f);                              // reference averages millimetres
      ++cnt;
    }
  }
  __shared__ double ssum[256];
  __shared__ int scnt[256];
  ssum[threadIdx.x] = sum; scnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[b * kDepthStrips + strip].sum = ssum[0]; part[b * kDepthStrips + strip].cnt = scnt[0]; }
}

__global__ void depth_box_final_kernel(const DepthPartial* part, const int* boxes, int n, float fx, float fy, float cx,
                                       float cy, float* depth_val, int* reliable, float* xyz) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  double s = 0.0;
  int c = 0;
  for (int k = 0; k < kDepthStrips; ++k) { s += part[b * kDepthStrips + k].sum; c += part[b * kDepthStrips + k].cnt; }
  const double dv = c > 0 ? (s / c) / 1000.0 : 0.0;
  depth_val[b] = (float)dv;
  reliable[b] = c >= 50 ? 1 : 0;
  // uv = centre of the (un-squared) box; depth is the ray length (mvg.py:387-408)
  const int x0 = boxes[b * 4], y0 = boxes[b * 4 + 1], x1 = boxes[b * 4 + 2], y1 = boxes[b * 4 + 3];
  const double u = (x0 + x1) * 0.5, v = (y0 + y1) * 0.5;
  const double xn = (u - cx) / fx, yn = (v - cy) / fy;
  const double z = dv / sqrt(xn * xn + yn * yn + 1.0);
  xyz[b * 3] = (float)(xn * z); xyz[b * 3 + 1] = (float)(yn * z); xyz[b * 3 + 2] = (float)z;
}

extern "C" int flope_depth_lift(const void* depth_dev, int depth_format, const uint8_t* mask_dev, int frame_h,
                                int frame_w, float depth_div, float near_plane, float far_plane, const int32_t* boxes_dev, int n,
                                const float* K4_host, uint8_t* scratch_dev, float* depth_val_dev,
                                int32_t* reliable_dev, float* xyz_dev, void* stream) {
  if (n < 0 || frame_h <= 0 || frame_w <= 0 || !(depth_div > 0.f) || !K4_host || depth_format < 0 || depth_format > 1) return -1;
  if (n == 0) return 0;
  if (!depth_dev || !mask_dev || !boxes_dev || !scratch_dev || !depth_val_dev || !reliable_dev || !xyz_dev) return -1;
  const size_t npix = (size_t)frame_h * frame_w;
  hipStream_t st = (hipStream_t)stream;
  // scratch: [H*W bytes, unused since the validity map became an LDS tile][16-byte aligned n * kDepthStrips partials]
  DepthPartial* part = (DepthPartial*)(scratch_dev + ((npix + 15) & ~(size_t)15));
  if (depth_format == 0)
    hipLaunchKernelGGL(depth_box_kernel<unsigned short>, dim3(n, kDepthStrips), dim3(256), 0, st, (const unsigned short*)depth_dev,
                       mask_dev, frame_h, frame_w, depth_div, near_plane, far_plane, boxes_dev, part);
  else
    hipLaunchKernelGGL(depth_box_kernel<float>, dim3(n, kDepthStrips), dim3(256), 0, st, (const float*)depth_dev, mask_dev,
                       frame_h, frame_w, depth_div, near_plane, far_plane, boxes_dev, part);
  hipLaunchKernelGGL(depth_box_final_kernel, dim3((n + 63) / 64), dim3(64), 0, st, part, boxes_dev, n, K4_host[0], K4_host[1],
                     K4_host[2], K4_host[3], depth_val_dev, reliable_dev, xyz_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- get_bbox_mask post-processing (reference fast_pose_predictor.py:50-54) ------------------------------------
// sum over instance masks -> clip [0,1] -> x255 -> uint8 (numpy astype truncation) at the detector's resolution
__global__ void merge_masks_kernel(const float* masks, int n, int hw, uint8_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += masks[(size_t)k * hw + i];
  s = fminf(fmaxf(s, 0.f), 1.f) * 255.f;
  out[i] = (uint8_t)s;
}

// One axis of cv2's INTER_LINEAR tables for 8-bit images (resize.cpp): source index, 11-bit fixed-point weights.
// The fraction is formed exactly as OpenCV does -- double product, float cast, float subtract -- with explicit
// round-to-nearest intrinsics so that no fused multiply-add changes a bit.
__device__ __forceinline__ void linear_tap(int d, int src, double scale, int* s0, int* s1, int* a0, int* a1) {
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

__global__ void resize_linear_u8_kernel(const uint8_t* in, int h, int w, uint8_t* out, int H, int W, double sx,
                                        double sy) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int x0, x1, a0, a1, y0, y1, b0, b1;
  linear_tap(x, w, sx, &x0, &x1, &a0, &a1);
  linear_tap(y, h, sy, &y0, &y1, &b0, &b1);
  const int S0 = in[(size_t)y0 * w + x0] * a0 + in[(size_t)y0 * w + x1] * a1;
  const int S1 = in[(size_t)y1 * w + x0] * a0 + in[(size_t)y1 * w + x1] * a1;
  out[(size_t)y * W + x] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}

// cv2.resize(img, (W, H)) (INTER_LINEAR, 8-bit) of a device image: shared with the detector front end (yolo_engine.hip)
extern "C" int flope_resize_linear_u8_launch(const uint8_t* in, int h, int w, uint8_t* out, int H, int W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (h == H && w == W) return (int)hipMemcpyAsync(out, in, (size_t)H * W, hipMemcpyDeviceToDevice, st);
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((W + 255) / 256, H), dim3(256), 0, st, in, h, w, out, H, W,
                     1.0 / ((double)W / w), 1.0 / ((double)H / h));
  return (int)hipGetLastError();
}

extern "C" int flope_merge_masks_resize(const float* masks_dev, int n, int h, int w, uint8_t* scratch_dev,
                                        uint8_t* out_dev, int H, int W, void* stream) {
  if (n < 0 || h < 1 || w < 1 || H < 1 || W < 1 || !out_dev || (n > 0 && !masks_dev)) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(out_dev, 0, (size_t)H * W, st) == hipSuccess ? 0 : -2;
  const bool same = h == H && w == W;                      // cv2.resize copies when the size does not change
  if (!same && !scratch_dev) return -1;
  uint8_t* merged = same ? out_dev : scratch_dev;
  hipLaunchKernelGGL(merge_masks_kernel, dim3((h * w + 255) / 256), dim3(256), 0, st, masks_dev, n, h * w, merged);
  if (!same)
    hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((W + 255) / 256, H), dim3(256), 0, st, merged, h, w, out_dev, H, W,
                       1.0 / ((double)W / w), 1.0 / ((double)H / h));    // cv::resize: scale = 1. / inv_scale
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
