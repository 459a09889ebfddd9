// Strict fp32 mode (FLOPE_DT_F32): plain direct convolution on zero-bordered NHWC
// float tensors -- no MFMA, no 16-bit storage.  It is the on-device cross-check for
// the MFMA kernels and the mode whose rotations sit at fp32 round-off from the oracle.
// One thread = one (pixel, 4 consecutive output channels); threads of a wave share a
// pixel neighbourhood so activation loads broadcast and weight loads coalesce
// ([ky][kx][ci][cout] layout).
#include "../../include/flope_amd.h"
#include "common.h"

__global__ __launch_bounds__(256) void naive_conv_kernel(const NaiveConvP p) {
  const int cq = p.Cout / 4;
  const size_t total = (size_t)p.B * p.Ho * p.Wo * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq) * 4;
    size_t r = i / cq;
    const int wo = (int)(r % p.Wo); r /= p.Wo;
    const int ho = (int)(r % p.Ho);
    const int b = (int)(r / p.Ho);
    f32x4 acc = *(const f32x4*)(p.bias + c4);
    for (int ky = 0; ky < p.KH; ++ky)
      for (int kx = 0; kx < p.KW; ++kx) {
        const float* xp = p.in + (((size_t)b * p.Hip + ho * p.stride + ky + p.in_off) * p.Wip +
                                  wo * p.stride + kx + p.in_off) * p.Cin_stored;
        const float* wp = p.w + ((size_t)(ky * p.KW + kx) * p.Cin) * p.Cout + c4;
        for (int ci = 0; ci < p.Cin; ++ci) {
          const float xv = xp[ci];
          const f32x4 wv = *(const f32x4*)(wp + (size_t)ci * p.Cout);
          acc[0] = fmaf(xv, wv[0], acc[0]); acc[1] = fmaf(xv, wv[1], acc[1]);
          acc[2] = fmaf(xv, wv[2], acc[2]); acc[3] = fmaf(xv, wv[3], acc[3]);
        }
      }
    const size_t o = (((size_t)b * p.Hop + ho + 1) * p.Wop + wo + 1) * p.Cout + c4;
    if (p.res) {
      const f32x4 rv = *(const f32x4*)(p.res + o);
      acc[0] += rv[0]; acc[1] += rv[1]; acc[2] += rv[2]; acc[3] += rv[3];
    }
    if (p.relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
    *(f32x4*)(p.out + o) = acc;
  }
}

extern "C" int flope_naive_conv_launch(const NaiveConvP* p, void* stream) {
  const size_t total = (size_t)p->B * p->Ho * p->Wo * (p->Cout / 4);
  const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(naive_conv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

// Developer aid: the packed 16-bit epilogue helpers of common.h on caller-supplied bit patterns, so that a test can run
// the DEVICE code over every pattern.  which: 0 pk_out16<bf16>(w, relu), 1 pk_out16<f16>(w, relu), 2 pk_relu16<bf16>(w), 3 pk_relu16<f16>(w),
// 4 pk_max16_nonneg(w, w2) (w2 = the next input word, cyclically)
__global__ void pk16_probe_kernel(const unsigned* __restrict__ in, unsigned* __restrict__ out, int n, int which, int relu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned w = in[i];
  unsigned r;
  if (which == 0) r = pk_out16<bf16_t>(w, relu != 0);
  else if (which == 1) r = pk_out16<f16_t>(w, relu != 0);
  else if (which == 2) r = pk_relu16<bf16_t>(w);
  else if (which == 3) r = pk_relu16<f16_t>(w);
  else r = pk_max16_nonneg(w, in[i + 1 < n ? i + 1 : 0]);
  out[i] = r;
}

extern "C" int flope_debug_pk16(int which, int relu, const void* in_dev, void* out_dev, int n, void* stream) {
  if (!in_dev || !out_dev || n < 1 || which < 0 || which > 4) return FLOPE_EINVAL;
  hipLaunchKernelGGL(pk16_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned*)in_dev, (unsigned*)out_dev, n, which, relu);
  return hipGetLastError() == hipSuccess ? FLOPE_OK : FLOPE_EHIP;
}
