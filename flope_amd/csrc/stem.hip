// Stem: Conv2d(3,64,k7,s2,p3,bias=False) -> BatchNorm2d(64) -> ReLU on MFMA.
// Reference call site: sunflower/models/posenet.py:25 -> torchvision
// ResNet._forward_impl (conv1, bn1, relu); BN folded on the host.
//
// The crop batch is first converted (prep.hip) to a 4-channel, 3-pixel-bordered
// NHWC tensor, so one kernel row (ky) of the 7x7 window is ONE 32-deep MFMA
// k-step: k = kx*4 + c with kx 0..7 (kx = 7 and c = 3 carry zero weights), i.e.
// 64 contiguous bytes starting at input pixel (2*ho + ky, 2*wo).  K = 7 x 32 = 224
// (147 real).  A block owns 256 consecutive output pixels of one image; the
// input rows it needs (<= 13 at 224^2) sit in LDS as a linear copy and every
// fragment read is a 16-byte window at a 16-byte pixel pitch -> conflict free
// (identical addresses broadcast).  The 7 x [64][32] weight images (28 KB) are
// LDS resident for the whole block.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const StemP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int BM = 256, MT = 4, NT = 4, W_BYTES = 7 * 64 * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ws = smem;
  char* const Ps = smem + W_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int img = lid / p.tiles_per_image, t = lid - img * p.tiles_per_image;
  const int HoWo = p.Ho * p.Wo;
  const int m0 = t * BM, mend = min(m0 + BM, HoWo);       // within the image
  const int ho0 = m0 / p.Wo, ho1 = (mend - 1) / p.Wo;
  const int nrows = 2 * (ho1 - ho0) + 7;
  const int row_bytes = p.Wip * 8;

  // stage weights + input rows (both linear copies)
  for (int q = tid; q < W_BYTES / 16; q += 256) *(u32x4*)(Ws + q * 16) = *(const u32x4*)((const char*)p.w + q * 16);
  {
    const char* src = (const char*)p.in + ((size_t)img * p.Hip + 2 * ho0) * row_bytes;
    const int pieces = nrows * row_bytes / 16;
    for (int q = tid; q < pieces; q += 256) *(u32x4*)(Ps + (size_t)q * 16) = *(const u32x4*)(src + (size_t)q * 16);
  }
  __syncthreads();

  int xo[MT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = min(m0 + wave * 64 + pt * 16 + r16, mend - 1);
    const int ho = m / p.Wo, wo = m - ho * p.Wo;
    xo[pt] = (2 * (ho - ho0) * p.Wip + 2 * wo) * 8 + g * 16;
  }
  // weight rows are 64 B: slot g of row r lives at g ^ h[(r >> 2) & 3], h = {0,2,3,1}
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wo_ = r16 * 64 + ((g ^ wsw) << 4);

  f32x4 acc[MT][NT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    frag wf[NT], xf[MT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(Ws + ky * 4096 + ct * 1024 + wo_);
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) xf[pt] = *(const frag*)(Ps + xo[pt] + ky * row_bytes);
#pragma unroll
    for (int pt = 0; pt < MT; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], acc[pt][ct]);
  }

  ConvP q;
  q.out = p.out; q.res = nullptr; q.Wo = p.Wo; q.Hop = p.Ho + 2; q.Wop = p.Wo + 2; q.Cout = 64; q.relu = 1;
  const int cb = g * 16;
  float bias[NT * 4];
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) bias[i] = p.bias[cb + i];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int m = m0 + wave * 64 + pt * 16 + r16;
    if (m < mend) conv_epilogue_at<T, NT>(q, acc[pt], img, m / p.Wo, m % p.Wo, cb, bias);
  }
}

extern "C" int flope_stem_init() {
  hipError_t e = hipFuncSetAttribute((const void*)stem_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)stem_mfma_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return (int)e;
}

extern "C" int flope_stem_launch(const StemP* p, int dtype, size_t lds, void* stream) {
  const dim3 grid(p->B * p->tiles_per_image), block(256);
  if (dtype == 0) hipLaunchKernelGGL(stem_mfma_kernel<bf16_t>, grid, block, lds, (hipStream_t)stream, *p);
  else            hipLaunchKernelGGL(stem_mfma_kernel<f16_t>, grid, block, lds, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}
