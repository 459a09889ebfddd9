// FETCH_SIZE calibration: how many bytes does rocprofv3's FETCH_SIZE report for
//   (a) a wide coalesced streaming read, 16 B per lane, 1 KiB per wave-instruction (the guide: exactly 1/2 of the bytes), and
//   (b) the residual-prefetch pattern of conv_stag's row-band kernel (conv_stag.hip, RES && ROWS, double step 6): lane
//       (g = lane >> 4, r = lane & 15) reads bytes [32 g, 32 g + 16) and [32 g + 16, 32 g + 32) of the 128-byte NHWC
//       row of pixel r with TWO 16-byte loads -- every wave-instruction touches 16 lines x 4 segments at a 32-byte stride?
// Each kernel reads every byte of a 512 MiB buffer exactly once (no reuse, larger than the Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o build/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- build/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream16_kernel(const u32x4* in, unsigned* out, size_t n16) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const u32x4 v = in[i];
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// one wave = 16 consecutive pixels (128-byte rows); lane (g, r): two 16-byte loads at 32 g and 32 g + 16 of pixel r
__global__ __launch_bounds__(256) void res32_kernel(const char* in, unsigned* out, size_t npix) {
  const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
  unsigned acc = 0;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
  for (size_t p0 = wave * 16; p0 < npix; p0 += nw * 16) {
    const char* rp = in + (p0 + r) * 128 + g * 32;
    const u32x4 a = *(const u32x4*)rp, b = *(const u32x4*)(rp + 16);
    acc ^= a[0] ^ a[3] ^ b[1] ^ b[2];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)512 << 20;
  char* buf; unsigned* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64);
  hipMemset(buf, 1, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream16_kernel, dim3(4096), dim3(256), 0, 0, (const u32x4*)buf, out, bytes / 16);
    hipLaunchKernelGGL(res32_kernel, dim3(4096), dim3(256), 0, 0, buf, out, bytes / 128);
  }
  hipDeviceSynchronize();
  printf("bytes per launch: %zu\n", bytes);
  return 0;
}
