// What a dependent chain of SMALL kernels costs per launch on this GPU (the detector is ~75 of them per frame).
//   hipcc --offload-arch=gfx950 -O3 -o build/launch_floor tools/calib/launch_floor.hip && build/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k_empty() {}
// each thread: one 16-byte load from `in`, one 16-byte store to `out` (a dependent chain when out of launch i = in of launch i+1)
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint4 v = in[i]; v.x += 1; out[i] = v; }
}
// two dependent memory round trips inside the kernel (index -> data), like params -> operands
__global__ void k_chase(const int* __restrict__ idx, const uint4* __restrict__ in, uint4* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int j = idx[i]; uint4 v = in[j]; v.x += 1; out[i] = v; }
}

template <typename F> static float timed(F f, int iters, hipStream_t st) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f(i);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < iters; ++i) f(i);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / iters;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  const int NB = 64, n = 920 * 16;                 // a 23 x 40 map of 128 channels in 16-byte granules
  std::vector<uint4*> buf(NB);
  for (auto& p : buf) { hipMalloc(&p, 4 << 20); hipMemset(p, 0, 4 << 20); }
  int* idx; hipMalloc(&idx, n * 4);
  std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 7) % n;
  hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int wg : {1, 58, 256, 1024}) {
    const int nn = wg == 1 ? 256 : (wg == 58 ? n : wg * 256);
    printf("grid %4d x 256:\n", wg);
    printf("  empty kernel                              %6.2f us per launch\n", timed([&](int) { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, st); }, iters, st));
    printf("  copy, ping-pong between two buffers       %6.2f\n", timed([&](int i) { hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, st, buf[i & 1], buf[(i + 1) & 1], nn); }, iters, st));
    printf("  copy, chain through 64 different buffers  %6.2f\n", timed([&](int i) { hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], nn); }, iters, st));
    if (nn <= n) printf("  index -> load -> store, 64 buffers        %6.2f\n", timed([&](int i) { hipLaunchKernelGGL(k_chase, dim3(wg), dim3(256), 0, st, idx, buf[i % NB], buf[(i + 1) % NB], nn); }, iters, st));
  }
  return 0;
}
