// What an epilogue's 16-byte stores cost by the SHAPE of one wave-instruction (r04).
//   hipcc --offload-arch=gfx950 -O3 -o build/store_probe tools/probes/store_probe.hip && build/store_probe
// Every wave stores the same bytes to the same place -- a tile of 128 pixels x 128 bytes inside rows of ROWB bytes (the padded
// NHWC output of a 64- or 128-channel layer), 16 store instructions of 16 bytes per lane -- in one of these lane orders:
//   0  MFMA order    lane (g = lane >> 4, r = lane & 15): pixel r of pixel tile pt, 16-byte chunks g (first store) and 4 + g
//                    (second store) of the pixel's 128 bytes: 64 lanes -> 16 pixels x 64 bytes per instruction (conv_w4's epilogue)
//   1  line order    lane l: pixel l >> 3 of the instruction's 8 pixels, chunk l & 7: 8 whole 128-byte lines per instruction
//   2  line order through LDS: the values start in MFMA order, go through a wave-private 2 KB LDS image (2 ds_write_b128 + 2
//      ds_read_b128 per pixel tile, XOR-swizzled: conflict-free both ways) and leave in line order
//   3  MFMA order with 32 contiguous bytes per lane (two adjacent 16-byte stores: what a different channel permutation would give)
// Reported: shader cycles of the store phase per workgroup (median over workgroups; entry -> all stores issued, and -> vmcnt(0)),
// and the kernel's wall time for a grid of `rounds` workgroups per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(char* out, unsigned long long* cyc, int rowb, int reps) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 2048];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpx = wave & 1, wch = wave >> 1;
  const int g = lane >> 4, r = lane & 15;
  // workgroup tile: 256 pixels x 256 bytes; wave (wpx, wch): pixels wpx * 128 .., bytes wch * 128 ..
  char* const base = out + ((size_t)blockIdx.x * 256 + wpx * 128) * rowb + wch * 128;
  u4 v[8][2];
#pragma unroll
  for (int pt = 0; pt < 8; ++pt)
#pragma unroll
    for (int c = 0; c < 2; ++c) v[pt][c] = u4{(unsigned)(blockIdx.x * 65536 + wave * 16384 + (pt * 16 + r) * 128 + (c * 4 + g) * 16), 1u, 2u, 3u};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t1 = t0;
  for (int rep = 0; rep < reps; ++rep) {
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int pt = 0; pt < 8; ++pt)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          *(u4*)(base + (size_t)(pt * 16 + r) * rowb + (MODE == 0 ? (c * 4 + g) * 16 : (g * 2 + c) * 16)) = v[pt][c];
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int pt = 0; pt < 8; ++pt)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          *(u4*)(base + (size_t)(pt * 16 + c * 8 + (lane >> 3)) * rowb + (lane & 7) * 16) = v[pt][c];
    } else {
      char* const scr = smem + wave * 2048;
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) *(u4*)(scr + r * 128 + (((c * 4 + g) ^ (r & 7)) << 4)) = v[pt][c];
        u4 w[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int p = c * 8 + (lane >> 3);
          w[c] = *(const u4*)(scr + p * 128 + (((lane & 7) ^ (p & 7)) << 4));
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
          *(u4*)(base + (size_t)(pt * 16 + c * 8 + (lane >> 3)) * rowb + (lane & 7) * 16) = w[c];
      }
    }
  }
  t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

template <int MODE>
static void run(const char* name, char* d, unsigned long long* dc, int rowb, int rounds, int ncu) {
  const int grid = rounds * ncu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, rowb, 1);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, rowb, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)grid * 8);
  hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> a, b;
  for (int i = 0; i < grid; ++i) {                      // per workgroup: its slowest wave
    unsigned long long ma = 0, mb = 0;
    for (int w = 0; w < 4; ++w) { ma = std::max(ma, h[(i * 4 + w) * 2]); mb = std::max(mb, h[(i * 4 + w) * 2 + 1]); }
    a.push_back(ma); b.push_back(mb);
  }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  const double bytes = (double)grid * 65536;
  printf("%-44s rowb %4d rounds %d: issue %6llu  landed %6llu cycles / workgroup (median);  %7.2f us / launch = %5.2f TB/s\n", name, rowb, rounds,
         a[a.size() / 2], b[b.size() / 2], ms * 100.0, bytes / (ms * 1e-4) / 1e12);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  char* d = nullptr; unsigned long long* dc = nullptr;
  const size_t bytes = (size_t)8 * ncu * 256 * 512 + (1 << 20);
  hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
  hipMalloc(&dc, (size_t)8 * ncu * 8 * 8);
  for (int rowb : {256, 512}) {
    for (int rounds : {1, 4}) {
      run<0>("MFMA order (16 px x 64 B per instruction)", d, dc, rowb, rounds, ncu);
      run<3>("MFMA order, 32 contiguous B per lane", d, dc, rowb, rounds, ncu);
      run<1>("line order (8 px x 128 B per instruction)", d, dc, rowb, rounds, ncu);
      run<2>("line order through LDS (from MFMA order)", d, dc, rowb, rounds, ncu);
    }
  }
  return 0;
}
