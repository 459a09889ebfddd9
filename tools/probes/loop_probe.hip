// What a conv_w4-shaped inner loop costs beyond its MFMAs, piece by piece (synthetic: LDS holds random bytes, results unused).
//   hipcc --offload-arch=gfx950 -O3 -o build/loop_probe tools/probes/loop_probe.hip && build/loop_probe
// One workgroup of 4 waves per CU.  A sub-step = 32 MFMA 16x16x32 f16 in 8 groups of 4 on 32 accumulators (128 AGPRs), operands
// from fragment set C; RD: the 12 ds_read_b128 of set 1-C ride in groups 0..5 (2 each); BAR: one s_barrier per two sub-steps;
// DMA: K LDS-DMA pieces per two sub-steps in the second one's groups (global_load_lds, L2-resident source).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <bool RD, bool BAR, int K, int STRIDE, bool GATHER = false>
__global__ __launch_bounds__(256, 1) void probe(const char* src, float* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // GATHER: a piece = 16 half cache lines (64 B of 16 different 128-B lines: one half-chunk of 16 pixels), not one contiguous KiB
  const char* base = src + (size_t)(blockIdx.x & 63) * 65536 + (GATHER ? (lane >> 2) * 128 + (lane & 3) * 16 : lane * 16);
  for (int i = threadIdx.x; i < 32768; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
  __syncthreads();
  h8 wf[2][4], xf[2][8];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < 4; ++i) wf[s][i] = *(const h8*)(smem + 65536 + (wave & 1) * 8192 + i * 1024 + lane * 16);
    for (int i = 0; i < 8; ++i) xf[s][i] = *(const h8*)(smem + (wave >> 1) * 16384 + i * 2048 + lane * 16);
  }
  f4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
  // fragment addresses: the conv kernels' conflict-free pattern is 16 rows x 128 B with a 4-slot swizzle; here rows of STRIDE bytes
  const int r16 = lane & 15, g = lane >> 4;
  int xo[8], wo;
  // STRIDE 64: the kernels' image -- 64-B rows (32 channels of a pixel), slot = g ^ {0, 2, 3, 1}[(row >> 2) & 3]: every ds_read_b128
  // lane group {0-3, 12-15, 20-27}, ... covers 16 distinct 16-B bank quads.  Other strides: g ^ ((row >> 2) & 3) (conflicts).
  for (int i = 0; i < 8; ++i) {
    const int row = i * 16 + r16;
    const int sw = STRIDE == 64 ? (0x1320 >> (((row >> 2) & 3) * 4)) & 3 : (row >> 2) & 3;
    xo[i] = (wave >> 1) * 16384 + row * STRIDE + ((g ^ sw) << 4);
  }
  wo = 65536 + (wave & 1) * 8192 + r16 * 64 + ((g ^ ((0x1320 >> ((r16 >> 2) * 4)) & 3)) << 4);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define PIECE(i_) (GATHER ? ((r & 3) * 8 + wave * 2 + ((i_) & 1)) * 2048 + (((i_) >> 1) & 1) * 64 : ((r & 3) * 16 + wave * 4 + ((i_) & 3)) * 1024)
#define GRP(P_, C_, N_, DMA_)                                                                                  \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                                           \
      acc[P_][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[C_][ct], xf[C_][P_], acc[P_][ct], 0, 0, 0);      \
    if constexpr (RD) {                                                                                        \
      if constexpr ((P_) < 2) {                                                                                \
        wf[N_][2 * (P_)] = *(const h8*)(smem + wo + koff + (2 * (P_)) * 1024);                                 \
        wf[N_][2 * (P_) + 1] = *(const h8*)(smem + wo + koff + (2 * (P_) + 1) * 1024);                         \
      } else if constexpr ((P_) < 6) {                                                                         \
        xf[N_][2 * ((P_) - 2)] = *(const h8*)(smem + xo[2 * ((P_) - 2)] + xk);                                 \
        xf[N_][2 * ((P_) - 2) + 1] = *(const h8*)(smem + xo[2 * ((P_) - 2) + 1] + xk);                         \
      }                                                                                                        \
    }                                                                                                          \
    if constexpr (DMA_ && (P_) < K) GLDS16(base + PIECE(P_), smem + 98304 + ((r & 3) * 16 + wave * 4 + ((P_) & 3)) * 1024); \
    if constexpr (DMA_ && (P_) + 8 < K) GLDS16(base + PIECE((P_) + 1), smem + 98304 + ((r & 3) * 16 + wave * 4 + (((P_) + 1) & 3)) * 1024); \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                         \
    if constexpr (RD && (P_) < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                           \
    if constexpr (DMA_ && (P_) < K) __builtin_amdgcn_sched_group_barrier(0x020, (P_) + 8 < K ? 2 : 1, 0);      \
  } while (0)
#define SUB(C_, N_, DMA_)                                                                                      \
  do { GRP(0, C_, N_, DMA_); GRP(1, C_, N_, DMA_); GRP(2, C_, N_, DMA_); GRP(3, C_, N_, DMA_);                 \
       GRP(4, C_, N_, DMA_); GRP(5, C_, N_, DMA_); GRP(6, C_, N_, DMA_); GRP(7, C_, N_, DMA_); } while (0)
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    const int koff = (r & 1) * 4096, xk = (r & 1) * 8192;
    SUB(0, 1, false);
    if constexpr (K > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K) : "memory");
    if constexpr (BAR) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    SUB(1, 0, true);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool RD, bool BAR, int K, int STRIDE, bool GATHER = false>
static void run(const char* src, float* out, unsigned long long* cyc, int reps, const char* name) {
  hipFuncSetAttribute((const void*)probe<RD, BAR, K, STRIDE, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<RD, BAR, K, STRIDE, GATHER>), dim3(256), dim3(256), 163840, 0, src, out, cyc, reps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-62s %8.1f cycles / 64 MFMA (floor 1024)\n", name, s / 1024 / reps);
  fflush(stdout);
}

int main() {
  char* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 2 * 64 * 65536); hipMemset(src, 0x3c, 2 * 64 * 65536);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8192);
  const int reps = 2000;
  run<false, false, 0, 64>(src, out, cyc, reps, "MFMA only");
  run<false, true, 0, 64>(src, out, cyc, reps, "MFMA + barrier");
  run<true, false, 0, 64>(src, out, cyc, reps, "MFMA + 24 ds_read_b128 (conflict-free image)");
  run<true, true, 0, 64>(src, out, cyc, reps, "MFMA + reads + barrier");
  run<true, true, 4, 64>(src, out, cyc, reps, "MFMA + reads + barrier + 4 DMA pieces");
  run<true, true, 6, 64>(src, out, cyc, reps, "MFMA + reads + barrier + 6 DMA pieces");
  run<true, true, 8, 64>(src, out, cyc, reps, "MFMA + reads + barrier + 8 DMA pieces");
  run<true, true, 12, 64>(src, out, cyc, reps, "MFMA + reads + barrier + 12 DMA pieces (a patch burst)");
  run<true, true, 16, 64>(src, out, cyc, reps, "MFMA + reads + barrier + 16 DMA pieces (a patch burst)");
  run<true, true, 4, 64, true>(src, out, cyc, reps, "MFMA + reads + barrier + 4 gathered pieces (16 half lines each)");
  run<true, true, 8, 64, true>(src, out, cyc, reps, "MFMA + reads + barrier + 8 gathered pieces");
  run<true, true, 12, 64, true>(src, out, cyc, reps, "MFMA + reads + barrier + 12 gathered pieces");
  run<false, false, 6, 64>(src, out, cyc, reps, "MFMA + 6 DMA pieces");
  run<true, false, 0, 128>(src, out, cyc, reps, "MFMA + 24 ds_read_b128 (128-B rows: 4-way conflicts on x)");
  run<true, true, 0, 128>(src, out, cyc, reps, "MFMA + reads (4-way conflicts on x) + barrier");
  return 0;
}
