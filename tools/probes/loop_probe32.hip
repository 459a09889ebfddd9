// conv_w4's inner-loop skeleton on the two 16-bit MFMA shapes, side by side in one process (r05):
//   SHAPE 16: a sub-step = 32 x v_mfma_f32_16x16x32_f16 in 8 groups of 4  (wave tile 128 px x 64 ch = 8 x 4 tiles of 16 x 16)
//   SHAPE 32: a sub-step = 16 x v_mfma_f32_32x32x16_f16 in 8 groups of 2  (the same wave tile = 4 x 2 tiles of 32 x 32, two k-steps of 16)
// Both read 12 ds_read_b128 per sub-step (4 weight + 8 pixel fragments of the NEXT sub-step, conflict-free 64-byte-row images),
// one barrier per double step, K LDS-DMA pieces per double step in the second sub-step's groups.
// Printed: shader cycles per double step (s_memtime, floor 1024), the in-kernel clock (s_memtime / s_memrealtime) and the wall
// time per double step -- the guide's DVFS note (MI355X_MICROARCH.md, give-back item 7) says the chip may hold a different clock
// on the two shapes, so cycles alone do not decide.  LDS holds full-range random f16 (both signs).
//   hipcc --offload-arch=gfx950 -O3 -o build/loop_probe32 tools/probes/loop_probe32.hip && build/loop_probe32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ void fill_lds(char* smem) {
  for (int i = threadIdx.x; i < 40960; i += 256) {         // 160 KB of f16 pairs: exponent 14..16 (0.5 .. 4), random mantissa and sign
    unsigned h = i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo = (h & 0x83ffu) | ((14u + ((h >> 10) & 1u)) << 10);
    const unsigned hi = ((h >> 16) & 0x83ffu) | ((14u + ((h >> 27) & 1u)) << 10);
    ((unsigned*)smem)[i] = lo | (hi << 16);
  }
}

template <int SHAPE, bool RD, bool BAR, int K>
__global__ __launch_bounds__(256, 1) void probe(const char* src, float* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  fill_lds(smem);
  __syncthreads();
  h8 wf[2][4], xf[2][8];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < 4; ++i) wf[s][i] = *(const h8*)(smem + 65536 + (wave & 1) * 8192 + i * 1024 + lane * 16);
    for (int i = 0; i < 8; ++i) xf[s][i] = *(const h8*)(smem + (wave >> 1) * 16384 + i * 2048 + lane * 16);
  }
  // fragment addresses on 64-byte rows (32 channels of a pixel / of a weight row), 16-byte slot swizzled by (row >> 2) & 3:
  //   16x16x32: lane (r16, g) reads slot g of row r16 of tile i, rows in the kernels' lane order (conflict-free with the 0x1320 map)
  //   32x32x16: lane (r32, h) reads slot 2 s + h of row r32 of tile i for k-step s: natural row order is conflict-free
  int xo[8], wo[4];
  if constexpr (SHAPE == 16) {
    const int r16 = lane & 15, g = lane >> 4;
    for (int i = 0; i < 8; ++i) {
      const int row = i * 16 + r16;
      xo[i] = (wave >> 1) * 16384 + row * 64 + ((g ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3)) << 4);
    }
    for (int i = 0; i < 4; ++i) wo[i] = 65536 + (wave & 1) * 8192 + (i * 16 + r16) * 64 + ((g ^ ((0x1320 >> ((r16 >> 2) * 4)) & 3)) << 4);
  } else {
    const int r32 = lane & 31, h = lane >> 5;
    for (int i = 0; i < 8; ++i) {                          // i = 2 pt + s
      const int row = (i >> 1) * 32 + r32, s = i & 1;
      xo[i] = (wave >> 1) * 16384 + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4);
    }
    for (int i = 0; i < 4; ++i) {                          // i = 2 ct + s
      const int row = (i >> 1) * 32 + r32, s = i & 1;
      wo[i] = 65536 + (wave & 1) * 8192 + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4);
    }
  }
  __syncthreads();

  f4 acc16[SHAPE == 16 ? 8 : 1][4];
  f16v acc32[SHAPE == 32 ? 4 : 1][2];
  if constexpr (SHAPE == 16) {
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc16[i][j] = f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc16[i][0]), "+a"(acc16[i][1]), "+a"(acc16[i][2]), "+a"(acc16[i][3]));
  } else {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc32[i][j][q] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc32[i][0]), "+a"(acc32[i][1]));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define PIECE(i_) (((r & 3) * 16 + wave * 4 + ((i_) & 3)) * 1024)
#define DMA_GRP(P_, DMA_)                                                                                      \
  do {                                                                                                         \
    if constexpr (DMA_ && (P_) < K) GLDS16(base + PIECE(P_), smem + 98304 + ((r & 3) * 16 + wave * 4 + ((P_) & 3)) * 1024); \
    if constexpr (DMA_ && (P_) + 8 < K) GLDS16(base + PIECE((P_) + 1), smem + 98304 + ((r & 3) * 16 + wave * 4 + (((P_) + 1) & 3)) * 1024); \
  } while (0)
  // group P_ of a sub-step on set C_; the reads of set N_ ride in groups 0..5, two each
#define GRP16(P_, C_, N_, DMA_)                                                                                \
  do {                                                                                                         \
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                                           \
      acc16[P_][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[C_][ct], xf[C_][P_], acc16[P_][ct], 0, 0, 0);  \
    if constexpr (RD) {                                                                                        \
      if constexpr ((P_) < 2) {                                                                                \
        wf[N_][2 * (P_)] = *(const h8*)(smem + wo[2 * (P_)] + koff);                                           \
        wf[N_][2 * (P_) + 1] = *(const h8*)(smem + wo[2 * (P_) + 1] + koff);                                   \
      } else if constexpr ((P_) < 6) {                                                                         \
        xf[N_][2 * ((P_) - 2)] = *(const h8*)(smem + xo[2 * ((P_) - 2)] + xk);                                 \
        xf[N_][2 * ((P_) - 2) + 1] = *(const h8*)(smem + xo[2 * ((P_) - 2) + 1] + xk);                         \
      }                                                                                                        \
    }                                                                                                          \
    DMA_GRP(P_, DMA_);                                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                         \
    if constexpr (RD && (P_) < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                           \
    if constexpr (DMA_ && (P_) < K) __builtin_amdgcn_sched_group_barrier(0x020, (P_) + 8 < K ? 2 : 1, 0);      \
  } while (0)
  // 32x32x16: group P_ = k-step s = P_ >> 2, pixel tile pt = P_ & 3, both channel tiles.  Fragment registers: wf[set][2 ct + s],
  // xf[set][2 pt + s].  Reads of the next set: groups 0, 1 the four weight fragments, groups 2..5 the eight pixel fragments -- the
  // s = 0 ones first (pt 0..3 at s = 0 in groups 2, 3; s = 1 in groups 4, 5).
#define GRP32(P_, C_, N_, DMA_)                                                                                \
  do {                                                                                                         \
    constexpr int s_ = (P_) >> 2, pt_ = (P_) & 3;                                                              \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                           \
      acc32[pt_][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[C_][2 * ct + s_], xf[C_][2 * pt_ + s_], acc32[pt_][ct], 0, 0, 0); \
    if constexpr (RD) {                                                                                        \
      if constexpr ((P_) == 0) {                                                                               \
        wf[N_][0] = *(const h8*)(smem + wo[0] + koff); wf[N_][2] = *(const h8*)(smem + wo[2] + koff);          \
      } else if constexpr ((P_) == 1) {                                                                        \
        wf[N_][1] = *(const h8*)(smem + wo[1] + koff); wf[N_][3] = *(const h8*)(smem + wo[3] + koff);          \
      } else if constexpr ((P_) < 6) {                                                                         \
        constexpr int q_ = (P_) - 2, ns_ = q_ >> 1, np_ = (q_ & 1) * 2;                                        \
        xf[N_][2 * np_ + ns_] = *(const h8*)(smem + xo[2 * np_ + ns_] + xk);                                   \
        xf[N_][2 * (np_ + 1) + ns_] = *(const h8*)(smem + xo[2 * (np_ + 1) + ns_] + xk);                       \
      }                                                                                                        \
    }                                                                                                          \
    DMA_GRP(P_, DMA_);                                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                         \
    if constexpr (RD && (P_) < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                           \
    if constexpr (DMA_ && (P_) < K) __builtin_amdgcn_sched_group_barrier(0x020, (P_) + 8 < K ? 2 : 1, 0);      \
  } while (0)
#define GRP(P_, C_, N_, DMA_) do { if constexpr (SHAPE == 16) GRP16(P_, C_, N_, DMA_); else GRP32(P_, C_, N_, DMA_); } while (0)
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
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  float s = 0;
  if constexpr (SHAPE == 16) { for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc16[i][j][0] + acc16[i][j][3]; }
  else { for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc32[i][j][0] + acc32[i][j][15]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

struct Res { double cyc, ghz, ns; };
template <int SHAPE, bool RD, bool BAR, int K>
static Res run1(const char* src, float* out, unsigned long long* cyc, int reps) {
  hipFuncSetAttribute((const void*)probe<SHAPE, RD, BAR, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<SHAPE, RD, BAR, K>), dim3(256), dim3(256), 163840, 0, src, out, cyc, reps);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<SHAPE, RD, BAR, K>), dim3(256), dim3(256), 163840, 0, src, out, cyc, reps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2048);
  hipMemcpy(h.data(), cyc, 16384, hipMemcpyDeviceToHost);
  double sc = 0, sr = 0;
  for (int i = 0; i < 1024; ++i) { sc += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
  Res r; r.cyc = sc / 1024 / reps; r.ghz = sc / sr * 0.1; r.ns = sr / 1024 / reps * 10.0;
  (void)ms;
  return r;
}
template <bool RD, bool BAR, int K>
static void run(const char* src, float* out, unsigned long long* cyc, int reps, const char* name) {
  // interleaved rounds of both shapes in one process (guide rule 24); medians
  std::vector<Res> a, b;
  for (int round = 0; round < 5; ++round) {
    a.push_back(run1<16, RD, BAR, K>(src, out, cyc, reps));
    b.push_back(run1<32, RD, BAR, K>(src, out, cyc, reps));
  }
  auto med = [](std::vector<Res> v) { std::sort(v.begin(), v.end(), [](const Res& x, const Res& y) { return x.ns < y.ns; }); return v[v.size() / 2]; };
  const Res ra = med(a), rb = med(b);
  printf("%-44s 16x16x32: %7.1f cyc  %5.3f GHz  %6.1f ns | 32x32x16: %7.1f cyc  %5.3f GHz  %6.1f ns | wall 32/16 = %.3f\n", name,
         ra.cyc, ra.ghz, ra.ns, rb.cyc, rb.ghz, rb.ns, rb.ns / ra.ns);
  fflush(stdout);
}

int main() {
  char* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 2 * 64 * 65536); hipMemset(src, 0x3c, 2 * 64 * 65536);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 16384);
  const int reps = 20000;                                   // ~10 ms per launch: long enough for the sustained clock
  printf("per double step (64 MFMA 16x16x32 = 32 MFMA 32x32x16 per wave; floor 1024 cycles)\n");
  run<false, false, 0>(src, out, cyc, reps, "MFMA only");
  run<true, false, 0>(src, out, cyc, reps, "MFMA + 24 ds_read_b128");
  run<true, true, 0>(src, out, cyc, reps, "MFMA + reads + barrier");
  run<true, true, 4>(src, out, cyc, reps, "MFMA + reads + barrier + 4 DMA pieces");
  run<true, true, 6>(src, out, cyc, reps, "MFMA + reads + barrier + 6 DMA pieces");
  run<true, true, 8>(src, out, cyc, reps, "MFMA + reads + barrier + 8 DMA pieces");
  run<true, true, 12>(src, out, cyc, reps, "MFMA + reads + barrier + 12 DMA pieces");
  run<false, false, 6>(src, out, cyc, reps, "MFMA + 6 DMA pieces");
  return 0;
}
