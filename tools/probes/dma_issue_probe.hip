// Issue cost of one LDS-DMA piece (1 KiB per wave-instruction) to the wave that also issues the MFMAs, by addressing form.
//   hipcc --offload-arch=gfx950 -O3 -o build/dma_issue_probe tools/probes/dma_issue_probe.hip && build/dma_issue_probe
// One workgroup of 4 waves per CU; every wave runs REPS x {32 MFMA 16x16x32 f16 in 8 groups of 4, K pieces spread over the groups}.
// Forms: 0 none | 1 global_load_lds (64-bit per-lane address) | 2 buffer_load ... offen lds (32-bit per-lane offset)
//        | 3 buffer_load ... off lds with ADD_TID_ENABLE in the descriptor (no address VGPR at all; checks the landed bytes too).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int FORM, int K>
__global__ __launch_bounds__(256, 1) void probe(const char* src, float* out, unsigned long long* cyc, unsigned* chk, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)(blockIdx.x & 63) * 65536;
  // descriptors: word1 = base[47:32] | stride << 16; word3: NUM_FORMAT 7 (<<12), DATA_FORMAT 4 (<<15) for the plain one;
  // ADD_TID_ENABLE = bit 23 (DATA_FORMAT then extends the stride: 0)
  const unsigned long long b64 = (unsigned long long)base;
  u4 srd_plain = {(unsigned)b64, (unsigned)(b64 >> 32) & 0xffffu, 0x7fffffffu, 0x00027000u};
  u4 srd_tid = {(unsigned)b64, ((unsigned)(b64 >> 32) & 0xffffu) | (16u << 16), 0x7fffffffu, 0x00007000u | (1u << 23)};
  for (int i = 0; i < 4; ++i) {
    srd_plain[i] = __builtin_amdgcn_readfirstlane(srd_plain[i]);
    srd_tid[i] = __builtin_amdgcn_readfirstlane(srd_tid[i]);
  }
  h8 a[4], b[8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (lane + i + j));
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(0.002f * (lane - i + j));
  f4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]), "+v"(b[i + 4]));
  const unsigned voff = lane * 16;
  const char* gaddr = base + lane * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    const unsigned soff = (unsigned)((r & 3) * 16384 + wave * 4096);   // uniform: the piece's byte offset in the source
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[g], acc[g][j], 0, 0, 0);
      if (g < K) {
        const unsigned ldsb = (unsigned)(wave * 16384 + (g & 3) * 1024 + ((r & 3) * 4096));
        if constexpr (FORM == 1) {
          const char* ga = gaddr + soff + (g & 3) * 1024;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ga), "s"(ldsb) : "memory");
        } else if constexpr (FORM == 2) {
          const unsigned so = soff + (g & 3) * 1024;
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(srd_plain), "s"(ldsb), "s"(so) : "memory");
        } else if constexpr (FORM == 3) {
          const unsigned so = soff + (g & 3) * 1024;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, %2 lds" ::"s"(srd_tid), "s"(ldsb), "s"(so) : "memory");
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (FORM != 0 && (r & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  // landed bytes of the last repetition's piece 0 of this wave: word `lane * 4` must equal the source word
  if (FORM != 0 && K > 0) {
    const int r = reps - 1;
    const unsigned ldsb = (unsigned)(wave * 16384 + ((r & 3) * 4096));
    const unsigned got = *(const unsigned*)(smem + ldsb + lane * 16);
    const unsigned want = *(const unsigned*)(base + (r & 3) * 16384 + wave * 4096 + lane * 16);
    if (got != want) atomicAdd(chk, 1u);
  }
}

template <int FORM, int K>
static void run(const char* src, float* out, unsigned long long* cyc, unsigned* chk, int reps, const char* name) {
  hipFuncSetAttribute((const void*)probe<FORM, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipMemset(chk, 0, 4);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<FORM, K>), dim3(256), dim3(256), 65536, 0, src, out, cyc, chk, reps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  unsigned bad = 0;
  hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
  hipMemcpy(&bad, chk, 4, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-44s K=%d  %8.1f cycles / 32 MFMA (floor 512)   mismatching lanes %u\n", name, K, s / 1024 / reps, bad);
  fflush(stdout);
}

int main() {
  char* src; float* out; unsigned long long* cyc; unsigned* chk;
  hipMalloc(&src, 2 * 64 * 65536);      // (twice the bytes used: slack behind the last block)
  hipMemset(src, 0, 2 * 64 * 65536); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8192); hipMalloc(&chk, 4);
  std::vector<unsigned> h(64 * 65536 / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int reps = 2000;
  run<0, 0>(src, out, cyc, chk, reps, "no DMA");
  run<1, 2>(src, out, cyc, chk, reps, "global_load_lds (64-bit lane address)");
  run<1, 4>(src, out, cyc, chk, reps, "global_load_lds (64-bit lane address)");
  run<2, 2>(src, out, cyc, chk, reps, "buffer_load offen lds (32-bit lane offset)");
  run<2, 4>(src, out, cyc, chk, reps, "buffer_load offen lds (32-bit lane offset)");
  run<3, 2>(src, out, cyc, chk, reps, "buffer_load off lds, ADD_TID descriptor");
  run<3, 4>(src, out, cyc, chk, reps, "buffer_load off lds, ADD_TID descriptor");
  run<3, 8>(src, out, cyc, chk, reps, "buffer_load off lds, ADD_TID descriptor");
  run<1, 8>(src, out, cyc, chk, reps, "global_load_lds (64-bit lane address)");
  return 0;
}
