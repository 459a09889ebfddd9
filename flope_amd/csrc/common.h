// Shared device-side types and launch-parameter structs for the flower-pose
// hot path (gfx950 only).  Activation tensors live in HBM as zero-bordered
// NHWC ("padded NHWC"): [B][H+2][W+2][C] with a one-pixel ring of zeros that is
// written once at flope_create and never touched again, so a 3x3 / pad-1
// convolution is pure address arithmetic -- no bounds checks in any inner loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// pack two floats into one 32-bit word of two T (low half = a)
// (a 2-vector conversion compiles to ONE v_cvt_pk_{f16,bf16}_f32 on gfx950, round-to-nearest-even like the scalar
// cast; two scalar casts cost two converts and a v_perm -- and every vector instruction is paid in MFMA issue time)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef T t2 __attribute__((ext_vector_type(2)));
  const f32x2_t f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, t2));
}
template <typename T> __device__ __forceinline__ float unpack_lo(unsigned w) {
  return to_f32<T>(__builtin_bit_cast(T, (unsigned short)(w & 0xffffu)));
}
template <typename T> __device__ __forceinline__ float unpack_hi(unsigned w) {
  return to_f32<T>(__builtin_bit_cast(T, (unsigned short)(w >> 16)));
}

// Packed 16-bit helpers on raw f16 / bf16 bit patterns (two values per 32-bit word):
//   ReLU  = signed-integer max with 0  (a negative float has its sign bit set, -0.0 becomes +0.0)
//   max of NON-NEGATIVE values = unsigned-integer max (IEEE ordering of same-sign values is the bit ordering)
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_relu16(unsigned a) {
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a), z));
}
__device__ __forceinline__ unsigned pk_max16_nonneg(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

// maximum of packed pairs as SIGNED 16-bit integers: equals the float maximum of the two f16 / bf16 values whenever that maximum is
// non-negative (non-negative patterns order like integers and sit above every negative one); when both are negative it returns a
// negative value -- which a following ReLU turns into the same 0 the float maximum would give.  (stem: vertical pool before ReLU)
__device__ __forceinline__ unsigned pk_max16_signed(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b)));
}

// 16-bit store of an epilogue pair: optional ReLU, and for float16 saturation at +-65504 -- a residual stream that
// outgrows the float16 range would otherwise become inf here, NaN one layer later and a garbage rotation with no error
// (bfloat16 has float32's range and needs nothing).  One v_pk_min_f16 (+ v_pk_max_f16 without ReLU) per two channels.
// Branch-free (r03: written as `relu ? a : b` the compiler turned every packed pair of an epilogue into its own branch on the
// uniform flag -- 64 branches per wave-tile, 1,250 instructions for the epilogue of one conv_w4 tile):
//   ReLU or identity = signed-integer max with 0 or with INT16_MIN;  float16: then clamp to +-65504 (for the ReLU case the
//   lower clamp is a no-op on a non-negative value).  Same results, bit for bit, as the two-path form.
template <typename T> __device__ __forceinline__ unsigned pk_out16(unsigned w, bool relu) {
  const short zl = relu ? (short)0 : (short)-32768;
  const i16x2_t z = {zl, zl};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z));
}
template <> __device__ __forceinline__ unsigned pk_out16<f16_t>(unsigned w, bool relu) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t hi = {(_Float16)65504.f, (_Float16)65504.f}, lo = {(_Float16)-65504.f, (_Float16)-65504.f};
  const short zl = relu ? (short)0 : (short)-32768;
  const i16x2_t z = {zl, zl};
  const unsigned x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_elementwise_min(__builtin_bit_cast(h2_t, x), hi), lo));
}
// (r03: the same clamp as integer mins -- three instructions per pair instead of five -- changed no conv's time: the conv epilogues
// are store-bound; the stem, which is bound by its instruction stream, uses pk_relu16 below)

// ReLU known at compile time (stem): after the integer max the value is a non-negative 16-bit float pattern, and those order like
// integers, so the clamp to 65504 (+inf and NaN patterns 0x7C00..0x7FFF included: pk_out16's float min returns the number for a NaN)
// is an integer min with 0x7BFF -- two instructions per pair instead of the five of the float form (r03 stamps: the stem is bound
// by its own instruction stream).  Same results as pk_out16(w, true), bit for bit, for every input.
template <typename T> __device__ __forceinline__ unsigned pk_relu16(unsigned w) {
  const i16x2_t z = {(short)0, (short)0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z));
}
template <> __device__ __forceinline__ unsigned pk_relu16<f16_t>(unsigned w) {
  const i16x2_t z = {(short)0, (short)0}, hi = {(short)0x7BFF, (short)0x7BFF};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z), hi));
}

// Bijective XCD-aware block remap (blocks b and b+8 share an XCD under the
// observed round-robin dispatch; speed only, never correctness): each XCD gets a
// contiguous run of logical tile ids so tiles that share halo rows / weight
// panels hit the same L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// n / d for 0 <= n < 2^31 as one 64-bit multiply and a shift: M = ceil(2^(31+l) / d), l = ceil(log2 d),
// q = (n * M) >> (31 + l)   (exact: the error term n*e / (d*2^(31+l)) is < 2^-l <= 1/d)
__device__ __forceinline__ int fastdiv(int n, unsigned mg, unsigned sh) {
  return (int)(((unsigned long long)(unsigned)n * mg) >> sh);
}

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution on MFMA (conv_mfma.hip)
struct ConvP {
  const void* in;      // padded NHWC  [B][Hip][Wip][Cin]
  void* out;           // padded NHWC  [B][Hop][Wop][Cout]
  const void* res;     // optional residual, same shape as out (nullptr = none)
  const void* w;       // packed weights [ntile][step][BN rows][64 k] in LDS image order
  const float* bias;   // [Cout] folded BN shift
  int B, Hip, Wip, Cin;
  int Ho, Wo, Hop, Wop, Cout;
  int stride;          // 1 or 2
  int ntaps;           // 9 (3x3, pad 1) or 1 (1x1, pad 0: centre tap of the padded window)
  int M;               // B*Ho*Wo
  int relu;
  int nchunks;         // Cin / 64
  int mtiles, ntiles;
  int per_image;       // 1: M tiles never straddle two images (tiles_per_image below)
  int tiles_per_image;
  int patch_rows_max;  // patch mode: LDS rows reserved
  int dbg;             // timing experiments only: bit0 skip weight staging, bit1 skip pixel staging, bit2 skip MFMA
  unsigned mg_hw, sh_hw, mg_w, sh_w;   // n / (Ho*Wo) and n / Wo as multiply-shift (host: fastdiv_magic), n < 2^31
  int total_tiles;     // conv_stag: persistent grid walks tiles blockIdx.x + k*gridDim.x < total_tiles
  // conv_stag, folded downsample (layerX.0.conv2): out += W_ds . ds_in(2*ho, 2*wo) -- the block's 1x1 stride-2 shortcut
  const void* ds_in;   // padded NHWC [B][ds_Hip][ds_Wip][ds_Cin] (the block input), nullptr = none
  const void* ds_w;    // [ntile][ds_Cin/32 half-chunks][128 rows][32 k] conv_stag image
  int ds_Hip, ds_Wip, ds_Cin;
  // conv_stag split-K (small batches: fewer tiles than CUs): workgroup blockIdx.x handles tile blockIdx.x / ksplit and
  // the (blockIdx.x % ksplit)-th share of the K loop, and writes raw fp32 partial sums to split_ws[ks][M][Cout];
  // conv_split_finalize adds them up with bias / residual / ReLU.  ksplit <= 1: off.
  int ksplit;
  float* split_ws;
  // conv_stag row bands on maps wider than 64 columns: tiles are 8 rows x one of nseg 64-column segments
  // (tiles_per_image = Ho / 8 * nseg); 0 / 1 = full-width bands
  int nseg;
  // conv_stag flat tiles: 1 = LDS patch rows at pitch W + 4 with the slot swizzle taken from i * W + c (conflict-free fragment
  // reads across row wraps, r03); 0 = the r02 image (natural pitch)
  int skew;
  // conv_stag flat 256 x 128 tiles with a residual input, one tile per workgroup: the residual arrives by LDS-DMA in the slots the
  // last body's look-ahead DMAs leave unused (conv_stag.hip) instead of register loads in front of the first tile
  int res_lds;
  int prio;            // conv_stag: 1 = s_setprio 1 for waves 4..7, 2 = for waves 0..3, 0 = none
  int dbg_lds_off;     // diagnostic builds, dbg & 128: LDS byte offset of the stamp area (behind the kernel's own image)
  unsigned mg_pitch, sh_pitch;   // conv_w4: n / (Wip + 2) as multiply-shift (the skewed patch image's row pitch in pixels)
  int cw_imgs;         // conv_w4 class walk (persistent workgroups): images a workgroup advances per tile of its walk (0: one tile per workgroup)
};

// Stem: 7x7 s2 p3 conv, Cin 3 (stored as 4) -> 64, + folded BN + ReLU
struct StemP {
  const void* in;      // [B][Hip][Wip][4]   3-pixel zero border (+ right/bottom slack)
  void* out;           // padded NHWC [B][Ho+2][Wo+2][64]
  const void* w;       // packed [7 ky][64 rows][32 k] LDS image order
  const float* bias;   // [64]
  int B, Hip, Wip;
  int Ho, Wo;
  int tiles_per_image;
  int patch_rows_max;
};

struct PoolP {         // 3x3 s2 p1 max-pool on padded NHWC
  const void* in; void* out;
  int B, Hip, Wip, C, Ho, Wo;
};

struct NaiveConvP {    // strict fp32 direct convolution on padded NHWC float tensors
  const float* in; float* out; const float* res; const float* w; const float* bias;
  int B, Hip, Wip, Cin_stored, Cin, Ho, Wo, Hop, Wop, Cout;
  int KH, KW, stride, in_off;  // input pixel = (ho*stride + ky + in_off, wo*stride + kx + in_off)
  int relu;
};

// ---------------------------------------------------------------------------
// Shared MFMA epilogue: one lane = one output pixel x (4*NT) consecutive channels.
// acc + bias (+ residual) (ReLU) -> 16-bit, written as 16-byte NHWC stores into the
// interior of the zero-bordered output tensor.
// ADD_BIAS = false: the accumulators were initialised to the bias (conv_mfma), nothing to add here.
template <typename T, int NT, bool ADD_BIAS = true>
__device__ __forceinline__ void conv_epilogue_at(const ConvP& p, const f32x4 (&acc)[NT], int b, int ho, int wo,
                                                 int cb, const float (&bias)[NT * 4]) {
  const size_t pix = ((size_t)b * p.Hop + ho + 1) * p.Wop + wo + 1;
  const size_t off = (pix * p.Cout + cb) * 2;
  float v[NT * 4];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) v[ct * 4 + q] = ADD_BIAS ? acc[ct][q] + bias[ct * 4 + q] : acc[ct][q];
  if (p.res) {
    const char* rp = (const char*)p.res + off;
#pragma unroll
    for (int c = 0; c < NT / 2; ++c) {
      const u32x4 rv = *(const u32x4*)(rp + c * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[c * 8 + q * 2] += unpack_lo<T>(rv[q]);
        v[c * 8 + q * 2 + 1] += unpack_hi<T>(rv[q]);
      }
    }
  }
  char* op = (char*)p.out + off;
#pragma unroll
  for (int c = 0; c < NT / 2; ++c) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned w = pack2<T>(v[c * 8 + q * 2], v[c * 8 + q * 2 + 1]);
      o[q] = pk_out16<T>(w, p.relu);                       // ReLU on the packed pair (sign test is rounding-invariant)
    }
    *(u32x4*)(op + c * 16) = o;
  }
}

// flat output-pixel index m -> (image, row, column) by multiply-shift (p.mg_* from the host's fastdiv_magic; a runtime
// '/' costs ~40 vector instructions, and vector instructions share the SIMD's issue slots with the MFMAs)
template <typename T, int NT, bool ADD_BIAS = true>
__device__ __forceinline__ void conv_epilogue_px(const ConvP& p, const f32x4 (&acc)[NT], int m, bool valid,
                                                 int cb, const float (&bias)[NT * 4], int HoWo) {
  if (!valid) return;
  const int b = fastdiv(m, p.mg_hw, p.sh_hw);
  const int r = m - b * HoWo;
  const int ho = fastdiv(r, p.mg_w, p.sh_w);
  conv_epilogue_at<T, NT, ADD_BIAS>(p, acc, b, ho, r - ho * p.Wo, cb, bias);
}

