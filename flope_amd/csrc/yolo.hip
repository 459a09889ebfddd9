// Kernels of the YOLO11-seg detector front end (reference sunflower/predictor/fast_pose_predictor.py:36,44-57:
// `self.yolo(image)` + the mask / box post-processing of get_bbox_mask; network arithmetic = ultralytics 8.3.27,
// not vendored by the reference -- restated from the published algorithm, PARITY UNPINNED against ultralytics).
//
//   yconv_kernel     Conv2d (1x1 / 3x3, stride 1 / 2) + folded BatchNorm + SiLU (+ residual) as an implicit GEMM on
//                    v_mfma_f32_16x16x32_{f16,bf16}: weights are the A operand, 16 pixels the B operand, so a lane
//                    ends with one pixel x 4*NT consecutive channels and the epilogue (bias, SiLU = x * sigmoid(x),
//                    residual, 16-bit pack) stays in registers.  Operand fragments are 16-byte global loads
//                    (pixels: 8 consecutive channels of one tap; weights: 8 consecutive k of one output row) -- this
//                    network is 100+ small layers at one frame, i.e. launch- and latency-bound, not MFMA-bound.
//                    Also: plain Conv2d with bias (float32 rows of the prediction tensor) and the 2x2 stride-2
//                    ConvTranspose2d of the Proto block (a 1x1 GEMM with a pixel-shuffle store).
//   ydw / ypool / yup / yattn   depthwise 3x3, SPPF's 5x5 max-pool, nearest 2x upsample, C2PSA attention
//   yletter          LetterBox (cv2 INTER_LINEAR 8-bit arithmetic) + BGR->RGB + /255
//   ydecode / ynms   Detect._inference (DFL expectation, dist2bbox) and ops.non_max_suppression
//   ymask_*          ops.process_mask + the sum/clip/x255 of get_bbox_mask
#include "common.h"
#include "yolo.h"
#include "host_pack.h"

#include <algorithm>

namespace {

// SiLU on the 16-bit paths: v * rcp(1 + exp2(-v log2 e)) -- v_exp_f32 and v_rcp_f32 are good to ~1 ulp of float32, far inside the
// rounding of the 16-bit store behind it, and 5 instructions instead of the ~16 of an IEEE division (r04 stamps: the activation math
// was ~2 k of a small-map conv's 5.3 k in-kernel cycles... per value and lane, with one wave per SIMD).  (The strict float32 mode has its own kernels.)
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

template <typename T> __device__ __forceinline__ float ld16(const void* p) { return to_f32<T>(*(const T*)p); }

// ---- epilogue: this lane = one pixel (flat index mm[pt]) of each tile x channels ch0 .. ch0 + 4 NT - 1
template <typename T, int NT, int MTW>
__device__ __forceinline__ void yconv_epilogue(const YConvP& p, const int nblk, const int g, const int (&mm)[MTW], const bool (&pv)[MTW],
                                               const f32x4 (&acc)[MTW][NT]) {
  constexpr int CB = 16 * NT;
  const int ch0 = nblk * CB + g * 4 * NT;
#pragma unroll
  for (int pt = 0; pt < MTW; ++pt) {
    if (!pv[pt]) continue;
    const int m = mm[pt];
    float v[4 * NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a = acc[pt][ct][q];
        v[ct * 4 + q] = p.act ? silu(a) : a;
      }
    if (p.out_mode == 1) {                             // float32 prediction rows (Detect / Segment heads)
      float* o = (float*)p.out + (size_t)m * p.ldo;
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout) o[ch0 + i] = v[i];
      continue;
    }
    size_t opix;
    int och = ch0;
    if (p.out_mode == 2) {                             // ConvTranspose2d 2x2 s2: row block -> (dy, dx) quadrant
      const int quad = (nblk * CB) / p.dc;
      const int oy = m / p.Wo, ox = m - oy * p.Wo;
      opix = (size_t)(2 * oy + (quad >> 1)) * (2 * p.Wo) + 2 * ox + (quad & 1);
      och = ch0 - quad * p.dc;
    } else {
      opix = (size_t)m;
    }
    char* o = (char*)p.out + (opix * p.ldo + och) * 2;
    const bool full = p.out_mode == 2 || ch0 + 4 * NT <= p.Cout;
    if (full) {
      if (p.res) {
        const char* r = (const char*)p.res + ((size_t)m * p.ldr + ch0) * 2;
        if (NT >= 2 && (p.ldr & 7) == 0 && ((size_t)p.res & 15) == 0 && ((nblk * CB) & 7) == 0) {     // 16-byte aligned rows (wave-uniform): vector loads
#pragma unroll
          for (int i = 0; i < NT / 2; ++i) {
            const u32x4 rv = *(const u32x4*)(r + i * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[i * 8 + q * 2] += unpack_lo<T>(rv[q]); v[i * 8 + q * 2 + 1] += unpack_hi<T>(rv[q]); }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4 * NT; ++i) v[i] += ld16<T>(r + i * 2);
        }
      }
      unsigned w[2 * NT];
#pragma unroll
      for (int i = 0; i < 2 * NT; ++i) w[i] = pk_out16<T>(pack2<T>(v[2 * i], v[2 * i + 1]), false);
      if constexpr (NT == 1) *(u32x2*)o = u32x2{w[0], w[1]};
      else {
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) *(u32x4*)(o + i * 16) = u32x4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout) {
          float x = v[i];
          if (p.res) x += ld16<T>((const char*)p.res + ((size_t)m * p.ldr + ch0 + i) * 2);
          *(T*)(o + i * 2) = from_f32<T>(x);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SPLITK: the four waves of a workgroup share ONE 32-pixel tile and each walks a quarter of the K steps; partial sums
// are added in wave order through LDS (deterministic).  Used for the small maps (<= 16 k pixels), where a layer is a
// handful of workgroups and its serial K loop -- one L2 / Infinity-Cache round trip per step -- IS its latency.
// Every variant keeps PD = 4 K steps of operand fragments in flight (the first version prefetched one step: 0.7 us per
// step on the 23 x 40 maps, 50 us for a 72-step layer).
// `red`: the workgroup's LDS scratch for the SPLITK combine, 3 * 2 * NT * 4 * 64 floats (24 KB at NT = 4).
#ifdef FLOPE_STAG_DBG
// diagnostic build: shader-clock stamps of workgroup (0, 0), wave 0 of every yconv_body launch, 8 words per launch in launch order
// {entry, loads issued, K loop done, split-K combined, epilogue issued, realtime entry, realtime exit, M}; flope_ydbg_read
__device__ unsigned long long g_ydbg[8 * 512];
__device__ unsigned g_ydbg_n;
#define YDBG_STAMP(i_) do { __builtin_amdgcn_sched_barrier(0); yst[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define YDBG_STAMP(i_) do {} while (0)
#endif

// One (pixel tile, channel tile) unit of an epilogue: this lane's 4 channels ch .. ch + 3 of pixel m.  Element for element the
// arithmetic of yconv_epilogue (activation, shortcut added behind it, one rounding): split-K hands a workgroup's units out to
// its four waves (yconv_body) instead of leaving all of them to wave 0.
template <typename T, int NT>
__device__ __forceinline__ void yconv_epilogue_unit(const YConvP& p, const int nblk, const int g, const int m, const int ct, const f32x4 a) {
  constexpr int CB = 16 * NT;
  const int ch = nblk * CB + g * 4 * NT + ct * 4;
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = p.act ? silu(a[q]) : a[q];
  if (p.out_mode == 1) {
    float* o = (float*)p.out + (size_t)m * p.ldo;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ch + q < p.Cout) o[ch + q] = v[q];
    return;
  }
  size_t opix;
  int och = ch;
  if (p.out_mode == 2) {
    const int quad = (nblk * CB) / p.dc;
    const int oy = fastdiv(m, p.wo_mg, p.wo_sh), ox = m - oy * p.Wo;
    opix = (size_t)(2 * oy + (quad >> 1)) * (2 * p.Wo) + 2 * ox + (quad & 1);
    och = ch - quad * p.dc;
  } else {
    opix = (size_t)m;
  }
  char* o = (char*)p.out + (opix * p.ldo + och) * 2;
  // (whole-lane-run criterion of yconv_epilogue: the run's last channels decide, so that both forms take the same path per element)
  const bool full = p.out_mode == 2 || nblk * CB + g * 4 * NT + 4 * NT <= p.Cout;
  if (full) {
    if (p.res) {
      const char* r = (const char*)p.res + ((size_t)m * p.ldr + ch) * 2;
      if ((p.ldr & 3) == 0 && ((size_t)p.res & 7) == 0) {
        const u32x2 rv = *(const u32x2*)r;
        v[0] += unpack_lo<T>(rv[0]); v[1] += unpack_hi<T>(rv[0]); v[2] += unpack_lo<T>(rv[1]); v[3] += unpack_hi<T>(rv[1]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += ld16<T>(r + q * 2);
      }
    }
    const unsigned w0 = pk_out16<T>(pack2<T>(v[0], v[1]), false), w1 = pk_out16<T>(pack2<T>(v[2], v[3]), false);
    if ((p.ldo & 3) == 0 && ((size_t)p.out & 7) == 0 && (och & 3) == 0) *(u32x2*)o = u32x2{w0, w1};
    else { *(unsigned short*)(o) = (unsigned short)(w0 & 0xffffu); *(unsigned short*)(o + 2) = (unsigned short)(w0 >> 16);
           *(unsigned short*)(o + 4) = (unsigned short)(w1 & 0xffffu); *(unsigned short*)(o + 6) = (unsigned short)(w1 >> 16); }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ch + q < p.Cout) {
        float x = v[q];
        if (p.res) x += ld16<T>((const char*)p.res + ((size_t)m * p.ldr + ch + q) * 2);
        *(T*)(o + q * 2) = from_f32<T>(x);
      }
  }
}

// (m_base: first pixel of the wave's two pixel tiles; nblk: channel block -- yconv_body derives them from the workgroup index, the
// chain kernel hands every wave its own channel block of one shared pixel tile)
template <typename T, int NT, bool K3, bool SPLITK>
__device__ __forceinline__ void yconv_body_at(const YConvP& p, const int m_base, const int nblk, float* const red, const int bx, const int by) {
  typedef typename Elem<T>::frag frag;
#ifdef FLOPE_STAG_DBG
  unsigned long long yst[5] = {0, 0, 0, 0, 0};
  const unsigned long long yrt0 = __builtin_amdgcn_s_memrealtime();
  YDBG_STAMP(0);
#endif
  constexpr int MTW = 2, CB = 16 * NT, PD = 4;       // pixel tiles per wave, channels per block, pipeline depth
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c16 = lane & 15;
  const int pad = K3 ? 1 : 0;
  // r04 (in-kernel stamps, tools/clock_probe_yolo.py: a 1x1 conv on the 23 x 40 map spent 4 k cycles before its first load and 4 k in
  // the epilogue of ONE wave, of 11.5 k in all): the bias goes into the first MFMA as its C operand (no wait for it in front of the
  // operand loads), the pixel decode is a multiply-shift, only the K steps a wave has are prefetched, and the split-K epilogue is
  // spread over the four waves.
  f32x4 bv[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    bv[ct] = *(const f32x4*)(p.bias + nblk * CB + ct * 16 + g * 4);              // bias is stored in packed-row order
    if (SPLITK && wave != 0) bv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int iy0[MTW], ix0[MTW];
  bool pv[MTW];
#pragma unroll
  for (int pt = 0; pt < MTW; ++pt) {
    const int m = m_base + pt * 16 + c16;
    pv[pt] = m < p.M;
    const int mm = pv[pt] ? m : p.M - 1;
    const int oy = fastdiv(mm, p.wo_mg, p.wo_sh), ox = mm - oy * p.Wo;
    iy0[pt] = oy * p.stride - pad;
    ix0[pt] = ox * p.stride - pad;
  }
  // weights are stored in fragment order [channel block][k step][channel tile][lane][16 B]: one wave-load = one contiguous
  // KiB (8 cache lines; row-major rows put its 64 lanes on 64 different lines and the L1's tag rate became the bound)
  const char* const wrow = (const char*)p.w + (size_t)nblk * p.ksteps * (NT * 1024) + lane * 16;
  const char* const in = (const char*)p.in;
  const char* const zero = (const char*)p.zero;
  const int ks0 = SPLITK ? (wave * p.ksteps) >> 2 : 0, ks1 = SPLITK ? ((wave + 1) * p.ksteps) >> 2 : p.ksteps;
  f32x4 acc[MTW][NT];
  auto load_step = [&](int ks, frag (&wf)[NT], frag (&xf)[MTW]) {
    const int kg = ks * 4 + g;
    int c8, ky = 0, kx = 0;
    bool tapok;
    if (K3) {
      const int tap = fastdiv(kg, p.cg_mg, p.cg_sh);
      c8 = kg - tap * p.cg;
      ky = (tap * 11) >> 5;
      kx = tap - 3 * ky;
      tapok = tap < 9;
    } else {
      c8 = kg;
      tapok = kg < p.cg;
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(wrow + (size_t)(ks * NT + ct) * 1024);
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt) {
      const int iy = iy0[pt] + ky, ix = ix0[pt] + kx;
      const bool ok = pv[pt] && tapok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      const char* a = ok ? in + ((size_t)(iy * p.Wi + ix) * p.ldi + c8 * 8) * 2 : zero;
      xf[pt] = *(const frag*)a;
    }
  };
  // software pipeline, PD steps deep; every condition below is wave-uniform
  frag wf[PD][NT], xf[PD][MTW];
  if (ks0 < ks1) {
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_)
      if (s_ == 0 || ks0 + s_ < ks1) load_step(ks0 + s_, wf[s_], xf[s_]);      // (a wave of a split-K 1x1 conv often has two steps)
    YDBG_STAMP(1);
    // first step: C = bias
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[0][ct], xf[0][pt], bv[ct]);
    if (ks0 + PD < ks1) load_step(ks0 + PD, wf[0], xf[0]);
#pragma unroll
    for (int s_ = 1; s_ < PD; ++s_) {
      if (ks0 + s_ < ks1) {
#pragma unroll
        for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[s_][ct], xf[s_][pt], acc[pt][ct]);
      }
      if (ks0 + s_ + PD < ks1) load_step(ks0 + s_ + PD, wf[s_], xf[s_]);
    }
    for (int ks = ks0 + PD; ks < ks1; ks += PD) {
#pragma unroll
      for (int s_ = 0; s_ < PD; ++s_) {
        if (ks + s_ < ks1) {
#pragma unroll
          for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[s_][ct], xf[s_][pt], acc[pt][ct]);
        }
        if (ks + s_ + PD < ks1) load_step(ks + s_ + PD, wf[s_], xf[s_]);
      }
    }
  } else {
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = bv[ct];
  }
#ifdef FLOPE_STAG_DBG
  { float keep_ = acc[0][0][0]; asm volatile("" : "+v"(keep_)); }
  YDBG_STAMP(2);
#endif
  int mm[MTW];
#pragma unroll
  for (int pt = 0; pt < MTW; ++pt) mm[pt] = m_base + pt * 16 + c16;
  if constexpr (SPLITK) {
    // unit u = pt * NT + ct belongs to wave u & 3 (NT = 4: channel tile w of both pixel tiles; NT = 2: one unit each; NT = 1: waves 0
    // and 1).  A wave parks the units it does not own ([source index among the three others][unit][q][lane]), the owner adds the
    // four partial sums in WAVE order -- ((P0 + P1) + P2) + P3, the order of the one-wave combine -- and runs the unit's epilogue.
    constexpr int RW = MTW * NT * 4;
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const int own = (pt * NT + ct) & 3;
        if (own != wave) {
          const int si = wave < own ? wave : wave - 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) red[(si * RW + (pt * NT + ct) * 4 + q) * 64 + lane] = acc[pt][ct][q];
        }
      }
    __syncthreads();
    YDBG_STAMP(3);
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        if (((pt * NT + ct) & 3) != wave || !pv[pt]) continue;
        f32x4 t;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          f32x4 part;
          if (s == wave) part = acc[pt][ct];
          else {
            const int si = s < wave ? s : s - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) part[q] = red[(si * RW + (pt * NT + ct) * 4 + q) * 64 + lane];
          }
          if (s == 0) t = part;
          else { t[0] += part[0]; t[1] += part[1]; t[2] += part[2]; t[3] += part[3]; }
        }
        yconv_epilogue_unit<T, NT>(p, nblk, g, mm[pt], ct, t);
      }
  } else {
    YDBG_STAMP(3);
    yconv_epilogue<T, NT, MTW>(p, nblk, g, mm, pv, acc);
  }
#ifdef FLOPE_STAG_DBG
  YDBG_STAMP(4);
  if (bx == 0 && by == 0 && threadIdx.x == 0) {
    const unsigned slot = atomicAdd(&g_ydbg_n, 1u) & 511u;
    for (int i = 0; i < 5; ++i) g_ydbg[slot * 8 + i] = yst[i];
    g_ydbg[slot * 8 + 5] = yrt0; g_ydbg[slot * 8 + 6] = __builtin_amdgcn_s_memrealtime(); g_ydbg[slot * 8 + 7] = (unsigned long long)p.M | ((unsigned long long)p.ksteps << 32);
  }
#endif
}

template <typename T, int NT, bool K3, bool SPLITK>
__device__ __forceinline__ void yconv_body(const YConvP& p, const int bx, const int by, float* const red) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m_base = SPLITK ? bx * 32 : (bx * 4 + wave) * 32;
  if (!SPLITK && m_base >= p.M) return;
  yconv_body_at<T, NT, K3, SPLITK>(p, m_base, by, red, bx, by);
}

template <typename T, int NT, int MTW>
__device__ __forceinline__ void ylds_kloop3w(const char* const wl, const int ksteps, const int cg, const unsigned cg_mg, const unsigned cg_sh,
                                             const char* const lds, const int pitch, const int PW, const int (&poff)[MTW],
                                             f32x4 (&acc)[MTW][NT]);

// ---------------------------------------------------------------------------------------------------------------
// Large maps (everything that is not SPLITK): one workgroup = an 8-row x 16-column tile of output pixels, wave w owns rows
// 2w and 2w+1.  The input patch ((7 s + k) x (15 s + k) pixels, all channels) is staged once into LDS with fully coalesced
// 16-byte loads (consecutive lanes = consecutive channels, then pixels) and the taps read their B fragments from there:
// fetched straight from global memory, the 16 pixels of a fragment load sit on 16 different cache lines (64 tag look-ups per
// wave-load), every input pixel is fetched nine times, and the L1's tag rate -- not bytes, not MFMA -- bounded these layers.
// Pixel pitch in LDS = Cin * 2 + 16 bytes: the 16 lanes of a ds_read_b128 group land on 16 different bank quads.
template <typename T, int NT, bool K3>
__device__ __forceinline__ void yconv_tile_body(const YConvP& p, const int bx, const int by, char* const lds) {
  typedef typename Elem<T>::frag frag;
  constexpr int MTW = 2, PD = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int ty = bx / p.tiles_x, tx = bx - ty * p.tiles_x;
  const int s = p.stride, pad = K3 ? 1 : 0, kk = K3 ? 3 : 1;
  const int PW = 15 * s + kk, PH = 7 * s + kk;
  const int iy00 = ty * 8 * s - pad, ix00 = tx * 16 * s - pad;
  const int pitch = p.Cin * 2 + 16;
  const int nblk = by;
  const char* const wrow = (const char*)p.w + (size_t)nblk * p.ksteps * (NT * 1024) + lane * 16;
  const int klast = p.ksteps - 1;
  if constexpr (K3) {
    // Long K (>= 8 steps): the layer's weights are cold for the one or two workgroups an XCD gets, and a K loop that fetches them
    // PD steps ahead pays a memory round trip every PD steps.  Stage the channel block's whole weight image into LDS together with
    // the patch (one round trip) and run the K loop on LDS operands only.
    if (p.wlds) {
      char* const wl = lds + PH * PW * pitch;
      const int nw = p.ksteps * NT * 64, total = PH * PW * p.cg;
      const char* const wsrc = (const char*)p.w + (size_t)nblk * p.ksteps * (NT * 1024);
      const char* const in = (const char*)p.in;
      for (int i0 = tid; i0 < nw + total; i0 += 8 * 256) {
        u32x4 v[8];
        int dst[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = i0 + j * 256;
          if (i < nw) { v[j] = *(const u32x4*)(wsrc + (size_t)i * 16); dst[j] = PH * PW * pitch + i * 16; continue; }
          const int ip = min(i - nw, total - 1);
          const int pp = fastdiv(ip, p.cg_mg, p.cg_sh), c8 = ip - pp * p.cg;
          const int py = fastdiv(pp, p.pw_mg, p.pw_sh), px = pp - py * PW;
          const int iy = iy00 + py, ix = ix00 + px;
          const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
          v[j] = ok ? *(const u32x4*)(in + ((size_t)(iy * p.Wi + ix) * p.ldi + c8 * 8) * 2) : u32x4{0u, 0u, 0u, 0u};
          dst[j] = i - nw < total ? pp * pitch + c8 * 16 : -1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (dst[j] >= 0) *(u32x4*)(lds + dst[j]) = v[j];
      }
      __syncthreads();
      int poff[MTW];
#pragma unroll
      for (int pt = 0; pt < MTW; ++pt) poff[pt] = ((wave * 2 + pt) * s) * PW + c16 * s;
      f32x4 accw[MTW][NT];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const f32x4 b = *(const f32x4*)(p.bias + nblk * (16 * NT) + ct * 16 + g * 4);
#pragma unroll
        for (int pt = 0; pt < MTW; ++pt) accw[pt][ct] = b;
      }
      ylds_kloop3w<T, NT, MTW>(wl, p.ksteps, p.cg, p.cg_mg, p.cg_sh, lds, pitch, PW, poff, accw);
      int mm[MTW];
      bool pv[MTW];
      const int ox = tx * 16 + c16;
#pragma unroll
      for (int pt = 0; pt < MTW; ++pt) {
        const int oy = ty * 8 + wave * 2 + pt;
        pv[pt] = oy < p.Ho && ox < p.Wo;
        mm[pt] = oy * p.Wo + ox;
      }
      yconv_epilogue<T, NT, MTW>(p, nblk, g, mm, pv, accw);
      return;
    }
  }
#ifdef FLOPE_STAG_DBG
  unsigned long long yst[5] = {0, 0, 0, 0, 0};
  const unsigned long long yrt0 = __builtin_amdgcn_s_memrealtime();
  YDBG_STAMP(0);
#endif
  frag wf[PD][NT];
#pragma unroll
  for (int s_ = 0; s_ < PD; ++s_)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) wf[s_][ct] = *(const frag*)(wrow + (size_t)(min(s_, klast) * NT + ct) * 1024);
  f32x4 acc[MTW][NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const f32x4 b = *(const f32x4*)(p.bias + nblk * (16 * NT) + ct * 16 + g * 4);
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt) acc[pt][ct] = b;
  }
  // ---- stage the patch (out-of-map pixels as zeros)
  {
    const int total = PH * PW * p.cg;
    const char* const in = (const char*)p.in;
    for (int i0 = tid; i0 < total; i0 += 4 * 256) {
      u32x4 v[4];
      int dst[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 256;
        const int pp = fastdiv(min(i, total - 1), p.cg_mg, p.cg_sh), c8 = min(i, total - 1) - pp * p.cg;
        const int py = fastdiv(pp, p.pw_mg, p.pw_sh), px = pp - py * PW;
        const int iy = iy00 + py, ix = ix00 + px;
        const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        v[j] = ok ? *(const u32x4*)(in + ((size_t)(iy * p.Wi + ix) * p.ldi + c8 * 8) * 2) : u32x4{0u, 0u, 0u, 0u};
        dst[j] = i < total ? pp * pitch + c8 * 16 : -1;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (dst[j] >= 0) *(u32x4*)(lds + dst[j]) = v[j];
    }
  }
  YDBG_STAMP(1);
  __syncthreads();
  YDBG_STAMP(2);
  // ---- K loop: B fragments from LDS (one step ahead), A fragments from global memory (PD steps ahead)
  const int prow = (wave * 2 * s) * PW + c16 * s;                       // patch pixel of (tile row 2w, column c16), tap (0, 0)
  auto xload = [&](int ks, frag (&xf)[MTW]) {
    const int kg = ks * 4 + g;
    int c8, ky = 0, kx = 0;
    bool tapok;
    if (K3) {
      const int tap = fastdiv(kg, p.cg_mg, p.cg_sh);
      c8 = kg - tap * p.cg;
      ky = (tap * 11) >> 5;
      kx = tap - 3 * ky;
      tapok = tap < 9;
    } else {
      c8 = kg;
      tapok = kg < p.cg;
    }
    const int off = tapok ? (prow + ky * PW + kx) * pitch + c8 * 16 : 0;
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt) {
      frag f = *(const frag*)(lds + off + pt * (s * PW) * pitch);
      if (!tapok) f = frag{};                                             // K padding: the weights are zero there, LDS is not
      xf[pt] = f;
    }
  };
  frag xa[MTW], xb[MTW];
  xload(0, xa);
  for (int ks = 0; ks < p.ksteps; ks += PD) {
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) {
      frag (&cur)[MTW] = (s_ & 1) ? xb : xa;
      frag (&nxt)[MTW] = (s_ & 1) ? xa : xb;
      if (ks + s_ < p.ksteps) {
        xload(min(ks + s_ + 1, klast), nxt);
#pragma unroll
        for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[s_][ct], cur[pt], acc[pt][ct]);
      }
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) wf[s_][ct] = *(const frag*)(wrow + (size_t)(min(ks + s_ + PD, klast) * NT + ct) * 1024);
    }
  }
  int mm[MTW];
  bool pv[MTW];
  const int ox = tx * 16 + c16;
#pragma unroll
  for (int pt = 0; pt < MTW; ++pt) {
    const int oy = ty * 8 + wave * 2 + pt;
    pv[pt] = oy < p.Ho && ox < p.Wo;
    mm[pt] = oy * p.Wo + ox;
  }
#ifdef FLOPE_STAG_DBG
  { float keep_ = acc[0][0][0]; asm volatile("" : "+v"(keep_)); }
  YDBG_STAMP(3);
#endif
  yconv_epilogue<T, NT, MTW>(p, nblk, g, mm, pv, acc);
#ifdef FLOPE_STAG_DBG
  YDBG_STAMP(4);
  if (bx == 0 && by == 0 && threadIdx.x == 0) {
    const unsigned slot = atomicAdd(&g_ydbg_n, 1u) & 511u;
    for (int i = 0; i < 5; ++i) g_ydbg[slot * 8 + i] = yst[i];
    g_ydbg[slot * 8 + 5] = yrt0; g_ydbg[slot * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    g_ydbg[slot * 8 + 7] = (unsigned long long)p.M | ((unsigned long long)p.ksteps << 32) | (1ull << 63);      // bit 63: tile path
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 K loop with the B fragments taken from an LDS image [pixel][channel] (pixel pitch `pitch` bytes, PW pixels per
// row): lane pixel offsets poff[pt] (in pixels, tap (0,0)); A fragments from global memory, PD steps ahead, wf preloaded.
template <typename T, int NT, int MTW>
__device__ __forceinline__ void ylds_kloop3(const char* const wrow, const int ksteps, const int cg, const unsigned cg_mg, const unsigned cg_sh,
                                            const char* const lds, const int pitch, const int PW, const int (&poff)[MTW],
                                            typename Elem<T>::frag (&wf)[4][NT], f32x4 (&acc)[MTW][NT]) {
  typedef typename Elem<T>::frag frag;
  constexpr int PD = 4;
  const int g = (threadIdx.x & 63) >> 4;
  const int klast = ksteps - 1;
  auto xload = [&](int ks, frag (&xf)[MTW]) {
    const int kg = ks * 4 + g;
    const int tap = fastdiv(kg, cg_mg, cg_sh), c8 = kg - tap * cg, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const bool tapok = tap < 9;
    const int off = tapok ? (ky * PW + kx) * pitch + c8 * 16 : 0;
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt) {
      frag f = *(const frag*)(lds + off + poff[pt] * pitch);
      if (!tapok) f = frag{};
      xf[pt] = f;
    }
  };
  frag xa[MTW], xb[MTW];
  xload(0, xa);
  for (int ks = 0; ks < ksteps; ks += PD) {
#pragma unroll
    for (int s_ = 0; s_ < PD; ++s_) {
      frag (&cur)[MTW] = (s_ & 1) ? xb : xa;
      frag (&nxt)[MTW] = (s_ & 1) ? xa : xb;
      if (ks + s_ < ksteps) {
        xload(min(ks + s_ + 1, klast), nxt);
#pragma unroll
        for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[s_][ct], cur[pt], acc[pt][ct]);
      }
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) wf[s_][ct] = *(const frag*)(wrow + (size_t)(min(ks + s_ + PD, klast) * NT + ct) * 1024);
    }
  }
}

// The same with BOTH operands in LDS: weights `wl` in fragment order ([k step][channel tile][lane][16 B]).
template <typename T, int NT, int MTW>
__device__ __forceinline__ void ylds_kloop3w(const char* const wl, const int ksteps, const int cg, const unsigned cg_mg, const unsigned cg_sh,
                                             const char* const lds, const int pitch, const int PW, const int (&poff)[MTW],
                                             f32x4 (&acc)[MTW][NT]) {
  typedef typename Elem<T>::frag frag;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int klast = ksteps - 1;
  auto xload = [&](int ks, frag (&xf)[MTW], frag (&wf)[NT]) {
    const int kg = ks * 4 + g;
    const int tap = fastdiv(kg, cg_mg, cg_sh), c8 = kg - tap * cg, ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const bool tapok = tap < 9;
    const int off = tapok ? (ky * PW + kx) * pitch + c8 * 16 : 0;
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt) {
      frag f = *(const frag*)(lds + off + poff[pt] * pitch);
      if (!tapok) f = frag{};
      xf[pt] = f;
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(wl + ((ks * NT + ct) * 64 + lane) * 16);
  };
  frag xa[MTW], xb[MTW], wa[NT], wb[NT];
  xload(0, xa, wa);
  for (int ks = 0; ks < ksteps; ks += 2) {
    xload(min(ks + 1, klast), xb, wb);
#pragma unroll
    for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wa[ct], xa[pt], acc[pt][ct]);
    if (ks + 1 < ksteps) {
      xload(min(ks + 2, klast), xa, wa);
#pragma unroll
      for (int pt = 0; pt < MTW; ++pt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wb[ct], xb[pt], acc[pt][ct]);
    }
  }
}

// Bottleneck (ultralytics nn/modules/block.py): out = cv2(cv1(x)) (+ x), both 3x3 stride 1 with <= 64 output channels.  One
// workgroup = an 8 x 16 output tile: the 12 x 20 input patch goes to LDS, cv1 is evaluated on the 10 x 18 pixels cv2 needs
// (pixels outside the map are cv2's zero padding, not cv1 outputs), its SiLU output is parked in LDS as 16-bit NHWC and cv2
// reads its taps from there -- one launch and no round trip of the intermediate map through memory.  The weights of a layer are
// cold for the handful of workgroups an XCD gets (every frame walks ~130 MB between two uses), so a K loop that fetches them
// step by step pays a memory round trip every few steps (first version: 0.4 us per K step, slower than the two launches it
// replaced); here cv1's whole weight image is staged into LDS together with the patch (ONE round trip) and cv2's travels in
// registers while cv1 computes.  Same MFMA sequence per output as the two separate launches whenever those do not split K (the
// split-K path adds its partial sums in another order).
template <typename T, int NT1, int NT2, int TH>
__device__ __forceinline__ void ybneck_body(const YBneckP& p, const int bx, char* const lds) {
  typedef typename Elem<T>::frag frag;
  // TH = 8 or 4 output rows per workgroup (4 on the small maps: twice the workgroups, shorter serial chains)
  constexpr int PW1 = 20, PH1 = TH + 4, PW2 = 18, NQ = (TH + 2) * 18, MT1 = (NQ + 63) / 64, MT2 = TH / 4;
  constexpr int WV1 = (18 * NT1 * 64 + 255) / 256, WV2 = (18 * NT2 * 64 + 255) / 256, PV = (PH1 * PW1 * 8 + 255) / 256;   // 16-byte chunks per thread (ksteps <= 18, Cin <= 64)
  const YConvP& c1 = p.c1;
  const YConvP& c2 = p.c2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  const int ty = bx / p.tiles_x, tx = bx - ty * p.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * 16;
  const int pitch1 = c1.Cin * 2 + 16, pitch2 = c2.Cin * 2 + 16;
  char* const mid = lds + PH1 * PW1 * pitch1;
  char* const wl = mid + NQ * pitch2;                       // weight image: cv1's, then cv2's
  const int nw1 = c1.ksteps * NT1 * 64, nw2 = c2.ksteps * NT2 * 64;
  {
    u32x4 wv[WV1], pv_[PV];
    const int total = PH1 * PW1 * c1.cg;
    const char* const in = (const char*)c1.in;
    int dst[PV];
#pragma unroll
    for (int j = 0; j < WV1; ++j) { const int i = tid + j * 256; if (i < nw1) wv[j] = *(const u32x4*)((const char*)c1.w + (size_t)i * 16); }
#pragma unroll
    for (int j = 0; j < PV; ++j) {
      const int i = tid + j * 256, ic = min(i, total - 1);
      const int pp = fastdiv(ic, c1.cg_mg, c1.cg_sh), c8 = ic - pp * c1.cg;
      const int py = pp / PW1, px = pp - py * PW1;
      const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
      const bool ok = i < total && (unsigned)iy < (unsigned)c1.Hi && (unsigned)ix < (unsigned)c1.Wi;
      pv_[j] = ok ? *(const u32x4*)(in + ((size_t)(iy * c1.Wi + ix) * c1.ldi + c8 * 8) * 2) : u32x4{0u, 0u, 0u, 0u};
      dst[j] = i < total ? pp * pitch1 + c8 * 16 : -1;
    }
#pragma unroll
    for (int j = 0; j < PV; ++j)
      if (dst[j] >= 0) *(u32x4*)(lds + dst[j]) = pv_[j];
#pragma unroll
    for (int j = 0; j < WV1; ++j) { const int i = tid + j * 256; if (i < nw1) *(u32x4*)(wl + (size_t)i * 16) = wv[j]; }
  }
  u32x4 wv2[WV2];                                           // cv2's weights: in flight while cv1 computes
#pragma unroll
  for (int j = 0; j < WV2; ++j) { const int i = tid + j * 256; if (i < nw2) wv2[j] = *(const u32x4*)((const char*)c2.w + (size_t)i * 16); }
  __syncthreads();
  // ---- cv1 on the 10 x 18 intermediate pixels: wave w owns pixel tiles w, w + 4, w + 8 (flat index q, row-major 18 wide)
  int q[MT1], poff1[MT1];
#pragma unroll
  for (int j = 0; j < MT1; ++j) {
    q[j] = (wave + 4 * j) * 16 + c16;
    const int qq = min(q[j], NQ - 1), qy = qq / PW2, qx = qq - qy * PW2;
    poff1[j] = qy * PW1 + qx;
  }
  f32x4 acc1[MT1][NT1];
#pragma unroll
  for (int ct = 0; ct < NT1; ++ct) {
    const f32x4 b = *(const f32x4*)(c1.bias + ct * 16 + g * 4);
#pragma unroll
    for (int j = 0; j < MT1; ++j) acc1[j][ct] = b;
  }
  ylds_kloop3w<T, NT1, MT1>(wl, c1.ksteps, c1.cg, c1.cg_mg, c1.cg_sh, lds, pitch1, PW1, poff1, acc1);
  __syncthreads();                                          // every wave is done with cv1's weights
#pragma unroll
  for (int j = 0; j < WV2; ++j) { const int i = tid + j * 256; if (i < nw2) *(u32x4*)(wl + (size_t)i * 16) = wv2[j]; }
  {
    const int ch0 = g * 4 * NT1;
#pragma unroll
    for (int j = 0; j < MT1; ++j) {
      if (q[j] >= NQ) continue;
      const int qy = q[j] / PW2, qx = q[j] - qy * PW2;
      const int gy = oy0 - 1 + qy, gx = ox0 - 1 + qx;
      const bool inside = (unsigned)gy < (unsigned)c1.Ho && (unsigned)gx < (unsigned)c1.Wo;
      char* const d = mid + q[j] * pitch2 + ch0 * 2;
#pragma unroll
      for (int i = 0; i < 2 * NT1; ++i) {
        const float a = acc1[j][i >> 1][(i & 1) * 2], b = acc1[j][i >> 1][(i & 1) * 2 + 1];
        const unsigned w = inside ? pk_out16<T>(pack2<T>(c1.act ? silu(a) : a, c1.act ? silu(b) : b), false) : 0u;
        if (ch0 + 2 * i < c2.Cin) *(unsigned*)(d + 4 * i) = w;
      }
    }
  }
  __syncthreads();
  // ---- cv2 on the TH x 16 output pixels: wave w owns rows MT2 w .. MT2 w + MT2 - 1
  int poff2[MT2];
#pragma unroll
  for (int pt = 0; pt < MT2; ++pt) poff2[pt] = (wave * MT2 + pt) * PW2 + c16;
  f32x4 acc2[MT2][NT2];
#pragma unroll
  for (int ct = 0; ct < NT2; ++ct) {
    const f32x4 b = *(const f32x4*)(c2.bias + ct * 16 + g * 4);
#pragma unroll
    for (int pt = 0; pt < MT2; ++pt) acc2[pt][ct] = b;
  }
  ylds_kloop3w<T, NT2, MT2>(wl, c2.ksteps, c2.cg, c2.cg_mg, c2.cg_sh, mid, pitch2, PW2, poff2, acc2);
  int mm[MT2];
  bool pv[MT2];
  const int ox = ox0 + c16;
#pragma unroll
  for (int pt = 0; pt < MT2; ++pt) {
    const int oy = oy0 + wave * MT2 + pt;
    pv[pt] = oy < c2.Ho && ox < c2.Wo;
    mm[pt] = oy * c2.Wo + ox;
  }
  yconv_epilogue<T, NT2, MT2>(c2, 0, g, mm, pv, acc2);
}

template <typename T, int NT1, int NT2>
__global__ __launch_bounds__(256) void ybneck_kernel(const YBneckP p) {
  extern __shared__ __attribute__((aligned(16))) char ylds[];
  if (p.th == 4) ybneck_body<T, NT1, NT2, 4>(p, blockIdx.x, ylds);
  else ybneck_body<T, NT1, NT2, 8>(p, blockIdx.x, ylds);
}

// Workgroups are handed to the 8 XCDs round-robin (flat id mod 8), each XCD with its own 4 MiB L2.  On the large maps the
// nine taps of a 3x3 conv re-read every input pixel nine times: with the natural order every XCD touches the whole map
// (7.5 MB at 184 x 320 x 64) and the re-reads fall out of its L2.  Remapped, XCD x owns one contiguous run of tiles (an
// image band): flat id b -> start(b mod 8) + b div 8 with start(x) = x * (n div 8) + min(x, n mod 8)  (a bijection on [0, n)).
__device__ __forceinline__ int xcd_band(const int b, const int n) {
  const int x = b & 7, q = n >> 3, r = n & 7;
  return x * q + min(x, r) + (b >> 3);
}

template <typename T, int NT, bool K3, bool SPLITK>
__global__ __launch_bounds__(256) void yconv_kernel(const YConvP p) {
  extern __shared__ __attribute__((aligned(16))) char ylds[];        // SPLITK: combine scratch; tile path: the input patch
  int bx = blockIdx.x, by = blockIdx.y;
  if (!SPLITK && p.xcd) {
    const int nbx = gridDim.x, lb = xcd_band(by * nbx + bx, nbx * gridDim.y);
    by = lb / nbx; bx = lb - by * nbx;
  }
  if constexpr (!SPLITK) {
    if (p.tile) { yconv_tile_body<T, NT, K3>(p, bx, by, ylds); return; }
  }
  yconv_body<T, NT, K3, SPLITK>(p, bx, by, (float*)ylds);
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ydw_body(const YDwP& p, const int bx) {
  const int c8n = p.C >> 3;
  const int idx = bx * 256 + threadIdx.x;
  if (idx >= p.H * p.W * c8n) return;
  const int pix = idx / c8n, c0 = (idx - pix * c8n) * 8;
  const int y = pix / p.W, x = pix - y * p.W;
  const int ci0 = p.blk ? (c0 / p.blk) * p.blk_stride + p.blk_off + c0 % p.blk : c0;
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = p.bias[c0 + i];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
    const u32x4 v = *(const u32x4*)((const char*)p.in + ((size_t)(iy * p.W + ix) * p.ldi + ci0) * 2);
    const float* w = p.w + t * p.C + c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[2 * q] += unpack_lo<T>(v[q]) * w[2 * q];
      a[2 * q + 1] += unpack_hi<T>(v[q]) * w[2 * q + 1];
    }
  }
  if (p.act)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = silu(a[i]);
  if (p.add) {
    const u32x4 v = *(const u32x4*)((const char*)p.add + ((size_t)pix * p.lda + c0) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[2 * q] += unpack_lo<T>(v[q]); a[2 * q + 1] += unpack_hi<T>(v[q]); }
  }
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = pk_out16<T>(pack2<T>(a[2 * q], a[2 * q + 1]), false);
  *(u32x4*)((char*)p.out + ((size_t)pix * p.ldo + c0) * 2) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void ydw_kernel(const YDwP p) { ydw_body<T>(p, blockIdx.x); }

// Several INDEPENDENT launches of the graph in one grid (the Segment head's box / class / coefficient branches, the Proto
// block beside them, the two 1x1 convs of a C3k that read the same map ...): every one of them is a few dozen workgroups with
// a serial K loop, far too small to fill 256 CUs, and each costs a whole dependent launch when queued alone.  Workgroup b
// belongs to op s with start[s] <= b < start[s+1]; the op's parameters are read from a table in device memory (uniform).
template <typename T>
__global__ __launch_bounds__(256) void ymulti_kernel(const YMultiP* __restrict__ Pd) {
  extern __shared__ __attribute__((aligned(16))) char ylds[];        // max over the ops: combine scratch / input patch
  float* const red = (float*)ylds;
  const YMultiP& P = *Pd;                                              // device memory, read with scalar loads (uniform)
  const int b = blockIdx.x;
  int s = 0;
  for (int i = 1; i < P.n; ++i) s = b >= P.op[i].start ? i : s;
  const YMultiOp& o = P.op[s];
  int lb = b - o.start;                                   // op starts are multiples of 8: lb mod 8 is the XCD
  if (lb >= o.nblocks) return;
  if (o.code == 12) { ydw_body<T>(o.u.d, lb); return; }
  if (o.code >= 16) {
    switch (o.code - 16) {                               // (NT1 index) * 4 + (NT2 index)
#define YB(N1, N2) do { if (o.u.b.th == 4) ybneck_body<T, N1, N2, 4>(o.u.b, lb, ylds); else ybneck_body<T, N1, N2, 8>(o.u.b, lb, ylds); } while (0)
      case 0: YB(1, 1); break;
      case 1: YB(1, 2); break;
      case 5: YB(2, 2); break;
      case 6: YB(2, 4); break;
      default: YB(4, 4); break;
#undef YB
    }
    return;
  }
  if (o.u.c.xcd) lb = xcd_band(lb, o.nblocks);
  const int by = lb / o.nbx, bx = lb - by * o.nbx;
  if (o.u.c.tile) {
    switch (o.code) {
      case 0: yconv_tile_body<T, 1, false>(o.u.c, bx, by, ylds); break;
      case 2: yconv_tile_body<T, 1, true>(o.u.c, bx, by, ylds); break;
      case 4: yconv_tile_body<T, 2, false>(o.u.c, bx, by, ylds); break;
      case 6: yconv_tile_body<T, 2, true>(o.u.c, bx, by, ylds); break;
      case 8: yconv_tile_body<T, 4, false>(o.u.c, bx, by, ylds); break;
      default: yconv_tile_body<T, 4, true>(o.u.c, bx, by, ylds); break;
    }
    return;
  }
  switch (o.code) {
    case 0: yconv_body<T, 1, false, false>(o.u.c, bx, by, red); break;
    case 1: yconv_body<T, 1, false, true>(o.u.c, bx, by, red); break;
    case 2: yconv_body<T, 1, true, false>(o.u.c, bx, by, red); break;
    case 3: yconv_body<T, 1, true, true>(o.u.c, bx, by, red); break;
    case 4: yconv_body<T, 2, false, false>(o.u.c, bx, by, red); break;
    case 5: yconv_body<T, 2, false, true>(o.u.c, bx, by, red); break;
    case 6: yconv_body<T, 2, true, false>(o.u.c, bx, by, red); break;
    case 7: yconv_body<T, 2, true, true>(o.u.c, bx, by, red); break;
    case 8: yconv_body<T, 4, false, false>(o.u.c, bx, by, red); break;
    case 9: yconv_body<T, 4, false, true>(o.u.c, bx, by, red); break;
    case 10: yconv_body<T, 4, true, false>(o.u.c, bx, by, red); break;
    default: yconv_body<T, 4, true, true>(o.u.c, bx, by, red); break;
  }
}

// A chain of 1x1 convs on a small map in one launch (yolo.h YChainP).  Workgroup = one 32-pixel tile; for every conv of the chain its
// four waves take channel blocks w, w + 4, ... (the full K loop each: no split-K partial sums), then a workgroup barrier -- which on
// this target also completes the wave's stores (vmcnt) and makes them visible to the workgroup's later loads (one CU, one L1).
template <typename T>
__global__ __launch_bounds__(256) void ychain_kernel(const YChainP* __restrict__ Pd) {
  const YChainP& P = *Pd;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m_base = blockIdx.x * 32;
  for (int i = 0; i < P.n; ++i) {
    const YConvP& p = P.op[i];
    const int nt = P.nt[i];
    const int nby = (p.Cout + 16 * nt - 1) / (16 * nt);
    for (int b = wave; b < nby; b += 4) {
      if (nt == 1) yconv_body_at<T, 1, false, false>(p, m_base, b, nullptr, 0, b);
      else if (nt == 2) yconv_body_at<T, 2, false, false>(p, m_base, b, nullptr, 0, b);
      else yconv_body_at<T, 4, false, false>(p, m_base, b, nullptr, 0, b);
    }
    __syncthreads();
  }
}

// SPPF: n cascaded 5x5 stride-1 max-pools (-inf border) of one map in one launch.  pool5 applied i+1 times is the maximum
// over the (4 i + 5)^2 window clipped to the map, so the n results are ring-wise maxima of one 13 x 13 sweep (n = 3); result i
// goes to channels [i C, (i + 1) C) of `out` (the concat buffer of SPPF.cv2).
template <typename T>
__global__ __launch_bounds__(256) void ypool_kernel(const YPoolP p) {
  const int c8n = p.C >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.H * p.W * c8n) return;
  const int pix = idx / c8n, c0 = (idx - pix * c8n) * 8;
  const int y = pix / p.W, x = pix - y * p.W;
  constexpr float kNegInf = -3.0e38f;
  float a[3][8];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[r][i] = kNegInf;
  const int R = 2 * p.n;
  for (int dy = -R; dy <= R; ++dy) {
    const int iy = y + dy;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    const int ry = dy < 0 ? -dy : dy;
    for (int dx = -R; dx <= R; ++dx) {
      const int ix = x + dx;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      const int rx = dx < 0 ? -dx : dx, ring = ry > rx ? ry : rx;
      const u32x4 v = *(const u32x4*)((const char*)p.in + ((size_t)(iy * p.W + ix) * p.ldi + c0) * 2);
      float f[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { f[2 * q] = unpack_lo<T>(v[q]); f[2 * q + 1] = unpack_hi<T>(v[q]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[2][i] = fmaxf(a[2][i], f[i]);
        if (ring <= 4) a[1][i] = fmaxf(a[1][i], f[i]);
        if (ring <= 2) a[0][i] = fmaxf(a[0][i], f[i]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r >= p.n) break;
    // windows of radius 2 (r + 1): a[0] always holds ring <= 2, a[1] ring <= 4, a[2] everything swept (ring <= 2 n)
    const float* src = (r == p.n - 1) ? a[2] : a[r];
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack2<T>(src[2 * q], src[2 * q + 1]);
    *(u32x4*)((char*)p.out + ((size_t)pix * p.ldo + r * p.C + c0) * 2) = o;
  }
}

// The same in LDS for maps that fit (2 x H W x 16 bytes): one workgroup = 8 channels of the whole map; each pool is a row pass
// and a column pass over LDS (5 reads per element and pass instead of 169 dependent global loads per thread), results
// streamed out after every column pass.  The ring kernel above stays as the fallback for larger maps.
template <typename T>
__global__ __launch_bounds__(256) void ysppf_lds_kernel(const YPoolP p) {
  extern __shared__ __attribute__((aligned(16))) char ylds[];
  const int HW = p.H * p.W, tid = threadIdx.x, c0 = blockIdx.x * 8;
  u32x4* const A = (u32x4*)ylds;
  u32x4* const B = A + HW;
  for (int i = tid; i < HW; i += 256) A[i] = *(const u32x4*)((const char*)p.in + ((size_t)i * p.ldi + c0) * 2);
  auto vmax = [](const u32x4& a, const u32x4& b) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack2<T>(fmaxf(unpack_lo<T>(a[q]), unpack_lo<T>(b[q])), fmaxf(unpack_hi<T>(a[q]), unpack_hi<T>(b[q])));
    return o;
  };
  for (int r = 0; r < p.n; ++r) {
    __syncthreads();
    for (int i = tid; i < HW; i += 256) {
      const int y = i / p.W, x = i - y * p.W;
      u32x4 m = A[i];
#pragma unroll
      for (int d = -2; d <= 2; ++d)
        if (d != 0 && (unsigned)(x + d) < (unsigned)p.W) m = vmax(m, A[i + d]);
      B[i] = m;
    }
    __syncthreads();
    for (int i = tid; i < HW; i += 256) {
      const int y = i / p.W;
      u32x4 m = B[i];
#pragma unroll
      for (int d = -2; d <= 2; ++d)
        if (d != 0 && (unsigned)(y + d) < (unsigned)p.H) m = vmax(m, B[i + d * p.W]);
      A[i] = m;
      *(u32x4*)((char*)p.out + ((size_t)i * p.ldo + r * p.C + c0) * 2) = m;
    }
  }
}

__global__ __launch_bounds__(256) void yup_kernel(const YUpP p) {
  const int c8n = p.C >> 3, W2 = 2 * p.W;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4 * p.H * p.W * c8n) return;
  const int pix = idx / c8n, c0 = (idx - pix * c8n) * 8;
  const int y = pix / W2, x = pix - y * W2;
  *(u32x4*)((char*)p.out + ((size_t)pix * p.ldo + c0) * 2) =
      *(const u32x4*)((const char*)p.in + ((size_t)((y >> 1) * p.W + (x >> 1)) * p.ldi + c0) * 2);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention of one C2PSA block: out[i, h*64 + d] = sum_j softmax_j(scale * q_i . k_j) v_j[d], float32 arithmetic.
// One workgroup = 16 queries of one head; 16 lanes share a query (keys j = lane mod 16 in the score passes, output
// dims 4*lane .. 4*lane+3 in the value pass); scores live in LDS.
template <typename T>
__global__ __launch_bounds__(256) void yattn_kernel(const YAttnP p) {
  extern __shared__ float S[];                         // [16][N]
  const int tid = threadIdx.x, ql = tid >> 4, kl = tid & 15;
  const int h = blockIdx.y;
  const int qi = min(blockIdx.x * 16 + ql, p.N - 1);
  const char* base = (const char*)p.qkv + (size_t)h * 128 * 2;
  float q[32];
  {
    const char* qp = base + (size_t)qi * p.ld * 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 v = *(const u32x4*)(qp + c * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) { q[c * 8 + 2 * k] = unpack_lo<T>(v[k]) * p.scale; q[c * 8 + 2 * k + 1] = unpack_hi<T>(v[k]) * p.scale; }
    }
  }
  float* Sq = S + (size_t)ql * p.N;
  float mx = -3.0e38f;
  for (int j = kl; j < p.N; j += 16) {
    const char* kp = base + (size_t)j * p.ld * 2 + 64;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 v = *(const u32x4*)(kp + c * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) s += q[c * 8 + 2 * k] * unpack_lo<T>(v[k]) + q[c * 8 + 2 * k + 1] * unpack_hi<T>(v[k]);
    }
    Sq[j] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
  float l = 0.f;
  for (int j = kl; j < p.N; j += 16) {
    const float e = __expf(Sq[j] - mx);
    Sq[j] = e;
    l += e;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 16);
  __syncthreads();
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const char* vp = base + 128 + kl * 8;
  for (int j = 0; j < p.N; ++j) {
    const float e = Sq[j];
    const u32x2 v = *(const u32x2*)(vp + (size_t)j * p.ld * 2);
    a0 += e * unpack_lo<T>(v[0]); a1 += e * unpack_hi<T>(v[0]);
    a2 += e * unpack_lo<T>(v[1]); a3 += e * unpack_hi<T>(v[1]);
  }
  if (blockIdx.x * 16 + ql < p.N) {
    const float inv = 1.f / l;
    *(u32x2*)((char*)p.out + ((size_t)qi * p.ldo + h * 64 + kl * 4) * 2) =
        u32x2{pack2<T>(a0 * inv, a1 * inv), pack2<T>(a2 * inv, a3 * inv)};
  }
}

// MFMA form for N <= 1280 tokens (imgsz 1280 at 16:9 gives 23 x 40 = 920).  One workgroup = 64 queries of one head.
//   * V^T of the head lives in LDS ([64 dims][N keys], 16-bit): the keys of every 32-key block are stored in the order
//     4g..4g+3, 16+4g..16+4g+3 (g = 0..3), which is exactly the order in which a lane of the S^T accumulators holds
//     its scores, so P^T feeds the second MFMA straight from registers and the V^T fragment is one ds_read_b128;
//   * S^T[key][query] = K . Q^T on v_mfma_16x16x32 (K rows and the query rows are 16-byte global loads: key_dim = 32
//     = one MFMA step); online softmax in fp32 in the accumulator layout (a query is a column: per-lane scalars, the
//     maximum crosses the four row groups with two shuffles); O^T[dim][query] += V^T . P^T.
template <typename T>
__global__ __launch_bounds__(256) void yattn_mfma_kernel(const YAttnP p) {
  typedef typename Elem<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char vt[];
  const int NB = (p.N + 31) >> 5, rowB = NB * 64 + 16;      // odd number of 16-byte slots per row: conflict-free column reads
  const int h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  const char* base = (const char*)p.qkv + (size_t)h * 256;
  const size_t ldB = (size_t)p.ld * 2;
  // V rows -> V^T in LDS, eight 16-byte loads in flight per thread (one at a time, each of the 29 rounds at 920 tokens
  // waited for its own L2 round trip: half of the kernel's time)
  for (int i0 = tid; i0 < NB * 32 * 8; i0 += 8 * 256) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256, j = i >> 3, c = i & 7;
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (j < p.N) v[u] = *(const u32x4*)(base + (size_t)j * ldB + 128 + c * 16);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256, j = i >> 3, c = i & 7;
      if (i >= NB * 32 * 8) break;
      const int jj = j & 31;
      const int pos = jj < 16 ? 8 * (jj >> 2) + (jj & 3) : 8 * ((jj - 16) >> 2) + 4 + (jj & 3);
      char* d = vt + (size_t)(8 * c) * rowB + ((j & ~31) + pos) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *(unsigned short*)(d + (size_t)(2 * e) * rowB) = (unsigned short)(v[u][e] & 0xffffu);
        *(unsigned short*)(d + (size_t)(2 * e + 1) * rowB) = (unsigned short)(v[u][e] >> 16);
      }
    }
  }
  __syncthreads();
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= p.N) return;
  const int qi = min(q0 + c16, p.N - 1);
  const frag qf = *(const frag*)(base + (size_t)qi * ldB + g * 16);
  f32x4 o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // K rows three key blocks ahead (a block per iteration otherwise exposes one L2 round trip: 29 of them at 920 tokens)
  constexpr int KPD = 3;
  frag kq[KPD][2];
  auto kload = [&](int blk, frag (&d)[2]) {
    const int kb_ = min(blk, NB - 1) * 32;
    d[0] = *(const frag*)(base + (size_t)min(kb_ + c16, p.N - 1) * ldB + 64 + g * 16);
    d[1] = *(const frag*)(base + (size_t)min(kb_ + 16 + c16, p.N - 1) * ldB + 64 + g * 16);
  };
#pragma unroll
  for (int i = 0; i < KPD; ++i) kload(i, kq[i]);
  for (int blk0 = 0; blk0 < NB; blk0 += KPD) {
#pragma unroll
   for (int bi = 0; bi < KPD; ++bi) {
    const int blk = blk0 + bi;
    if (blk >= NB) break;
    const int kbase = blk * 32;
    const frag ka = kq[bi][0], kb = kq[bi][1];
    kload(blk + KPD, kq[bi]);
    f32x4 s0 = Elem<T>::mfma(ka, qf, z4), s1 = Elem<T>::mfma(kb, qf, z4);    // s0[r]: key kbase + 4g + r, query q0 + c16
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0[r] = kbase + 4 * g + r < p.N ? s0[r] * p.scale : -3.0e38f;
      s1[r] = kbase + 16 + 4 * g + r < p.N ? s1[r] * p.scale : -3.0e38f;
      mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx), alpha = __expf(m - mn);
    float pr[8], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pr[r] = __expf(s0[r] - mn); pr[4 + r] = __expf(s1[r] - mn); }
#pragma unroll
    for (int r = 0; r < 8; ++r) ps += pr[r];
    l = l * alpha + ps;
    m = mn;
    const u32x4 pw = u32x4{pack2<T>(pr[0], pr[1]), pack2<T>(pr[2], pr[3]), pack2<T>(pr[4], pr[5]), pack2<T>(pr[6], pr[7])};
    const frag pf = __builtin_bit_cast(frag, pw);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const frag vf = *(const frag*)(vt + (size_t)(t * 16 + c16) * rowB + (kbase + 8 * g) * 2);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] *= alpha;
      o[t] = Elem<T>::mfma(vf, pf, o[t]);                                      // D[dim t*16 + 4g + r][query c16]
    }
   }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (q0 + c16 < p.N) {
    const float inv = 1.f / l;
    char* op = (char*)p.out + ((size_t)(q0 + c16) * p.ldo + h * 64 + 4 * g) * 2;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *(u32x2*)(op + t * 32) = u32x2{pack2<T>(o[t][0] * inv, o[t][1] * inv), pack2<T>(o[t][2] * inv, o[t][3] * inv)};
  }
}

// ---------------------------------------------------------------------------------------------------------------
// cv2 INTER_LINEAR tap of an 8-bit image (same arithmetic as prep.hip's resize_linear_u8_kernel)
__device__ __forceinline__ void lin_tap(int d, int src, double scale, int* s0, int* s1, int* a0, int* a1) {
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

template <typename T>
__global__ __launch_bounds__(256) void yletter_kernel(const YLetterP p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.h * p.w) return;
  const int y = idx / p.w, x = idx - y * p.w;
  int bgr[3] = {114, 114, 114};
  const int cy = y - p.top, cx = x - p.left;
  if ((unsigned)cy < (unsigned)p.nh && (unsigned)cx < (unsigned)p.nw) {
    if (p.nh == p.H && p.nw == p.W) {
      const uint8_t* s = p.frame + ((size_t)cy * p.W + cx) * 3;
      bgr[0] = s[0]; bgr[1] = s[1]; bgr[2] = s[2];
    } else {
      int x0, x1, a0, a1, y0, y1, b0, b1;
      lin_tap(cx, p.W, p.sx, &x0, &x1, &a0, &a1);
      lin_tap(cy, p.H, p.sy, &y0, &y1, &b0, &b1);
      const uint8_t* r0 = p.frame + (size_t)y0 * p.W * 3;
      const uint8_t* r1 = p.frame + (size_t)y1 * p.W * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
        const int S1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
        bgr[c] = ((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2) & 255;
      }
    }
  }
  // BGR -> RGB, /255 in float32 (BasePredictor.preprocess), then the network's 16-bit input type
  const float r = __fdiv_rn((float)bgr[2], 255.f), gch = __fdiv_rn((float)bgr[1], 255.f), b = __fdiv_rn((float)bgr[0], 255.f);
  *(u32x4*)((char*)p.out + (size_t)idx * 16) = u32x4{pack2<T>(r, gch), pack2<T>(b, 0.f), 0u, 0u};
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ydecode_kernel(const YDecodeP p) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= p.A) return;
  const int lvl = a >= p.lvl_a0[2] ? 2 : (a >= p.lvl_a0[1] ? 1 : 0);
  const int la = a - p.lvl_a0[lvl];
  const int gy = la / p.lvl_w[lvl], gx = la - gy * p.lvl_w[lvl];
  const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f, st = (float)p.lvl_stride[lvl];
  const float* row = p.pred + (size_t)a * p.no;
  float d[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float mx = row[s * 16];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, row[s * 16 + i]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = expf(row[s * 16 + i] - mx);
      den += e;
      num += e * (float)i;
    }
    d[s] = num / den;
  }
  // dist2bbox(xywh=True) * stride, then ops.xywh2xyxy (the order the reference pipeline applies them)
  const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
  const float cx = __fmul_rn(__fmul_rn(__fadd_rn(x1, x2), 0.5f), st), cy = __fmul_rn(__fmul_rn(__fadd_rn(y1, y2), 0.5f), st);
  const float w = __fmul_rn(__fsub_rn(x2, x1), st), h = __fmul_rn(__fsub_rn(y2, y1), st);
  const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);
  float* b = p.cand_box + (size_t)a * 4;
  b[0] = __fsub_rn(cx, hw); b[1] = __fsub_rn(cy, hh); b[2] = __fadd_rn(cx, hw); b[3] = __fadd_rn(cy, hh);
  float best = -1.f;
  int bj = 0;
  for (int j = 0; j < p.nc; ++j) {
    const float s = 1.f / (1.f + expf(-row[64 + j]));
    if (s > best) { best = s; bj = j; }
  }
  p.cand_conf[a] = best;
  p.cand_cls[a] = bj;
}

// ops.non_max_suppression (single label per box) + torchvision.ops.nms, one workgroup:
// candidates sorted by (confidence descending, anchor ascending) -- torchvision's stable descending sort over
// candidates that ultralytics leaves in anchor order -- then greedy suppression with
// inter / (area_i + area_j - inter) > iou in round-to-nearest float32, boxes shifted by cls * max_wh.
constexpr int kNmsCap = 4096;
__global__ __launch_bounds__(1024) void ynms_kernel(const YNmsP p) {
  __shared__ float sconf[kNmsCap];
  __shared__ int sanc[kNmsCap];
  __shared__ float sbox[kNmsCap][4];
  __shared__ float sarea[kNmsCap];
  __shared__ unsigned char dead[kNmsCap];
  __shared__ int sscan[1024];
  __shared__ int skept[300];                         // kept candidates in NMS order (max_det <= 300)
  __shared__ int kept_n;
  const int tid = threadIdx.x;
  // ---- candidates: every anchor whose confidence passes the threshold, gathered in anchor order (no atomics: the set and
  // its order are deterministic).  More than kNmsCap of them (a threshold far below ultralytics' 0.25 default): the kNmsCap
  // most confident ones -- a bisection on the float bit pattern finds the cut, ties at the cut enter in anchor order
  // (ultralytics keeps the 30,000 most confident; documented deviation for 4,096 < n <= 30,000).
  // Wave w owns the contiguous anchor segment [w seg, (w + 1) seg) and walks it 64 anchors at a time (coalesced loads);
  // ballots give the per-step counts and each lane's rank, so the compaction keeps anchor order with one block barrier.
  const int lane = tid & 63, wave = tid >> 6;
  const int seg = (p.A + 16 * 64 - 1) / (16 * 64) * 64, s0 = wave * seg, s1 = min(s0 + seg, p.A);
  int* const wtot = sscan;                           // [16] per-wave counts
  // kind 0: bits > cut, kind 1: bits == cut.  Eight loads in flight per lane (a step at a time, each ballot waited for its own
  // L2 round trip: 19 of them per pass at 19,320 anchors).
  auto test = [&](int a, float v, unsigned cut_, int kind) {
    const unsigned u = __float_as_uint(v);
    return a < s1 && v > p.conf && (kind ? u == cut_ : u > cut_);
  };
  auto count_pass = [&](unsigned cut_, int kind) {   // -> total over the block; wtot[] holds the per-wave counts afterwards
    int c = 0;
    for (int a0 = s0 + lane; a0 < s0 + seg; a0 += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int a = a0 + u * 64; v[u] = a < s1 ? p.cand_conf[a] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) c += __popcll(__ballot(test(a0 + u * 64, v[u], cut_, kind)));
    }
    __syncthreads();                                 // previous readers of wtot are done
    if (lane == 0) wtot[wave] = c;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wtot[w];
    return t;
  };
  auto gather_pass = [&](unsigned cut_, int kind, int base, int limit) {   // after count_pass(cut_, kind): ranks base + 0 .. limit - 1
    int w0 = base;
    for (int w = 0; w < wave; ++w) w0 += wtot[w];
    const int stop = base + limit;
    for (int a0 = s0 + lane; a0 < s0 + seg; a0 += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int a = a0 + u * 64; v[u] = a < s1 ? p.cand_conf[a] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int a = a0 + u * 64;
        const bool ok = test(a, v[u], cut_, kind);
        const unsigned long long m = __ballot(ok);
        const int r = w0 + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && r < stop) { sanc[r] = a; sconf[r] = v[u]; }
        w0 += __popcll(m);
      }
    }
  };
  unsigned cut = 0;                                  // candidates: conf > p.conf and bits(conf) > cut (+ ties at the cut)
  int total = count_pass(0u, 0);
  int n_eq_take = 0;
  if (total > kNmsCap) {
    unsigned lo = 0u, hi = 0x3f800000u;              // confidences are sigmoids in (0, 1): bit order = value order
    while (lo < hi) {                                // smallest cut with count(bits > cut) <= kNmsCap
      const unsigned mid = lo + ((hi - lo) >> 1);
      if (count_pass(mid, 0) <= kNmsCap) hi = mid; else lo = mid + 1;
    }
    cut = lo;
    total = count_pass(cut, 0);
    n_eq_take = kNmsCap - total;                     // free slots for confidences exactly at the cut
  }
  int n = total;
  gather_pass(cut, 0, 0, kNmsCap);
  if (n_eq_take > 0) {
    const int teq = count_pass(cut, 1);
    gather_pass(cut, 1, n, n_eq_take);
    n += min(teq, n_eq_take);
  }
  __syncthreads();
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = n + tid; i < np2; i += 1024) { sanc[i] = 0x7fffffff; sconf[i] = -1.f; }
  if (tid == 0) kept_n = 0;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const float ci = sconf[i], cl = sconf[l];
          const int ai = sanc[i], al = sanc[l];
          const bool i_first = ci > cl || (ci == cl && ai < al);      // desired order: i before l
          const bool asc_block = (i & k) != 0;                         // this block sorts in the reverse order
          if (i_first == asc_block) { sconf[i] = cl; sconf[l] = ci; sanc[i] = al; sanc[l] = ai; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += 1024) {
    const int a = sanc[i];
    const float off = __fmul_rn((float)p.cand_cls[a], p.max_wh);
    const float* b = p.cand_box + (size_t)a * 4;
    const float x1 = __fadd_rn(b[0], off), y1 = __fadd_rn(b[1], off), x2 = __fadd_rn(b[2], off), y2 = __fadd_rn(b[3], off);
    sbox[i][0] = x1; sbox[i][1] = y1; sbox[i][2] = x2; sbox[i][3] = y2;
    sarea[i] = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    dead[i] = 0;
  }
  __syncthreads();
  // greedy pass: nothing but LDS inside the serial loop (the kept list); the detection rows are written afterwards, one
  // thread per row (writing them from thread 0 inside the loop put two global round trips on every kept box: 25 of 35 us)
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;                              // uniform: every thread reads the same byte after the barrier
    const int slot = kept_n;
    __syncthreads();
    if (tid == 0) { skept[slot] = i; kept_n = slot + 1; }
    if (slot + 1 >= p.max_det) { __syncthreads(); break; }
    const float ix1 = sbox[i][0], iy1 = sbox[i][1], ix2 = sbox[i][2], iy2 = sbox[i][3], ia = sarea[i];
    for (int j = i + 1 + tid; j < n; j += 1024) {
      if (dead[j]) continue;
      const float w = fmaxf(0.f, __fsub_rn(fminf(ix2, sbox[j][2]), fmaxf(ix1, sbox[j][0])));
      const float h = fmaxf(0.f, __fsub_rn(fminf(iy2, sbox[j][3]), fmaxf(iy1, sbox[j][1])));
      const float inter = __fmul_rn(w, h);
      const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ia, sarea[j]), inter));
      if (ovr > p.iou) dead[j] = 1;
    }
    __syncthreads();
  }
  __syncthreads();
  for (int slot = tid; slot < kept_n; slot += 1024) {
    const int i = skept[slot], a = sanc[i];
    const float* b = p.cand_box + (size_t)a * 4;
    float* d = p.det + (size_t)slot * 8;
    const float fx[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float padv = (c & 1) ? p.pad_y : p.pad_x, lim = (c & 1) ? (float)p.frame_h : (float)p.frame_w;
      const float v = __fdiv_rn(__fsub_rn(fx[c], padv), p.gain);
      d[c] = fminf(fmaxf(v, 0.f), lim);
      p.det_lb[slot * 4 + c] = fx[c];
    }
    d[4] = sconf[i]; d[5] = (float)p.cand_cls[a]; d[6] = (float)a; d[7] = 0.f;
    p.det_anchor[slot] = a;
  }
  if (tid == 0) *p.det_count = kept_n;
}

// ---------------------------------------------------------------------------------------------------------------
// ops.process_mask, step 1: masks = coef @ proto, cropped to the box at proto resolution
template <typename T>
__global__ __launch_bounds__(256) void ymask_low_kernel(const YMaskP p) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.mh * p.mw) return;
  const int n = min(*p.det_count, p.max_det);          // one thread = one proto pixel, all instances (a grid over
  const int y = pix / p.mw, x = pix - y * p.mw;        // max_det x pixels spent 18 us starting 69 k empty workgroups)
  const float wr = (float)((double)p.mw / (double)p.iw), hr = (float)((double)p.mh / (double)p.ih);
  float pr[32];
  {
    const char* ps = (const char*)p.proto + (size_t)pix * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 v = *(const u32x4*)(ps + c * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) { pr[c * 8 + 2 * k] = unpack_lo<T>(v[k]); pr[c * 8 + 2 * k + 1] = unpack_hi<T>(v[k]); }
    }
  }
  for (int i = 0; i < n; ++i) {
    const float* bl = p.det_lb + i * 4;
    const float x1 = __fmul_rn(bl[0], wr), y1 = __fmul_rn(bl[1], hr), x2 = __fmul_rn(bl[2], wr), y2 = __fmul_rn(bl[3], hr);
    float m = 0.f;
    if ((float)x >= x1 && (float)x < x2 && (float)y >= y1 && (float)y < y2) {
      const float* coef = p.pred + (size_t)p.det_anchor[i] * p.no + 64 + p.nc;
#pragma unroll
      for (int k = 0; k < 32; ++k) m += coef[k] * pr[k];
    }
    p.low[(size_t)i * p.mh * p.mw + pix] = m;
  }
}

// step 2 + get_bbox_mask's sum / clip / x255: bilinear upsample (F.interpolate, align_corners=False) of every cropped
// mask to the letterboxed input, `> 0`, OR over instances -> 255 / 0
__global__ __launch_bounds__(256) void ymask_merge_kernel(const YMaskP p) {
  __shared__ float sb[300][4];
  const int n = min(*p.det_count, p.max_det);
  for (int i = threadIdx.x; i < n * 4; i += 256) sb[i >> 2][i & 3] = p.det_lb[i];
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.ih * p.iw) return;
  const int Y = idx / p.iw, X = idx - Y * p.iw;
  const float sch = (float)p.mh / (float)p.ih, scw = (float)p.mw / (float)p.iw;
  float fy = __fsub_rn(__fmul_rn(sch, (float)Y + 0.5f), 0.5f), fx = __fsub_rn(__fmul_rn(scw, (float)X + 0.5f), 0.5f);
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < p.mh - 1 ? 1 : 0), x1 = x0 + (x0 < p.mw - 1 ? 1 : 0);
  const float ly1 = __fsub_rn(fy, (float)y0), lx1 = __fsub_rn(fx, (float)x0), ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
  const float margin = 2.f * (float)p.iw / (float)p.mw;          // a cropped mask is zero further than one proto pixel outside its box
  bool any = false;
  for (int i = 0; i < n && !any; ++i) {
    if ((float)X < sb[i][0] - margin || (float)X > sb[i][2] + margin || (float)Y < sb[i][1] - margin || (float)Y > sb[i][3] + margin) continue;
    const float* L = p.low + (size_t)i * p.mh * p.mw;
    const float a = L[y0 * p.mw + x0], b = L[y0 * p.mw + x1], c = L[y1 * p.mw + x0], d = L[y1 * p.mw + x1];
    const float v = __fadd_rn(__fmul_rn(ly0, __fadd_rn(__fmul_rn(lx0, a), __fmul_rn(lx1, b))),
                              __fmul_rn(ly1, __fadd_rn(__fmul_rn(lx0, c), __fmul_rn(lx1, d))));
    any = v > 0.f;
  }
  p.merged[idx] = any ? 255 : 0;
}

// parity taps: a 16-bit / float32 / uint8 NHWC view -> float32 [C][H][W]
template <typename T>
__global__ __launch_bounds__(256) void yread_kernel(const void* src, int kind, int H, int W, int C, int ld, float* dst) {
  const size_t total = (size_t)C * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    const size_t r = i / W;
    const int y = (int)(r % H), c = (int)(r / H);
    const size_t s = ((size_t)y * W + x) * ld + c;
    dst[i] = kind == 1 ? ((const float*)src)[s] : (kind == 2 ? (float)((const uint8_t*)src)[s] : (kind == 3 ? (float)((const int*)src)[s] : to_f32<T>(((const T*)src)[s])));
  }
}

}  // namespace

extern "C" int flope_yread_launch(const void* src, int kind, int H, int W, int C, int ld, int dtype, float* dst, void* stream) {
  const size_t total = (size_t)C * H * W;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  if (dtype == 0) hipLaunchKernelGGL(yread_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, kind, H, W, C, ld, dst);
  else hipLaunchKernelGGL(yread_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, kind, H, W, C, ld, dst);
  return (int)hipGetLastError();
}

// ---- launch wrappers ---------------------------------------------------------------------------------------------
#define YDISPATCH(dt_, KERN, grid, block, lds, st, ...)                                                   \
  do {                                                                                                    \
    if ((dt_) == 0) hipLaunchKernelGGL(KERN<bf16_t>, grid, block, lds, st, __VA_ARGS__);                  \
    else hipLaunchKernelGGL(KERN<f16_t>, grid, block, lds, st, __VA_ARGS__);                              \
  } while (0)

template <typename T>
static void yconv_go(const YConvP& p, int nt, bool splitk, dim3 grid, int lds, hipStream_t st) {
  const bool k3 = p.k == 3;
#define GO(NT_)                                                                                              \
  do {                                                                                                       \
    if (k3 && splitk) hipLaunchKernelGGL((yconv_kernel<T, NT_, true, true>), grid, dim3(256), lds, st, p);   \
    else if (k3) hipLaunchKernelGGL((yconv_kernel<T, NT_, true, false>), grid, dim3(256), lds, st, p);       \
    else if (splitk) hipLaunchKernelGGL((yconv_kernel<T, NT_, false, true>), grid, dim3(256), lds, st, p);   \
    else hipLaunchKernelGGL((yconv_kernel<T, NT_, false, false>), grid, dim3(256), lds, st, p);              \
  } while (0)
  if (nt == 1) GO(1); else if (nt == 2) GO(2); else GO(4);
#undef GO
}

static int g_splitk_max_m = 8192;                           // split-K (one 32-pixel tile per workgroup, K over its 4 waves) up to this map size
extern "C" int flope_yconv_splitk_max_m(int m) { const int prev = g_splitk_max_m; if (m >= 0) g_splitk_max_m = m; return prev; }
static int g_wlds_mode = 0;                                  // 1: tile path, 3x3, >= 8 K steps keeps the weight image in LDS too (measured
                                                             // slower: 0.78 vs 0.74 ms -- 96 KB of LDS leave one workgroup per CU on the large maps)
extern "C" int flope_yconv_wlds_mode(int mode) { const int prev = g_wlds_mode; if (mode == 0 || mode == 1) g_wlds_mode = mode; return prev; }
static int g_tile_mode = 1;                                  // 1: LDS-staged 8 x 16 tiles for every non-split-K conv; 0: fragments from global
extern "C" int flope_yconv_tile_mode(int mode) { const int prev = g_tile_mode; if (mode == 0 || mode == 1) g_tile_mode = mode; return prev; }
constexpr int kYTileLdsMax = 96 * 1024, kYBneckLdsMax = 144 * 1024;   // patch alone; patch + weight image(s)

// fills the launch-derived fields of *q (tile, tiles_x, pw_*), -> split-K?, grid, dynamic LDS bytes
static bool yconv_geometry(const YConvP* p, int nt, YConvP* q, bool* splitk, int* nbx, int* nby, int* lds) {
  if ((p->k != 1 && p->k != 3) || p->Cin % 8 || (nt != 1 && nt != 2 && nt != 4) || p->M < 1) return false;
  const int rows = p->out_mode == 2 ? 4 * p->dc : p->Cout;
  if (p->out_mode == 2 && (p->dc % (16 * nt) || p->res)) return false;
  *q = *p;
  *splitk = p->M <= g_splitk_max_m && p->ksteps >= 8;        // small map, long K: one tile per workgroup, K over its 4 waves
  *nby = (rows + 16 * nt - 1) / (16 * nt);
  const int PW = 15 * p->stride + p->k, PH = 7 * p->stride + p->k, patch = PH * PW * (p->Cin * 2 + 16);
  q->tile = (!*splitk && g_tile_mode && patch <= kYTileLdsMax) ? 1 : 0;
  q->wlds = 0;
  if (q->tile) {
    q->tiles_x = (p->Wo + 15) / 16;
    *nbx = q->tiles_x * ((p->Ho + 7) / 8);
    flope_host::fastdiv_magic((unsigned)PW, &q->pw_mg, &q->pw_sh);
    *lds = patch;
    const int wbytes = p->ksteps * nt * 1024;
    if (g_wlds_mode && p->k == 3 && p->ksteps >= 8 && patch + wbytes <= kYBneckLdsMax) { q->wlds = 1; *lds = patch + wbytes; }
  } else {
    *nbx = *splitk ? (p->M + 31) / 32 : (p->M + 127) / 128;
    *lds = *splitk ? 3 * 2 * nt * 4 * 64 * 4 : 0;
  }
  return true;
}
static bool yconv_wants_xcd_bands(const YConvP* p, bool splitk, int nbx, int nby, int mode) {
  return mode != 0 && !splitk && nbx * nby >= 64 && (mode == 2 || p->k == 3);
}
static int g_xcd_mode = 0;                                   // 0 never, 1 3x3 convs on >= 64 workgroups, 2 those and 1x1 convs
extern "C" int flope_yconv_xcd_mode(int mode) { const int prev = g_xcd_mode; if (mode >= 0 && mode <= 2) g_xcd_mode = mode; return prev; }

// nt = channel tiles of 16 per workgroup column (1, 2 or 4): rows of p->w / p->bias = ceil(Cout / (16 nt)) * 16 nt
extern "C" int flope_yconv_launch(const YConvP* p, int dtype, int nt, void* stream) {
  bool splitk; int nbx, nby, lds;
  YConvP q;
  if (!yconv_geometry(p, nt, &q, &splitk, &nbx, &nby, &lds)) return (int)hipErrorInvalidValue;
  const dim3 grid(nbx, nby);
  q.xcd = yconv_wants_xcd_bands(p, splitk, nbx, nby, g_xcd_mode) ? 1 : 0;
  if (dtype == 0) yconv_go<bf16_t>(q, nt, splitk, grid, lds, (hipStream_t)stream); else yconv_go<f16_t>(q, nt, splitk, grid, lds, (hipStream_t)stream);
  return (int)hipGetLastError();
}

// ---- several independent ops in one grid (ymulti_kernel) ---------------------------------------------------------
extern "C" int flope_ymulti_add_conv(YMultiP* m, const YConvP* p, int nt) {
  bool splitk; int nbx, nby, lds;
  YConvP q;
  if (m->n >= kYMultiMax || !yconv_geometry(p, nt, &q, &splitk, &nbx, &nby, &lds)) return (int)hipErrorInvalidValue;
  YMultiOp& o = m->op[m->n++];
  o.code = (nt == 1 ? 0 : nt == 2 ? 4 : 8) + (p->k == 3 ? 2 : 0) + (splitk ? 1 : 0);
  o.nbx = nbx; o.start = m->total; o.nblocks = nbx * nby; o.u.c = q;
  m->lds = std::max(m->lds, lds);
  o.u.c.xcd = yconv_wants_xcd_bands(p, splitk, nbx, nby, g_xcd_mode) ? 1 : 0;
  m->total += (o.nblocks + 7) / 8 * 8;
  return 0;
}

// ---- fused Bottleneck ------------------------------------------------------------------------------------------------
// fusable: both 3x3 stride 1 on the same map, <= 64 output channels each, (NT1, NT2) one of the instantiated pairs, LDS fits
static int ybneck_code(const YConvP* c1, int nt1, const YConvP* c2, int nt2, int* lds, int* tiles) {
  if (c1->k != 3 || c2->k != 3 || c1->stride != 1 || c2->stride != 1 || c1->out_mode || c2->out_mode || c1->res) return -1;
  if (c1->Ho != c2->Hi || c1->Wo != c2->Wi || c1->Cout > 16 * nt1 || c2->Cout > 16 * nt2 || c2->Cin != c1->Cout) return -1;
  const int i1 = nt1 == 1 ? 0 : nt1 == 2 ? 1 : 2, i2 = nt2 == 1 ? 0 : nt2 == 2 ? 1 : 2, code = i1 * 4 + i2;
  if (code != 0 && code != 1 && code != 5 && code != 6 && code != 10) return -1;
  if (c1->ksteps > 18 || c2->ksteps > 18 || c1->Cin > 64) return -1;
  const int th = c2->M <= 4096 ? 4 : 8;                       // small maps: 4-row tiles (twice the workgroups, shorter chains)
  *lds = (th + 4) * 20 * (c1->Cin * 2 + 16) + (th + 2) * 18 * (c2->Cin * 2 + 16) + std::max(c1->ksteps * nt1, c2->ksteps * nt2) * 1024;
  if (*lds > kYBneckLdsMax) return -1;
  *tiles = ((c2->Wo + 15) / 16) * ((c2->Ho + th - 1) / th);
  return code;
}
extern "C" int flope_ybneck_fusable(const YConvP* c1, int nt1, const YConvP* c2, int nt2) {
  int lds, tiles;
  return ybneck_code(c1, nt1, c2, nt2, &lds, &tiles) >= 0 ? 1 : 0;
}
static void ybneck_fill(YBneckP* b, const YConvP* c1, const YConvP* c2) {
  b->c1 = *c1; b->c2 = *c2; b->tiles_x = (c2->Wo + 15) / 16; b->th = c2->M <= 4096 ? 4 : 8;
}
template <typename T>
static void ybneck_go(const YBneckP& b, int code, int tiles, int lds, hipStream_t st) {
  switch (code) {
    case 0: hipLaunchKernelGGL((ybneck_kernel<T, 1, 1>), dim3(tiles), dim3(256), lds, st, b); break;
    case 1: hipLaunchKernelGGL((ybneck_kernel<T, 1, 2>), dim3(tiles), dim3(256), lds, st, b); break;
    case 5: hipLaunchKernelGGL((ybneck_kernel<T, 2, 2>), dim3(tiles), dim3(256), lds, st, b); break;
    case 6: hipLaunchKernelGGL((ybneck_kernel<T, 2, 4>), dim3(tiles), dim3(256), lds, st, b); break;
    default: hipLaunchKernelGGL((ybneck_kernel<T, 4, 4>), dim3(tiles), dim3(256), lds, st, b); break;
  }
}
extern "C" int flope_ybneck_launch(const YConvP* c1, int nt1, const YConvP* c2, int nt2, int dtype, void* stream) {
  int lds, tiles;
  const int code = ybneck_code(c1, nt1, c2, nt2, &lds, &tiles);
  if (code < 0) return (int)hipErrorInvalidValue;
  YBneckP b; ybneck_fill(&b, c1, c2);
  if (dtype == 0) ybneck_go<bf16_t>(b, code, tiles, lds, (hipStream_t)stream); else ybneck_go<f16_t>(b, code, tiles, lds, (hipStream_t)stream);
  return (int)hipGetLastError();
}
extern "C" int flope_ymulti_add_bneck(YMultiP* m, const YConvP* c1, int nt1, const YConvP* c2, int nt2) {
  int lds, tiles;
  const int code = ybneck_code(c1, nt1, c2, nt2, &lds, &tiles);
  if (m->n >= kYMultiMax || code < 0) return (int)hipErrorInvalidValue;
  YMultiOp& o = m->op[m->n++];
  o.code = 16 + code; o.nbx = tiles; o.start = m->total; o.nblocks = tiles;
  ybneck_fill(&o.u.b, c1, c2);
  m->lds = std::max(m->lds, lds);
  m->total += (tiles + 7) / 8 * 8;
  return 0;
}

extern "C" int flope_ymulti_add_dw(YMultiP* m, const YDwP* p) {
  if (m->n >= kYMultiMax || p->C % 8) return (int)hipErrorInvalidValue;
  YMultiOp& o = m->op[m->n++];
  o.code = 12; o.nbx = 1; o.start = m->total; o.nblocks = (p->H * p->W * (p->C / 8) + 255) / 256; o.u.d = *p;
  m->total += (o.nblocks + 7) / 8 * 8;
  return 0;
}

// m: the host copy (grid, LDS size); m_dev: the same table in device memory, read by the kernel
extern "C" int flope_ymulti_launch(const YMultiP* m, const YMultiP* m_dev, int dtype, void* stream) {
  if (!m_dev || m->n < 1 || m->n > kYMultiMax || m->total < 1) return (int)hipErrorInvalidValue;
  YDISPATCH(dtype, ymulti_kernel, dim3(m->total), dim3(256), m->lds, (hipStream_t)stream, m_dev);
  return (int)hipGetLastError();
}

// can this conv be a member of a chain?  (1x1, stride 1, 16-bit output view, a small map)
extern "C" int flope_ychain_ok(const YConvP* p, int nt) {
  return p->k == 1 && p->stride == 1 && p->out_mode == 0 && p->M >= 1 && p->M <= 4096 && (nt == 1 || nt == 2 || nt == 4) && p->Hi == p->Ho && p->Wi == p->Wo;
}
extern "C" int flope_ychain_launch(const YChainP* c, const YChainP* c_dev, int dtype, void* stream) {
  if (!c_dev || c->n < 2 || c->n > kYChainMax || c->tiles < 1) return (int)hipErrorInvalidValue;
  YDISPATCH(dtype, ychain_kernel, dim3(c->tiles), dim3(256), 0, (hipStream_t)stream, c_dev);
  return (int)hipGetLastError();
}

extern "C" int flope_ydw_launch(const YDwP* p, int dtype, void* stream) {
  if (p->C % 8) return (int)hipErrorInvalidValue;
  const int total = p->H * p->W * (p->C / 8);
  YDISPATCH(dtype, ydw_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

constexpr size_t kYSppfLdsMax = 144 * 1024;
static int g_pool_lds = 1;                                   // 0: the ring kernel for every map (A/B, parity of the two)
extern "C" int flope_ypool_lds_mode(int mode) { const int prev = g_pool_lds; if (mode == 0 || mode == 1) g_pool_lds = mode; return prev; }
extern "C" int flope_ypool_launch(const YPoolP* p, int dtype, void* stream) {
  if (p->C % 8 || p->n < 1 || p->n > 3) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)2 * p->H * p->W * 16;
  if (g_pool_lds && lds <= kYSppfLdsMax) {
    YDISPATCH(dtype, ysppf_lds_kernel, dim3(p->C / 8), dim3(256), lds, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
  }
  const int total = p->H * p->W * (p->C / 8);
  YDISPATCH(dtype, ypool_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_yup_launch(const YUpP* p, void* stream) {
  if (p->C % 8) return (int)hipErrorInvalidValue;
  const int total = 4 * p->H * p->W * (p->C / 8);
  hipLaunchKernelGGL(yup_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_yattn_init() {
  hipError_t e = hipFuncSetAttribute((const void*)yattn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)yattn_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)yattn_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)yattn_mfma_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define YLDS(K) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, kYBneckLdsMax)
#define YLDSB(K) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, kYBneckLdsMax)
#define YLDS_T(T) YLDSB((ybneck_kernel<T, 1, 1>)); YLDSB((ybneck_kernel<T, 1, 2>)); YLDSB((ybneck_kernel<T, 2, 2>)); YLDSB((ybneck_kernel<T, 2, 4>)); YLDSB((ybneck_kernel<T, 4, 4>)); YLDSB(ymulti_kernel<T>); \
  YLDS((yconv_kernel<T, 1, false, false>)); YLDS((yconv_kernel<T, 1, true, false>)); YLDS((yconv_kernel<T, 2, false, false>)); \
  YLDS((yconv_kernel<T, 2, true, false>)); YLDS((yconv_kernel<T, 4, false, false>)); YLDS((yconv_kernel<T, 4, true, false>))
  YLDS_T(bf16_t); YLDS_T(f16_t);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ysppf_lds_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kYSppfLdsMax);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ysppf_lds_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kYSppfLdsMax);
#undef YLDS_T
#undef YLDS
#undef YLDSB
  return (int)e;
}

// variant: 0 = automatic (MFMA kernel while the head's V^T fits LDS, N <= 1280), 1 = force the generic fp32 kernel
extern "C" int flope_yattn_launch(const YAttnP* p, int dtype, int variant, void* stream) {
  if (p->N < 1) return (int)hipErrorInvalidValue;
  const int NB = (p->N + 31) / 32;
  const size_t lds_m = (size_t)64 * (NB * 64 + 16);
  if (variant == 0 && lds_m <= 160 * 1024) {
    YDISPATCH(dtype, yattn_mfma_kernel, dim3((p->N + 63) / 64, p->heads), dim3(256), lds_m, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
  }
  const size_t lds = (size_t)16 * p->N * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;                   // N <= 2560 tokens (imgsz up to ~1600)
  YDISPATCH(dtype, yattn_kernel, dim3((p->N + 15) / 16, p->heads), dim3(256), lds, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_yletter_launch(const YLetterP* p, int dtype, void* stream) {
  YDISPATCH(dtype, yletter_kernel, dim3((p->h * p->w + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_ydecode_launch(const YDecodeP* p, void* stream) {
  hipLaunchKernelGGL(ydecode_kernel, dim3((p->A + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_ynms_launch(const YNmsP* p, void* stream) {
  if (p->max_det < 1 || p->max_det > 300) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ynms_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

extern "C" int flope_y32_mask_low_launch(const YMaskP* p, void* stream);   // yolo_f32.hip: float32 proto map
extern "C" int flope_ymask_launch(const YMaskP* p, int dtype, void* stream) {
  if (p->max_det < 1 || p->max_det > 300) return (int)hipErrorInvalidValue;
  if (dtype == 2) { if (int s = flope_y32_mask_low_launch(p, stream)) return s; }
  else YDISPATCH(dtype, ymask_low_kernel, dim3((p->mh * p->mw + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  hipLaunchKernelGGL(ymask_merge_kernel, dim3((p->ih * p->iw + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p);
  return (int)hipGetLastError();
}

#ifdef FLOPE_STAG_DBG
// diagnostic build: copies the stamp records of the last launches to host memory and resets the counter; returns the count
extern "C" int flope_ydbg_read(unsigned long long* dst_host, int cap_records) {
  unsigned n = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ydbg_n), sizeof(n)) != hipSuccess) return -1;
  const int m = (int)(n < 512u ? n : 512u) < cap_records ? (int)(n < 512u ? n : 512u) : cap_records;
  if (m > 0 && hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_ydbg), (size_t)m * 64) != hipSuccess) return -1;
  n = 0;
  hipMemcpyToSymbol(HIP_SYMBOL(g_ydbg_n), &n, sizeof(n));
  return m;
}
#endif
