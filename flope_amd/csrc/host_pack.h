// Host-side weight preparation shared by the engine (engine.hip) and the CPU test
// harness (tests/host_harness): 16-bit conversions (round to nearest even) and the
// MFMA/LDS-image weight packers.  Plain C++, no HIP.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

#define FLOPE_DT_BF16_ 0

namespace flope_host {

// multiply-shift constants for n / d, 0 <= n < 2^31, 1 <= d < 2^31 (device: common.h fastdiv)
inline void fastdiv_magic(unsigned d, unsigned* mg, unsigned* sh) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  const unsigned long long num = 1ull << (31 + l);
  *mg = (unsigned)((num + d - 1) / d);
  *sh = 31 + l;
}

// ---- host-side 16-bit conversions (round to nearest even) ---------------------
inline uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

inline uint16_t f32_to_f16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x0200u : 0u));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          // overflows to inf after rounding
  if (x < 0x33000001u) return (uint16_t)sign;                        // rounds to zero
  int exp = (int)(x >> 23) - 127;
  uint32_t man = (x & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t hexp;
  if (exp < -14) { shift = 13 + (-14 - exp); hexp = 0; }            // subnormal half
  else { shift = 13; hexp = (uint32_t)(exp + 15); }
  const uint32_t lsb = 1u << shift, half = lsb >> 1;
  uint32_t q = man >> shift;
  const uint32_t rem = man & (lsb - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  uint32_t h;
  if (hexp == 0) h = q;                                              // may carry into exponent 1: fine
  else h = ((hexp - 1) << 10) + q;                                   // q includes the implicit bit (0x400)
  return (uint16_t)(sign | h);
}

inline uint16_t cvt16(float f, int dtype) { return dtype == FLOPE_DT_BF16_ ? f32_to_bf16(f) : f32_to_f16(f); }


// ---- weight packing -----------------------------------------------------------
// Row order inside a 64-channel wave range (NT = 4 channel tiles): LDS row ct*16 + r
// holds channel (r>>2)*16 + ct*4 + (r&3), so that MFMA D rows 4g..4g+3 of tile ct are
// channels 16g + 4ct .. +3 and a lane's 16 accumulators are 16 consecutive channels.
inline int lds_row_to_channel(int rl) {
  const int range = rl / 64, in = rl % 64, ct = in / 16, r = in % 16;
  return range * 64 + (r >> 2) * 16 + ct * 4 + (r & 3);
}

// folded conv weights wf[cout][cin][k][k] -> [ntile][chunk*ntaps + tap][BN rows][64 k] images
inline std::vector<uint16_t> pack_conv(const std::vector<float>& wf, int cout, int cin, int k, int dtype) {
  const int BN = cout == 64 ? 64 : 128, ntiles = cout / BN, ntaps = k * k, nchunks = cin / 64;
  std::vector<uint16_t> out((size_t)cout * cin * ntaps);
  for (int nt = 0; nt < ntiles; ++nt)
    for (int ch = 0; ch < nchunks; ++ch)
      for (int tap = 0; tap < ntaps; ++tap) {
        const size_t tile = ((size_t)nt * nchunks * ntaps + (size_t)ch * ntaps + tap) * BN * 64;
        const int ky = tap / k, kx = tap % k;
        for (int rl = 0; rl < BN; ++rl) {
          const int co = nt * BN + lds_row_to_channel(rl);
          for (int kk = 0; kk < 64; ++kk) {
            const int ci = ch * 64 + kk;
            const float v = wf[(((size_t)co * cin + ci) * k + ky) * k + kx];
            const int slot = (kk >> 3) ^ ((rl >> 1) & 7);
            out[tile + (size_t)rl * 64 + slot * 8 + (kk & 7)] = cvt16(v, dtype);
          }
        }
      }
  return out;
}

// Row order of the conv_stag / conv_gstag images (r02).  Same idea as lds_row_to_channel, but a lane's 16 accumulators are TWO
// runs of 8 consecutive channels, 32 channels apart: MFMA rows 4g..4g+3 of channel tile ct are channels
// (ct >> 1) * 32 + 8 g + (ct & 1) * 4 .. + 3.  The epilogue's two 16-byte stores (and residual loads) of the four lane groups
// of a pixel then cover bytes [0, 64) and [64, 128) of the pixel's 128-byte channel block contiguously -- 16 requests of 64 bytes
// per wave-instruction instead of 64 pieces of 16 bytes at a 32-byte stride (measured: -1.5 ... -2.6 % per conv launch).
inline int stag_row_to_channel(int rl) {
  const int range = rl / 64, in = rl % 64, ct = in / 16, r = in % 16, g = r >> 2, q = r & 3;
  return range * 64 + (ct >> 1) * 32 + g * 8 + (ct & 1) * 4 + q;
}

// conv_stag image: folded 3x3 weights wf[cout][cin][3][3] -> [ntile (cout/BN)][hc*9 + tap][BN rows][32 k], BN = min(cout,128),
// 64-byte rows, 16-byte slot g of row r at g ^ h[(r>>2)&3], h = {0,2,3,1}; rows permuted by stag_row_to_channel
inline std::vector<uint16_t> pack_conv32(const std::vector<float>& wf, int cout, int cin, int dtype) {
  static const int h[4] = {0, 2, 3, 1};
  const int BN = cout == 64 ? 64 : 128, ntiles = cout / BN, nhc = cin / 32;
  std::vector<uint16_t> out((size_t)cout * cin * 9);
  for (int nt = 0; nt < ntiles; ++nt)
    for (int hc = 0; hc < nhc; ++hc)
      for (int tap = 0; tap < 9; ++tap) {
        const size_t tile = ((size_t)nt * nhc * 9 + (size_t)hc * 9 + tap) * BN * 32;
        const int ky = tap / 3, kx = tap % 3;
        for (int rl = 0; rl < BN; ++rl) {
          const int co = nt * BN + stag_row_to_channel(rl);
          for (int kk = 0; kk < 32; ++kk) {
            const float v = wf[(((size_t)co * cin + hc * 32 + kk) * 3 + ky) * 3 + kx];
            const int slot = (kk >> 3) ^ h[(rl >> 2) & 3];
            out[tile + (size_t)rl * 32 + slot * 8 + (kk & 7)] = cvt16(v, dtype);
          }
        }
      }
  return out;
}

// 1x1 weights wf[cout][cin] in the conv_stag image: [ntile (cout/128)][hc][128 rows][32 k] (one tap per half-chunk)
inline std::vector<uint16_t> pack_conv32_1x1(const std::vector<float>& wf, int cout, int cin, int dtype) {
  static const int h[4] = {0, 2, 3, 1};
  const int BN = 128, ntiles = cout / BN, nhc = cin / 32;
  std::vector<uint16_t> out((size_t)cout * cin);
  for (int nt = 0; nt < ntiles; ++nt)
    for (int hc = 0; hc < nhc; ++hc) {
      const size_t tile = ((size_t)nt * nhc + hc) * BN * 32;
      for (int rl = 0; rl < BN; ++rl) {
        const int co = nt * BN + stag_row_to_channel(rl);
        for (int kk = 0; kk < 32; ++kk) {
          const int slot = (kk >> 3) ^ h[(rl >> 2) & 3];
          out[tile + (size_t)rl * 32 + slot * 8 + (kk & 7)] = cvt16(wf[(size_t)co * cin + hc * 32 + kk], dtype);
        }
      }
    }
  return out;
}

// stem weights wf[64][3][7][7] -> [7 ky][64 rows][32 k = kx*4 + c] images (64-byte rows,
// slot g of row r at g ^ h[(r>>2)&3], h = {0,2,3,1})
inline std::vector<uint16_t> pack_stem(const std::vector<float>& wf, int dtype) {
  static const int h[4] = {0, 2, 3, 1};
  std::vector<uint16_t> out((size_t)7 * 64 * 32, cvt16(0.f, dtype));
  for (int ky = 0; ky < 7; ++ky)
    for (int rl = 0; rl < 64; ++rl) {
      const int co = lds_row_to_channel(rl);
      for (int kk = 0; kk < 32; ++kk) {
        const int kx = kk >> 2, c = kk & 3;
        const float v = (kx < 7 && c < 3) ? wf[(((size_t)co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
        const int slot = (kk >> 3) ^ h[(rl >> 2) & 3];
        out[((size_t)ky * 64 + rl) * 32 + slot * 8 + (kk & 7)] = cvt16(v, dtype);
      }
    }
  return out;
}

// stem weights for stem_pool_r_kernel (r05): wave w owns channels 16 w .. 16 w + 15 and keeps its seven A fragments in registers:
// [4 waves][7 ky][64 lanes][8]  <-  W[16 w + (lane & 15)][ky][k = 8 (lane >> 4) + j],  k = kx * 4 + c  (zero for kx = 7 or c = 3)
inline std::vector<uint16_t> pack_stem_frag(const std::vector<float>& wf, int dtype) {
  std::vector<uint16_t> out((size_t)4 * 7 * 64 * 8, cvt16(0.f, dtype));
  for (int w = 0; w < 4; ++w)
    for (int ky = 0; ky < 7; ++ky)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int co = 16 * w + (lane & 15), kk = 8 * (lane >> 4) + j, kx = kk >> 2, c = kk & 3;
          const float v = (kx < 7 && c < 3) ? wf[(((size_t)co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
          out[(((size_t)w * 7 + ky) * 64 + lane) * 8 + j] = cvt16(v, dtype);
        }
  return out;
}

// conv_s2r (layer2.0.conv1, r05): wave w owns output channels 32 w .. 32 w + 31 as two MFMA row tiles ct and reads its A fragments
// from global memory, one 16-byte load per lane:  [4 waves][18 steps = half-chunk * 9 + tap][2 ct][64 lanes][8]  <-
// W[32 w + 8 (i >> 2) + 4 ct + (i & 3)][ci = 32 hc + 8 (lane >> 4) + j][tap],  i = lane & 15  -- MFMA D rows 4 g .. 4 g + 3 of tile ct
// are then channels 32 w + 8 g + 4 ct .. + 3, so a lane's eight outputs of a pixel are eight consecutive channels (one 16-byte store).
inline std::vector<uint16_t> pack_s2r(const std::vector<float>& wf, int cout, int cin, int dtype) {
  std::vector<uint16_t> out((size_t)4 * 18 * 2 * 64 * 8);
  for (int w = 0; w < 4; ++w)
    for (int st = 0; st < 18; ++st)
      for (int ct = 0; ct < 2; ++ct)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int i = lane & 15, hc = st / 9, tap = st % 9, ky = tap / 3, kx = tap % 3;
            const int co = 32 * w + 8 * (i >> 2) + 4 * ct + (i & 3), ci = 32 * hc + 8 * (lane >> 4) + j;
            (void)cout;
            out[((((size_t)w * 18 + st) * 2 + ct) * 64 + lane) * 8 + j] = cvt16(wf[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx], dtype);
          }
  return out;
}

// conv_s1r (layer2.1.conv1 / conv2, r05): wave (cg, kh) owns output channels 32 cg .. 32 cg + 31 and input channels 64 kh .. 64 kh + 63:
// [4 cg][2 kh][18 steps = half-chunk * 9 + tap][2 ct][64 lanes][8]  <-
// W[32 cg + 8 (i >> 2) + 4 ct + (i & 3)][ci = 64 kh + 32 hc + 8 (lane >> 4) + j][tap],  i = lane & 15   (row order as pack_s2r)
inline std::vector<uint16_t> pack_s1r(const std::vector<float>& wf, int cin, int dtype) {
  std::vector<uint16_t> out((size_t)4 * 2 * 18 * 2 * 64 * 8);
  for (int cg = 0; cg < 4; ++cg)
    for (int kh = 0; kh < 2; ++kh)
      for (int st = 0; st < 18; ++st)
        for (int ct = 0; ct < 2; ++ct)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int i = lane & 15, hc = st / 9, tap = st % 9, ky = tap / 3, kx = tap % 3;
              const int co = 32 * cg + 8 * (i >> 2) + 4 * ct + (i & 3), ci = 64 * kh + 32 * hc + 8 * (lane >> 4) + j;
              out[(((((size_t)cg * 2 + kh) * 18 + st) * 2 + ct) * 64 + lane) * 8 + j] = cvt16(wf[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx], dtype);
            }
  return out;
}

// conv_s1r with the folded 1x1 stride-2 shortcut (layer2.0.conv2): one more A fragment pair per wave,
// [4 cg][2 kh][2 ct][64 lanes][8]  <-  W_ds[32 cg + 8 (i >> 2) + 4 ct + (i & 3)][ci = 32 kh + 8 (lane >> 4) + j]   (64 input channels)
inline std::vector<uint16_t> pack_s1r_ds(const std::vector<float>& wf, int cin, int dtype) {
  std::vector<uint16_t> out((size_t)4 * 2 * 2 * 64 * 8);
  for (int cg = 0; cg < 4; ++cg)
    for (int kh = 0; kh < 2; ++kh)
      for (int ct = 0; ct < 2; ++ct)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int i = lane & 15, co = 32 * cg + 8 * (i >> 2) + 4 * ct + (i & 3), ci = 32 * kh + 8 * (lane >> 4) + j;
            out[((((size_t)cg * 2 + kh) * 2 + ct) * 64 + lane) * 8 + j] = cvt16(wf[(size_t)co * cin + ci], dtype);
          }
  return out;
}

inline std::vector<float> naive_layout(const std::vector<float>& wf, int cout, int cin, int k) {
  std::vector<float> out((size_t)cout * cin * k * k);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
          out[(((size_t)ky * k + kx) * cin + ci) * cout + co] = wf[(((size_t)co * cin + ci) * k + ky) * k + kx];
  return out;
}


// fc.0 weights W1 [N][K] (N % 16 == 0, K % 32 == 0) in the A-fragment order of fc1_packed_kernel:
// [n tile][K / 32][2][lane = g * 16 + r][4]  <-  W1[(tile * 16 + r) * K + kb * 32 + h * 16 + 4 g + s]
inline std::vector<float> pack_fc1(const float* W1, int N, int K) {
  std::vector<float> out((size_t)N * K);
  const int KB = K / 32;
  for (int t = 0; t < N / 16; ++t)
    for (int kb = 0; kb < KB; ++kb)
      for (int h = 0; h < 2; ++h)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s)
            out[((((size_t)t * KB + kb) * 2 + h) * 64 + lane) * 4 + s] =
                W1[(size_t)(t * 16 + (lane & 15)) * K + kb * 32 + h * 16 + 4 * (lane >> 4) + s];
  return out;
}

}  // namespace flope_host
