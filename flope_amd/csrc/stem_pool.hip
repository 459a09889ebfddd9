// Fused front of the trunk: crop batch (any FLOPE_IN_* format) -> Conv2d(3,64,k7,s2,p3) + folded
// BatchNorm + ReLU -> MaxPool2d(k3,s2,p1) -> zero-bordered NHWC [B][Hq+2][Wq+2][64].
// Reference call site: sunflower/models/posenet.py:25 -> torchvision ResNet._forward_impl
// (conv1, bn1, relu, maxpool).
//
// Why fused: unfused, the 112x112x64 stem activation is written (205 MB at B=256) and read back
// by the pool, and a separate pass converts the input -- ~450 MB of HBM traffic and two launch
// boundaries around a kernel whose real output is 51 MB.  Here one workgroup owns an 8x8 tile of
// POOLED pixels: it converts the 39x40 input window it needs straight from the caller's tensor
// into LDS (4-channel pixels, zero outside the image), runs the 17x17 conv outputs that feed its
// pool windows on MFMA (K = 7 kernel rows x [8 px x 4 ch]; see stem.hip), parks them post-ReLU in
// LDS (aliasing the weight/input staging), max-pools there and writes 16-byte NHWC vectors.
// The one-pixel conv halo is recomputed by the neighbour tile (289 conv px per 256 consumed).
#include "common.h"
#include "host_pack.h"
#include <type_traits>

#define GLDS16(gptr, lptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct StemPoolP {
  const void* x;       // crop batch in in_format
  void* out;           // padded NHWC [B][Hq+2][Wq+2][64]
  const void* w;       // packed [7 ky][64 rows][32 k] LDS image (host_pack.h pack_stem)
  const void* w2;      // the same weights in per-wave A-fragment order (host_pack.h pack_stem_frag): stem_pool_r_kernel
  int* q;              // stem_pool_r_kernel: eight tile-queue heads + a count of finished workgroups, 256 bytes apart (one word saturates
                       // at ~88 pulls / us, and heads on ONE line behaved like one word: 10 us per pull); zero between launches
  const float* bias;   // [64]
  int in_format;       // FLOPE_IN_*
  int B, H, W;         // crop size
  int Hs, Ws;          // conv output size
  int Hq, Wq;          // pooled output size
  int tiles_y, tiles_x;
  unsigned mg_tx, sh_tx, mg_ty, sh_ty;   // n / tiles_x and n / tiles_y as multiply-shift (host_pack.h fastdiv_magic)
#ifdef FLOPE_STAG_DBG
  unsigned long long* dbg;   // diagnostic build: per workgroup {cycles in 5 phases, tiles, total} (tools/clock_probe_stem.py)
#endif
};

template <typename T> __device__ __forceinline__ u32x2 pack4(float a, float b, float c) {
  return u32x2{pack2<T>(a, b), pack2<T>(c, 0.f)};
}

template <typename T>
__global__ __launch_bounds__(256, 3) void stem_pool_kernel(const StemPoolP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int W_BYTES = 7 * 64 * 64;            // 28672
  constexpr int PR = 39, PC = 42;                 // input window rows x (40 used + 2 pad) cols, 8 B per pixel;
                                                  // the 21-slot row pitch keeps the column tiles at <= 2-way conflicts
  constexpr int P_BYTES = ((PR * PC * 8 + 15) / 16) * 16;
  constexpr int CR = 17;                          // conv region is CR x CR
  constexpr int NQ = CR * CR;                     // 289 conv pixels -> 19 MFMA pixel tiles
  constexpr int MT = 5, NT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ws = smem;
  char* const Ps = smem + W_BYTES;
  char* const Cs = smem;                          // [NQ][64 ch] conv outputs, aliases Ws|Ps after the MFMA phase

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = lid % p.tiles_x; lid /= p.tiles_x;
  const int ty = lid % p.tiles_y;
  const int img = lid / p.tiles_y;
  const int cr0 = 2 * (ty * 8) - 1, cc0 = 2 * (tx * 8) - 1;       // conv coords of region (0,0)
  const int py0 = 2 * cr0 - 3, px0 = 2 * cc0 - 3;                  // input coords of window (0,0)

  // ---- weights: linear LDS-DMA copy (28 pieces of 1 KiB; wave w moves pieces w, w+4, ...)
  for (int i = wave; i < W_BYTES / 1024; i += 4) GLDS16((const char*)p.w + i * 1024 + lane * 16, Ws + i * 1024);

  // ---- input window: convert from the caller's layout, zero outside the image.  Thread t < 252 owns window column
  // t % 42 and rows t / 42 + 6k (k = 0..6): one divide per thread, the column test / clamp once, and the LDS slot of
  // item k is simply t + 252 k.  All loads of a thread are issued before the first conversion (7 x 3 independent
  // loads in flight) -- a load->convert->store loop exposes the HBM latency once per iteration.  Vector-ALU work
  // here is paid in MFMA issue time (they share the SIMD's issue slots), hence the care.
  if (tid < 6 * PC) {
    constexpr int NI = 7;
    const int r0 = tid / PC, c = tid - r0 * PC;
    const int x = px0 + c;
    const bool okx = x >= 0 && x < p.W;
    const int estep = p.in_format == 0 ? 1 : 3;
    const int xoffs = min(max(x, 0), p.W - 1) * estep;
    int off[NI];                                              // element offset inside this image (< 2^31)
    bool ok[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int r = r0 + 6 * k, y = py0 + r;
      ok[k] = okx && r < PR && y >= 0 && y < p.H;
      off[k] = min(max(y, 0), p.H - 1) * p.W * estep + xoffs;  // always a legal address
    }
    const size_t img_elems = (size_t)3 * p.H * p.W;
    u32x2 px[NI];                                             // the 8-byte LDS pixels: c0 c1 | c2 0
#if defined(FLOPE_STEM_ABL) && (FLOPE_STEM_ABL & 1)      // timing experiments only
    for (int k = 0; k < NI; ++k) px[k] = pack4<T>(0.5f, 0.5f, 0.5f);
    if (false) {
#else
    if (p.in_format == 0) {
#endif
      const float* s = (const float*)p.x + (size_t)img * img_elems;
      const int plane = p.H * p.W;
      float v0[NI], v1[NI], v2[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) { v0[k] = s[off[k]]; v1[k] = s[off[k] + plane]; v2[k] = s[off[k] + 2 * plane]; }
#pragma unroll
      for (int k = 0; k < NI; ++k) px[k] = pack4<T>(v0[k], v1[k], v2[k]);
    } else if (p.in_format == 3) {
      const unsigned char* s = (const unsigned char*)p.x + (size_t)img * img_elems;
      unsigned char a[NI], b[NI], cc[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) { a[k] = s[off[k]]; b[k] = s[off[k] + 1]; cc[k] = s[off[k] + 2]; }
#pragma unroll
      for (int k = 0; k < NI; ++k) px[k] = pack4<T>((float)a[k] / 255.0f, (float)b[k] / 255.0f, (float)cc[k] / 255.0f);
    } else {
      const unsigned short* s = (const unsigned short*)p.x + (size_t)img * img_elems;
      unsigned short a[NI], b[NI], cc[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) { a[k] = s[off[k]]; b[k] = s[off[k] + 1]; cc[k] = s[off[k] + 2]; }
      const bool same = (p.in_format == 1) == (sizeof(T) == 2 && std::is_same<T, bf16_t>::value);
      if (same) {                                             // input dtype == trunk dtype: the bits pass through
#pragma unroll
        for (int k = 0; k < NI; ++k) px[k] = u32x2{(unsigned)a[k] | ((unsigned)b[k] << 16), (unsigned)cc[k]};
      } else if (p.in_format == 1) {
#pragma unroll
        for (int k = 0; k < NI; ++k)
          px[k] = pack4<T>(to_f32(__builtin_bit_cast(bf16_t, a[k])), to_f32(__builtin_bit_cast(bf16_t, b[k])),
                           to_f32(__builtin_bit_cast(bf16_t, cc[k])));
      } else {
#pragma unroll
        for (int k = 0; k < NI; ++k)
          px[k] = pack4<T>(to_f32(__builtin_bit_cast(f16_t, a[k])), to_f32(__builtin_bit_cast(f16_t, b[k])),
                           to_f32(__builtin_bit_cast(f16_t, cc[k])));
      }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (r0 + 6 * k < PR) *(u32x2*)(Ps + (tid + 6 * PC * k) * 8) = ok[k] ? px[k] : u32x2{0u, 0u};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- MFMA: wave w owns pixel tiles 5w .. 5w+4 of the 19 that cover the 17x17 conv region:
  // tile t < 17 = conv row t, columns 0..15 (16 consecutive 16-byte windows: conflict-free reads);
  // tile 17 = column 16 of rows 0..15; tile 18 = the corner (16,16).  Tile 19 does not exist.
  int xo[MT], qidx[MT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int t = wave * MT + pt;
    int qr, qc;
    if (t < CR) { qr = t; qc = r16; }
    else if (t == CR) { qr = r16; qc = CR - 1; }
    else { qr = CR - 1; qc = CR - 1; }
    qidx[pt] = (t < CR + 1 || (t == CR + 1 && r16 == 0)) ? qr * CR + qc : -1;
    xo[pt] = (2 * qr * PC + 2 * qc) * 8 + g * 16;
  }
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;       // 64-byte weight rows: slot g at g ^ h[(r>>2)&3]
  const int wo_ = r16 * 64 + ((g ^ wsw) << 4);
  f32x4 acc[MT][NT];                                      // accumulators start at the folded-BN bias
  {
    f32x4 b4[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) b4[ct] = *(const f32x4*)(p.bias + g * 16 + ct * 4);
#pragma unroll
    for (int pt = 0; pt < MT; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = b4[ct];
  }
#if defined(FLOPE_STEM_ABL) && (FLOPE_STEM_ABL & 2)
  for (int ky = 0; ky < 0; ++ky) {
#else
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
#endif
    frag wf[NT], xf[MT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(Ws + ky * 4096 + ct * 1024 + wo_);
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) xf[pt] = *(const frag*)(Ps + xo[pt] + ky * (PC * 8));
#pragma unroll
    for (int pt = 0; pt < MT; ++pt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], acc[pt][ct]);
  }
  __syncthreads();                                          // everyone is done reading Ws / Ps

  // ---- ReLU -> Cs, packed (cvt_pk + one v_pk_max_i16 per channel pair).  Conv positions outside the feature
  // map (only the -1 row / column of top / left tiles, or the tail of a ragged last tile) are forced to 0:
  // neutral for a max over ReLU outputs.  The test is wave-uniform per block edge, so interior tiles skip it.
  {
    const bool edge = cr0 < 0 || cc0 < 0 || cr0 + CR > p.Hs || cc0 + CR > p.Ws;
#pragma unroll
    for (int pt = 0; pt < MT; ++pt) {
      const int q = qidx[pt];
      if (q >= 0) {
        u32x4 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            const int c = h * 8 + w2 * 2;                   // channel pair c, c+1 of this lane's 16
            o[h][w2] = pk_relu16<T>(pack2<T>(acc[pt][c >> 2][c & 3], acc[pt][(c + 1) >> 2][(c + 1) & 3]));
          }
        if (edge) {
          const int qr = q / CR, qc = q - qr * CR;
          const int cr = cr0 + qr, cc = cc0 + qc;
          if (!(cr >= 0 && cr < p.Hs && cc >= 0 && cc < p.Ws)) { o[0] = u32x4{0u, 0u, 0u, 0u}; o[1] = o[0]; }
        }
        // row q, 16-byte slots 2g and 2g+1, stored at slot ^ (q & 7)
        *(u32x4*)(Cs + q * 128 + (((2 * g) ^ (q & 7)) << 4)) = o[0];
        *(u32x4*)(Cs + q * 128 + (((2 * g + 1) ^ (q & 7)) << 4)) = o[1];
      }
    }
  }
  __syncthreads();

  // ---- 3x3 / s2 max-pool out of LDS: item = (pooled pixel, 8-channel group)
  for (int i = tid; i < 64 * 8; i += 256) {
    const int cg = i & 7, pp = i >> 3;
    const int pr = pp >> 3, pc = pp & 7;
    const int oy = ty * 8 + pr, ox = tx * 8 + pc;
    if (oy >= p.Hq || ox >= p.Wq) continue;
    u32x4 o = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int q = (2 * pr + dy) * CR + 2 * pc + dx;
        const u32x4 v = *(const u32x4*)(Cs + q * 128 + ((cg ^ (q & 7)) << 4));
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pk_max16_nonneg(o[k], v[k]);
      }
    char* dst = (char*)p.out + ((((size_t)img * (p.Hq + 2) + oy + 1) * (p.Wq + 2) + ox + 1) * 64 + cg * 8) * 2;
#if defined(FLOPE_STEM_ABL) && (FLOPE_STEM_ABL & 4)
    if (o[0] == 0x12345678u)
#endif
    *(u32x4*)dst = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form (r02, default).  The r02 ablations of the kernel above (B = 256, 141 us on that box): without the MFMA
// phase 92 us, without MFMA + input loads + stores still 82 us -- the skeleton, not the arithmetic, was the cost, and its
// largest item was the 28 KB weight image that each of the 12,544 workgroups streams L2 -> LDS for itself (351 MB per
// launch at the chip's ~7 TB/s DMA rate).  Here a grid of two workgroups per CU walks the tiles: the weights are staged
// once per workgroup, the conv-output buffer no longer aliases them (LDS = 28 + 13 + 37 KB), the input window of the
// NEXT tile is loaded into registers before the MFMA phase and written to LDS after it (its latency hides behind the
// MFMAs, and LDS is written while the other phase's buffer is idle), and a tile costs two barriers instead of four.
// FMT = the crop batch's FLOPE_IN_* format as a template argument: with a runtime branch around the window loads the
// compiler has to merge the branches' destination registers right behind them, i.e. waits for the loads BEFORE the MFMA
// phase -- the prefetch became a stall (measured: 166 us instead of 143).
template <typename T, int FMT>
__global__ __launch_bounds__(256, 2) void stem_pool_persist_kernel(const StemPoolP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int W_BYTES = 7 * 64 * 64;
  constexpr int PR = 39, PC = 42;
  constexpr int P_BYTES = ((PR * PC * 8 + 15) / 16) * 16;
  constexpr int CR = 17, NQ = CR * CR;
  constexpr int MT = 5, NT = 4, NI = 7;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ws = smem;
  char* const Ps = smem + W_BYTES;
  char* const Cs = smem + W_BYTES + P_BYTES;       // [NQ][64 ch] conv outputs (own region: Ws stays resident)
  (void)NQ;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int G = gridDim.x, total = p.B * p.tiles_y * p.tiles_x;
  const int lb = xcd_remap(blockIdx.x, G);

  for (int i = wave; i < W_BYTES / 1024; i += 4) GLDS16((const char*)p.w + i * 1024 + lane * 16, Ws + i * 1024);

  // ---- staging geometry of this thread (tile independent): window column c, rows r0 + 6k
  // every thread runs the loads (threads 252..255 repeat thread 251's: no divergent region around them), 252 store
  const bool stager = tid < 6 * PC;
  const int st = min(tid, 6 * PC - 1);
  const int r0 = st / PC, c = st - r0 * PC;
  constexpr int estep = FMT == 0 ? 1 : 3;
  const size_t img_elems = (size_t)3 * p.H * p.W;
  const int plane = p.H * p.W;
  // the next tile's window exactly as the load instructions deliver it (f32: three dwords; 16-bit: one dword c0|c1 and one
  // short c2; u8: one short c0|c1 and one byte c2) -- any unpacking here would have to wait for the loads, before the MFMAs
  unsigned raw[NI][3];
  unsigned okmask = 0;
  unsigned rowmask = 0;                              // rows r0 + 6k that exist in the 39-row window
#pragma unroll
  for (int k = 0; k < NI; ++k) if (r0 + 6 * k < PR) rowmask |= 1u << k;
  auto tile_origin = [&](int tile, int& tx, int& ty, int& img) {     // (multiply-shift: two runtime divisions were ~80 instructions per tile)
    const int q = fastdiv(tile, p.mg_tx, p.sh_tx);
    tx = tile - q * p.tiles_x;
    img = fastdiv(q, p.mg_ty, p.sh_ty);
    ty = q - img * p.tiles_y;
  };
  auto issue_loads = [&](int tile) {
    int tx, ty, img;
    tile_origin(tile, tx, ty, img);
    const int py0 = 2 * (2 * (ty * 8) - 1) - 3, px0 = 2 * (2 * (tx * 8) - 1) - 3;
    const int x = px0 + c;
    int off[NI];
    // a window that lies inside the image (wave-uniform test; about half of the tiles of a 224 x 224 crop) needs neither the
    // clamps nor the per-row masks: one multiply, six adds
    if (py0 >= 0 && py0 + PR <= p.H && px0 >= 0 && px0 + PC <= p.W) {
      okmask = rowmask;
      const int rs = 6 * p.W * estep;
      off[0] = ((py0 + r0) * p.W + x) * estep;
#pragma unroll
      for (int k = 1; k < NI; ++k) off[k] = off[k - 1] + rs;
      if (r0 + 6 * (NI - 1) >= PR) off[NI - 1] = off[NI - 2];     // the row behind the window: any legal address
    } else {
      const bool okx = x >= 0 && x < p.W;
      const int xoffs = min(max(x, 0), p.W - 1) * estep;
      okmask = 0;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int r = r0 + 6 * k, y = py0 + r;
        if (okx && r < PR && y >= 0 && y < p.H) okmask |= 1u << k;
        off[k] = min(max(y, 0), p.H - 1) * p.W * estep + xoffs;
      }
    }
    if constexpr (FMT == 0) {
      const float* s = (const float*)p.x + (size_t)img * img_elems;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        raw[k][0] = __builtin_bit_cast(unsigned, s[off[k]]); raw[k][1] = __builtin_bit_cast(unsigned, s[off[k] + plane]);
        raw[k][2] = __builtin_bit_cast(unsigned, s[off[k] + 2 * plane]);
      }
    } else if constexpr (FMT == 3) {
      const unsigned char* s = (const unsigned char*)p.x + (size_t)img * img_elems;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        unsigned short w01; __builtin_memcpy(&w01, s + off[k], 2);
        raw[k][0] = w01; raw[k][1] = s[off[k] + 2];
      }
    } else {
      const unsigned short* s = (const unsigned short*)p.x + (size_t)img * img_elems;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        unsigned w01; __builtin_memcpy(&w01, s + off[k], 4);
        raw[k][0] = w01; raw[k][1] = s[off[k] + 2];
      }
    }
  };
  auto write_window = [&]() {
    if (!stager) return;
    constexpr bool same = (FMT == 1) == std::is_same<T, bf16_t>::value;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      u32x2 px;
      if constexpr (FMT == 0)
        px = pack4<T>(__builtin_bit_cast(float, raw[k][0]), __builtin_bit_cast(float, raw[k][1]), __builtin_bit_cast(float, raw[k][2]));
      else if constexpr (FMT == 3)
        px = pack4<T>((float)(raw[k][0] & 0xffu) / 255.0f, (float)(raw[k][0] >> 8) / 255.0f, (float)raw[k][1] / 255.0f);
      else if constexpr (same)
        px = u32x2{raw[k][0], raw[k][1]};
      else if constexpr (FMT == 1)
        px = pack4<T>(to_f32(__builtin_bit_cast(bf16_t, (unsigned short)(raw[k][0] & 0xffffu))), to_f32(__builtin_bit_cast(bf16_t, (unsigned short)(raw[k][0] >> 16))),
                      to_f32(__builtin_bit_cast(bf16_t, (unsigned short)raw[k][1])));
      else
        px = pack4<T>(to_f32(__builtin_bit_cast(f16_t, (unsigned short)(raw[k][0] & 0xffffu))), to_f32(__builtin_bit_cast(f16_t, (unsigned short)(raw[k][0] >> 16))),
                      to_f32(__builtin_bit_cast(f16_t, (unsigned short)raw[k][1])));
      if (r0 + 6 * k < PR) *(u32x2*)(Ps + (tid + 6 * PC * k) * 8) = ((okmask >> k) & 1u) ? px : u32x2{0u, 0u};
    }
  };

  // ---- MFMA geometry of this lane (tile independent)
  int xo[MT], qidx[MT];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int t = wave * MT + pt;
    int qr, qc;
    if (t < CR) { qr = t; qc = r16; }
    else if (t == CR) { qr = r16; qc = CR - 1; }
    else { qr = CR - 1; qc = CR - 1; }
    qidx[pt] = (t < CR + 1 || (t == CR + 1 && r16 == 0)) ? qr * CR + qc : -1;
    xo[pt] = (2 * qr * PC + 2 * qc) * 8 + g * 16;
  }
  const int wsw = (0x1320 >> ((r16 >> 2) * 4)) & 3;
  const int wo_ = r16 * 64 + ((g ^ wsw) << 4);
  f32x4 b4[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) b4[ct] = *(const f32x4*)(p.bias + g * 16 + ct * 4);

  // tile-independent LDS offsets (r03: the stamps say a wave's own instruction stream, not a shared unit, bounds this kernel):
  // where this lane parks its conv outputs, and the nine conv outputs under each of its two pool items
  int cso[MT][2];
#pragma unroll
  for (int pt = 0; pt < MT; ++pt) {
    const int q = qidx[pt] < 0 ? 0 : qidx[pt];
    cso[pt][0] = q * 128 + (((2 * g) ^ (q & 7)) << 4);
    cso[pt][1] = q * 128 + (((2 * g + 1) ^ (q & 7)) << 4);
  }
  int pro[2][9];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = tid + it * 256, cg = i & 7, pp = i >> 3, pr = pp >> 3, pc = pp & 7;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int q = (2 * pr + t / 3) * CR + 2 * pc + t % 3;
      pro[it][t] = q * 128 + ((cg ^ (q & 7)) << 4);
    }
  }
  int tile = lb;
  if (tile < total) { issue_loads(tile); write_window(); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef FLOPE_STAG_DBG
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime(), t_first = t_prev;
  int ntile_dbg = 0;
  unsigned long long ph_pre = 0;                   // part of phase 0 in front of the MFMA loop (tile decode + window loads issue)
  if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[8192 + (p.B & 1023)] = __builtin_amdgcn_s_memrealtime();   // launch start, by batch size (100 MHz clock): when does each slice's first kernel start?
#define ST_PH(i_) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i_] += t_ - t_prev; t_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ST_PH(i_) do {} while (0)
#endif

  for (; tile < total; tile += G) {
    int tx, ty, img;
    tile_origin(tile, tx, ty, img);
    const int cr0 = 2 * (ty * 8) - 1, cc0 = 2 * (tx * 8) - 1;
    const bool has_next = tile + G < total;
    if (has_next) issue_loads(tile + G);           // in flight during the MFMA phase
    asm volatile("" ::: "memory");
#ifdef FLOPE_STAG_DBG
    { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_pre += t_ - t_prev; __builtin_amdgcn_sched_barrier(0); }
#endif
    f32x4 acc[MT][NT];                               // row 0 takes the folded-BN bias as its C operand: no 80-register init
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      frag wf[NT], xf[MT];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) wf[ct] = *(const frag*)(Ws + ky * 4096 + ct * 1024 + wo_);
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) xf[pt] = *(const frag*)(Ps + xo[pt] + ky * (PC * 8));
#pragma unroll
      for (int pt = 0; pt < MT; ++pt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[pt][ct] = Elem<T>::mfma(wf[ct], xf[pt], ky == 0 ? b4[ct] : acc[pt][ct]);
    }
    ST_PH(0);
    __syncthreads();                               // Ps is free (and the previous tile's pool has finished with Cs)
    ST_PH(1);
    if (has_next) write_window();
    {
      const bool edge = cr0 < 0 || cc0 < 0 || cr0 + CR > p.Hs || cc0 + CR > p.Ws;
#pragma unroll
      for (int pt = 0; pt < MT; ++pt) {
        const int q = qidx[pt];
        if (q >= 0) {
          u32x4 o[2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
              const int ch = h * 8 + w2 * 2;
              o[h][w2] = pk_relu16<T>(pack2<T>(acc[pt][ch >> 2][ch & 3], acc[pt][(ch + 1) >> 2][(ch + 1) & 3]));
            }
          if (edge) {
            const int qr = q / CR, qc = q - qr * CR;
            const int cr = cr0 + qr, cc = cc0 + qc;
            if (!(cr >= 0 && cr < p.Hs && cc >= 0 && cc < p.Ws)) { o[0] = u32x4{0u, 0u, 0u, 0u}; o[1] = o[0]; }
          }
          *(u32x4*)(Cs + cso[pt][0]) = o[0];
          *(u32x4*)(Cs + cso[pt][1]) = o[1];
        }
      }
    }
    ST_PH(2);
    __syncthreads();
    ST_PH(3);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * 256;
      const int cg = i & 7, pp = i >> 3;
      const int pr = pp >> 3, pc = pp & 7;
      const int oy = ty * 8 + pr, ox = tx * 8 + pc;
      if (oy >= p.Hq || ox >= p.Wq) continue;
      u32x4 o = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const u32x4 v = *(const u32x4*)(Cs + pro[it][t]);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pk_max16_nonneg(o[k], v[k]);
      }
      char* dst = (char*)p.out + ((((size_t)img * (p.Hq + 2) + oy + 1) * (p.Wq + 2) + ox + 1) * 64 + cg * 8) * 2;
      *(u32x4*)dst = o;
    }
    ST_PH(4);
#ifdef FLOPE_STAG_DBG
    ++ntile_dbg;
#endif
  }
#ifdef FLOPE_STAG_DBG
  if (p.dbg && tid == 0) {
    unsigned long long* d_ = p.dbg + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 5; ++i) d_[i] = ph[i];
    d_[5] = (unsigned long long)ntile_dbg; d_[6] = __builtin_amdgcn_s_memtime() - t_first; d_[7] = ph_pre;
  }
#endif
#undef ST_PH
}


// ---------------------------------------------------------------------------------------------------------------
// Register-weight form (r05; VERDICT r4 item 5: "change the decomposition").  What the persistent kernel above leaves on the table
// (r03 / r04 stamps): a tile is a chain of phases -- MFMA 4.9 k cycles for 2.2 k of matrix work, window + ReLU + conv-output
// writes 2.6 k, pool 2.2 k -- on a CU that holds only TWO workgroups (78 KB of LDS, 210 registers), so whenever both are outside
// their MFMA phase the matrix pipe idles; and every pool read was a 2-way bank conflict (two pooled pixels of a 16-lane group sit
// two conv columns apart = two 128-byte rows of the same bank half: the 25 % replays the counters have shown since r02).
//   * The OUTPUT CHANNELS are split over the waves instead of the pixel tiles: wave w owns channels 16 w .. 16 w + 15 and keeps
//     its seven weight fragments (7 ky x 16 B per lane) in REGISTERS for the life of the workgroup -- no weight image in LDS
//     (28 KB), no weight reads in the loop (4 of every 9 fragment reads).  Every wave walks all 19 pixel tiles of the 17 x 17
//     conv region, ONE tile at a time: 7 fragment reads (one base register + immediate offsets), 7 MFMAs on one accumulator
//     tile, ReLU + pack, one 8-byte LDS write -- 4 live accumulator registers instead of 80.
//   * VERTICAL pooling happens in registers (a lane holds one conv column in every row tile): only the 8 pooled rows go to LDS.
//   * LDS = 13 KB window + 18 KB pooled rows + 2 KB conv column 16 = 34 KB, 141 registers: THREE workgroups per CU (a fourth fits
//     with 128 registers and six spills and measured the same: 121.5 vs 118.4 us; so did static wave priorities by co-resident
//     workgroup or by hardware wave slot, built to break a suspected convoy of the MFMA phases: 123.6 / 124.1 / 128.7 us).
//   * Pooled rows in LDS: pixel (pooled row pr, column c) at slot pr * 18 + pi(c), pi = evens then odds, so that the two conv columns a
//     16-lane pool read touches (c and c + 2) fall into different bank halves; 16-byte chunk cg of a pixel at cg ^ k(c) with k a
//     function of the column only (tile-independent write offsets), chosen so that the 16 pixels of a write instruction cover 16
//     distinct bank groups.  Pool reads and conv-output writes are conflict-free (column 16's once-per-tile writes: 2-way).
// Same arithmetic per output as the kernels above (bias as the first MFMA's C operand, ky = 0 .. 6 in order, k = kx * 4 + c):
// bit-identical (tests/test_gpu_parity.py).
__device__ __forceinline__ int stem_pi(int c) { return (c >> 1) + (c & 1) * 9; }
__device__ __forceinline__ int stem_key(int c, int t) { return c == 16 ? (t & 7) : (((c & 1) << 2) + ((c >> 2) & 3)); }

#ifndef FLOPE_STEM_WPE
#define FLOPE_STEM_WPE 3          // workgroups per CU the register-weight stem is compiled for (3: 141 registers; 4: 128 with 6 spilled -- measured equal)
#endif
template <typename T, int FMT>
__global__ __launch_bounds__(256, FLOPE_STEM_WPE) void stem_pool_r_kernel(const StemPoolP p) {
  typedef typename Elem<T>::frag frag;
  constexpr int PR = 39, PC = 42;
  constexpr int P_BYTES = ((PR * PC * 8 + 15) / 16) * 16;
  constexpr int CR = 17, CS = 18, NI = 7;                     // conv region 17 x 17; 18 pixel slots per row of the LDS image
  constexpr int V_BYTES = 8 * CS * 128;                       // vertically pooled rows: [8 pooled rows][18 slots][64 ch]
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FLOPE_STAG_DBG
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
  char* const Ps = smem;
  char* const Cs = smem + P_BYTES;                            // the pooled-row image, then [17 rows][64 ch] of conv column 16

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int G = gridDim.x, total = p.B * p.tiles_y * p.tiles_x;
  // Static walk first, work queues for the tail (r05 stamps: with a purely static partition the workgroups of a launch finished anywhere
  // between 61 and 98 us -- co-resident waves are not served equally -- and the CUs idled behind their fastest workgroups; with every
  // tile pulled from a queue the finish times met within 10 us but each pull cost ~1.8 k cycles of a tile).  The first nstat rounds of
  // the grid (3/4 of the tiles) are walked as before, tile = lb + j G; the rest sits in eight queues, one per XCD: queue x holds the
  // contiguous range [S + x R / 8, S + (x + 1) R / 8) (neighbouring tiles share window rows: same L2), a workgroup pulls from the queue
  // of the XCD it runs on (HW_REG_XCC_ID) and, once that is empty, from the others in turn.  p.q = eight heads + a count of finished
  // workgroups, 256 bytes apart; the last workgroup out zeroes them for the next launch.  A pull is issued at the start of a tile and
  // read behind its first barrier.
  const int lb = xcd_remap(blockIdx.x, G);
  const int nstat = (int)((long long)total * 3 / 4 / G);            // static rounds
  const int S0 = nstat * G, RT = total - S0;                          // queue region [S0, total)
  const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  int* const qn = (int*)(smem + P_BYTES + V_BYTES + CR * 128);      // one word of LDS: the next tile index, from thread 0 to everybody
  const int qlo = S0 + (int)((long long)xcc * RT / 8), qhi = S0 + (int)((long long)(xcc + 1) * RT / 8);
  auto pull_issue = [&]() -> int {                                  // (thread 0) the atomic only: its value is looked at a phase later
    return __hip_atomic_fetch_add(p.q + xcc * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto pull_resolve = [&](int raw) -> int {                         // (thread 0) own queue, or -- once it is empty -- the others in turn
    if (qlo + raw < qhi) return qlo + raw;
    for (int k = 1; k < 8; ++k) {
      const int x = (xcc + k) & 7;
      const int lo = S0 + (int)((long long)x * RT / 8), hi = S0 + (int)((long long)(x + 1) * RT / 8);
      if (lo >= hi) continue;
      const int t = lo + __hip_atomic_fetch_add(p.q + x * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t < hi) return t;
    }
    return total;
  };

  // this wave's weights: seven fragments + its bias, in fragment order in global memory (one contiguous KiB per wave-load).  They
  // take a detour through LDS (each lane parks its own 8 x 16 bytes and reads them back: no barrier in between): registers that a
  // GLOBAL load wrote in front of the loop make hipcc's wait-count pass put `s_waitcnt vmcnt(0)` in front of the loop's first MFMA --
  // i.e. behind the next window's fourteen loads, whose flight time the MFMA phase is there to hide.
  frag wf[7];
  f32x4 b4;
  {
    char* const wl = smem + wave * 8192 + lane * 16;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) *(frag*)(wl + ky * 1024) = *(const frag*)((const char*)p.w2 + ((size_t)(wave * 7 + ky) * 64 + lane) * 16);
    *(f32x4*)(wl + 7 * 1024) = *(const f32x4*)(p.bias + wave * 16 + g * 4);
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) wf[ky] = *(const frag*)(wl + ky * 1024);
    b4 = *(const f32x4*)(wl + 7 * 1024);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                               // the staging area is the window / pooled-row image from here on
  }

  // ---- staging geometry of this thread (tile independent): window column c, rows r0 + 6k   (as stem_pool_persist_kernel)
  const bool stager = tid < 6 * PC;
  const int st = min(tid, 6 * PC - 1);
  const int r0 = st / PC, c = st - r0 * PC;
  constexpr int estep = FMT == 0 ? 1 : 3;
  const size_t img_elems = (size_t)3 * p.H * p.W;
  const int plane = p.H * p.W;
  unsigned raw[NI][FMT == 0 ? 3 : 2];
  unsigned okmask = 0, rowmask = 0;
#pragma unroll
  for (int k = 0; k < NI; ++k) if (r0 + 6 * k < PR) rowmask |= 1u << k;
  auto tile_origin = [&](int tile, int& tx, int& ty, int& img) {
    const int q = fastdiv(tile, p.mg_tx, p.sh_tx);
    tx = tile - q * p.tiles_x;
    img = fastdiv(q, p.mg_ty, p.sh_ty);
    ty = q - img * p.tiles_y;
  };
  auto issue_loads = [&](int tile) {
    int tx, ty, img;
    tile_origin(tile, tx, ty, img);
    const int py0 = 2 * (2 * (ty * 8) - 1) - 3, px0 = 2 * (2 * (tx * 8) - 1) - 3;
    const int x = px0 + c;
    int off[NI];
    if (py0 >= 0 && py0 + PR <= p.H && px0 >= 0 && px0 + PC <= p.W) {
      okmask = rowmask;
      const int rs = 6 * p.W * estep;
      off[0] = ((py0 + r0) * p.W + x) * estep;
#pragma unroll
      for (int k = 1; k < NI; ++k) off[k] = off[k - 1] + rs;
      if (r0 + 6 * (NI - 1) >= PR) off[NI - 1] = off[NI - 2];
    } else {
      const bool okx = x >= 0 && x < p.W;
      const int xoffs = min(max(x, 0), p.W - 1) * estep;
      okmask = 0;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int r = r0 + 6 * k, y = py0 + r;
        if (okx && r < PR && y >= 0 && y < p.H) okmask |= 1u << k;
        off[k] = min(max(y, 0), p.H - 1) * p.W * estep + xoffs;
      }
    }
    if constexpr (FMT == 0) {
      const float* s = (const float*)p.x + (size_t)img * img_elems;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        raw[k][0] = __builtin_bit_cast(unsigned, s[off[k]]); raw[k][1] = __builtin_bit_cast(unsigned, s[off[k] + plane]);
        raw[k][2] = __builtin_bit_cast(unsigned, s[off[k] + 2 * plane]);
      }
    } else if constexpr (FMT == 3) {
      const unsigned char* s = (const unsigned char*)p.x + (size_t)img * img_elems;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        unsigned short w01; __builtin_memcpy(&w01, s + off[k], 2);
        raw[k][0] = w01; raw[k][1] = s[off[k] + 2];
      }
    } else {
      const unsigned short* s = (const unsigned short*)p.x + (size_t)img * img_elems;
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        unsigned w01; __builtin_memcpy(&w01, s + off[k], 4);
        raw[k][0] = w01; raw[k][1] = s[off[k] + 2];
      }
    }
  };
  auto write_window = [&]() {
    if (!stager) return;
    constexpr bool same = (FMT == 1) == std::is_same<T, bf16_t>::value;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      u32x2 px;
      if constexpr (FMT == 0)
        px = pack4<T>(__builtin_bit_cast(float, raw[k][0]), __builtin_bit_cast(float, raw[k][1]), __builtin_bit_cast(float, raw[k][FMT == 0 ? 2 : 0]));
      else if constexpr (FMT == 3)
        px = pack4<T>((float)(raw[k][0] & 0xffu) / 255.0f, (float)(raw[k][0] >> 8) / 255.0f, (float)raw[k][1] / 255.0f);
      else if constexpr (same)
        px = u32x2{raw[k][0], raw[k][1]};
      else if constexpr (FMT == 1)
        px = pack4<T>(to_f32(__builtin_bit_cast(bf16_t, (unsigned short)(raw[k][0] & 0xffffu))), to_f32(__builtin_bit_cast(bf16_t, (unsigned short)(raw[k][0] >> 16))),
                      to_f32(__builtin_bit_cast(bf16_t, (unsigned short)raw[k][1])));
      else
        px = pack4<T>(to_f32(__builtin_bit_cast(f16_t, (unsigned short)(raw[k][0] & 0xffffu))), to_f32(__builtin_bit_cast(f16_t, (unsigned short)(raw[k][0] >> 16))),
                      to_f32(__builtin_bit_cast(f16_t, (unsigned short)raw[k][1])));
      if (r0 + 6 * k < PR) *(u32x2*)(Ps + (tid + 6 * PC * k) * 8) = ((okmask >> k) & 1u) ? px : u32x2{0u, 0u};
    }
  };

  // ---- tile-independent LDS offsets of this lane
  // fragment reads: tile t < 17 = conv row t, columns r16 (+ t * 2 PC 8 as an immediate); tile 17 = column 16 of rows r16;
  // tile 18 = the corner (16, 16): every lane reads it, lane r16 == 0 owns it
  const int xb_row = (2 * r16) * 8 + g * 16;
  const int xb_col = (2 * r16 * PC + 2 * (CR - 1)) * 8 + g * 16;
  const int xb_cor = (2 * (CR - 1) * PC + 2 * (CR - 1)) * 8 + g * 16;
  // conv-output writes: this lane's 4 channels 16 w + 4 g .. of pixel (t, c): 8-byte granule (g & 1) of chunk cg = 2 w + (g >> 1)
  const int cgw = 2 * wave + (g >> 1), subw = (g & 1) * 8;
  const int cw_row = stem_pi(r16) * 128 + ((cgw ^ stem_key(r16, 0)) << 4) + subw;                       // + pooled row * CS * 128
  const int cw_col = V_BYTES + r16 * 128 + ((cgw ^ (r16 & 7)) << 4) + subw;                             // column 16, conv row r16
  const int cw_cor = V_BYTES + (CR - 1) * 128 + ((cgw ^ ((CR - 1) & 7)) << 4) + subw;
  // pool reads of this thread's two items (pooled pixel, 8-channel group): the three vertically pooled columns under it; the last
  // pooled column takes its third one from the three conv rows of column 16
  int pro[2][3], pc16[2][3];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = tid + it * 256, cg = i & 7, pp = i >> 3, pr = pp >> 3, pc = pp & 7;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int qc = min(2 * pc + t, CR - 2);
      pro[it][t] = (pr * CS + stem_pi(qc)) * 128 + ((cg ^ stem_key(qc, 0)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) pc16[it][t] = V_BYTES + (2 * pr + t) * 128 + ((cg ^ ((2 * pr + t) & 7)) << 4);
  }

  if (tid == 0) {
    int t0_ = lb, t1_ = lb + G;                                     // static rounds 0 and 1, or pulls where there are none
    if (nstat < 1) { const int r0_ = pull_issue(), r1_ = pull_issue(); t0_ = pull_resolve(r0_); t1_ = pull_resolve(r1_); }
    else if (nstat < 2) t1_ = pull_resolve(pull_issue());
    qn[0] = t0_; qn[1] = t1_;
  }
  __syncthreads();
  int tile = qn[0], tile_next = qn[1];
  int jt = 0;                                                       // tiles this workgroup has started
  if (tile < total) { issue_loads(tile); write_window(); }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef FLOPE_STAG_DBG
  // diagnostic build: cycles of wave 0 in {MFMA phase, barrier, window write + pool, barrier} per workgroup (tools/clock_probe_stem.py)
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime(), t_first = t_prev;
  const unsigned long long rt_first = __builtin_amdgcn_s_memrealtime();
  int ntile_dbg = 0;
#define ST_PH(i_) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i_] += t_ - t_prev; t_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ST_PH(i_) do {} while (0)
#endif

  for (; tile < total; tile = tile_next, tile_next = qn[0], ++jt) {
    int tx, ty, img;
    tile_origin(tile, tx, ty, img);
    const int cr0 = 2 * (ty * 8) - 1, cc0 = 2 * (tx * 8) - 1;
    const bool has_next = tile_next < total;
    if (has_next) issue_loads(tile_next);          // in flight during the MFMA phase
    const bool dyn2 = jt + 2 >= nstat;             // the tile after that: still a static round, or a pull (its round trip behind the MFMA phase)
    int praw = 0;
    if (tid == 0 && has_next && dyn2) praw = pull_issue();
    asm volatile("" ::: "memory");
    // Per row tile the wave's vector instructions, not its MFMAs, are what the SIMD runs out of (r05 counters: ~1000 instructions per
    // tile and wave around 133 MFMAs; three or four workgroups per CU, priorities, a one-barrier software pipeline: all the same
    // time), so the epilogue is cut to the bone:
    //   * ReLU and the float16 clamp commute with the maximum, and a SIGNED 16-bit integer maximum orders IEEE bit patterns correctly
    //     whenever the result is non-negative (and any negative result is a 0 after ReLU): the vertical pool runs on the raw packed
    //     conversions -- 2 cvt_pk + 2 v_pk_max_i16 per conv row -- and ReLU + clamp are applied once per POOLED row (same bits);
    //   * conv positions outside the feature map (the -1 row / column of top / left tiles, the tail of a ragged last tile) must
    //     contribute 0.  Only tiles on the border have any: the tile body exists twice, and the interior version carries no masks
    //     at all; the border version ANDs a per-lane column mask, selected per row by one scalar bit.  (Branch-free inside a body: a
    //     branch per pixel tile cuts the stream into basic blocks and the compiler then issues every fragment read right in front
    //     of its MFMA.)
    const bool edge = cr0 < 0 || cc0 < 0 || cr0 + CR > p.Hs || cc0 + CR > p.Ws;
    auto body = [&](auto edge_c) {
      constexpr bool EDGE = decltype(edge_c)::value;
      unsigned colm = 0xffffffffu, c16m = 0xffffffffu, corm = 0xffffffffu, rowbits = 0x1ffffu;
      if constexpr (EDGE) {
        colm = (cc0 + r16 >= 0 && cc0 + r16 < p.Ws) ? 0xffffffffu : 0u;
        c16m = (cc0 + CR - 1 >= 0 && cc0 + CR - 1 < p.Ws && cr0 + r16 >= 0 && cr0 + r16 < p.Hs) ? 0xffffffffu : 0u;
        corm = (cr0 + CR - 1 >= 0 && cr0 + CR - 1 < p.Hs && cc0 + CR - 1 >= 0 && cc0 + CR - 1 < p.Ws) ? 0xffffffffu : 0u;
        const int lo_ = max(0, -cr0), hi_ = min(CR, p.Hs - cr0);                  // valid conv rows of this tile: [lo_, hi_)
        rowbits = hi_ > lo_ ? ((1u << hi_) - 1u) & ~((1u << lo_) - 1u) : 0u;
      }
      auto raw_pack = [&](const f32x4 a_) { return u32x2{pack2<T>(a_[0], a_[1]), pack2<T>(a_[2], a_[3])}; };
      auto relu_clamp = [&](const u32x2 v_) { return u32x2{pk_relu16<T>(v_[0]), pk_relu16<T>(v_[1])}; };
      u32x2 vm = u32x2{0u, 0u};
      auto row_done = [&](int t, const f32x4 a_) {            // conv row t (compile-time)
        u32x2 o_ = raw_pack(a_);
        if constexpr (EDGE) {
          const unsigned m_ = ((rowbits >> t) & 1u) ? colm : 0u;
          o_ = u32x2{o_[0] & m_, o_[1] & m_};
        }
        if (t == 0) { vm = o_; return; }
        vm = u32x2{pk_max16_signed(vm[0], o_[0]), pk_max16_signed(vm[1], o_[1])};
        if ((t & 1) == 0) { *(u32x2*)(Cs + cw_row + (t / 2 - 1) * (CS * 128)) = relu_clamp(vm); vm = o_; }
      };
      // The 17 row tiles share window rows: conv row t reads rows 2 t .. 2 t + 6, so every row fragment is read ONCE (39 reads, not
      // 119), two new ones per tile, issued a whole tile ahead of their first MFMA; a tile's epilogue runs behind the NEXT tile's
      // MFMAs (its accumulators have long landed).  The column tile and the corner (7 reads each) go first.
      frag xr[PR];
#pragma unroll
      for (int r = 0; r < 7; ++r) xr[r] = *(const frag*)(Ps + xb_row + r * (PC * 8));
      f32x4 a_col, a_cor, a_prev;
      {
        frag xc[7], xk[7];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) { xc[ky] = *(const frag*)(Ps + xb_col + ky * (PC * 8)); xk[ky] = *(const frag*)(Ps + xb_cor + ky * (PC * 8)); }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) a_col = Elem<T>::mfma(wf[ky], xc[ky], ky == 0 ? b4 : a_col);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) a_cor = Elem<T>::mfma(wf[ky], xk[ky], ky == 0 ? b4 : a_cor);
      }
#pragma unroll
      for (int t = 0; t < CR; ++t) {
        if (2 * t + 7 < PR) xr[2 * t + 7] = *(const frag*)(Ps + xb_row + (2 * t + 7) * (PC * 8));
        if (2 * t + 8 < PR) xr[2 * t + 8] = *(const frag*)(Ps + xb_row + (2 * t + 8) * (PC * 8));
        __builtin_amdgcn_sched_barrier(0);           // the reads stay HERE, a tile ahead of their first use (left alone, the scheduler sinks them in front of it)
        f32x4 a_;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) a_ = Elem<T>::mfma(wf[ky], xr[2 * t + ky], ky == 0 ? b4 : a_);
        if (t == 0) {
          u32x2 oc_ = relu_clamp(raw_pack(a_col)), ok_ = relu_clamp(raw_pack(a_cor));
          if constexpr (EDGE) { oc_ = u32x2{oc_[0] & c16m, oc_[1] & c16m}; ok_ = u32x2{ok_[0] & corm, ok_[1] & corm}; }
          *(u32x2*)(Cs + cw_col) = oc_;
          if (r16 == 0) *(u32x2*)(Cs + cw_cor) = ok_;
        } else row_done(t - 1, a_prev);
        a_prev = a_;
        __builtin_amdgcn_sched_barrier(0);
      }
      row_done(CR - 1, a_prev);
    };
    if (edge) body(std::true_type()); else body(std::false_type());
    ST_PH(0);
    if (tid == 0) qn[0] = !has_next ? total : (dyn2 ? pull_resolve(praw) : lb + (jt + 2) * G);
    __syncthreads();                               // conv outputs complete; Ps is free; the next-but-one tile index is published
    ST_PH(1);
    if (has_next) write_window();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * 256;
      const int cg = i & 7, pp = i >> 3;
      const int pr = pp >> 3, pc = pp & 7;
      const int oy = ty * 8 + pr, ox = tx * 8 + pc;
      if (oy >= p.Hq || ox >= p.Wq) continue;
      u32x4 o = *(const u32x4*)(Cs + pro[it][0]);
#pragma unroll
      for (int t = 1; t < 3; ++t) {
        const u32x4 v = *(const u32x4*)(Cs + pro[it][t]);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pk_max16_nonneg(o[k], v[k]);
      }
      if (pc == 7) {                                 // conv column 16: its three rows, from the side image
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const u32x4 v = *(const u32x4*)(Cs + pc16[it][t]);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = pk_max16_nonneg(o[k], v[k]);
        }
      }
      char* dst = (char*)p.out + ((((size_t)img * (p.Hq + 2) + oy + 1) * (p.Wq + 2) + ox + 1) * 64 + cg * 8) * 2;
      *(u32x4*)dst = o;
    }
    ST_PH(2);
    __syncthreads();                               // pool done with Cs, window written: next tile
    ST_PH(3);
#ifdef FLOPE_STAG_DBG
    ++ntile_dbg;
#endif
  }
#ifdef FLOPE_STAG_DBG
  if (p.dbg && tid == 0) {
    unsigned long long* d_ = p.dbg + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 4; ++i) d_[i] = ph[i];
    d_[4] = __builtin_amdgcn_s_memrealtime() - rt_first;      // 100 MHz ticks over the same span as d_[6]: the shader clock
    d_[5] = (unsigned long long)ntile_dbg; d_[6] = __builtin_amdgcn_s_memtime() - t_first; d_[7] = 0;
    unsigned long long* a_ = p.dbg + 16384 + (size_t)blockIdx.x * 4;      // absolute 100 MHz times: kernel entry, loop start, exit; XCC id
    a_[0] = rt_entry; a_[1] = rt_first; a_[2] = __builtin_amdgcn_s_memrealtime(); a_[3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
  }
#endif
#undef ST_PH
  if (tid == 0) {                                  // the last workgroup out resets the queues for the next launch on this slice
    if (__hip_atomic_fetch_add(p.q + 8 * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) __hip_atomic_store(p.q + x * 64, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// (r02 also built a register-pool form -- 7 x 7 pooled pixels per workgroup, pooling by v_pk_max across accumulator tiles and DPP
// row shifts, no conv-output buffer in LDS, three workgroups per CU.  Bit-identical and slower: 197 vs 150 us at 224 x 224, 983 vs
// 697 us at 512 x 512 (+30 % MFMAs for the conv rows computed twice).  Removed in r04; numbers in DESIGN.md 4.2.)

extern "C" int flope_stem_pool_r_blocks_per_cu() { return FLOPE_STEM_WPE; }

extern "C" size_t flope_stem_pool_lds() { return 7 * 64 * 64 + ((39 * 42 * 8 + 15) / 16) * 16; }

extern "C" int flope_stem_pool_init() {
  hipError_t e = hipFuncSetAttribute((const void*)stem_pool_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)stem_pool_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
#define PA(T_, F_) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_pool_persist_kernel<T_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
                   if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_pool_r_kernel<T_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  PA(bf16_t, 0) PA(bf16_t, 1) PA(bf16_t, 2) PA(bf16_t, 3) PA(f16_t, 0) PA(f16_t, 1) PA(f16_t, 2) PA(f16_t, 3)
#undef PA
  return (int)e;
}

// persist_blocks > 0: a persistent kernel on that many workgroups; 0: one workgroup per tile.
// w2 and queues != nullptr (and persist_blocks > 0): the register-weight form (stem_pool_r_kernel; persist_blocks = 3 per CU; queues =
// 9 x 64 zeroed ints -- heads 256 bytes apart -- that no other launch in flight uses).
#ifdef FLOPE_STAG_DBG
static unsigned long long* g_stem_dbg = nullptr;
extern "C" void flope_stem_pool_set_dbg(void* ptr) { g_stem_dbg = (unsigned long long*)ptr; }
#endif

extern "C" int flope_stem_pool_launch(const void* x, int in_format, int B, int H, int W, int Hs, int Ws_, int Hq,
                                      int Wq, const void* w, const void* w2, int* queues, const float* bias, void* out, int dtype, int persist_blocks,
                                      void* stream) {
  StemPoolP p;
#ifdef FLOPE_STAG_DBG
  p.dbg = g_stem_dbg;
#endif
  p.x = x; p.out = out; p.w = w; p.w2 = w2; p.q = queues; p.bias = bias; p.in_format = in_format;
  p.B = B; p.H = H; p.W = W; p.Hs = Hs; p.Ws = Ws_; p.Hq = Hq; p.Wq = Wq;
  p.tiles_y = (Hq + 7) / 8; p.tiles_x = (Wq + 7) / 8;
  flope_host::fastdiv_magic((unsigned)p.tiles_x, &p.mg_tx, &p.sh_tx);
  flope_host::fastdiv_magic((unsigned)p.tiles_y, &p.mg_ty, &p.sh_ty);
  const dim3 block(256);
  if (persist_blocks > 0) {
    const int total = B * p.tiles_y * p.tiles_x;
    const dim3 pgrid(persist_blocks < total ? persist_blocks : total);
    if (w2 && queues) {
      const size_t rlds = ((39 * 42 * 8 + 15) / 16) * 16 + 8 * 18 * 128 + 17 * 128 + 16;
#define RL(T_, F_) hipLaunchKernelGGL((stem_pool_r_kernel<T_, F_>), pgrid, block, rlds, (hipStream_t)stream, p)
      if (dtype == 0) { if (in_format == 0) RL(bf16_t, 0); else if (in_format == 1) RL(bf16_t, 1); else if (in_format == 2) RL(bf16_t, 2); else RL(bf16_t, 3); }
      else            { if (in_format == 0) RL(f16_t, 0); else if (in_format == 1) RL(f16_t, 1); else if (in_format == 2) RL(f16_t, 2); else RL(f16_t, 3); }
#undef RL
      return (int)hipGetLastError();
    }
    const size_t plds = 7 * 64 * 64 + ((39 * 42 * 8 + 15) / 16) * 16 + 17 * 17 * 128;
#define PL(T_, F_) hipLaunchKernelGGL((stem_pool_persist_kernel<T_, F_>), pgrid, block, plds, (hipStream_t)stream, p)
    if (dtype == 0) { if (in_format == 0) PL(bf16_t, 0); else if (in_format == 1) PL(bf16_t, 1); else if (in_format == 2) PL(bf16_t, 2); else PL(bf16_t, 3); }
    else            { if (in_format == 0) PL(f16_t, 0); else if (in_format == 1) PL(f16_t, 1); else if (in_format == 2) PL(f16_t, 2); else PL(f16_t, 3); }
#undef PL
    return (int)hipGetLastError();
  }
  const dim3 grid(B * p.tiles_y * p.tiles_x);
  const size_t lds = flope_stem_pool_lds();
  if (dtype == 0) hipLaunchKernelGGL(stem_pool_kernel<bf16_t>, grid, block, lds, (hipStream_t)stream, p);
  else            hipLaunchKernelGGL(stem_pool_kernel<f16_t>, grid, block, lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
