// C-ABI of the MI355X flower-pose hot path (include/flope_amd.h): engine handle,
// launch plan, BatchNorm folding + MFMA weight packing, and the forward pass
//   crop batch -> PoseResNet trunk -> fp32 head -> special Procrustes.
// Reference behaviour restated: sunflower/models/posenet.py:5-34 (network),
// sunflower/utils/conversion.py:54-58 (Procrustes), eval-mode semantics throughout
// (BatchNorm running statistics, dropout = identity).
#include "../../include/flope_amd.h"
#include "common.h"
#include "host_pack.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

// kernels (other translation units)
extern "C" int flope_conv_mfma_init();
extern "C" int flope_conv_mfma_launch(const ConvP* p, int dtype, int cfg, int patch, int nbuf, size_t lds, void* stream);
extern "C" int flope_stem_init();
extern "C" int flope_stem_launch(const StemP* p, int dtype, size_t lds, void* stream);
extern "C" int flope_maxpool_launch(const PoolP* p, int dtype, void* stream);
extern "C" int flope_avgpool_launch(const void* in, float* out, int B, int h, int w, int C, int dtype, void* stream);
extern "C" int flope_fc1_launch(const float* feat, const float* W1, const float* W1p, const float* b1, float* hidden, int B, int K, int N, void* stream);
extern "C" int flope_fc2_procrustes_launch(const float* hidden, const float* W2, const float* b2, float* r9, float* R, int B, int K, const float* xyz, int nullify, float* Rt, void* stream);
extern "C" int flope_fc2_procrustes_k4_launch(const float* hidden, const float* W2, const float* b2, float* r9, float* R, int B, int K, const float* xyz, int nullify, float* Rt, void* stream);
extern "C" int flope_prep_input_launch(const void* x, int in_format, int B, int H, int W, void* out, int Hip, int Wip, int dtype, void* stream);
extern "C" int flope_read_stage_launch(const void* in, float* out, int B, int C, int h, int w, int dtype, void* stream);
extern "C" int flope_naive_conv_launch(const NaiveConvP* p, void* stream);
extern "C" int flope_conv_stag_init();
extern "C" int flope_conv_gstag_init();
extern "C" int flope_conv_w4_init();
extern "C" int flope_conv_r4_init();
extern "C" int flope_conv_r4_ok(const ConvP* p);
extern "C" int flope_conv_r4_launch(const ConvP* p, int dtype, int grid_blocks, void* stream);
extern "C" int flope_conv_s1r_init();
extern "C" int flope_conv_s1r_ok(const ConvP* p);
extern "C" int flope_conv_s1r_launch(const ConvP* p, const void* w, int dtype, int grid, void* stream);
extern "C" int flope_conv_s2r_init();
extern "C" int flope_conv_s2r_ok(const ConvP* p);
extern "C" int flope_conv_s2r_launch(const ConvP* p, const void* w, int dtype, int grid, void* stream);
extern "C" int flope_conv_w4_launch(const ConvP* p, int dtype, int grid_blocks, int mt, void* stream);
extern "C" size_t flope_conv_w4_lds(int pt, int mt, int dsf, int pers);
extern "C" int flope_conv_gstag_launch(const ConvP* p, int dtype, void* stream);
extern "C" int flope_conv_stag_launch(const ConvP* p, int dtype, int grid_blocks, size_t lds, void* stream);
extern "C" int flope_conv_split_finalize_launch(const ConvP* p, int dtype, void* stream);
extern "C" int flope_stem_pool_init();
extern "C" int flope_stem_pool_r_blocks_per_cu();
#ifdef FLOPE_STAG_DBG
extern "C" void flope_stem_pool_set_dbg(void* ptr);
#endif
extern "C" int flope_stem_pool_launch(const void* x, int in_format, int B, int H, int W, int Hs, int Ws_, int Hq, int Wq,
                                      const void* w, const void* w2, int* queues, const float* bias, void* out, int dtype, int persist_blocks, void* stream);

using namespace flope_host;

namespace {

thread_local std::string g_last_error;

constexpr double kBnEps = 1e-5;
constexpr size_t kLdsTwoBlocks = 80 * 1024;   // <= this: two workgroups per CU
constexpr size_t kLdsMax = 160 * 1024;
constexpr size_t kDbgRegion = 1 << 20;        // diagnostic builds: bytes of the split-K workspace per conv launch (clock stamps)
constexpr size_t kBufSlack = 1 << 20;         // conv_stag's fixed-size patch DMA may read this far past the last pixel (zeros)

struct Conv {
  std::string name, bn;
  int cin = 0, cout = 0, k = 0, stride = 1;
  int in_buf = -1, out_buf = -1, res_buf = -1;
  int hin = 0, win = 0, hout = 0, wout = 0;   // unpadded
  int relu = 0;
  // plan
  int cfg = 0, patch = 0, nbuf = 2, per_image = 0, tiles_per_image = 0, mtiles = 0, ntiles = 0, rows_max = 0;
  size_t lds = 0;
  // device weights
  void* w_packed = nullptr;    // MFMA image (16-bit)
  void* w_stag = nullptr;      // conv_stag image (16-bit), 3x3 s1 Cout >= 128 only
  void* w_s2r = nullptr;       // conv_s2r fragment image (16-bit), the 3x3 stride-2 64 -> 128 conv only
  void* w_s1r = nullptr;       // conv_s1r fragment image (16-bit), the 3x3 stride-1 128 -> 128 convs
  int stag = 0, stag_patch_bytes = 0, nseg = 1; size_t stag_lds = 0;
  int w4_patch[9] = {0};               // conv_w4 on 32 mt-pixel tiles, mt = 4..7: patch rounds of such a tile (0: not available)
  float* w_naive = nullptr;    // [ky][kx][ci][cout]
  float* bias = nullptr;
  // folded shortcut (conv_stag DSF): on a 1x1 downsample conv, folded = 1 means "computed inside layerX.0.conv2";
  // on that conv2, ds_conv is the index of the downsample and bias_fused = bias + bias of the downsample
  int folded = 0, ds_conv = -1;
  void* w_ds_stag = nullptr;   // downsample conv only: its weights as a conv_stag image
  void* w_ds_s1r = nullptr;    // the 64 -> 128 downsample conv only: its weights as conv_s1r's extra fragment pair
  float* bias_fused = nullptr; // conv2 only
  // what the LAST forward launched for this conv (run_slice): kernel family as flope_launch_info names it, and the launch's shape
  mutable std::string last_kernel, last_detail;
};

struct Buf { void* ptr = nullptr; size_t bytes = 0; int C = 0, h = 0, w = 0; };

}  // namespace

struct flope_engine {
  int device = 0, H = 0, W = 0, maxB = 0, dtype = 0, bod = 2048;
  int esz = 2;                       // bytes per trunk element
  // stem
  int sHip = 0, sWip = 0, Hs = 0, Ws = 0, stem_tiles = 0, stem_rows = 0;
  size_t stem_lds = 0;
  void* stem_in = nullptr; size_t stem_in_bytes = 0;
  int* stem_q = nullptr;               // tile queues of the register-weight stem: 1024 ints per batch slice (heads 256 bytes apart; zero between launches)
  void* stem_w = nullptr; void* stem_w2 = nullptr; float* stem_w_naive = nullptr; float* stem_bias = nullptr;   // stem_w2: per-wave fragment order (stem_pool_r_kernel)
  std::vector<Buf> bufs;             // 0 stem_out, 1 pool, then per block: mid, [ds], out
  std::vector<Conv> convs;
  int stage_buf[10];                 // FLOPE_STAGE_* (0..9) -> buffer index
  int final_buf = -1;
  float *feat = nullptr, *hidden = nullptr, *W1 = nullptr, *W1p = nullptr, *b1 = nullptr, *W2 = nullptr, *b2 = nullptr;
  float* r9_scratch = nullptr;
  bool weights_loaded = false;
  int opt_patch = 1, opt_bm256 = 1, opt_profile = 0, opt_nbuf = 2, opt_dbg = 0, opt_ldspad = 0, opt_fuse_stem = 1, opt_streams = 2, opt_persist = 0, num_cus = 256, opt_stag = 3, cur_slices = 1, plan_slices = 1, cur_batch = 1, opt_rows_grid = 0, opt_split = 0, opt_dsfuse = 1, opt_ksplit = 1, opt_fc1_packed = 1, opt_gstag = 1, opt_rowseg = 1, opt_stem_persist = 1, opt_skew = 1, opt_reslds = 1, opt_prio = 0, opt_r4 = 1, opt_w4cw = 4, opt_w4cwf = 0, opt_fc2_k4 = 1, opt_w4mt = 0, opt_w4mtlo = 0, opt_lag = 20, opt_w4 = 1, opt_stem_r = 1, opt_s2r = 1, opt_s2r_grid = 0, opt_s1r = 1;   // w4: 0 = conv_stag for the flat 256 x 128 tiles, 1 = conv_w4 (4 waves); w4cw: class walk of conv_w4 (persistent workgroups), tiles per workgroup aimed at (0 = one tile per workgroup)
  float* split_ws = nullptr; size_t split_ws_bytes = 0;   // fp32 partial sums of the split-K path (small batches)   // stag: 0 off, 1 Cout >= 128 layers, 2 also the 64-channel layer (512 x 64 tiles), 3 (default) 64-channel layer as 8-row bands where the shape allows
  hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ev;        // profile mode: one event before every launch + one after the last
  int ev_n = 0;
  std::vector<int> ev_slice;         // profile = 2: the slice whose stream recorded event i
  int mark_slice = 0;
  int last_batch = 0;
  bool last_fused = false;
  std::string err;
};

namespace {

int fail(flope_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  g_last_error = msg;
  return code;
}

#define HIP_TRY(e, call)                                                                   \
  do {                                                                                     \
    hipError_t _s = (call);                                                                \
    if (_s != hipSuccess)                                                                  \
      return fail(e, FLOPE_EHIP, std::string(#call) + ": " + hipGetErrorString(_s));       \
  } while (0)

#define K_TRY(e, what, call)                                                               \
  do {                                                                                     \
    int _s = (call);                                                                       \
    if (_s != 0)                                                                           \
      return fail(e, FLOPE_EHIP, std::string(what) + ": " + hipGetErrorString((hipError_t)_s)); \
  } while (0)

#define MARK(e, stream)                                                                    \
  do {                                                                                     \
    if ((e)->opt_profile && (e)->ev_n < (int)(e)->ev.size()) {                             \
      (e)->ev_slice[(e)->ev_n] = (e)->mark_slice;                                          \
      HIP_TRY(e, hipEventRecord((e)->ev[(e)->ev_n++], (hipStream_t)(stream)));             \
    }                                                                                      \
  } while (0)

// option "lag": one wave that sleeps ~us microseconds at the head of the last slice's stream (bounded by the real-time clock)
__global__ void lag_kernel(int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
}

int out_dim(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

// ---- plan ----------------------------------------------------------------------
void tile_dims(int cfg, int* BM, int* BN) {
  *BM = cfg == 2 ? 256 : 128;
  *BN = cfg == 1 ? 128 : 64;
}

// exact worst-case number of padded input rows a patch tile needs
int patch_rows(const Conv& c, int B, int BM, bool per_image) {
  const int HoWo = c.hout * c.wout, Hip = c.hin + 2;
  long M = (long)B * HoWo;
  int worst = 0;
  auto rows_of = [&](long m0, long mend) {
    const long ml = mend - 1;
    const long b0 = m0 / HoWo, ho0 = (m0 - b0 * HoWo) / c.wout;
    const long b1 = ml / HoWo, ho1 = (ml - b1 * HoWo) / c.wout;
    return (int)((b1 * Hip + ho1 * c.stride + 2) - (b0 * Hip + ho0 * c.stride) + 1);
  };
  if (per_image) {
    const int tpi = (HoWo + BM - 1) / BM;
    for (int t = 0; t < tpi; ++t) {
      const long m0 = (long)t * BM, mend = std::min<long>(m0 + BM, HoWo);
      worst = std::max(worst, rows_of(m0, mend));
    }
  } else {
    for (long m0 = 0; m0 < M; m0 += BM) worst = std::max(worst, rows_of(m0, std::min<long>(m0 + BM, M)));
  }
  return worst;
}

void plan_conv(flope_engine* e, Conv& c) {
  const int B = e->maxB;
  const int HoWo = c.hout * c.wout;
  const int Wip = c.win + 2;
  struct Cand { int cfg, patch, per_image, rows, nbuf; size_t lds; };
  std::vector<Cand> cands;
  const bool can_patch = e->opt_patch && c.k == 3 && c.stride == 1;
  std::vector<int> cfgs;
  if (c.cout == 64) { if (e->opt_bm256) cfgs.push_back(2); cfgs.push_back(0); }
  else cfgs.push_back(1);
  // a 3-deep ring pays where the weight tile is 16 KB per step (Cout >= 128); for the 64-channel layers
  // it would push the 256-pixel tile out of LDS and cost more than it hides (r01 layer timings)
  std::vector<int> depths;
  if (e->opt_nbuf == 3 && c.cout >= 128) depths.push_back(3);
  depths.push_back(2);
  // preference: deepest ring first, then patch before gather, flat tiles before per-image tiles
  for (int nb : depths)
    for (int cfg : cfgs) {
      int BM, BN; tile_dims(cfg, &BM, &BN);
      if (can_patch)
        for (int pi = 0; pi < 2; ++pi) {
          const int rows = patch_rows(c, B, BM, pi != 0);
          cands.push_back({cfg, 1, pi, rows, nb, (size_t)nb * BN * 128 + (((size_t)rows * Wip * 128 + 4095) & ~(size_t)4095)});
        }
    }
  for (int nb : depths)
    for (int cfg : cfgs) {
      int BM, BN; tile_dims(cfg, &BM, &BN);
      cands.push_back({cfg, 0, 0, 0, nb, (size_t)nb * BN * 128 + (size_t)nb * BM * 128});
    }
  // first candidate that lets two workgroups share a CU; else the smallest that fits at all
  const Cand* pick = nullptr;
  for (const Cand& cd : cands)
    if (cd.lds <= kLdsTwoBlocks) { pick = &cd; break; }
  if (!pick)
    for (const Cand& cd : cands)
      if (cd.lds <= kLdsMax && (!pick || cd.lds < pick->lds)) pick = &cd;
  int BM, BN; tile_dims(pick->cfg, &BM, &BN);
  c.cfg = pick->cfg; c.patch = pick->patch; c.nbuf = pick->nbuf; c.per_image = pick->per_image; c.rows_max = pick->rows; c.lds = pick->lds;
  c.tiles_per_image = (HoWo + BM - 1) / BM;
  c.ntiles = c.cout / BN;
  // second-generation kernel (conv_stag.hip): 256 x 128 tiles, 32-channel steps, double-buffered patch
  c.stag = 0;
  if (e->opt_stag && c.k == 3 && c.stride == 1 && c.cin % 64 == 0 && (c.cout >= 128 || (c.cout == 64 && e->opt_stag >= 2))) {
    const int sbm = c.cout == 64 ? 512 : 256, sbn = c.cout == 64 ? 64 : 128;
    const int rows = patch_rows(c, B, sbm, false);
    const long pieces = (long)rows * (Wip + (e->opt_skew ? 2 : 0)) * 4;   // skew: LDS row pitch W + 4, conflict-free fragment reads across row wraps (conv_stag.hip)
    int P = (int)((pieces + 511) / 512);
    if (P < 2) P = 2;                                  // kernel instantiations: 2..6 and 8 DMA rounds per patch burst
    if (P < 4 && c.cout >= 128 && e->opt_dsfuse) P = 4;  // 32 KB buffers: room for a folded shortcut's gathered pixel tiles
    if (P == 7) P = 8;                                 // (every round is 8 KB of L2 -> LDS traffic per half-chunk and tile)
    const size_t lds = (size_t)6 * sbn * 64 + (size_t)2 * P * 8192;   // 3 double tiles + 2 patch buffers
    if (P <= 8 && lds <= kLdsMax) { c.stag = 1; c.stag_patch_bytes = P; c.stag_lds = lds; }
    if (c.cout >= 128)                                  // the 4-wave kernel's smaller tiles (conv_w4.hip, MT = 5..7)
      for (int mt = 5; mt <= 7; ++mt) {
        const long pcs = (long)patch_rows(c, B, 32 * mt, false) * (Wip + 2) * 4;
        const int Pm = std::max(4, (int)((pcs + 511) / 512));
        c.w4_patch[mt] = (mt == 7 ? Pm <= 6 : Pm == 4) ? Pm : 0;      // below 7: the 4-round instantiations only
      }
    // layer-1 shape: 8-row bands of one image per tile (constant tile geometry, 7 bands per 56-row image)
    c.nseg = 1;
    if (c.cout == 64 && e->opt_stag >= 3 && c.hout % 8 == 0 && c.wout <= 64) {
      int Pr = (int)(((long)10 * Wip * 4 + 511) / 512);
      // odd Pr (3, 5) = the instantiations that keep the 72 KB weight panel of a 64 -> 64 layer resident in LDS
      if (c.cin == 64 && Pr <= 5) Pr = Pr <= 3 ? 3 : 5; else Pr = Pr <= 6 ? 6 : 8;
      const size_t ldsr = ((Pr & 1) ? (size_t)18 * 4096 : (size_t)6 * sbn * 64) + (size_t)2 * Pr * 8192;
      if (Pr <= 8 && ldsr <= kLdsMax) { c.stag = 2; c.stag_patch_bytes = Pr; c.stag_lds = ldsr; }
    } else if (c.cout == 64 && e->opt_stag >= 3 && e->opt_rowseg && c.hout % 8 == 0 && c.wout > 64) {
      // wide maps (512 x 512 crops: layer 1 is 128 x 128): 8-row bands cut into 64-column segments, 10 x 66-pixel patches
      // (2640 pieces -> 6 DMA rounds), ring weights
      c.stag = 2; c.stag_patch_bytes = 6; c.nseg = (c.wout + 63) / 64;
      c.stag_lds = (size_t)6 * sbn * 64 + (size_t)2 * 6 * 8192;
    }
  }
  // 3x3 stride-2 convs: the gathered-tile variant of the same 8-wave structure (conv_gstag)
  if (e->opt_stag && e->opt_gstag && c.k == 3 && c.stride == 2 && c.cin % 64 == 0 && c.cin >= (e->opt_gstag >= 2 ? 64 : 128) && c.cout % 128 == 0) c.stag = 3;   // gstag: 1 = Cin >= 128 (K = 576 is too short to amortise the 8-wave prologue), 2 = every stride-2 3x3
}

void conv_params(const flope_engine* e, const std::vector<Buf>& bufs, const Conv& c, int batch, ConvP* p) {
  int BM, BN; tile_dims(c.cfg, &BM, &BN);
  memset(p, 0, sizeof(*p));
  p->in = bufs[c.in_buf].ptr; p->out = bufs[c.out_buf].ptr;
  p->res = c.res_buf >= 0 ? bufs[c.res_buf].ptr : nullptr;
  p->w = c.w_packed; p->bias = c.bias;
  p->B = batch; p->Hip = c.hin + 2; p->Wip = c.win + 2; p->Cin = c.cin;
  p->Ho = c.hout; p->Wo = c.wout; p->Hop = c.hout + 2; p->Wop = c.wout + 2; p->Cout = c.cout;
  p->stride = c.stride; p->ntaps = c.k == 3 ? 9 : 1;
  p->M = batch * c.hout * c.wout; p->relu = c.relu; p->nchunks = c.cin / 64;
  p->per_image = c.per_image; p->tiles_per_image = c.tiles_per_image;
  p->mtiles = c.per_image ? batch * c.tiles_per_image : (p->M + BM - 1) / BM;
  p->ntiles = c.ntiles; p->patch_rows_max = c.rows_max; p->dbg = e->opt_dbg;
  fastdiv_magic((unsigned)(c.hout * c.wout), &p->mg_hw, &p->sh_hw);
  fastdiv_magic((unsigned)c.wout, &p->mg_w, &p->sh_w);
}

template <typename V>
int upload(flope_engine* e, const std::vector<V>& host, void** dev) {
  if (*dev) { hipFree(*dev); *dev = nullptr; }
  HIP_TRY(e, hipMalloc(dev, host.size() * sizeof(V)));
  HIP_TRY(e, hipMemcpy(*dev, host.data(), host.size() * sizeof(V), hipMemcpyHostToDevice));
  return 0;
}

struct Tensors {
  std::map<std::string, std::pair<const float*, std::vector<int64_t>>> t;
  const float* get(flope_engine* e, const std::string& name, const std::vector<int64_t>& shape, int* rc) const {
    auto it = t.find(name);
    if (it == t.end()) { *rc = fail(e, FLOPE_EWEIGHTS, "state_dict entry missing: " + name); return nullptr; }
    if (it->second.second != shape) {
      std::string got, want;
      for (auto d : it->second.second) got += std::to_string(d) + ",";
      for (auto d : shape) want += std::to_string(d) + ",";
      *rc = fail(e, FLOPE_EWEIGHTS, "size mismatch for " + name + ": got [" + got + "] expected [" + want + "]");
      return nullptr;
    }
    return it->second.first;
  }
};

// fold eval-mode BN into a bias-free conv: w' = w * g / sqrt(v + eps), b' = beta - mean * g / sqrt(v + eps)
int fold(flope_engine* e, const Tensors& ts, const std::string& conv, const std::string& bn, int cout, int cin,
         int k, std::vector<float>* wf, std::vector<float>* bf) {
  int rc = 0;
  const float* w = ts.get(e, conv + ".weight", {cout, cin, k, k}, &rc); if (!w) return rc;
  const float* g = ts.get(e, bn + ".weight", {cout}, &rc); if (!g) return rc;
  const float* b = ts.get(e, bn + ".bias", {cout}, &rc); if (!b) return rc;
  const float* m = ts.get(e, bn + ".running_mean", {cout}, &rc); if (!m) return rc;
  const float* v = ts.get(e, bn + ".running_var", {cout}, &rc); if (!v) return rc;
  const size_t per = (size_t)cin * k * k;
  wf->resize((size_t)cout * per); bf->resize(cout);
  double wmax = 0.0;
  for (int co = 0; co < cout; ++co) {
    const double scale = (double)g[co] / sqrt((double)v[co] + kBnEps);
    (*bf)[co] = (float)((double)b[co] - (double)m[co] * scale);
    for (size_t i = 0; i < per; ++i) {
      const double x = (double)w[co * per + i] * scale;
      (*wf)[co * per + i] = (float)x;
      if (!(fabs(x) <= 3.0e38)) return fail(e, FLOPE_EWEIGHTS, "non-finite folded weight in " + conv);
      wmax = std::max(wmax, fabs(x));
    }
    if (!std::isfinite((*bf)[co])) return fail(e, FLOPE_EWEIGHTS, "non-finite folded bias in " + bn);
  }
  if (e->dtype == FLOPE_DT_F16 && wmax > 6.0e4)
    return fail(e, FLOPE_EWEIGHTS, "folded weights of " + conv + " exceed the float16 range; use bf16 or f32");
  return 0;
}

}  // namespace

// ================================================================================
// A HIP stream whose kernels may only occupy the compute units set in `mask` (bit i of word i / 32 = CU i in the runtime's
// enumeration, which walks the XCDs round-robin).  The live loop gives the detector -- a chain of ~80 short, narrow
// launches -- its own CUs so that it never queues behind the pose network's full-chip grids (and vice versa).
extern "C" int flope_stream_create_cu_mask(int device_id, const uint32_t* mask, int words, void** out_stream) {
  if (!mask || words < 1 || words > 32 || !out_stream) return FLOPE_EINVAL;
  *out_stream = nullptr;
  bool any = false;
  for (int i = 0; i < words; ++i) any = any || mask[i] != 0;
  if (!any) return FLOPE_EINVAL;
  if (hipSetDevice(device_id) != hipSuccess) return FLOPE_EHIP;
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); return FLOPE_EHIP; }
  *out_stream = st;
  return FLOPE_OK;
}

extern "C" int flope_stream_destroy(int device_id, void* stream) {
  if (!stream) return FLOPE_OK;
  if (hipSetDevice(device_id) != hipSuccess) return FLOPE_EHIP;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? FLOPE_OK : FLOPE_EHIP;
}

extern "C" const char* flope_version(void) { return "flope_amd 0.1 (gfx950; mfma_f32_16x16x32 bf16/f16; fp32 head)"; }

extern "C" int flope_engine_geometry(flope_handle h, int* max_batch, int* dtype, int* height, int* width, int* device_id) {
  if (!h) return FLOPE_EINVAL;
  if (max_batch) *max_batch = h->maxB;
  if (dtype) *dtype = h->dtype;
  if (height) *height = h->H;
  if (width) *width = h->W;
  if (device_id) *device_id = h->device;
  return FLOPE_OK;
}

extern "C" const char* flope_last_error(flope_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

static int rebuild_plan(flope_engine* e) {
  for (Conv& c : e->convs) { plan_conv(e, c); c.folded = 0; c.ds_conv = -1; }
  // fold each block's 1x1 stride-2 shortcut into the conv2 that consumes it when that conv2 runs on conv_stag 256x128
  // tiles with >= 32 KB patch buffers (one gathered 64-channel pixel tile pair fits one buffer)
  if (e->opt_dsfuse && e->dtype != FLOPE_DT_F32)
    for (size_t i = 0; i + 1 < e->convs.size(); ++i) {
      Conv& cd = e->convs[i];
      Conv& c2 = e->convs[i + 1];
      if (cd.k == 1 && c2.k == 3 && c2.res_buf == cd.out_buf && c2.stag == 1 && c2.cout >= 128 && c2.stag_patch_bytes >= 4 &&
          c2.stag_patch_bytes != 7 && cd.cin % 64 == 0 && cd.stride == 2) {
        cd.folded = 1;
        c2.ds_conv = (int)i;
      }
    }
  return 0;
}

extern "C" int flope_create(int device_id, int height, int width, int max_batch, int dtype, int backbone_out_dim,
                            flope_handle* out) {
  if (!out) return fail(nullptr, FLOPE_EINVAL, "flope_create: out is NULL");
  *out = nullptr;
  if (height < 32 || width < 32 || height > 4096 || width > 4096)
    return fail(nullptr, FLOPE_EINVAL, "flope_create: crop size must be within 32..4096");
  if (max_batch < 1) return fail(nullptr, FLOPE_EINVAL, "flope_create: max_batch must be >= 1");
  if (dtype < 0 || dtype > 2) return fail(nullptr, FLOPE_EINVAL, "flope_create: unknown dtype");
  if (backbone_out_dim < 1) return fail(nullptr, FLOPE_EINVAL, "flope_create: backbone_out_dim must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, FLOPE_EHIP, "flope_create: no HIP device visible (the product path has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, FLOPE_EINVAL, "flope_create: bad device id");
  flope_engine* e = new flope_engine();
  e->device = device_id; e->H = height; e->W = width; e->maxB = max_batch; e->dtype = dtype; e->bod = backbone_out_dim;
  e->esz = dtype == FLOPE_DT_F32 ? 4 : 2;
#define CREATE_TRY(call)                                                                         \
  do {                                                                                           \
    hipError_t _s = (call);                                                                      \
    if (_s != hipSuccess) {                                                                      \
      int _rc = fail(nullptr, FLOPE_EHIP, std::string(#call) + ": " + hipGetErrorString(_s));    \
      flope_destroy(e);                                                                          \
      return _rc;                                                                                \
    }                                                                                            \
  } while (0)
  CREATE_TRY(hipSetDevice(device_id));
  {
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, device_id));
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (dtype != FLOPE_DT_F32) {
    int s = flope_conv_mfma_init();
    if (s == 0) s = flope_stem_init();
    if (s == 0) s = flope_stem_pool_init();
    if (s == 0) s = flope_conv_stag_init();
    if (s == 0) s = flope_conv_gstag_init();
    if (s == 0) s = flope_conv_w4_init();
    if (s == 0) s = flope_conv_r4_init();
    if (s == 0) s = flope_conv_s2r_init();
    if (s == 0) s = flope_conv_s1r_init();
    if (s != 0) { int rc = fail(nullptr, FLOPE_EHIP, std::string("kernel attribute setup: ") + hipGetErrorString((hipError_t)s)); flope_destroy(e); return rc; }
  }
  const size_t B = (size_t)max_batch;
  // stem input: 4 channels, 3-pixel border, slack on the right/bottom for the kx=7 / ky pad taps
  e->sHip = height + 6; e->sWip = (width + 8 + 1) & ~1;
  e->Hs = out_dim(height, 7, 2, 3); e->Ws = out_dim(width, 7, 2, 3);
  e->stem_in_bytes = B * e->sHip * e->sWip * 4 * e->esz;
  CREATE_TRY(hipMalloc(&e->stem_in, e->stem_in_bytes));
  CREATE_TRY(hipMemset(e->stem_in, 0, e->stem_in_bytes));
  {
    const int HoWo = e->Hs * e->Ws;
    e->stem_tiles = (HoWo + 255) / 256;
    int span = 1;
    for (int t = 0; t < e->stem_tiles; ++t) {
      const int m0 = t * 256, me = std::min(m0 + 256, HoWo);
      span = std::max(span, (me - 1) / e->Ws - m0 / e->Ws + 1);
    }
    e->stem_rows = 2 * (span - 1) + 7;
    e->stem_lds = (size_t)7 * 64 * 64 + (size_t)e->stem_rows * e->sWip * 8;
    if (dtype != FLOPE_DT_F32 && e->stem_lds > kLdsMax) {
      int rc = fail(nullptr, FLOPE_EINVAL, "flope_create: crop too wide for the stem kernel's LDS patch"); flope_destroy(e); return rc;
    }
  }
  auto add_buf = [&](int C, int h, int w) {
    Buf b; b.C = C; b.h = h; b.w = w; b.bytes = B * (h + 2) * (w + 2) * C * e->esz + kBufSlack;
    e->bufs.push_back(b);
    return (int)e->bufs.size() - 1;
  };
  const int b_stem = add_buf(64, e->Hs, e->Ws);
  const int Hq = out_dim(e->Hs, 3, 2, 1), Wq = out_dim(e->Ws, 3, 2, 1);
  const int b_pool = add_buf(64, Hq, Wq);
  e->stage_buf[FLOPE_STAGE_STEM] = b_stem; e->stage_buf[FLOPE_STAGE_POOL] = b_pool;
  int cur = b_pool, ch = 64, hh = Hq, ww = Wq;
  const int couts[4] = {64, 128, 256, 512}, strides[4] = {1, 2, 2, 2};
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < 2; ++bi) {
      const int s = bi == 0 ? strides[li] : 1, co = couts[li];
      const int ho = out_dim(hh, 3, s, 1), wo = out_dim(ww, 3, s, 1);
      const std::string p = "base.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      const int b_mid = add_buf(co, ho, wo);
      Conv c1; c1.name = p + ".conv1"; c1.bn = p + ".bn1"; c1.cin = ch; c1.cout = co; c1.k = 3; c1.stride = s;
      c1.in_buf = cur; c1.out_buf = b_mid; c1.hin = hh; c1.win = ww; c1.hout = ho; c1.wout = wo; c1.relu = 1;
      e->convs.push_back(c1);
      int res = cur;
      if (bi == 0 && (s != 1 || ch != co)) {
        const int b_ds = add_buf(co, ho, wo);
        Conv cd; cd.name = p + ".downsample.0"; cd.bn = p + ".downsample.1"; cd.cin = ch; cd.cout = co; cd.k = 1; cd.stride = s;
        cd.in_buf = cur; cd.out_buf = b_ds; cd.hin = hh; cd.win = ww; cd.hout = ho; cd.wout = wo; cd.relu = 0;
        e->convs.push_back(cd);
        res = b_ds;
      }
      const int b_out = add_buf(co, ho, wo);
      Conv c2; c2.name = p + ".conv2"; c2.bn = p + ".bn2"; c2.cin = co; c2.cout = co; c2.k = 3; c2.stride = 1;
      c2.in_buf = b_mid; c2.out_buf = b_out; c2.res_buf = res; c2.hin = ho; c2.win = wo; c2.hout = ho; c2.wout = wo; c2.relu = 1;
      e->convs.push_back(c2);
      e->stage_buf[FLOPE_STAGE_LAYER(li + 1, bi)] = b_out;
      cur = b_out; ch = co; hh = ho; ww = wo;
    }
  e->final_buf = cur;
  if (hh < 1 || ww < 1) { int rc = fail(nullptr, FLOPE_EINVAL, "flope_create: crop too small"); flope_destroy(e); return rc; }
  for (Buf& b : e->bufs) {
    CREATE_TRY(hipMalloc(&b.ptr, b.bytes));
    CREATE_TRY(hipMemset(b.ptr, 0, b.bytes));          // the zero ring is written exactly once
  }
  CREATE_TRY(hipMalloc((void**)&e->feat, B * 512 * sizeof(float)));
  CREATE_TRY(hipMalloc((void**)&e->hidden, B * (size_t)e->bod * sizeof(float)));
  CREATE_TRY(hipMalloc((void**)&e->r9_scratch, B * 9 * sizeof(float)));
  if (dtype != FLOPE_DT_F32) {          // split-K partials: tiles * ksplit <= num_cus, 256 x 128 fp32 per tile share
    e->split_ws_bytes = (size_t)e->num_cus * 256 * 128 * sizeof(float);
    CREATE_TRY(hipMalloc((void**)&e->split_ws, e->split_ws_bytes));
    CREATE_TRY(hipMalloc((void**)&e->stem_q, 4 * 1024 * sizeof(int)));
    CREATE_TRY(hipMemset(e->stem_q, 0, 4 * 1024 * sizeof(int)));
  }
  for (int i = 0; i < 4; ++i) {
    CREATE_TRY(hipStreamCreateWithFlags(&e->side[i], hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
  }
  CREATE_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  e->ev.resize(4 * (e->convs.size() + 8) + 2);       // profile = 2: up to four slices' marks + the fork
  e->ev_slice.assign(e->ev.size(), 0);
  for (hipEvent_t& ev : e->ev) CREATE_TRY(hipEventCreate(&ev));
  CREATE_TRY(hipDeviceSynchronize());
#undef CREATE_TRY
  rebuild_plan(e);
  *out = e;
  return FLOPE_OK;
}

extern "C" int flope_destroy(flope_handle e) {
  if (!e) return FLOPE_OK;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  for (Buf& b : e->bufs) if (b.ptr) hipFree(b.ptr);
  for (Conv& c : e->convs) { if (c.w_packed) hipFree(c.w_packed); if (c.w_stag) hipFree(c.w_stag); if (c.w_s2r) hipFree(c.w_s2r); if (c.w_s1r) hipFree(c.w_s1r); if (c.w_naive) hipFree(c.w_naive); if (c.bias) hipFree(c.bias); if (c.w_ds_stag) hipFree(c.w_ds_stag); if (c.w_ds_s1r) hipFree(c.w_ds_s1r); if (c.bias_fused) hipFree(c.bias_fused); }
  void* singles[] = {e->stem_in, e->stem_q, e->stem_w, e->stem_w2, e->stem_w_naive, e->stem_bias, e->feat, e->hidden, e->W1, e->W1p, e->b1, e->W2, e->b2, e->r9_scratch, e->split_ws};
  for (void* p : singles) if (p) hipFree(p);
  for (hipEvent_t ev : e->ev) hipEventDestroy(ev);
  for (int i = 0; i < 4; ++i) { if (e->side[i]) hipStreamDestroy(e->side[i]); if (e->ev_join[i]) hipEventDestroy(e->ev_join[i]); }
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  delete e;
  return FLOPE_OK;
}

// developer aid (diagnostic builds, -DFLOPE_STAG_DBG + option dbg = 64): copies `bytes` of the split-K workspace, where conv launch i
// of the last forward left its clock stamps at byte offset i * 1 MiB (kDbgRegion), to host memory
extern "C" int flope_debug_read_ws(flope_handle e, void* dst_host, size_t offset, size_t bytes) {
  if (!e || !dst_host) return fail(e, FLOPE_EINVAL, "flope_debug_read_ws: NULL argument");
  if (!e->split_ws || offset + bytes > e->split_ws_bytes) return fail(e, FLOPE_EINVAL, "flope_debug_read_ws: out of range");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipDeviceSynchronize());
  HIP_TRY(e, hipMemcpy(dst_host, (const char*)e->split_ws + offset, bytes, hipMemcpyDeviceToHost));
  return FLOPE_OK;
}

extern "C" int flope_set_option(flope_handle e, const char* name, int value) {
  if (!e || !name) return fail(e, FLOPE_EINVAL, "flope_set_option: NULL argument");
  int prev;
  if (!strcmp(name, "patch")) { prev = e->opt_patch; e->opt_patch = value != 0; }
  else if (!strcmp(name, "bm256")) { prev = e->opt_bm256; e->opt_bm256 = value != 0; }
  else if (!strcmp(name, "persist")) { prev = e->opt_persist; e->opt_persist = value != 0; return prev; }
  else if (!strcmp(name, "rows_grid")) { prev = e->opt_rows_grid; e->opt_rows_grid = value < 0 ? -1 : value; return prev; }   // layer-1 persistent grid: 0 = one workgroup per CU, -1 = the slice's share of the CUs, > 0 = that many
  else if (!strcmp(name, "split")) { prev = e->opt_split; e->opt_split = value < 0 ? 0 : value; return prev; }   // 0: default 3/8 : 5/8; 1..100: percent of the batch in slice 0; > 100: (value - 100) images
  else if (!strcmp(name, "fc1_packed")) { prev = e->opt_fc1_packed; e->opt_fc1_packed = value != 0; return prev; }
  else if (!strcmp(name, "w4mtlo")) { prev = e->opt_w4mtlo; e->opt_w4mtlo = value <= 0 ? 0 : (value < 5 ? 5 : (value > 8 ? 8 : value)); return prev; }   // smallest tile height the per-launch choice may take (0: 7 with two slices in flight, 5 alone)
  else if (!strcmp(name, "lag")) { prev = e->opt_lag; e->opt_lag = value < 0 ? 0 : (value > 500 ? 500 : value); return prev; }   // microseconds by which the last batch slice starts late (default 20; 0 = off)
  else if (!strcmp(name, "w4mt")) { prev = e->opt_w4mt; e->opt_w4mt = (value >= 5 && value <= 8) ? value : 0; return prev; }   // conv_w4 tile height: 0 = per launch, 5..8 = 160..256 pixels (where the shape has that instantiation)
  else if (!strcmp(name, "fc2_k4")) { prev = e->opt_fc2_k4; e->opt_fc2_k4 = value != 0; return prev; }            // fc_rot: K split over the four waves of a workgroup per image
  else if (!strcmp(name, "ksplit")) { prev = e->opt_ksplit; e->opt_ksplit = value < 0 ? 0 : (value > 2 ? 2 : value); return prev; }   // 2: always the largest split (r02a rule)
  else if (!strcmp(name, "stem_r")) { prev = e->opt_stem_r; e->opt_stem_r = value != 0; return prev; }   // 1: the register-weight stem (three workgroups per CU, r05); 0: the r02 forms (stem_persist)
  else if (!strcmp(name, "stem_persist")) { prev = e->opt_stem_persist; e->opt_stem_persist = value < 0 ? 0 : (value > 2 ? 2 : value); return prev; }
  else if (!strcmp(name, "rowseg")) { prev = e->opt_rowseg; e->opt_rowseg = value != 0; }
  else if (!strcmp(name, "skew")) { prev = e->opt_skew; e->opt_skew = value != 0; }
  else if (!strcmp(name, "r4")) { prev = e->opt_r4; e->opt_r4 = value != 0; return prev; }
  else if (!strcmp(name, "s1r")) { prev = e->opt_s1r; e->opt_s1r = value != 0; return prev; }   // 1: layer2.1.conv1 / conv2 on conv_s1r (224^2 crops), 0: conv_w4
  else if (!strcmp(name, "s2r_grid")) { prev = e->opt_s2r_grid; e->opt_s2r_grid = value; return prev; }   // workgroups of a conv_s2r launch (0: one per CU)
  else if (!strcmp(name, "s2r")) { prev = e->opt_s2r; e->opt_s2r = value != 0; return prev; }   // 1: layer2.0.conv1 on conv_s2r (224^2 crops), 0: conv_mfma<gather>
  else if (!strcmp(name, "w4")) { prev = e->opt_w4; e->opt_w4 = value != 0; return prev; }
  else if (!strcmp(name, "w4cw")) { prev = e->opt_w4cw; e->opt_w4cw = value < 0 ? 0 : (value > 64 ? 64 : value); return prev; }   // conv_w4 class walk: tiles per persistent workgroup aimed at (0 / 1 = one tile per workgroup)
  else if (!strcmp(name, "w4cwf")) { prev = e->opt_w4cwf; e->opt_w4cwf = value & 3; return prev; }   // ... bit 0: also with several batch slices in flight, bit 1: also where the walk fills < 85 % of the slice's CUs
  else if (!strcmp(name, "prio")) { prev = e->opt_prio; e->opt_prio = value < 0 ? 0 : (value > 2 ? 2 : value); return prev; }
  else if (!strcmp(name, "reslds")) { prev = e->opt_reslds; e->opt_reslds = value != 0; return prev; }
  else if (!strcmp(name, "gstag")) { prev = e->opt_gstag; e->opt_gstag = value < 0 ? 0 : (value > 2 ? 2 : value); }
  else if (!strcmp(name, "dsfuse")) { prev = e->opt_dsfuse; e->opt_dsfuse = value != 0; }
  else if (!strcmp(name, "stag")) { prev = e->opt_stag; e->opt_stag = value < 0 ? 0 : (value > 3 ? 3 : value); }
  else if (!strcmp(name, "streams")) { prev = e->opt_streams; e->opt_streams = value < 1 ? 1 : (value > 4 ? 4 : value); return prev; }
  else if (!strcmp(name, "fuse_stem")) { prev = e->opt_fuse_stem; e->opt_fuse_stem = value != 0; return prev; }
  else if (!strcmp(name, "ldspad")) { prev = e->opt_ldspad; e->opt_ldspad = value; return prev; }
  else if (!strcmp(name, "dbg")) { prev = e->opt_dbg; e->opt_dbg = value; return prev; }
  else if (!strcmp(name, "nbuf")) { prev = e->opt_nbuf; e->opt_nbuf = value == 2 ? 2 : 3; }
  else if (!strcmp(name, "profile")) { prev = e->opt_profile; e->opt_profile = value < 0 ? 0 : (value > 2 ? 2 : value); e->ev_n = 0; return prev; }   // 1: one slice, an event around every launch (flope_profile_read); 2: the slices as in production, events on every slice's stream (flope_profile_timeline)
  else return fail(e, FLOPE_EINVAL, std::string("flope_set_option: unknown option ") + name);
  rebuild_plan(e);
  return prev;
}

extern "C" int flope_load_weights(flope_handle e, int n, const char* const* names, const float* const* host_ptrs,
                                  const int* ndims, const int64_t* const* shapes) {
  if (!e) return fail(nullptr, FLOPE_EINVAL, "flope_load_weights: NULL handle");
  if (n < 0 || (n > 0 && (!names || !host_ptrs || !ndims || !shapes)))
    return fail(e, FLOPE_EINVAL, "flope_load_weights: NULL argument");
  HIP_TRY(e, hipSetDevice(e->device));
  e->weights_loaded = false;
  Tensors ts;
  for (int i = 0; i < n; ++i) {
    if (!names[i] || !host_ptrs[i] || ndims[i] < 0 || ndims[i] > 8 || (ndims[i] > 0 && !shapes[i]))
      return fail(e, FLOPE_EINVAL, "flope_load_weights: malformed entry " + std::to_string(i));
    std::vector<int64_t> shp(shapes[i], shapes[i] + ndims[i]);
    ts.t[names[i]] = std::make_pair(host_ptrs[i], shp);
  }
  std::vector<float> wf, bf;
  std::vector<std::vector<float>> host_bias;   // folded-BN bias of every conv, in e->convs order
  int rc;
  // stem
  if ((rc = fold(e, ts, "base.conv1", "base.bn1", 64, 3, 7, &wf, &bf)) != 0) return rc;
  if ((rc = upload(e, bf, (void**)&e->stem_bias)) != 0) return rc;
  if (e->dtype == FLOPE_DT_F32) { if ((rc = upload(e, naive_layout(wf, 64, 3, 7), (void**)&e->stem_w_naive)) != 0) return rc; }
  else {
    if ((rc = upload(e, pack_stem(wf, e->dtype), &e->stem_w)) != 0) return rc;
    if ((rc = upload(e, pack_stem_frag(wf, e->dtype), &e->stem_w2)) != 0) return rc;
  }
  for (Conv& c : e->convs) {
    if ((rc = fold(e, ts, c.name, c.bn, c.cout, c.cin, c.k, &wf, &bf)) != 0) return rc;
    if ((rc = upload(e, bf, (void**)&c.bias)) != 0) return rc;
    if (e->dtype == FLOPE_DT_F32) { if ((rc = upload(e, naive_layout(wf, c.cout, c.cin, c.k), (void**)&c.w_naive)) != 0) return rc; }
    else {
      if ((rc = upload(e, pack_conv(wf, c.cout, c.cin, c.k, e->dtype), &c.w_packed)) != 0) return rc;
      if (c.k == 3 && c.cin % 64 == 0 && (rc = upload(e, pack_conv32(wf, c.cout, c.cin, e->dtype), &c.w_stag)) != 0) return rc;
      if (c.k == 3 && c.stride == 1 && c.cin == 128 && c.cout == 128 && (rc = upload(e, pack_s1r(wf, c.cin, e->dtype), &c.w_s1r)) != 0) return rc;
      if (c.k == 3 && c.stride == 2 && c.cin == 64 && c.cout == 128 && (rc = upload(e, pack_s2r(wf, c.cout, c.cin, e->dtype), &c.w_s2r)) != 0) return rc;
      if (c.k == 1 && c.stride == 2 && c.cin == 64 && c.cout == 128 && (rc = upload(e, pack_s1r_ds(wf, c.cin, e->dtype), &c.w_ds_s1r)) != 0) return rc;
      if (c.k == 1 && c.cout >= 128 && c.cin % 64 == 0 && (rc = upload(e, pack_conv32_1x1(wf, c.cout, c.cin, e->dtype), &c.w_ds_stag)) != 0) return rc;
    }
    host_bias.push_back(bf);
  }
  // conv2 of a block with a shortcut conv: bias2 + bias_ds for the folded form (whether or not the plan uses it)
  for (size_t i = 0; i + 1 < e->convs.size(); ++i)
    if (e->convs[i].k == 1 && e->convs[i + 1].res_buf == e->convs[i].out_buf) {
      std::vector<float> sum = host_bias[i + 1];
      for (size_t j = 0; j < sum.size(); ++j) sum[j] += host_bias[i][j];
      if ((rc = upload(e, sum, (void**)&e->convs[i + 1].bias_fused)) != 0) return rc;
    }
  // head (fp32 as stored)
  const float* w1 = ts.get(e, "base.fc.0.weight", {e->bod, 512}, &rc); if (!w1) return rc;
  const float* b1 = ts.get(e, "base.fc.0.bias", {e->bod}, &rc); if (!b1) return rc;
  const float* w2 = ts.get(e, "fc_rot.weight", {9, e->bod}, &rc); if (!w2) return rc;
  const float* b2 = ts.get(e, "fc_rot.bias", {9}, &rc); if (!b2) return rc;
  auto chk = [&](const float* p, size_t cnt, const char* nm) {
    for (size_t i = 0; i < cnt; ++i) if (!std::isfinite(p[i])) return fail(e, FLOPE_EWEIGHTS, std::string("non-finite value in ") + nm);
    return 0;
  };
  if ((rc = chk(w1, (size_t)e->bod * 512, "base.fc.0.weight")) || (rc = chk(b1, e->bod, "base.fc.0.bias")) ||
      (rc = chk(w2, (size_t)9 * e->bod, "fc_rot.weight")) || (rc = chk(b2, 9, "fc_rot.bias"))) return rc;
  if ((rc = upload(e, std::vector<float>(w1, w1 + (size_t)e->bod * 512), (void**)&e->W1)) != 0) return rc;
  if (e->bod % 16 == 0 && (rc = upload(e, flope_host::pack_fc1(w1, e->bod, 512), (void**)&e->W1p)) != 0) return rc;
  if ((rc = upload(e, std::vector<float>(b1, b1 + e->bod), (void**)&e->b1)) != 0) return rc;
  if ((rc = upload(e, std::vector<float>(w2, w2 + (size_t)9 * e->bod), (void**)&e->W2)) != 0) return rc;
  if ((rc = upload(e, std::vector<float>(b2, b2 + 9), (void**)&e->b2)) != 0) return rc;
  HIP_TRY(e, hipDeviceSynchronize());
  e->weights_loaded = true;
  return FLOPE_OK;
}

// trunk: crop batch -> last BasicBlock output + pooled features
// One slice [start, start+batch) of the crop batch through the trunk + fc.0 on `stream`.  Every tensor is
// batch-major, so a slice is just an offset view of the same buffers.
// head: fc_rot + Procrustes of this slice on the slice's own stream (so a slice's head overlaps the other slice's
// trunk instead of running after the join); r9_dev / R_dev are the caller's full-batch buffers, head = false skips it.
// static part of the choice between conv_w4 (4 waves) and conv_stag for a flat 256 x 128 conv (split-K at small batches and
// persistent grids still take conv_stag at launch time)
static bool w4_eligible(const flope_engine* e, const Conv& c) {
  return e->opt_w4 && !e->opt_persist && c.stag == 1 && c.cout >= 128 && e->opt_skew && c.stag_patch_bytes >= 4 && c.stag_patch_bytes <= 6;
}

struct PoseOut { const float* xyz = nullptr; int nullify = 0; float* Rt = nullptr; };   // optional [B,16] pose assembly

static int run_slice(flope_engine* e, const void* x_dev, int in_format, int start, int batch, void* stream, bool marks,
                     bool head, float* r9_dev, float* R_dev, const PoseOut& po = PoseOut()) {
  const int dt = e->dtype;
  const size_t in_img_bytes = (size_t)e->H * e->W * 3 * (in_format == 0 ? 4 : (in_format == 3 ? 1 : 2));
  const char* x = (const char*)x_dev + (size_t)start * in_img_bytes;
  std::vector<Buf> vb = e->bufs;
  for (Buf& b : vb) b.ptr = (char*)b.ptr + (size_t)start * (b.h + 2) * (b.w + 2) * b.C * e->esz;
  char* stem_in = (char*)e->stem_in + (size_t)start * e->sHip * e->sWip * 4 * e->esz;
  float* feat = e->feat + (size_t)start * 512;
  float* hidden = e->hidden + (size_t)start * e->bod;
#define SMARK() do { if (marks) MARK(e, stream); } while (0)
  const Buf& bs = vb[e->stage_buf[FLOPE_STAGE_STEM]];
  const Buf& bp = vb[e->stage_buf[FLOPE_STAGE_POOL]];
  const bool fused = e->opt_fuse_stem && dt != FLOPE_DT_F32;
  if (fused) {
    SMARK();
#ifdef FLOPE_STAG_DBG
    flope_stem_pool_set_dbg(((e->opt_dbg & 64) && e->split_ws) ? (void*)(e->split_ws + (size_t)30 * (kDbgRegion / 4)) : nullptr);
#endif
    // r05: the register-weight form (weights in VGPRs, 51 KB of LDS: three workgroups per CU) where the option asks for it; else the
    // r02 forms -- persistent where that measured faster (same-run A/B at B = 256: 224 x 224 crops +2.7 % on the step; 512 x 512
    // crops -7 % on the kernel.  stem_persist: 1 = auto, 2 = always, 0 = never)
    const bool stem_r = e->opt_stem_r && e->stem_w2 && e->stem_q;
    K_TRY(e, "stem+maxpool", flope_stem_pool_launch(x, in_format, batch, e->H, e->W, e->Hs, e->Ws, bp.h, bp.w, e->stem_w, stem_r ? e->stem_w2 : nullptr,
                                                   stem_r ? e->stem_q + 1024 * (e->cur_slices > 1 ? e->mark_slice : 0) : nullptr,
                                                   e->stem_bias, bp.ptr, dt,
                                                   stem_r ? flope_stem_pool_r_blocks_per_cu() * e->num_cus
                                                          : (e->opt_stem_persist == 2 || (e->opt_stem_persist == 1 && bp.h * bp.w <= 64 * 64)) ? 2 * e->num_cus : 0,
                                                   stream));
  } else {
    SMARK();
    K_TRY(e, "prep_input", flope_prep_input_launch(x, in_format, batch, e->H, e->W, stem_in, e->sHip, e->sWip, dt, stream));
    if (dt == FLOPE_DT_F32) {
      NaiveConvP p; memset(&p, 0, sizeof(p));
      p.in = (const float*)stem_in; p.out = (float*)bs.ptr; p.w = e->stem_w_naive; p.bias = e->stem_bias;
      p.B = batch; p.Hip = e->sHip; p.Wip = e->sWip; p.Cin_stored = 4; p.Cin = 3; p.Ho = e->Hs; p.Wo = e->Ws;
      p.Hop = e->Hs + 2; p.Wop = e->Ws + 2; p.Cout = 64; p.KH = 7; p.KW = 7; p.stride = 2; p.in_off = 0; p.relu = 1;
      SMARK();
      K_TRY(e, "stem (fp32)", flope_naive_conv_launch(&p, stream));
    } else {
      StemP p; memset(&p, 0, sizeof(p));
      p.in = stem_in; p.out = bs.ptr; p.w = e->stem_w; p.bias = e->stem_bias;
      p.B = batch; p.Hip = e->sHip; p.Wip = e->sWip; p.Ho = e->Hs; p.Wo = e->Ws;
      p.tiles_per_image = e->stem_tiles; p.patch_rows_max = e->stem_rows;
      SMARK();
      K_TRY(e, "stem", flope_stem_launch(&p, dt, e->stem_lds, stream));
    }
    PoolP pp; pp.in = bs.ptr; pp.out = bp.ptr; pp.B = batch; pp.Hip = bs.h + 2; pp.Wip = bs.w + 2; pp.C = 64; pp.Ho = bp.h; pp.Wo = bp.w;
    SMARK();
    K_TRY(e, "maxpool", flope_maxpool_launch(&pp, dt, stream));
  }
  for (const Conv& c : e->convs) {
    if (c.folded) continue;                          // computed inside the next launch (conv_stag DSF)
    if (dt == FLOPE_DT_F32) {
      NaiveConvP p; memset(&p, 0, sizeof(p));
      p.in = (const float*)vb[c.in_buf].ptr; p.out = (float*)vb[c.out_buf].ptr;
      p.res = c.res_buf >= 0 ? (const float*)vb[c.res_buf].ptr : nullptr;
      p.w = c.w_naive; p.bias = c.bias;
      p.B = batch; p.Hip = c.hin + 2; p.Wip = c.win + 2; p.Cin_stored = c.cin; p.Cin = c.cin; p.Ho = c.hout; p.Wo = c.wout;
      p.Hop = c.hout + 2; p.Wop = c.wout + 2; p.Cout = c.cout; p.KH = c.k; p.KW = c.k; p.stride = c.stride;
      p.in_off = c.k == 3 ? 0 : 1; p.relu = c.relu;
      SMARK();
      K_TRY(e, c.name.c_str(), flope_naive_conv_launch(&p, stream));
    } else if (c.stag == 3) {
      ConvP p; conv_params(e, vb, c, batch, &p);
      p.w = c.w_stag; p.per_image = 0; p.mtiles = (p.M + 255) / 256; p.ntiles = c.cout / 128; p.total_tiles = p.mtiles * p.ntiles;
      SMARK();
      c.last_kernel = "conv_gstag_kernel<256x128,s2>"; c.last_detail.clear();
      K_TRY(e, c.name.c_str(), flope_conv_gstag_launch(&p, dt, stream));
    } else if (c.stag) {
      ConvP p; conv_params(e, vb, c, batch, &p);
      const int sbm = c.cout == 64 ? 512 : 256;
      p.w = c.w_stag; p.per_image = 0; p.mtiles = (p.M + sbm - 1) / sbm; p.ntiles = c.cout == 64 ? 1 : c.cout / 128; p.patch_rows_max = c.stag_patch_bytes;
      p.total_tiles = p.mtiles * p.ntiles;
      p.skew = e->opt_skew; p.prio = e->opt_prio;
      if (c.ds_conv >= 0) {
        const Conv& cd = e->convs[c.ds_conv];
        p.res = nullptr; p.bias = c.bias_fused;
        p.ds_in = vb[cd.in_buf].ptr; p.ds_w = cd.w_ds_stag; p.ds_Hip = cd.hin + 2; p.ds_Wip = cd.win + 2; p.ds_Cin = cd.cin;
      }
      if (e->opt_s1r && c.w_s1r && c.stag == 1 && (c.ds_conv < 0 || e->convs[c.ds_conv].w_ds_s1r)) {   // r05: weights in registers, K split over wave pairs (conv_s1r.hip)
        ConvP q = p;
        if (c.ds_conv >= 0) q.ds_w = e->convs[c.ds_conv].w_ds_s1r;
        if (flope_conv_s1r_ok(&q)) {
          SMARK();
          c.last_kernel = "conv_s1r_kernel<4rows x28>"; c.last_detail = c.ds_conv >= 0 ? "[shortcut folded in]" : "";
          if ((e->opt_dbg & 64) && e->split_ws) q.split_ws = e->split_ws + (size_t)(&c - &e->convs[0]) * (kDbgRegion / 4);
          K_TRY(e, c.name.c_str(), flope_conv_s1r_launch(&q, c.w_s1r, dt, e->num_cus, stream));
          continue;
        }
      }
      if (c.stag == 2) { p.per_image = 2; p.nseg = c.nseg; p.tiles_per_image = c.hout / 8 * c.nseg; p.mtiles = batch * p.tiles_per_image; p.total_tiles = p.mtiles; }
      // persistent grid: one workgroup per CU (a multiple of ntiles so a workgroup keeps its channel tile); the
      // row-band kernel is always persistent and shares the CUs with the other batch slices in flight
      int gridb = (e->opt_persist && c.ds_conv < 0)
                      ? std::min(p.total_tiles, std::max(1, (int)((long)e->num_cus * batch / std::max(1, e->cur_batch))))
                      : p.total_tiles;
      if (c.stag == 2)    // one workgroup per CU.  (r02 sized this grid to the slice's share of the CUs; two whole-chip grids
        // interleave more evenly: +0.7 .. +2.5 % on the two-slice step in un-profiled same-run pairs, DESIGN.md 9.7c.
        // rows_grid = -1 restores the share, > 0 sets the grid.)
        gridb = std::min(p.total_tiles, e->opt_rows_grid > 0 ? e->opt_rows_grid
                                          : e->opt_rows_grid < 0 ? std::max(1, (int)((long)e->num_cus * batch / std::max(1, e->cur_batch)))
                                                                 : e->num_cus);
      gridb -= gridb % p.ntiles;
      if (gridb < p.ntiles) gridb = p.ntiles;
      // r03: layer 1 (64 -> 64 on the 56-wide map) on the 4-wave row-band kernel (conv_r4.hip)
      if (c.stag == 2 && e->opt_r4 && c.nseg <= 1 && !(e->opt_dbg & 128) && flope_conv_r4_ok(&p)) {
        fastdiv_magic((unsigned)(p.Wip + 2), &p.mg_pitch, &p.sh_pitch);
        if ((e->opt_dbg & 64) && e->split_ws) p.split_ws = e->split_ws + (size_t)(&c - &e->convs[0]) * (kDbgRegion / 4);
        SMARK();
        c.last_kernel = "conv_r4_kernel<8rows x56>"; c.last_detail.clear();
        K_TRY(e, c.name.c_str(), flope_conv_r4_launch(&p, dt, gridb, stream));
        continue;
      }
      // split-K for small batches: with fewer tiles than half the CUs a tile's serial K loop (up to 72 double steps) is the
      // layer's latency; give every tile ksplit workgroups, each a share of the input channels.  A split pays its fp32
      // partial sums (128 KB per workgroup, written and read back) and a finalize launch, so the factor is chosen by a small
      // cost model fitted to B = 16 / 31 @ 512^2 (layer 3, 128 tiles, x2: 40 vs 37 us -- a loss; layer 4, 124 tiles, x2: a win;
      // layer 4, 64 tiles, x4: 37 vs 55 us): gain = T (1 - 1/s) - (5 us + 0.066 us * tiles * s), T = 0.75 us per double step.
      int ksp = 1;
      if (e->opt_ksplit && c.stag == 1 && e->plan_slices == 1 && !e->opt_persist && p.total_tiles * 2 <= e->num_cus) {   // plan_slices: the profile pass (one stream) times the kernels the production schedule of this batch runs
        const int bodies = c.cin / 64;
        const double T = 0.75 * 9.0 * bodies;
        double best = 0.0;
        for (int sp = 2; sp <= bodies && bodies % sp == 0 && p.total_tiles * sp <= e->num_cus; sp *= 2) {
          const double gain = e->opt_ksplit == 2 ? sp : T * (1.0 - 1.0 / sp) - (5.0 + 0.066 * p.total_tiles * sp);
          if (gain > best) { best = gain; ksp = sp; }
        }
      }
      ConvP pf;
      if (ksp > 1) {
        pf = p;                                    // the finalize kernel owns bias / residual / ReLU
        p.ksplit = ksp; p.split_ws = e->split_ws; p.res = nullptr;
        pf.ksplit = ksp; pf.split_ws = e->split_ws;
        gridb = p.total_tiles * ksp;
      }
      p.res_lds = (e->opt_reslds && p.res && c.stag == 1 && c.cout >= 128 && c.stag_patch_bytes >= 4 && ksp == 1 && gridb == p.total_tiles) ? 1 : 0;
      if ((e->opt_dbg & (64 | 128)) && ksp == 1 && e->split_ws)      // diagnostic build: clock stamps of this launch (flope_debug_read_ws)
        p.split_ws = e->split_ws + (size_t)(&c - &e->convs[0]) * (kDbgRegion / 4);
      // r03: flat 256 x 128 tiles, no split-K -> the 4-wave kernel (conv_w4.hip)
      if (w4_eligible(e, c) && ksp == 1 && gridb == p.total_tiles && !(e->opt_dbg & 128)) {
        // workgroup tiles of 256 .. 128 pixels (8 .. 4 pixel tiles per wave; one tile per workgroup): the cheapest by whole rounds
        // of the chip x the time of a tile -- ~15 k cycles of prologue + epilogue, and per double step 128 cycles of MFMAs per
        // pixel tile + ~500 of everything else (r03 stamps: 1.52 k at 8, 1.4 k at 7); ties go to the larger tile
        int mt = 8, ptr = c.stag_patch_bytes;
        if (e->opt_w4mt != 8) {
          const double dsteps = 9.0 * (c.cin / 64 + (p.ds_in ? 1 : 0));
          // Measured (profiles/r03_conv_w4_tile_height_ab.txt): with two batch slices in flight only 224 against 256 pays (+1.8 % on
          // the step; letting the choice go down to 128 or sizing it to the slice's share of the CUs loses 2 - 5 %: the other slice's
          // launches fill what a coarse tiling leaves idle).  A launch that has the chip to itself (one slice: batches below 64, the
          // profile pass) gains another ~12 % from 192 / 160-pixel tiles where they save a round.
          const int cus = e->num_cus;
          const int mt_lo = e->opt_w4mtlo ? e->opt_w4mtlo : (e->plan_slices == 1 ? 5 : 7);
          auto cost = [&](int m) {
            const int t = (p.M + 32 * m - 1) / (32 * m) * p.ntiles;
            return (double)((t + cus - 1) / cus) * (15000.0 + dsteps * (128.0 * m + 500.0));
          };
          double best = e->opt_w4mt ? 1e30 : cost(8);
          for (int m = 7; m >= mt_lo; --m) {
            if (!c.w4_patch[m] || (e->opt_w4mt && e->opt_w4mt != m)) continue;
            const double cm = cost(m);
            if (cm < best) { best = cm; mt = m; }
          }
        }
        // class walk (option w4cw = tiles per workgroup aimed at): 224-pixel tiles, every tile whole, a walk step of G tiles = whole
        // images (G a multiple of the tiling's period lcm(Ho Wo, 224) / 224), G | mtiles.  The largest k <= w4cw that allows it.
        int gw = 0, cw_imgs = 0;
        // Measured at B = 256 (profiles/r04_conv_w4_class_walk_ab.txt): one slice -2.3 % per step (layer 2 at 4 tiles per workgroup
        // -14 %, layer 3 at 2 tiles -5 %); with two slices in flight +0.5 % -- 896 tiles of 224 pixels are 3.5 per CU, equal walks
        // leave 32 CUs idle where the one-tile-per-workgroup launches of the two slices fill each other's gaps.  So: where a launch
        // has the chip to itself (option w4cwf overrides).
        if (e->opt_w4cw >= 2 && (e->plan_slices == 1 || (e->opt_w4cwf & 1)) && (e->opt_w4mt == 0 || e->opt_w4mt == 7) && c.w4_patch[7] && p.M % 224 == 0 &&
            flope_conv_w4_lds(c.w4_patch[7], 7, p.ds_in ? 1 : 0, 1) != 0) {
          const long hw = (long)c.hout * c.wout;
          long a_ = hw, b_ = 224; while (b_) { const long t_ = a_ % b_; a_ = b_; b_ = t_; }   // gcd
          const int period = (int)(hw / a_), mt7 = p.M / 224;
          // ... and that still fills this slice's share of the CUs (a walk of 112 workgroups on 256 CUs loses more than the tile
          // boundaries it saves: profiles/r04_conv_w4_class_walk_layers.txt)
          const long share = std::max(1L, (long)e->num_cus * batch / std::max(1, e->cur_batch));
          for (int k = std::min(e->opt_w4cw, mt7); k >= 2 && !gw; --k)
            if (mt7 % k == 0 && (mt7 / k) % period == 0 && ((e->opt_w4cwf & 2) || (long)(mt7 / k) * p.ntiles * 100 >= share * 85)) { gw = mt7 / k; cw_imgs = (int)((long)gw * 224 / hw); }
          if (gw) mt = 7;
        }
        if (mt != 8) { ptr = c.w4_patch[mt]; p.mtiles = (p.M + 32 * mt - 1) / (32 * mt); p.total_tiles = p.mtiles * p.ntiles; p.patch_rows_max = ptr; }
        const int grid_w4 = gw ? gw * p.ntiles : p.total_tiles;
        p.cw_imgs = gw ? cw_imgs : 0;
        if (flope_conv_w4_lds(ptr, mt, p.ds_in ? 1 : 0, gw ? 1 : 0) != 0) {
          fastdiv_magic((unsigned)(p.Wip + 2), &p.mg_pitch, &p.sh_pitch);
          SMARK();
          {
            char d_[96];
            if (gw) snprintf(d_, sizeof d_, "[%d px tiles, walk: %d workgroups x %d tiles]", 32 * mt, grid_w4, p.mtiles / gw);
            else snprintf(d_, sizeof d_, "[%d px tiles]", 32 * mt);
            c.last_kernel = "conv_w4_kernel<256x128>"; c.last_detail = d_;
          }
          K_TRY(e, c.name.c_str(), flope_conv_w4_launch(&p, dt, grid_w4, mt, stream));
          continue;
        }
        p.cw_imgs = 0;
        if (mt != 8) { p.mtiles = (p.M + 255) / 256; p.total_tiles = p.mtiles * p.ntiles; p.patch_rows_max = c.stag_patch_bytes; }
      }
      size_t lds_bytes = c.stag_lds;
      if ((e->opt_dbg & 128) && lds_bytes + 2048 <= kLdsMax) { p.dbg_lds_off = (int)lds_bytes; lds_bytes += 2048; }
      else if (e->opt_dbg & 128) p.dbg &= ~128;
      SMARK();
      c.last_kernel = c.stag == 2 ? "conv_stag_kernel<8rows x64>" : (c.cout == 64 ? "conv_stag_kernel<512x64>" : "conv_stag_kernel<256x128>");
      c.last_detail = ksp > 1 ? "[split-K x" + std::to_string(ksp) + "]" : std::string();
      K_TRY(e, c.name.c_str(), flope_conv_stag_launch(&p, dt, gridb, lds_bytes, stream));
      if (ksp > 1) K_TRY(e, c.name.c_str(), flope_conv_split_finalize_launch(&pf, dt, stream));
    } else {
      ConvP p; conv_params(e, vb, c, batch, &p);
      if (e->opt_s2r && c.w_s2r && flope_conv_s2r_ok(&p)) {     // r05: patch in LDS, weights through registers (conv_s2r.hip)
        if ((e->opt_dbg & 64) && e->split_ws) p.split_ws = e->split_ws + (size_t)(&c - &e->convs[0]) * (kDbgRegion / 4);
        SMARK();
        c.last_kernel = "conv_s2r_kernel<4rows x28>"; c.last_detail.clear();
        K_TRY(e, c.name.c_str(), flope_conv_s2r_launch(&p, c.w_s2r, dt, e->opt_s2r_grid > 0 ? e->opt_s2r_grid : e->num_cus, stream));
        continue;
      }
      SMARK();
      c.last_kernel.clear(); c.last_detail.clear();       // (flope_launch_info derives conv_mfma's label from the static plan)
      K_TRY(e, c.name.c_str(), flope_conv_mfma_launch(&p, dt, c.cfg, c.patch, c.nbuf, c.lds + (size_t)e->opt_ldspad * 1024, stream));
    }
  }
  const Buf& bl = vb[e->final_buf];
  SMARK();
  K_TRY(e, "avgpool", flope_avgpool_launch(bl.ptr, feat, batch, bl.h, bl.w, 512, dt, stream));
  SMARK();
  K_TRY(e, "fc1", flope_fc1_launch(feat, e->W1, e->opt_fc1_packed ? e->W1p : nullptr, e->b1, hidden, batch, 512, e->bod, stream));
  if (head) {
    SMARK();
    float* r9 = (r9_dev ? r9_dev : e->r9_scratch) + (size_t)start * 9;
    float* Rp = R_dev ? R_dev + (size_t)start * 9 : nullptr;
    const float* xyzp = po.xyz ? po.xyz + (size_t)start * 3 : nullptr;
    float* Rtp = po.Rt ? po.Rt + (size_t)start * 16 : nullptr;
    int k4 = 0;
    if (e->opt_fc2_k4) {
      k4 = flope_fc2_procrustes_k4_launch(hidden, e->W2, e->b2, r9, Rp, batch, e->bod, xyzp, po.nullify, Rtp, stream);
      if (k4 < 0) return fail(e, FLOPE_EHIP, "fc_rot+procrustes: launch failed");
    }
    if (!k4) K_TRY(e, "fc_rot+procrustes", flope_fc2_procrustes_launch(hidden, e->W2, e->b2, r9, Rp, batch, e->bod, xyzp, po.nullify, Rtp, stream));
  }
  SMARK();
#undef SMARK
  return FLOPE_OK;
}

// trunk + fc.0 for the whole batch.  With the "streams" option (default 2) and a large enough batch the
// crops are split into two halves that run the same launch sequence on two internal streams forked from /
// joined to the caller's stream: the tail of one half's kernel (the last, partly filled round of
// workgroups -- up to 24 % of a launch at B = 256) overlaps the head of the other half's.
static int run_trunk(flope_engine* e, const void* x_dev, int in_format, int batch, void* stream, bool head = false,
                     float* r9_dev = nullptr, float* R_dev = nullptr, const PoseOut& po = PoseOut()) {
  if (!e->weights_loaded) return fail(e, FLOPE_ESTATE, "forward before flope_load_weights");
  if (!x_dev) return fail(e, FLOPE_EINVAL, "forward: x_dev is NULL");
  if (batch < 1 || batch > e->maxB) return fail(e, FLOPE_EINVAL, "forward: batch must be within 1..max_batch");
  if (in_format < 0 || in_format > 3) return fail(e, FLOPE_EINVAL, "forward: unknown input format");
  HIP_TRY(e, hipSetDevice(e->device));               // the handle's device, whatever the caller's current device is
  e->ev_n = 0;
  e->last_fused = e->opt_fuse_stem && e->dtype != FLOPE_DT_F32;
  e->last_batch = batch;
  int ns = e->opt_streams >= 2 ? e->opt_streams : 1;
  while (ns > 1 && batch / ns < 32) --ns;              // keep every slice large enough to fill the chip
  // profile = 1 times every launch on ONE stream -- but with the kernel variants (tile heights, class walk) the production
  // schedule of this batch picks, so that the per-launch table describes the kernels the un-profiled step runs
  e->plan_slices = ns;
  if (e->opt_profile == 1) ns = 1;
  e->cur_slices = ns;
  e->cur_batch = batch;
  e->mark_slice = 0;
  if (ns == 1) return run_slice(e, x_dev, in_format, 0, batch, stream, true, head, r9_dev, R_dev, po);
  hipStream_t user = (hipStream_t)stream;
  if (e->opt_profile == 2) MARK(e, user);             // time zero of flope_profile_timeline
  HIP_TRY(e, hipEventRecord(e->ev_fork, user));
  // slices are launched layer-interleaved?  No: each slice's whole sequence goes to its own stream; the
  // hardware queues interleave them, and a slice's short tail round overlaps another slice's next launch.
  int rc_all = FLOPE_OK, forked = 0;
  for (int s = 0; s < ns && rc_all == FLOPE_OK; ++s) {
    if (hipStreamWaitEvent(e->side[s], e->ev_fork, 0) != hipSuccess) { rc_all = fail(e, FLOPE_EHIP, "hipStreamWaitEvent(fork) failed"); break; }

    forked = s + 1;
    // slice boundaries on multiples of 8 images (whole tiles in every layer) when the batch allows it
    auto bound = [&](int k) { const int b_ = (int)((long)batch * k / ns); return (batch >= 16 * ns && k > 0 && k < ns) ? ((b_ + 4) & ~7) : b_; };
    int start = bound(s), cnt = bound(s + 1) - start;
    if (ns == 2) {
      // Two slices of 3/8 and 5/8 of the batch (multiples of 8 images, so every layer's tiles stay whole) instead of
      // two halves (B = 256: 96/160 1.199 ms vs 128/128 1.215 ms, same-run A/B in r02; "split" overrides).  r03 time line
      // (option profile = 2, tools/slice_timeline.py): the two slices walk the same layers side by side and finish within
      // microseconds of each other -- the uneven sizes change the tile counts that share the chip, not the phase.
      int first = e->opt_split > 100 ? e->opt_split - 100 : (int)((long)batch * e->opt_split / 100);
      // r04: equal halves by default -- with the r04 kernels 128/128 beats 96/160 by 1 - 2.5 % in every autotune run
      // (profiles/r04_autotune_runs.txt); PoseEngine.autotune still tries 3/8 and 7/16
      if (e->opt_split == 0) first = batch >= 128 ? (batch / 2) & ~7 : batch / 2;
      first = std::max(1, std::min(batch - 1, first));
      start = s == 0 ? 0 : first; cnt = s == 0 ? first : batch - first;
    }
    e->mark_slice = s;
    // the last slice starts ~20 us late (one sleeping wave): the time line (profile = 2) shows the slices walking the same layers
    // side by side, every pair of launches starting in the same microsecond -- i.e. their prologue fills and epilogue drains
    // coincide; a small offset is +0.7 .. +1.4 % on the step (5 .. 30 us all do; 80 us and more lose: profiles/r03_slice_lag.txt)
    // (measured at B = 256 x 224 x 224 only: applied from 192 crops up, where a step is >= 35 x the offset)
    if (e->opt_lag && s == ns - 1 && batch >= 192) {
      hipLaunchKernelGGL(lag_kernel, dim3(1), dim3(64), 0, e->side[s], e->opt_lag);
      if (hipGetLastError() != hipSuccess) { rc_all = fail(e, FLOPE_EHIP, "lag kernel launch failed"); break; }
    }
    rc_all = run_slice(e, x_dev, in_format, start, cnt, e->side[s], e->opt_profile == 2, head, r9_dev, R_dev, po);
  }
  // join every stream that was forked -- also after a failed launch, so that work already queued on the side
  // streams stays ordered before the caller's next use of x / r9 / R / Rt
  const std::string first_err = rc_all != FLOPE_OK ? e->err : std::string();
  for (int s = 0; s < forked; ++s) {
    if (hipEventRecord(e->ev_join[s], e->side[s]) != hipSuccess || hipStreamWaitEvent(user, e->ev_join[s], 0) != hipSuccess) {
      hipStreamSynchronize(e->side[s]);
      if (rc_all == FLOPE_OK) rc_all = fail(e, FLOPE_EHIP, "joining the batch-slice streams failed");
    }
  }
  if (rc_all != FLOPE_OK) {
    if (!first_err.empty()) { e->err = first_err; g_last_error = first_err; }
    e->cur_slices = 1; e->plan_slices = 1; e->cur_batch = 1; e->last_batch = 0;
  }
  return rc_all;
}

extern "C" int flope_forward(flope_handle e, const void* x_dev, int in_format, int batch, float* r9_dev, float* R_dev,
                             void* stream) {
  if (!e) return fail(nullptr, FLOPE_EINVAL, "flope_forward: NULL handle");
  return run_trunk(e, x_dev, in_format, batch, stream, true, r9_dev, R_dev);
}

extern "C" int flope_forward_poses(flope_handle e, const void* x_dev, int in_format, int batch, const float* xyz_dev,
                                   int nullify_yaw, float* r9_dev, float* R_dev, float* Rt_dev, void* stream) {
  if (!e) return fail(nullptr, FLOPE_EINVAL, "flope_forward_poses: NULL handle");
  if (!Rt_dev) return fail(e, FLOPE_EINVAL, "flope_forward_poses: Rt_dev is NULL");
  PoseOut po; po.xyz = xyz_dev; po.nullify = nullify_yaw != 0; po.Rt = Rt_dev;
  return run_trunk(e, x_dev, in_format, batch, stream, true, r9_dev, R_dev, po);
}

extern "C" int flope_extract_features(flope_handle e, const void* x_dev, int in_format, int batch, float* feat_dev,
                                      void* stream) {
  if (!e) return fail(nullptr, FLOPE_EINVAL, "flope_extract_features: NULL handle");
  if (!feat_dev) return fail(e, FLOPE_EINVAL, "flope_extract_features: feat_dev is NULL");
  int rc = run_trunk(e, x_dev, in_format, batch, stream);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(feat_dev, e->hidden, (size_t)batch * e->bod * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return FLOPE_OK;
}

extern "C" int flope_read_stage(flope_handle e, int stage, int batch, float* dst_dev, int64_t* dims_out, void* stream) {
  if (!e) return fail(nullptr, FLOPE_EINVAL, "flope_read_stage: NULL handle");
  if (!dst_dev || !dims_out) return fail(e, FLOPE_EINVAL, "flope_read_stage: NULL argument");
  if (batch < 1 || batch > e->maxB) return fail(e, FLOPE_EINVAL, "flope_read_stage: bad batch");
  HIP_TRY(e, hipSetDevice(e->device));
  if (stage == FLOPE_STAGE_FEAT || stage == FLOPE_STAGE_HIDDEN) {
    const int nn = stage == FLOPE_STAGE_FEAT ? 512 : e->bod;
    dims_out[0] = batch; dims_out[1] = nn; dims_out[2] = 1; dims_out[3] = 1;
    HIP_TRY(e, hipMemcpyAsync(dst_dev, stage == FLOPE_STAGE_FEAT ? e->feat : e->hidden, (size_t)batch * nn * sizeof(float),
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FLOPE_OK;
  }
  if (stage < 0 || stage > 9) return fail(e, FLOPE_EINVAL, "flope_read_stage: unknown stage");
  if (stage == FLOPE_STAGE_STEM && e->last_fused)
    return fail(e, FLOPE_ESTATE, "flope_read_stage: the stem activation is not materialised by the fused stem+maxpool kernel (set option fuse_stem=0)");
  const Buf& b = e->bufs[e->stage_buf[stage]];
  dims_out[0] = batch; dims_out[1] = b.C; dims_out[2] = b.h; dims_out[3] = b.w;
  K_TRY(e, "read_stage", flope_read_stage_launch(b.ptr, dst_dev, batch, b.C, b.h, b.w, e->dtype, stream));
  return FLOPE_OK;
}

extern "C" double flope_forward_flops(flope_handle e, int batch) {
  if (!e) return 0.0;
  double macs = (double)e->Hs * e->Ws * 64 * 147;
  for (const Conv& c : e->convs) macs += (double)c.hout * c.wout * c.cout * c.cin * c.k * c.k;
  macs += 512.0 * e->bod + 9.0 * e->bod;
  return 2.0 * macs * batch;
}

static int head_launches(const flope_engine* e) { return (e->opt_fuse_stem && e->dtype != FLOPE_DT_F32) ? 1 : 3; }   // front of the trunk: input + stem + maxpool
static int tail_launches(const flope_engine*) { return 3; }                                                           // avgpool, fc.0, fc_rot + Procrustes
extern "C" int flope_forward_launches(flope_handle e) {
  if (!e) return 0;
  int n = (int)e->convs.size() + tail_launches(e) + head_launches(e);
  for (const Conv& c : e->convs) n -= c.folded;
  return n;
}

// profile mode ("profile" option): per-launch GPU time of the LAST flope_forward, from HIP
// events recorded on the caller's stream around every launch.  Synchronises on the last event.
extern "C" int flope_profile_read(flope_handle e, float* ms_out, int cap) {
  if (!e || !ms_out) return fail(e, FLOPE_EINVAL, "flope_profile_read: NULL argument");
  if (e->opt_profile != 1 || e->ev_n < 2) return fail(e, FLOPE_ESTATE, "flope_profile_read: no forward with option profile = 1 (profile = 2 records a time line: flope_profile_timeline)");
  HIP_TRY(e, hipEventSynchronize(e->ev[e->ev_n - 1]));
  const int n = std::min(cap, e->ev_n - 1);
  for (int i = 0; i < n; ++i) HIP_TRY(e, hipEventElapsedTime(&ms_out[i], e->ev[i], e->ev[i + 1]));
  return n;
}

// profile = 2: the last forward as it ran in production (slices on their own streams), as a time line: event i was recorded on
// slice slice_out[i]'s stream in front of that slice's next launch (behind its last one), ms_out[i] = milliseconds since the fork.
// Event 0 is the fork itself.  Returns the number of events.
extern "C" int flope_profile_timeline(flope_handle e, float* ms_out, int* slice_out, int cap) {
  if (!e || !ms_out || !slice_out) return fail(e, FLOPE_EINVAL, "flope_profile_timeline: NULL argument");
  if (e->opt_profile != 2 || e->ev_n < 2) return fail(e, FLOPE_ESTATE, "flope_profile_timeline: no forward with option profile = 2");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipDeviceSynchronize());
  const int n = std::min(cap, e->ev_n);
  for (int i = 0; i < n; ++i) {
    ms_out[i] = 0.f;
    if (i > 0) HIP_TRY(e, hipEventElapsedTime(&ms_out[i], e->ev[0], e->ev[i]));
    slice_out[i] = e->ev_slice[i];
  }
  return n;
}

// launch idx of flope_forward: "layer|kernel" label and its algorithmic FLOPs for `batch` crops
extern "C" int flope_launch_info(flope_handle e, int idx, int batch, char* name, int name_cap, double* flops) {
  if (!e || !name || name_cap < 1 || !flops) return fail(e, FLOPE_EINVAL, "flope_launch_info: NULL argument");
  std::vector<int> live;                             // convs that are launched (folded shortcuts are not)
  for (size_t i = 0; i < e->convs.size(); ++i) if (!e->convs[i].folded) live.push_back((int)i);
  const int nc = (int)live.size();
  const int nh = head_launches(e);
  if (idx < 0 || idx >= nc + tail_launches(e) + nh) return fail(e, FLOPE_EINVAL, "flope_launch_info: bad index");
  const bool f32 = e->dtype == FLOPE_DT_F32;
  std::string s; double f = 0.0;
  if (nh == 1) {
    if (idx == 0) { s = "input+stem+maxpool|stem_pool_kernel"; f = 2.0 * e->Hs * e->Ws * 64 * 147; }
  } else {
    if (idx == 0) s = "prep_input|prep_input_kernel";
    else if (idx == 1) { s = f32 ? "stem|naive_conv_kernel" : "stem|stem_mfma_kernel"; f = 2.0 * e->Hs * e->Ws * 64 * 147; }
    else if (idx == 2) s = "maxpool|maxpool_kernel";
  }
  idx += 3 - nh;
  if (idx < 3) { /* named above */ }
  else if (idx < 3 + nc) {
    const Conv& c = e->convs[live[idx - 3]];
    int BM, BN; tile_dims(c.cfg, &BM, &BN);
    char k[96];
    if (f32) snprintf(k, sizeof k, "naive_conv_kernel");
    else if (c.stag == 3) snprintf(k, sizeof k, "conv_gstag_kernel<256x128,s2>");
    else if (e->opt_s1r && c.w_s1r && (c.ds_conv < 0 || e->convs[c.ds_conv].w_ds_s1r) && c.stag == 1 && c.wout == 28 && c.hout % 4 == 0) snprintf(k, sizeof k, "conv_s1r_kernel<4rows x28>");
    else if (w4_eligible(e, c)) snprintf(k, sizeof k, "conv_w4_kernel<256x128>");
    else if (c.stag == 2 && e->opt_r4 && c.nseg <= 1 && c.cin == 64 && c.cout == 64 && c.wout == 56 && c.hout % 8 == 0) snprintf(k, sizeof k, "conv_r4_kernel<8rows x56>");
    else if (c.stag) snprintf(k, sizeof k, c.stag == 2 ? "conv_stag_kernel<8rows x64>" : (c.cout == 64 ? "conv_stag_kernel<512x64>" : "conv_stag_kernel<256x128>"));
    else if (e->opt_s2r && c.w_s2r && c.wout == 28 && c.hout % 4 == 0) snprintf(k, sizeof k, "conv_s2r_kernel<4rows x28>");
    else snprintf(k, sizeof k, "conv_mfma_kernel<%dx%d,%s,ring%d>", BM, BN, c.patch ? "patch" : "gather", c.nbuf);
    // a forward has run: the kernel it actually launched (split-K and shapes the 4-wave kernels do not take go to conv_stag at
    // launch time, whatever the static plan says) and the launch's tiling
    if (!f32 && e->last_batch > 0 && !c.last_kernel.empty()) snprintf(k, sizeof k, "%s", c.last_kernel.c_str());
    s = c.name + ((!f32 && e->last_batch > 0) ? c.last_detail : std::string()) + "|" + k;
    f = 2.0 * c.hout * c.wout * c.cout * c.cin * c.k * c.k;
    if (c.ds_conv >= 0) {
      const Conv& cd = e->convs[c.ds_conv];
      s = c.name + "+shortcut" + ((!f32 && e->last_batch > 0) ? c.last_detail : std::string()) + "|" + k;
      f += 2.0 * cd.hout * cd.wout * cd.cout * cd.cin;
    }
  } else if (idx == 3 + nc) s = "avgpool|avgpool_kernel";
  else if (idx == 4 + nc) { s = "fc1|fc1_kernel"; f = 2.0 * 512 * e->bod; }
  else { s = "fc_rot+procrustes|fc2_procrustes_kernel"; f = 2.0 * 9 * e->bod; }
  snprintf(name, name_cap, "%s", s.c_str());
  *flops = f * batch;
  return FLOPE_OK;
}

// plan introspection for DESIGN.md / tests: writes one line per conv into buf
extern "C" int flope_describe_plan(flope_handle e, char* buf, int buflen) {
  if (!e || !buf || buflen < 1) return FLOPE_EINVAL;
  std::string s;
  char line[256];
  snprintf(line, sizeof line, "stem: tiles/img=%d rows=%d lds=%zu\n", e->stem_tiles, e->stem_rows, e->stem_lds);
  s += line;
  for (const Conv& c : e->convs) {
    if (c.stag == 3) { snprintf(line, sizeof line, "%s: 3x3 s2 %d->%d out %dx%d conv_gstag 256x128 (gathered tiles) lds=147456\n", c.name.c_str(), c.cin, c.cout, c.hout, c.wout); s += line; continue; }
    if (c.folded) { snprintf(line, sizeof line, "%s: 1x1 s2 %d->%d out %dx%d folded into the next conv (conv_stag DSF)\n", c.name.c_str(), c.cin, c.cout, c.hout, c.wout); s += line; continue; }
    if (c.stag && c.ds_conv >= 0) { snprintf(line, sizeof line, "%s: 3x3 s1 %d->%d out %dx%d %s 256x128 patch_rounds=%d lds=%zu, shortcut folded in (+%d K)\n", c.name.c_str(), c.cin, c.cout, c.hout, c.wout, w4_eligible(e, c) ? "conv_w4" : "conv_stag", c.stag_patch_bytes, c.stag_lds, e->convs[c.ds_conv].cin); s += line; continue; }
    if (c.stag == 2 && c.nseg > 1) { snprintf(line, sizeof line, "%s: 3x3 s1 %d->%d out %dx%d conv_stag 8-row bands x %d column segments of 64 patch_rounds=%d lds=%zu\n", c.name.c_str(), c.cin, c.cout, c.hout, c.wout, c.nseg, c.stag_patch_bytes, c.stag_lds); s += line; continue; }
    if (c.stag == 1 && w4_eligible(e, c)) { snprintf(line, sizeof line, "%s: 3x3 s1 %d->%d out %dx%d conv_w4 256x128 patch_rounds=%d lds=%zu\n", c.name.c_str(), c.cin, c.cout, c.hout, c.wout, c.stag_patch_bytes, flope_conv_w4_lds(c.stag_patch_bytes, 8, 0, 0)); s += line; continue; }
    if (c.stag) { snprintf(line, sizeof line, c.stag == 2 ? "%s: 3x3 s1 %d->%d out %dx%d conv_stag 8-row bands x 64 patch_rounds=%d lds=%zu\n" : "%s: 3x3 s1 %d->%d out %dx%d conv_stag 256x128 patch_rounds=%d lds=%zu\n", c.name.c_str(), c.cin, c.cout, c.hout, c.wout, c.stag_patch_bytes, c.stag_lds); s += line; continue; }
    snprintf(line, sizeof line, "%s: %dx%d s%d %d->%d out %dx%d cfg=%d patch=%d ring=%d per_image=%d rows=%d lds=%zu\n", c.name.c_str(),
             c.k, c.k, c.stride, c.cin, c.cout, c.hout, c.wout, c.cfg, c.patch, c.nbuf, c.per_image, c.rows_max, c.lds);
    s += line;
  }
  snprintf(buf, buflen, "%s", s.c_str());
  return FLOPE_OK;
}
