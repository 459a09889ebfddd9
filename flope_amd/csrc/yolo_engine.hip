// C-ABI of the YOLO11-seg detector front end (include/flope_amd.h, flope_yolo_*): replaces
//   self.yolo = YOLO(yolo_path)                      sunflower/predictor/fast_pose_predictor.py:36
//   results = self.yolo(image); masks / boxes ...    :44-57 (get_bbox_mask)
// The network definition, pre- and post-processing are ultralytics 8.3.27's (environment.yml:231; not vendored by the
// reference, absent here): the graph below is built from the checkpoint's own state_dict -- module names
// `model.<i>. ...`, repeat counts and the C3k switch from the key set, channel widths from the tensor shapes -- so it
// covers the yolo11{n,s,m,l,x}-seg family without a scale table.  PARITY UNPINNED against ultralytics itself.
//
// Memory: every intermediate map is allocated once at flope_yolo_load_weights (the graph is known then); concat /
// chunk / split are channel-slice views into shared buffers; flope_yolo_detect allocates and synchronises nothing.
#include "../../include/flope_amd.h"
#include "common.h"
#include "host_pack.h"
#include "yolo.h"

#include <math.h>
#include <algorithm>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

extern "C" int flope_yconv_launch(const YConvP* p, int dtype, int nt, void* stream);
extern "C" int flope_ydw_launch(const YDwP* p, int dtype, void* stream);
extern "C" int flope_ymulti_add_conv(YMultiP* m, const YConvP* p, int nt);
extern "C" int flope_ymulti_add_dw(YMultiP* m, const YDwP* p);
extern "C" int flope_ymulti_add_bneck(YMultiP* m, const YConvP* c1, int nt1, const YConvP* c2, int nt2);
extern "C" int flope_ybneck_fusable(const YConvP* c1, int nt1, const YConvP* c2, int nt2);
extern "C" int flope_ybneck_launch(const YConvP* c1, int nt1, const YConvP* c2, int nt2, int dtype, void* stream);
extern "C" int flope_ymulti_launch(const YMultiP* m, const YMultiP* m_dev, int dtype, void* stream);
extern "C" int flope_yconv_xcd_mode(int mode);
extern "C" int flope_yconv_tile_mode(int mode);
extern "C" int flope_yconv_splitk_max_m(int m);
extern "C" int flope_ypool_lds_mode(int mode);
extern "C" int flope_yconv_wlds_mode(int mode);
extern "C" int flope_ypool_launch(const YPoolP* p, int dtype, void* stream);
extern "C" int flope_yup_launch(const YUpP* p, void* stream);
extern "C" int flope_yattn_init();
extern "C" int flope_yattn_launch(const YAttnP* p, int dtype, int variant, void* stream);
extern "C" int flope_yletter_launch(const YLetterP* p, int dtype, void* stream);
extern "C" int flope_ydecode_launch(const YDecodeP* p, void* stream);
extern "C" int flope_ynms_launch(const YNmsP* p, void* stream);
extern "C" int flope_ymask_launch(const YMaskP* p, int dtype, void* stream);
extern "C" int flope_resize_linear_u8_launch(const uint8_t* in, int h, int w, uint8_t* out, int H, int W, void* stream);
extern "C" int flope_yread_launch(const void* src, int is_f32, int H, int W, int C, int ld, int dtype, float* dst, void* stream);
// strict float32 mode (yolo_f32.hip): the same graph on float32 maps, plain fused-multiply-add convolutions
extern "C" int flope_y32_conv_launch(const YConvP* p, void* stream);
extern "C" int flope_y32m_conv_launch(const YConvP* p, void* stream);
extern "C" int flope_y32m_conv_ok(const YConvP* p);
extern "C" int flope_ychain_ok(const YConvP* p, int nt);
extern "C" int flope_ychain_launch(const YChainP* c, const YChainP* c_dev, int dtype, void* stream);
extern "C" int flope_y32m_chain_ok(const YConvP* p);
extern "C" int flope_y32m_chain_launch(const YChainP* c, const YChainP* c_dev, void* stream);
extern "C" int flope_y32m_multi_add_conv(YMultiP* m, const YConvP* p);
extern "C" int flope_y32m_multi_add_dw(YMultiP* m, const YDwP* p);
extern "C" int flope_y32m_multi_launch(const YMultiP* m, const YMultiP* m_dev, void* stream);
extern "C" int flope_y32_dw_launch(const YDwP* p, void* stream);
extern "C" int flope_y32_pool_launch(const YPoolP* p, void* stream);
extern "C" int flope_y32_up_launch(const YUpP* p, void* stream);
extern "C" int flope_y32_attn_init();
extern "C" int flope_y32_pool_init();
extern "C" int flope_y32_attn_launch(const YAttnP* p, void* stream);
extern "C" int flope_y32m_attn_launch(const YAttnP* p, void* stream);
extern "C" int flope_y32_letter_launch(const YLetterP* p, void* stream);

using namespace flope_host;

namespace {

thread_local std::string g_yolo_error;
constexpr double kYoloBnEps = 1e-3;            // ultralytics: BatchNorm2d(eps=0.001)
constexpr int kRegMax = 16, kNm = 32, kMaxDet = 300;
constexpr size_t kGraphCache = 8;              // one captured sequence per (frame, thresholds, output buffers) tuple in use

struct View { int t = -1, off = 0, C = 0; };   // channel slice [off, off + C) of tensor t
struct Tensor { void* ptr = nullptr; int H = 0, W = 0, C = 0; };
struct Rng { int t, c0, c1; };                 // what an op touches: channels [c0, c1) of tensor t (t < 0: prediction-row columns)

struct Op {
  enum Kind { CONV, DW, POOL, UP, ATTN, BNECK } kind;     // BNECK: Bottleneck cv1 -> cv2 (conv, conv2), one fused launch
  int nt = 4, nt2 = 4;
  int level = 0;                  // longest dependency chain before this op (ops of one level are independent)
  YConvP conv, conv2; YDwP dw; YPoolP pool; YUpP up; YAttnP attn;
  std::vector<Rng> reads, writes;
  std::string name;
};

// One launch of the schedule: a single op, or up to kYMultiMax independent conv / depthwise ops of one level in one grid.
// (r05) ... or a CHAIN of consecutive 1x1 convs on one small map, run back to back by one grid (yolo.h YChainP).
struct Launch { int op = -1; YMultiP multi; YMultiP* multi_dev = nullptr; std::vector<int> members; YChainP chain; YChainP* chain_dev = nullptr; };

struct Tap { int is_f32 = 0; const void* ptr = nullptr; int H = 0, W = 0, C = 0, ld = 0; };

struct GraphKey {                 // what a captured detect sequence bakes in
  const void* frame; void* det; void* count; void* mask; float conf, iou; int max_det, batch, generic_attn;
};

}  // namespace

struct flope_yolo {
  int device = 0, H = 0, W = 0, imgsz = 0, dtype = 1;
  int h = 0, w = 0, nh = 0, nw = 0, top = 0, left = 0;      // letterbox geometry
  bool loaded = false;
  int nc = 0, no = 0, A = 0;
  std::vector<Tensor> tensors;
  std::vector<void*> owned;                                   // weights, biases, scratch
  std::vector<Op> ops;                                        // program order of the ultralytics graph
  std::vector<Launch> sched[2];                               // [0]: one launch per op in program order; [1]: levels, batched
  std::map<std::string, Tap> taps;
  void* zero = nullptr;
  float* pred = nullptr;
  YLetterP letter; YDecodeP dec; YNmsP nms; YMaskP mask;
  uint8_t* merged = nullptr;
  int opt_generic_attn = 0;                                   // A/B + parity of the two attention kernels
  // The detector is a chain of ~100 launches of 5-12 us each on one frame: the GPU, not the host, is the bound (hipGraph
  // replay 1.196 vs eager 1.188 ms, r02), and independent branches on side streams cost more in cross-stream dependencies
  // than the overlap returned (1.31 ms; removed).  What does pay is putting the independent ops of one dependency level
  // into ONE grid (ymulti_kernel): option "batch", default on.
  int opt_batch = 1;
  int opt_chain = 0;                                          // runs of 1x1 convs on one small map as one launch (r05: built, measured no faster -- profiles/r05_yolo_chain.txt; off)
  int opt_f32mfma = 1;                                        // float32 mode: convolutions on the exact-fp32 MFMA kernel (0: the plain fused-multiply-add kernels, its checker)
  int opt_bneck = 1;                                          // 1: Bottleneck pairs as one fused launch (ybneck_kernel); 0: two conv launches
  int opt_graph = 0;                                          // 1: flope_yolo_detect replays a captured hipGraph; 0 (default)
  struct Captured { GraphKey key; hipGraphExec_t exec; hipEvent_t done; };   // done: recorded behind the last replay
  std::vector<Captured> graphs;                               // captured detect sequences, most recently used last (<= kGraphCache)
  double flops = 0.0;
  std::string err;
};

namespace {

int yfail(flope_yolo* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  g_yolo_error = msg;
  return code;
}

#define Y_TRY(e, call)                                                                          \
  do {                                                                                          \
    hipError_t _s = (call);                                                                     \
    if (_s != hipSuccess) return yfail(e, FLOPE_EHIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
  } while (0)

struct Builder {
  flope_yolo* e;
  std::map<std::string, std::pair<const float*, std::vector<int64_t>>> sd;
  int rc = 0;
  int n_pred = 0;                             // prediction-row column blocks written so far (each its own pseudo tensor)
  void push(Op& op) { e->ops.push_back(op); }
  bool f32() const { return e->dtype == FLOPE_DT_F32; }
  size_t esz() const { return f32() ? 4 : 2; }              // bytes per map element
  static Rng rng(const View& v) { return Rng{v.t, v.off, v.off + v.C}; }

  bool has(const std::string& k) const { return sd.count(k) != 0; }
  const std::vector<int64_t>* shape(const std::string& k) {
    auto it = sd.find(k);
    if (it == sd.end()) { if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "state_dict entry missing: " + k); return nullptr; }
    return &it->second.second;
  }
  const float* data(const std::string& k, const std::vector<int64_t>& want) {
    auto it = sd.find(k);
    if (it == sd.end()) { if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "state_dict entry missing: " + k); return nullptr; }
    if (it->second.second != want) {
      std::string g, w;
      for (auto d : it->second.second) g += std::to_string(d) + ",";
      for (auto d : want) w += std::to_string(d) + ",";
      if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "size mismatch for " + k + ": got [" + g + "] expected [" + w + "]");
      return nullptr;
    }
    size_t n = 1;
    for (auto d : want) n *= (size_t)d;
    for (size_t i = 0; i < n; ++i)
      if (!std::isfinite(it->second.first[i])) { if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "non-finite value in " + k); return nullptr; }
    return it->second.first;
  }
  int cout(const std::string& p) { const auto* s = shape(p + ".conv.weight"); return s && s->size() == 4 ? (int)(*s)[0] : 0; }
  int ksize(const std::string& p) { const auto* s = shape(p + ".conv.weight"); return s && s->size() == 4 ? (int)(*s)[2] : 0; }
  int count(const std::string& p) { int n = 0; while (has(p + "." + std::to_string(n) + ".cv1.conv.weight") || has(p + "." + std::to_string(n) + ".attn.qkv.conv.weight")) ++n; return n; }

  int tensor(int H, int W, int C) {
    Tensor t; t.H = H; t.W = W; t.C = C;
    const size_t bytes = (size_t)H * W * C * esz() + 256;
    if (hipMalloc(&t.ptr, bytes) != hipSuccess || hipMemset(t.ptr, 0, bytes) != hipSuccess) { if (!rc) rc = yfail(e, FLOPE_EHIP, "hipMalloc (activation map) failed"); t.ptr = nullptr; }
    e->tensors.push_back(t);
    return (int)e->tensors.size() - 1;
  }
  View full(int t) { View v; v.t = t; v.off = 0; v.C = e->tensors[t].C; return v; }
  View slice(int t, int off, int C) { View v; v.t = t; v.off = off; v.C = C; return v; }
  void* vptr(const View& v) { return (char*)e->tensors[v.t].ptr + (size_t)v.off * esz(); }
  int vld(const View& v) { return e->tensors[v.t].C; }
  int vH(const View& v) { return e->tensors[v.t].H; }
  int vW(const View& v) { return e->tensors[v.t].W; }

  template <typename V> void* upload(const std::vector<V>& host) {
    void* d = nullptr;
    if (hipMalloc(&d, host.size() * sizeof(V) + 64) != hipSuccess ||
        hipMemcpy(d, host.data(), host.size() * sizeof(V), hipMemcpyHostToDevice) != hipSuccess) {
      if (!rc) rc = yfail(e, FLOPE_EHIP, "weight upload failed");
      return nullptr;
    }
    e->owned.push_back(d);
    return d;
  }
  void tap(const std::string& name, const View& v) {
    Tap t; t.ptr = vptr(v); t.H = vH(v); t.W = vW(v); t.C = v.C; t.ld = vld(v);
    t.is_f32 = f32() ? 1 : 0;
    e->taps[name] = t;
  }

  // folded weights wf[cout][cin][k][k] + bias -> device images of yconv_kernel: rows padded to 16*nt and permuted so that
  // MFMA D rows 4g..4g+3 of channel tile ct are channels g*4nt + 4ct .. +3; k = tap * cin_pad + ci, zero padded to 32;
  // stored in MFMA A-fragment order [channel block][k step][channel tile][lane = kq * 16 + row][8 k]: a wave-load is 1 KiB
  void pack(const std::vector<float>& wf, const std::vector<float>& bf, int cout_, int cin, int cin_pad, int k, int nt,
            const void** w_dev, const float** b_dev, int* ksteps, const void** w32m = nullptr, const float** b32m = nullptr, int* k16steps = nullptr) {
    if (f32()) {            // strict mode: float32 [rows][tap * cin_pad + ci], rows in channel order (yolo_f32.hip)
      const int K = k * k * cin_pad;
      std::vector<float> w((size_t)cout_ * K, 0.f);
      for (int co = 0; co < cout_; ++co)
        for (int tap = 0; tap < k * k; ++tap)
          for (int ci = 0; ci < cin; ++ci) w[(size_t)co * K + tap * cin_pad + ci] = wf[((size_t)co * cin + ci) * k * k + tap];
      *w_dev = upload(w); *b_dev = (const float*)upload(bf); *ksteps = (K + 31) / 32;
      if (w32m) {           // ... and the image of the exact-fp32 MFMA kernel (yolo.h: YConvP::w32m), rows permuted as below
        const int CB = 16 * nt, rows = (cout_ + CB - 1) / CB * CB, K16 = (K + 15) / 16;
        std::vector<float> wm((size_t)rows * K16 * 16, 0.f), bm(rows, 0.f);
        for (int r = 0; r < rows; ++r) {
          const int blk = r / CB, in = r % CB, ct = in / 16, rr = in % 16, g = rr >> 2, q = rr & 3;
          const int co = blk * CB + g * 4 * nt + ct * 4 + q;
          if (co >= cout_) continue;
          bm[r] = bf[co];
          for (int tap = 0; tap < k * k; ++tap)
            for (int ci = 0; ci < cin; ++ci) {
              const int kk = tap * cin_pad + ci, ks = kk / 16, kq = (kk % 16) / 4, el = kk % 4;
              wm[((((size_t)blk * K16 + ks) * nt + ct) * 64 + kq * 16 + rr) * 4 + el] = wf[((size_t)co * cin + ci) * k * k + tap];
            }
        }
        *w32m = upload(wm); *b32m = (const float*)upload(bm); *k16steps = K16;
      }
      return;
    }
    const int CB = 16 * nt, rows = (cout_ + CB - 1) / CB * CB, K = k * k * cin_pad, Kp = (K + 31) / 32 * 32;
    std::vector<uint16_t> w((size_t)rows * Kp, 0);
    std::vector<float> b(rows, 0.f);
    const int ks_n = Kp / 32;
    for (int r = 0; r < rows; ++r) {
      const int blk = r / CB, in = r % CB, ct = in / 16, rr = in % 16, g = rr >> 2, q = rr & 3;
      const int co = blk * CB + g * 4 * nt + ct * 4 + q;
      if (co >= cout_) continue;
      b[r] = bf[co];
      for (int tap = 0; tap < k * k; ++tap)
        for (int ci = 0; ci < cin; ++ci) {
          const int kk = tap * cin_pad + ci, ks = kk / 32, kq = (kk % 32) / 8, kr = kk % 8;
          w[((((size_t)blk * ks_n + ks) * nt + ct) * 64 + kq * 16 + rr) * 8 + kr] = cvt16(wf[((size_t)co * cin + ci) * k * k + tap], e->dtype);
        }
    }
    *w_dev = upload(w); *b_dev = (const float*)upload(b); *ksteps = Kp / 32;
  }

  static int pick_nt(int cout_) { return cout_ <= 16 ? 1 : (cout_ <= 32 ? 2 : 4); }

  // Conv2d(bias=False) + BatchNorm2d (+ SiLU) (+ residual added after the activation)
  void conv(const std::string& p, const View& in, const View& out, int stride, int act, const View* res = nullptr,
            int cin_real = -1) {
    if (rc) return;
    const int co = cout(p), k = ksize(p);
    const int cin = cin_real > 0 ? cin_real : in.C;
    const float* w = data(p + ".conv.weight", {co, cin, k, k});
    const float* g = data(p + ".bn.weight", {co}); const float* bb = data(p + ".bn.bias", {co});
    const float* mu = data(p + ".bn.running_mean", {co}); const float* var = data(p + ".bn.running_var", {co});
    if (rc) return;
    if (co != out.C) { rc = yfail(e, FLOPE_EWEIGHTS, "graph mismatch at " + p + ": output channels"); return; }
    if ((k != 1 && k != 3) || in.C % 8) { rc = yfail(e, FLOPE_EWEIGHTS, "unsupported conv shape at " + p); return; }
    std::vector<float> wf((size_t)co * cin * k * k), bf(co);
    for (int o = 0; o < co; ++o) {
      const double s = (double)g[o] / sqrt((double)var[o] + kYoloBnEps);
      bf[o] = (float)((double)bb[o] - (double)mu[o] * s);
      for (size_t i = 0; i < (size_t)cin * k * k; ++i) wf[(size_t)o * cin * k * k + i] = (float)((double)w[(size_t)o * cin * k * k + i] * s);
    }
    emit_conv(p, wf, bf, co, cin, k, in, out, stride, act, res, 0, nullptr, 0);
  }

  void emit_conv(const std::string& p, const std::vector<float>& wf, const std::vector<float>& bf, int co, int cin, int k,
                 const View& in, const View& out, int stride, int act, const View* res, int out_mode, float* f32_out, int f32_ld) {
    Op op; op.kind = Op::CONV; op.name = p;
    const int rows = out_mode == 2 ? 4 * (co / 4) : co;
    op.nt = pick_nt(out_mode == 2 ? co / 4 : co);
    YConvP& c = op.conv; memset(&c, 0, sizeof c);
    pack(wf, bf, rows, cin, in.C, k, op.nt, &c.w, &c.bias, &c.ksteps, &c.w32m, &c.bias32m, &c.k16steps);
    c.nt32m = op.nt;
    c.in = vptr(in); c.Hi = vH(in); c.Wi = vW(in); c.Cin = in.C; c.ldi = vld(in);
    const int Ho = (c.Hi + 2 * (k / 2) - k) / stride + 1, Wo = (c.Wi + 2 * (k / 2) - k) / stride + 1;
    c.Ho = Ho; c.Wo = Wo; c.M = Ho * Wo; c.Cout = co;
    if (out_mode == 1) { c.out = f32_out; c.ldo = f32_ld; }
    else {
      c.out = vptr(out); c.ldo = vld(out);
      const int eh = out_mode == 2 ? 2 * Ho : Ho, ew = out_mode == 2 ? 2 * Wo : Wo;
      if (vH(out) != eh || vW(out) != ew) { rc = yfail(e, FLOPE_EWEIGHTS, "graph mismatch at " + p + ": output size"); return; }
    }
    if (res) { c.res = vptr(*res); c.ldr = vld(*res); }
    c.zero = e->zero; c.k = k; c.stride = stride; c.act = act; c.cg = in.C / 8;
    fastdiv_magic((unsigned)c.cg, &c.cg_mg, &c.cg_sh);
    fastdiv_magic((unsigned)Wo, &c.wo_mg, &c.wo_sh);
    c.out_mode = out_mode; c.dc = out_mode == 2 ? co / 4 : 0;
    e->flops += 2.0 * c.M * (double)rows * cin * k * k;
    op.reads.push_back(rng(in));
    if (res) op.reads.push_back(rng(*res));
    if (out_mode == 1) op.writes.push_back(Rng{-(++n_pred), 0, 1}); else op.writes.push_back(rng(out));
    push(op);
  }

  // nn.Conv2d(cin, cout, 1) with bias -> float32 columns [col0, col0 + cout) of the prediction rows [a0, a0 + H*W)
  void plain(const std::string& p, const View& in, int a0, int col0) {
    if (rc) return;
    const auto* s = shape(p + ".weight"); if (!s) return;
    const int co = (int)(*s)[0];
    const float* w = data(p + ".weight", {co, in.C, 1, 1}); const float* b = data(p + ".bias", {co});
    if (rc) return;
    std::vector<float> wf(w, w + (size_t)co * in.C), bf(b, b + co);
    View none;
    emit_conv(p, wf, bf, co, in.C, 1, in, none, 1, 0, nullptr, 1, e->pred + (size_t)a0 * e->no + col0, e->no);
  }

  // nn.ConvTranspose2d(c, c, 2, 2, 0, bias=True): weight [cin][cout][2][2]
  void deconv(const std::string& p, const View& in, const View& out) {
    if (rc) return;
    const int c = in.C;
    const float* w = data(p + ".weight", {c, out.C, 2, 2}); const float* b = data(p + ".bias", {out.C});
    if (rc) return;
    const int dc = out.C;
    if (dc % 16) { rc = yfail(e, FLOPE_EWEIGHTS, "unsupported transposed-conv width at " + p); return; }
    std::vector<float> wf((size_t)4 * dc * c), bf(4 * dc);
    for (int q = 0; q < 4; ++q)
      for (int o = 0; o < dc; ++o) {
        bf[q * dc + o] = b[o];
        for (int ci = 0; ci < c; ++ci) wf[((size_t)q * dc + o) * c + ci] = w[(((size_t)ci * dc + o) * 2 + (q >> 1)) * 2 + (q & 1)];
      }
    emit_conv(p, wf, bf, 4 * dc, c, 1, in, out, 1, 0, nullptr, 2, nullptr, 0);
  }

  // depthwise Conv2d(c, c, 3, 1, 1, groups=c, bias=False) + BatchNorm (+ SiLU) (+ add); blk != 0: input channel of output
  // channel ch is (ch / blk) * blk_stride + blk_off + ch % blk (the v rows of a qkv map)
  void dw(const std::string& p, const View& in, const View& out, int act, const View* add = nullptr, int blk = 0,
          int blk_stride = 0, int blk_off = 0) {
    if (rc) return;
    const int c = out.C;
    const float* w = data(p + ".conv.weight", {c, 1, 3, 3});
    const float* g = data(p + ".bn.weight", {c}); const float* bb = data(p + ".bn.bias", {c});
    const float* mu = data(p + ".bn.running_mean", {c}); const float* var = data(p + ".bn.running_var", {c});
    if (rc) return;
    if (c % 8 || (blk && blk % 8)) { rc = yfail(e, FLOPE_EWEIGHTS, "unsupported depthwise width at " + p); return; }
    std::vector<float> wf((size_t)9 * c), bf(c);
    for (int o = 0; o < c; ++o) {
      const double s = (double)g[o] / sqrt((double)var[o] + kYoloBnEps);
      bf[o] = (float)((double)bb[o] - (double)mu[o] * s);
      for (int t = 0; t < 9; ++t) wf[(size_t)t * c + o] = (float)((double)w[(size_t)o * 9 + t] * s);
    }
    Op op; op.kind = Op::DW; op.name = p;
    YDwP& d = op.dw; memset(&d, 0, sizeof d);
    d.in = vptr(in); d.H = vH(in); d.W = vW(in); d.C = c; d.ldi = vld(in);
    d.out = vptr(out); d.ldo = vld(out);
    if (add) { d.add = vptr(*add); d.lda = vld(*add); }
    d.w = (const float*)upload(wf); d.bias = (const float*)upload(bf); d.act = act;
    d.blk = blk; d.blk_stride = blk_stride; d.blk_off = blk_off;
    e->flops += 2.0 * d.H * d.W * (double)c * 9;
    op.reads.push_back(rng(in));
    if (add) op.reads.push_back(rng(*add));
    op.writes.push_back(rng(out));
    push(op);
  }

  // n cascaded MaxPool2d(5, 1, 2) of `in`; result i lands in channels [i C, (i + 1) C) of `out` (out.C = n * in.C)
  void pool(const View& in, const View& out, int n) {
    Op op; op.kind = Op::POOL; op.name = "maxpool5 x" + std::to_string(n);
    op.pool.in = vptr(in); op.pool.H = vH(in); op.pool.W = vW(in); op.pool.C = in.C; op.pool.ldi = vld(in);
    op.pool.out = vptr(out); op.pool.ldo = vld(out); op.pool.n = n;
    op.reads.push_back(rng(in)); op.writes.push_back(rng(out));
    push(op);
  }
  void upsample(const View& in, const View& out) {
    Op op; op.kind = Op::UP; op.name = "upsample2x";
    op.up.in = vptr(in); op.up.H = vH(in); op.up.W = vW(in); op.up.C = in.C; op.up.ldi = vld(in);
    op.up.out = vptr(out); op.up.ldo = vld(out);
    op.reads.push_back(rng(in)); op.writes.push_back(rng(out));
    push(op);
  }

  // ---- modules (ultralytics nn/modules/block.py) -------------------------------------------------------------------
  void bottleneck(const std::string& p, const View& in, const View& out) {          // x + cv2(cv1(x))
    const int t = tensor(vH(in), vW(in), cout(p + ".cv1"));
    conv(p + ".cv1", in, full(t), 1, 1);
    conv(p + ".cv2", full(t), out, 1, 1, &in);
    if (rc || e->ops.size() < 2) return;
    const Op& o2 = e->ops.back();
    const Op& o1 = e->ops[e->ops.size() - 2];
    if (f32() || o1.kind != Op::CONV || o2.kind != Op::CONV || !flope_ybneck_fusable(&o1.conv, o1.nt, &o2.conv, o2.nt)) return;
    Op f; f.kind = Op::BNECK; f.name = p; f.conv = o1.conv; f.conv2 = o2.conv; f.nt = o1.nt; f.nt2 = o2.nt;
    f.reads = o1.reads;                                   // the intermediate map never leaves the workgroup
    for (const Rng& r : o2.reads)
      if (r.t != t) f.reads.push_back(r);
    f.writes = o2.writes;
    e->ops.pop_back(); e->ops.pop_back();
    e->ops.push_back(f);
  }
  void c3k(const std::string& p, const View& in, const View& out) {                 // cv3(cat(m(cv1(x)), cv2(x)))
    const int c_ = cout(p + ".cv1"), H = vH(in), W = vW(in);
    const int z = tensor(H, W, 2 * c_), a = tensor(H, W, c_);
    conv(p + ".cv1", in, full(a), 1, 1);
    const int n = count(p + ".m");
    View cur = full(a);
    for (int j = 0; j < n; ++j) {
      const View dst = j == n - 1 ? slice(z, 0, c_) : full(tensor(H, W, c_));
      bottleneck(p + ".m." + std::to_string(j), cur, dst);
      cur = dst;
    }
    conv(p + ".cv2", in, slice(z, c_, c_), 1, 1);
    conv(p + ".cv3", full(z), out, 1, 1);
  }
  void c3k2(const std::string& p, const View& in, const View& out) {                // C2f with Bottleneck / C3k inner modules
    const int c = cout(p + ".cv1") / 2, n = count(p + ".m"), H = vH(in), W = vW(in);
    if (c < 8 || n < 1) { if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "unsupported C3k2 at " + p); return; }
    const int y = tensor(H, W, (2 + n) * c);
    conv(p + ".cv1", in, slice(y, 0, 2 * c), 1, 1);
    View prev = slice(y, c, c);
    for (int j = 0; j < n; ++j) {
      const View dst = slice(y, (2 + j) * c, c);
      const std::string q = p + ".m." + std::to_string(j);
      if (has(q + ".cv3.conv.weight")) c3k(q, prev, dst); else bottleneck(q, prev, dst);
      prev = dst;
    }
    conv(p + ".cv2", full(y), out, 1, 1);
  }
  void sppf(const std::string& p, const View& in, const View& out) {
    const int c_ = cout(p + ".cv1"), y = tensor(vH(in), vW(in), 4 * c_);
    conv(p + ".cv1", in, slice(y, 0, c_), 1, 1);
    pool(slice(y, 0, c_), slice(y, c_, 3 * c_), 3);          // y1 = m(x), y2 = m(y1), y3 = m(y2) in one sweep
    conv(p + ".cv2", full(y), out, 1, 1);
  }
  void c2psa(const std::string& p, const View& in, const View& out) {
    const int c = cout(p + ".cv1") / 2, H = vH(in), W = vW(in), n = count(p + ".m");
    if (c % 64) { if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "unsupported C2PSA width at " + p); return; }
    const int heads = c / 64;
    const int ab = tensor(H, W, 2 * c);
    conv(p + ".cv1", in, full(ab), 1, 1);
    const View b = slice(ab, c, c);
    for (int j = 0; j < n; ++j) {
      const std::string q = p + ".m." + std::to_string(j);
      const int qkv = tensor(H, W, cout(q + ".attn.qkv")), att = tensor(H, W, c), att2 = tensor(H, W, c), f = tensor(H, W, cout(q + ".ffn.0"));
      if ((size_t)16 * H * W * sizeof(float) > 160 * 1024) {   // the generic attention kernel keeps 16 score rows in LDS
        if (!rc) rc = yfail(e, FLOPE_EINVAL, "C2PSA attention: " + std::to_string(H * W) + " tokens at this frame size / imgsz exceed the supported 2560 (imgsz up to ~1600 for 16:9 frames)");
        return;
      }
      if (cout(q + ".attn.qkv") != heads * 128) { if (!rc) rc = yfail(e, FLOPE_EWEIGHTS, "unsupported attention layout at " + q); return; }
      conv(q + ".attn.qkv", b, full(qkv), 1, 0);
      Op op; op.kind = Op::ATTN; op.name = q + ".attn";
      op.attn.qkv = vptr(full(qkv)); op.attn.N = H * W; op.attn.heads = heads; op.attn.ld = heads * 128;
      op.attn.out = vptr(full(att)); op.attn.ldo = c; op.attn.scale = 1.0f / sqrtf(32.f);
      e->flops += 2.0 * heads * (double)H * W * H * W * (32 + 64);
      op.reads.push_back(rng(full(qkv))); op.writes.push_back(rng(full(att)));
      push(op);
      const View va = full(att);
      dw(q + ".attn.pe", full(qkv), full(att2), 0, &va, 64, 128, 64);            // (v @ attn^T) + pe(v)
      conv(q + ".attn.proj", full(att2), b, 1, 0, &b);                            // b = b + proj(...)
      conv(q + ".ffn.0", b, full(f), 1, 1);
      conv(q + ".ffn.1", full(f), b, 1, 0, &b);                                   // b = b + ffn(b)
    }
    conv(p + ".cv2", full(ab), out, 1, 1);
  }
};

int launch_op(flope_yolo* e, const Op& op, hipStream_t st) {
  if (e->dtype == FLOPE_DT_F32) {
    switch (op.kind) {
      case Op::CONV: return (e->opt_f32mfma && flope_y32m_conv_ok(&op.conv)) ? flope_y32m_conv_launch(&op.conv, st) : flope_y32_conv_launch(&op.conv, st);   // (a shape the MFMA kernel does not take: the plain kernel)
      case Op::DW: return flope_y32_dw_launch(&op.dw, st);
      case Op::POOL: return flope_y32_pool_launch(&op.pool, st);
      case Op::UP: return flope_y32_up_launch(&op.up, st);
      case Op::ATTN: return e->opt_f32mfma ? flope_y32m_attn_launch(&op.attn, st) : flope_y32_attn_launch(&op.attn, st);
      case Op::BNECK: return (int)hipErrorInvalidValue;      // never built in this mode
    }
  }
  switch (op.kind) {
    case Op::CONV: return flope_yconv_launch(&op.conv, e->dtype, op.nt, st);
    case Op::DW: return flope_ydw_launch(&op.dw, e->dtype, st);
    case Op::POOL: return flope_ypool_launch(&op.pool, e->dtype, st);
    case Op::UP: return flope_yup_launch(&op.up, st);
    case Op::ATTN: return flope_yattn_launch(&op.attn, e->dtype, e->opt_generic_attn, st);
    case Op::BNECK: {
      if (e->opt_bneck) return flope_ybneck_launch(&op.conv, op.nt, &op.conv2, op.nt2, e->dtype, st);
      const int s = flope_yconv_launch(&op.conv, e->dtype, op.nt, st);
      return s ? s : flope_yconv_launch(&op.conv2, e->dtype, op.nt2, st);
    }
  }
  return (int)hipErrorInvalidValue;
}

// float32 mode: the level-batched schedule belongs to the MFMA kernels; their checker (f32mfma = 0) runs in program order, one
// launch per op of the ultralytics yaml
inline int sched_index(const flope_yolo* e) { return (e->opt_batch && (e->dtype != FLOPE_DT_F32 || e->opt_f32mfma)) ? 1 : 0; }

int launch_one(flope_yolo* e, const Launch& L, hipStream_t st) {
  if (L.chain_dev) return e->dtype == FLOPE_DT_F32 ? flope_y32m_chain_launch(&L.chain, L.chain_dev, st) : flope_ychain_launch(&L.chain, L.chain_dev, e->dtype, st);
  if (L.op >= 0) return launch_op(e, e->ops[L.op], st);
  return e->dtype == FLOPE_DT_F32 ? flope_y32m_multi_launch(&L.multi, L.multi_dev, st) : flope_ymulti_launch(&L.multi, L.multi_dev, e->dtype, st);
}

std::string launch_name(const flope_yolo* e, const Launch& L) {
  if (L.op >= 0) return e->ops[L.op].name;
  std::string n;
  for (int i : L.members) n += (n.empty() ? "" : (L.chain_dev ? " -> " : " | ")) + e->ops[i].name;
  return n;
}

int run_ops(flope_yolo* e, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  for (const Launch& L : e->sched[sched_index(e)]) {
    const int s = launch_one(e, L, st);
    if (s != 0) return yfail(e, FLOPE_EHIP, launch_name(e, L) + ": " + hipGetErrorString((hipError_t)s));
  }
  return FLOPE_OK;
}

// Dependency levels from the read / write sets (RAW, WAW, WAR on overlapping channel ranges of one tensor), then the two
// schedules: program order, and level order with the conv / depthwise ops of a level sharing grids of <= kYMultiMax ops.
int build_schedules(flope_yolo* e) {
  auto hit = [](const std::vector<Rng>& a, const std::vector<Rng>& b) {
    for (const Rng& x : a)
      for (const Rng& y : b)
        if (x.t == y.t && x.c0 < y.c1 && y.c0 < x.c1) return true;
    return false;
  };
  const int n = (int)e->ops.size();
  int depth = 0;
  for (int i = 0; i < n; ++i) {
    Op& oi = e->ops[i];
    oi.level = 0;
    for (int j = 0; j < i; ++j) {
      const Op& oj = e->ops[j];
      if (hit(oj.writes, oi.reads) || hit(oj.writes, oi.writes) || hit(oj.reads, oi.writes)) oi.level = std::max(oi.level, oj.level + 1);
    }
    depth = std::max(depth, oi.level + 1);
  }
  e->sched[0].clear(); e->sched[1].clear();
  for (int i = 0; i < n; ++i) { Launch L; L.op = i; e->sched[0].push_back(L); }
  const bool f32 = e->dtype == FLOPE_DT_F32;
  for (int lv = 0; lv < depth; ++lv) {
    // one grid's dynamic LDS is that of its hungriest op, so a fused Bottleneck (up to 134 KB) leaves the plain convs it shares a
    // grid with one workgroup per CU -- measured, sharing still wins (0.741 vs 0.757 ms per frame: a launch less per level);
    // bneck = 2 keeps them in grids of their own
    std::vector<int> classes[2], single;
    for (int i = 0; i < n; ++i)
      if (e->ops[i].level == lv) {
        const Op& op = e->ops[i];
        if (f32 && op.kind == Op::CONV && !flope_y32m_conv_ok(&op.conv)) single.push_back(i);   // launch_op falls back to the plain kernel
        else if (op.kind == Op::CONV || op.kind == Op::DW) classes[0].push_back(i);
        else if (op.kind == Op::BNECK && e->opt_bneck) classes[e->opt_bneck == 2 ? 1 : 0].push_back(i);
        else single.push_back(i);
      }
    for (int i : single) { Launch L; L.op = i; e->sched[1].push_back(L); }
    for (const std::vector<int>& batchable : classes)
      for (size_t at = 0; at < batchable.size(); at += kYMultiMax) {
        const size_t m = std::min(batchable.size() - at, (size_t)kYMultiMax);
        Launch L;
        if (m == 1) { L.op = batchable[at]; e->sched[1].push_back(L); continue; }
        memset(&L.multi, 0, sizeof L.multi);
        for (size_t k = 0; k < m; ++k) {
          const Op& op = e->ops[batchable[at + k]];
          const int s = f32 ? (op.kind == Op::CONV ? flope_y32m_multi_add_conv(&L.multi, &op.conv) : flope_y32m_multi_add_dw(&L.multi, &op.dw))
                      : op.kind == Op::CONV ? flope_ymulti_add_conv(&L.multi, &op.conv, op.nt)
                      : op.kind == Op::DW ? flope_ymulti_add_dw(&L.multi, &op.dw)
                                          : flope_ymulti_add_bneck(&L.multi, &op.conv, op.nt, &op.conv2, op.nt2);
          if (s) return yfail(e, FLOPE_EINVAL, "schedule: cannot batch " + op.name);
          L.members.push_back(batchable[at + k]);
        }
        if (hipMalloc((void**)&L.multi_dev, sizeof(YMultiP)) != hipSuccess ||
            hipMemcpy(L.multi_dev, &L.multi, sizeof(YMultiP), hipMemcpyHostToDevice) != hipSuccess)
          return yfail(e, FLOPE_EHIP, "schedule: parameter table upload failed");
        e->owned.push_back(L.multi_dev);
        e->sched[1].push_back(L);
      }
  }
  // r05: chains.  Consecutive single launches of the level schedule that are 1x1 stride-1 convs on the same small map become ONE launch
  // (the 23 x 40 map's conv -> conv -> conv runs cost a dependent launch each, ~8 us for ~3 us of kernel): a 1x1 conv reads nothing but
  // its own pixel, so a workgroup can push its pixel tile through the whole run; everything else the run reads was complete before it.
  if (e->opt_chain) {
    std::vector<Launch> out;
    std::vector<int> run;
    auto chainable = [&](const Launch& L) {
      if (L.op < 0 || L.chain_dev) return false;
      const Op& op = e->ops[L.op];
      if (op.kind != Op::CONV) return false;
      return f32 ? (e->opt_f32mfma && flope_y32m_chain_ok(&op.conv) != 0) : flope_ychain_ok(&op.conv, op.nt) != 0;
    };
    auto flush = [&]() -> int {
      for (size_t at = 0; at < run.size();) {
        const size_t m = std::min(run.size() - at, (size_t)kYChainMax);
        if (m < 2) { Launch L; L.op = run[at]; out.push_back(L); at += m; continue; }
        Launch L;
        memset(&L.chain, 0, sizeof L.chain);
        L.chain.n = (int)m;
        const int M = e->ops[run[at]].conv.M;
        L.chain.tiles = f32 ? (M + 15) / 16 : (M + 31) / 32;
        for (size_t k = 0; k < m; ++k) { const Op& op = e->ops[run[at + k]]; L.chain.op[k] = op.conv; L.chain.nt[k] = op.nt; L.members.push_back(run[at + k]); }
        if (hipMalloc((void**)&L.chain_dev, sizeof(YChainP)) != hipSuccess || hipMemcpy(L.chain_dev, &L.chain, sizeof(YChainP), hipMemcpyHostToDevice) != hipSuccess)
          return yfail(e, FLOPE_EHIP, "schedule: chain table upload failed");
        e->owned.push_back(L.chain_dev);
        out.push_back(L);
        at += m;
      }
      run.clear();
      return FLOPE_OK;
    };
    for (const Launch& L : e->sched[1]) {
      if (chainable(L) && (run.empty() || e->ops[run.back()].conv.M == e->ops[L.op].conv.M)) { run.push_back(L.op); continue; }
      if (int rc = flush()) return rc;
      if (chainable(L)) run.push_back(L.op); else out.push_back(L);
    }
    if (int rc = flush()) return rc;
    e->sched[1].swap(out);
  }
  return FLOPE_OK;
}

// a captured sequence may still be replaying on some stream: wait for the event recorded behind its last launch
void drop_graph(flope_yolo::Captured& g) {
  if (g.done) { hipEventSynchronize(g.done); hipEventDestroy(g.done); }
  hipGraphExecDestroy(g.exec);
}
void drop_graphs(flope_yolo* e) {
  for (auto& g : e->graphs) drop_graph(g);
  e->graphs.clear();
}

}  // namespace

// ====================================================================================================================
extern "C" const char* flope_yolo_last_error(flope_yolo_handle h) { return h ? h->err.c_str() : g_yolo_error.c_str(); }

extern "C" int flope_yolo_create(int device_id, int frame_h, int frame_w, int imgsz, int dtype, flope_yolo_handle* out) {
  if (!out) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_create: out is NULL");
  *out = nullptr;
  if (frame_h < 32 || frame_w < 32 || frame_h > 8192 || frame_w > 8192) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_create: frame size must be within 32..8192");
  if (imgsz < 32 || imgsz > 2560 || imgsz % 32) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_create: imgsz must be a multiple of 32 within 32..2560");
  if (dtype != FLOPE_DT_BF16 && dtype != FLOPE_DT_F16 && dtype != FLOPE_DT_F32) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_create: dtype must be FLOPE_DT_F16, FLOPE_DT_BF16 or FLOPE_DT_F32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return yfail(nullptr, FLOPE_EHIP, "flope_yolo_create: no HIP device visible (the product path has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_create: bad device id");
  flope_yolo* e = new flope_yolo();
  e->device = device_id; e->H = frame_h; e->W = frame_w; e->imgsz = imgsz; e->dtype = dtype;
  // LetterBox(new_shape=(imgsz, imgsz), auto=True, scaleup=True, stride=32)
  const double r = std::min((double)imgsz / frame_h, (double)imgsz / frame_w);
  e->nw = (int)nearbyint(frame_w * r); e->nh = (int)nearbyint(frame_h * r);
  const double dw = ((imgsz - e->nw) % 32) / 2.0, dh = ((imgsz - e->nh) % 32) / 2.0;
  e->top = (int)nearbyint(dh - 0.1); e->left = (int)nearbyint(dw - 0.1);
  e->h = e->nh + e->top + (int)nearbyint(dh + 0.1); e->w = e->nw + e->left + (int)nearbyint(dw + 0.1);
  if (e->h % 32 || e->w % 32) { delete e; return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_create: letterboxed size is not a multiple of 32"); }
  if (hipSetDevice(device_id) != hipSuccess || flope_yattn_init() != 0 || flope_y32_attn_init() != 0 || flope_y32_pool_init() != 0) { delete e; return yfail(nullptr, FLOPE_EHIP, "flope_yolo_create: device setup failed"); }
  *out = e;
  return FLOPE_OK;
}

extern "C" int flope_yolo_destroy(flope_yolo_handle e) {
  if (!e) return FLOPE_OK;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  drop_graphs(e);
  for (Tensor& t : e->tensors) if (t.ptr) hipFree(t.ptr);
  for (void* p : e->owned) if (p) hipFree(p);
  delete e;
  return FLOPE_OK;
}

extern "C" int flope_yolo_input_size(flope_yolo_handle e, int* h, int* w) {
  if (!e || !h || !w) return yfail(e, FLOPE_EINVAL, "flope_yolo_input_size: NULL argument");
  *h = e->h; *w = e->w;
  return FLOPE_OK;
}

extern "C" int flope_yolo_load_weights(flope_yolo_handle e, int n, const char* const* names, const float* const* host_ptrs,
                                       const int* ndims, const int64_t* const* shapes) {
  if (!e) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_load_weights: NULL handle");
  if (e->loaded) return yfail(e, FLOPE_ESTATE, "flope_yolo_load_weights: weights already loaded (create a new handle)");
  if (n <= 0 || !names || !host_ptrs || !ndims || !shapes) return yfail(e, FLOPE_EINVAL, "flope_yolo_load_weights: NULL argument");
  Y_TRY(e, hipSetDevice(e->device));
  Builder b; b.e = e;
  for (int i = 0; i < n; ++i) {
    if (!names[i] || !host_ptrs[i] || ndims[i] < 0 || ndims[i] > 8 || (ndims[i] > 0 && !shapes[i]))
      return yfail(e, FLOPE_EINVAL, "flope_yolo_load_weights: malformed entry " + std::to_string(i));
    b.sd[names[i]] = std::make_pair(host_ptrs[i], std::vector<int64_t>(shapes[i], shapes[i] + ndims[i]));
  }
  const std::string hd = "model.23";
  for (const char* k : {"model.0.conv.weight", "model.22.cv2.conv.weight", "model.23.cv3.0.2.weight", "model.23.proto.cv3.conv.weight"})
    if (!b.has(k)) return yfail(e, FLOPE_EWEIGHTS, std::string("state_dict entry missing: ") + k + " (expected an ultralytics yolo11-seg state_dict)");
  e->nc = (int)(*b.shape(hd + ".cv3.0.2.weight"))[0];
  if ((int)(*b.shape(hd + ".cv2.0.2.weight"))[0] != 4 * kRegMax || (int)(*b.shape(hd + ".cv4.0.2.weight"))[0] != kNm)
    return yfail(e, FLOPE_EWEIGHTS, "Segment head: expected 64 DFL and 32 mask-coefficient outputs");
  e->no = 4 * kRegMax + e->nc + kNm;
  const int h = e->h, w = e->w;
  const int H8 = h / 8, W8 = w / 8, H16 = h / 16, W16 = w / 16, H32 = h / 32, W32 = w / 32;
  e->A = H8 * W8 + H16 * W16 + H32 * W32;
  {
    std::vector<uint16_t> z(64, 0);
    e->zero = b.upload(z);
    float* p = nullptr;
    if (hipMalloc((void**)&p, (size_t)e->A * e->no * sizeof(float)) != hipSuccess) return yfail(e, FLOPE_EHIP, "hipMalloc (prediction rows) failed");
    e->owned.push_back(p); e->pred = p;
  }
  // ---- graph (ultralytics cfg/models/11/yolo11-seg.yaml) ------------------------------------------------------------
  const int x = b.tensor(h, w, 8);
  const int c0 = b.cout("model.0"), c1 = b.cout("model.1"), c2 = b.cout("model.2.cv2"), c3 = b.cout("model.3"), c4 = b.cout("model.4.cv2"),
            c5 = b.cout("model.5"), c6 = b.cout("model.6.cv2"), c7 = b.cout("model.7"), c8 = b.cout("model.8.cv2"), c9 = b.cout("model.9.cv2"),
            c10 = b.cout("model.10.cv2"), c13 = b.cout("model.13.cv2"), c16 = b.cout("model.16.cv2"), c17 = b.cout("model.17"),
            c19 = b.cout("model.19.cv2"), c20 = b.cout("model.20"), c22 = b.cout("model.22.cv2");
  if (b.rc) return b.rc;
  for (int c : {c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c13, c16, c17, c19, c20, c22})
    if (c % 8) return yfail(e, FLOPE_EWEIGHTS, "channel widths must be multiples of 8");
  const int t0 = b.tensor(h / 2, w / 2, c0), t1 = b.tensor(h / 4, w / 4, c1), t2 = b.tensor(h / 4, w / 4, c2), t3 = b.tensor(H8, W8, c3);
  const int cat15 = b.tensor(H8, W8, c13 + c4), cat12 = b.tensor(H16, W16, c10 + c6), cat18 = b.tensor(H16, W16, c17 + c13),
            cat21 = b.tensor(H32, W32, c20 + c10);
  const int t5 = b.tensor(H16, W16, c5), t7 = b.tensor(H32, W32, c7), t8 = b.tensor(H32, W32, c8), t9 = b.tensor(H32, W32, c9);
  const int t16 = b.tensor(H8, W8, c16), t19 = b.tensor(H16, W16, c19), t22 = b.tensor(H32, W32, c22);
  const View o4 = b.slice(cat15, c13, c4), o6 = b.slice(cat12, c10, c6), o10 = b.slice(cat21, c20, c10), o13 = b.slice(cat18, c17, c13);
  b.conv("model.0", b.full(x), b.full(t0), 2, 1, nullptr, 3);
  b.conv("model.1", b.full(t0), b.full(t1), 2, 1);
  b.c3k2("model.2", b.full(t1), b.full(t2));
  b.conv("model.3", b.full(t2), b.full(t3), 2, 1);
  b.c3k2("model.4", b.full(t3), o4);
  b.conv("model.5", o4, b.full(t5), 2, 1);
  b.c3k2("model.6", b.full(t5), o6);
  b.conv("model.7", o6, b.full(t7), 2, 1);
  b.c3k2("model.8", b.full(t7), b.full(t8));
  b.sppf("model.9", b.full(t8), b.full(t9));
  b.c2psa("model.10", b.full(t9), o10);
  b.upsample(o10, b.slice(cat12, 0, c10));
  b.c3k2("model.13", b.full(cat12), o13);
  b.upsample(o13, b.slice(cat15, 0, c13));
  // The Segment head's branches (box, class, coefficients per level, Proto) only depend on their level's feature map:
  // build_schedules() finds that from the read / write sets and lets them share grids with each other and with the neck.
  const char* nm3[3] = {"box", "cls", "coef"};
  int a0 = 0;
  auto head_level = [&](int i, const View f) {
    const int H = b.vH(f), W = b.vW(f);
    const std::string si = std::to_string(i);
    const int cb = b.cout(hd + ".cv2." + si + ".0"), cc = b.cout(hd + ".cv3." + si + ".0.1"), cm = b.cout(hd + ".cv4." + si + ".0");
    if (b.rc) return;
    const int b1 = b.tensor(H, W, cb), b2 = b.tensor(H, W, cb);
    b.conv(hd + ".cv2." + si + ".0", f, b.full(b1), 1, 1);
    b.conv(hd + ".cv2." + si + ".1", b.full(b1), b.full(b2), 1, 1);
    b.plain(hd + ".cv2." + si + ".2", b.full(b2), a0, 0);
    const int d1 = b.tensor(H, W, f.C), e1 = b.tensor(H, W, cc), d2 = b.tensor(H, W, cc), e2 = b.tensor(H, W, cc);
    b.dw(hd + ".cv3." + si + ".0.0", f, b.full(d1), 1);
    b.conv(hd + ".cv3." + si + ".0.1", b.full(d1), b.full(e1), 1, 1);
    b.dw(hd + ".cv3." + si + ".1.0", b.full(e1), b.full(d2), 1);
    b.conv(hd + ".cv3." + si + ".1.1", b.full(d2), b.full(e2), 1, 1);
    b.plain(hd + ".cv3." + si + ".2", b.full(e2), a0, 4 * kRegMax);
    const int m1 = b.tensor(H, W, cm), m2 = b.tensor(H, W, cm);
    b.conv(hd + ".cv4." + si + ".0", f, b.full(m1), 1, 1);
    b.conv(hd + ".cv4." + si + ".1", b.full(m1), b.full(m2), 1, 1);
    b.plain(hd + ".cv4." + si + ".2", b.full(m2), a0, 4 * kRegMax + e->nc);
    const int col[3] = {0, 4 * kRegMax, 4 * kRegMax + e->nc}, cw[3] = {4 * kRegMax, e->nc, kNm};
    for (int k = 0; k < 3; ++k) {
      Tap t; t.is_f32 = 1; t.ptr = e->pred + (size_t)a0 * e->no + col[k]; t.H = H; t.W = W; t.C = cw[k]; t.ld = e->no;
      e->taps[std::string(nm3[k]) + si] = t;
    }
    e->dec.lvl_a0[i] = a0; e->dec.lvl_w[i] = W; e->dec.lvl_stride[i] = 8 << i;
    a0 += H * W;
  };
  b.c3k2("model.16", b.full(cat15), b.full(t16));
  const int npr = b.cout(hd + ".proto.cv1");
  if (b.rc) return b.rc;
  const int p1 = b.tensor(H8, W8, npr), pu = b.tensor(2 * H8, 2 * W8, npr), p2 = b.tensor(2 * H8, 2 * W8, npr), pr = b.tensor(2 * H8, 2 * W8, kNm);
  b.conv(hd + ".proto.cv1", b.full(t16), b.full(p1), 1, 1);
  b.deconv(hd + ".proto.upsample", b.full(p1), b.full(pu));
  b.conv(hd + ".proto.cv2", b.full(pu), b.full(p2), 1, 1);
  b.conv(hd + ".proto.cv3", b.full(p2), b.full(pr), 1, 1);
  head_level(0, b.full(t16));
  b.conv("model.17", b.full(t16), b.slice(cat18, 0, c17), 2, 1);
  b.c3k2("model.19", b.full(cat18), b.full(t19));
  head_level(1, b.full(t19));
  b.conv("model.20", b.full(t19), b.slice(cat21, 0, c20), 2, 1);
  b.c3k2("model.22", b.full(cat21), b.full(t22));
  head_level(2, b.full(t22));
  e->dec.lvl_a0[3] = a0;
  if (b.rc) return b.rc;
  b.tap("input", b.full(x));
  const std::pair<const char*, View> named[] = {{"0", b.full(t0)}, {"1", b.full(t1)}, {"2", b.full(t2)}, {"3", b.full(t3)}, {"4", o4},
      {"5", b.full(t5)}, {"6", o6}, {"7", b.full(t7)}, {"8", b.full(t8)}, {"9", b.full(t9)}, {"10", o10}, {"13", o13},
      {"16", b.full(t16)}, {"17", b.slice(cat18, 0, c17)}, {"19", b.full(t19)}, {"20", b.slice(cat21, 0, c20)}, {"22", b.full(t22)}};
  for (const auto& kv : named) b.tap(kv.first, kv.second);
  b.tap("proto_up", b.full(pu));
  b.tap("proto", b.full(pr));
  // ---- post-processing buffers ---------------------------------------------------------------------------------------
  auto dalloc = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes + 64) != hipSuccess || hipMemset(p, 0, bytes + 64) != hipSuccess) { if (!b.rc) b.rc = yfail(e, FLOPE_EHIP, "hipMalloc (post-processing) failed"); return nullptr; }
    e->owned.push_back(p);
    return p;
  };
  YDecodeP& d = e->dec;
  d.pred = e->pred; d.A = e->A; d.no = e->no; d.nc = e->nc; d.conf = 0.25f;
  d.cand_box = (float*)dalloc((size_t)e->A * 16); d.cand_conf = (float*)dalloc((size_t)e->A * 4); d.cand_cls = (int*)dalloc((size_t)e->A * 4);
  YNmsP& q = e->nms; memset(&q, 0, sizeof q);
  q.cand_box = d.cand_box; q.cand_conf = d.cand_conf; q.cand_cls = d.cand_cls; q.A = e->A;
  q.max_wh = 7680.f;
  // ops.scale_boxes(img1_shape = letterboxed, boxes, img0_shape = frame): gain, pad with python's round()
  {
    const double gain = std::min((double)h / e->H, (double)w / e->W);
    q.gain = (float)gain;
    q.pad_x = (float)nearbyint((w - e->W * gain) / 2 - 0.1); q.pad_y = (float)nearbyint((h - e->H * gain) / 2 - 0.1);
    q.frame_w = e->W; q.frame_h = e->H;
  }
  q.det_lb = (float*)dalloc((size_t)kMaxDet * 16); q.det_anchor = (int*)dalloc((size_t)kMaxDet * 4);
  YMaskP& m = e->mask; memset(&m, 0, sizeof m);
  m.proto = e->tensors[pr].ptr; m.mh = 2 * H8; m.mw = 2 * W8; m.pred = e->pred; m.no = e->no; m.nc = e->nc;
  m.det_lb = q.det_lb; m.det_anchor = q.det_anchor; m.ih = h; m.iw = w;
  m.low = (float*)dalloc((size_t)kMaxDet * m.mh * m.mw * 4);
  e->merged = (uint8_t*)dalloc((size_t)h * w); m.merged = e->merged;
  {
    Tap t; t.is_f32 = 2; t.ptr = e->merged; t.H = h; t.W = w; t.C = 1; t.ld = 1;
    e->taps["mask_lb"] = t;
    Tap c; c.is_f32 = 1; c.ptr = d.cand_box; c.H = 1; c.W = e->A; c.C = 4; c.ld = 4;
    e->taps["cand_box"] = c;                        // decoded xyxy of every anchor (letterbox pixels), after flope_yolo_detect
    c.ptr = d.cand_conf; c.C = 1; c.ld = 1;
    e->taps["cand_conf"] = c;                       // best class confidence of every anchor
    c.is_f32 = 3; c.ptr = d.cand_cls;
    e->taps["cand_cls"] = c;                        // its class index
  }
  YLetterP& L = e->letter; memset(&L, 0, sizeof L);
  L.H = e->H; L.W = e->W; L.out = e->tensors[x].ptr; L.h = h; L.w = w; L.nh = e->nh; L.nw = e->nw; L.top = e->top; L.left = e->left;
  L.sx = 1.0 / ((double)e->nw / e->W); L.sy = 1.0 / ((double)e->nh / e->H);
  if (b.rc) return b.rc;
  if (int rc = build_schedules(e)) return rc;
  Y_TRY(e, hipDeviceSynchronize());
  e->loaded = true;
  return FLOPE_OK;
}

extern "C" int flope_yolo_forward(flope_yolo_handle e, const uint8_t* frame_dev, void* stream) {
  if (!e) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_forward: NULL handle");
  if (!e->loaded) return yfail(e, FLOPE_ESTATE, "detect before flope_yolo_load_weights");
  if (!frame_dev) return yfail(e, FLOPE_EINVAL, "flope_yolo_forward: frame_dev is NULL");
  Y_TRY(e, hipSetDevice(e->device));
  YLetterP L = e->letter; L.frame = frame_dev;
  int s = e->dtype == FLOPE_DT_F32 ? flope_y32_letter_launch(&L, stream) : flope_yletter_launch(&L, e->dtype, stream);
  if (s) return yfail(e, FLOPE_EHIP, std::string("letterbox: ") + hipGetErrorString((hipError_t)s));
  return run_ops(e, stream);
}

static int detect_body(flope_yolo* e, const uint8_t* frame_dev, float conf, float iou, int max_det, float* det_dev,
                       int32_t* count_dev, uint8_t* mask_dev, void* stream) {
  int rc = flope_yolo_forward(e, frame_dev, stream);
  if (rc) return rc;
  YDecodeP d = e->dec; d.conf = conf;
  int s = flope_ydecode_launch(&d, stream);
  YNmsP q = e->nms; q.conf = conf; q.iou = iou; q.max_det = max_det; q.det = det_dev; q.det_count = count_dev;
  if (!s) s = flope_ynms_launch(&q, stream);
  YMaskP m = e->mask; m.det_count = count_dev; m.max_det = max_det;
  if (!s) s = flope_ymask_launch(&m, e->dtype, stream);
  if (!s) s = flope_resize_linear_u8_launch(e->merged, e->h, e->w, mask_dev, e->H, e->W, stream);
  if (s) return yfail(e, FLOPE_EHIP, std::string("post-processing: ") + hipGetErrorString((hipError_t)s));
  return FLOPE_OK;
}

// "graph" option: the launch sequence of one (frame buffer, thresholds, output buffers) tuple is captured once into a
// hipGraph and replayed afterwards.  Frees the host; does not shorten the frame on MI355X (see the option defaults above).
extern "C" int flope_yolo_detect(flope_yolo_handle e, const uint8_t* frame_dev, float conf, float iou, int max_det,
                                 float* det_dev, int32_t* count_dev, uint8_t* mask_dev, void* stream) {
  if (!e) return yfail(nullptr, FLOPE_EINVAL, "flope_yolo_detect: NULL handle");
  if (!e->loaded) return yfail(e, FLOPE_ESTATE, "detect before flope_yolo_load_weights");
  if (!frame_dev || !det_dev || !count_dev || !mask_dev) return yfail(e, FLOPE_EINVAL, "flope_yolo_detect: NULL argument");
  if (max_det < 1 || max_det > kMaxDet) return yfail(e, FLOPE_EINVAL, "flope_yolo_detect: max_det must be within 1..300");
  if (!(conf >= 0.f && conf < 1.f) || !(iou > 0.f && iou <= 1.f)) return yfail(e, FLOPE_EINVAL, "flope_yolo_detect: bad thresholds");
  if (!e->opt_graph) return detect_body(e, frame_dev, conf, iou, max_det, det_dev, count_dev, mask_dev, stream);
  Y_TRY(e, hipSetDevice(e->device));
  hipStream_t st = (hipStream_t)stream;
  GraphKey key;
  memset(&key, 0, sizeof key);                              // the struct has tail padding and is compared bytewise
  key.frame = frame_dev; key.det = det_dev; key.count = count_dev; key.mask = mask_dev; key.conf = conf; key.iou = iou;
  key.max_det = max_det; key.batch = e->opt_batch * 4 + e->opt_bneck + e->opt_f32mfma * 16 + e->opt_chain * 32; key.generic_attn = e->opt_generic_attn;
  hipGraphExec_t exec = nullptr;
  for (size_t i = 0; i < e->graphs.size(); ++i)
    if (memcmp(&key, &e->graphs[i].key, sizeof key) == 0) {
      exec = e->graphs[i].exec;
      if (i + 1 != e->graphs.size()) std::rotate(e->graphs.begin() + i, e->graphs.begin() + i + 1, e->graphs.end());
      break;
    }
  if (!exec) {
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();                              // e.g. the legacy null stream cannot be captured: run eagerly
      return detect_body(e, frame_dev, conf, iou, max_det, det_dev, count_dev, mask_dev, stream);
    }
    const int rc = detect_body(e, frame_dev, conf, iou, max_det, det_dev, count_dev, mask_dev, stream);
    const hipError_t ec = hipStreamEndCapture(st, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (ec != hipSuccess || !g) return yfail(e, FLOPE_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
    const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (ei != hipSuccess) return yfail(e, FLOPE_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
    if (e->graphs.size() >= kGraphCache) { drop_graph(e->graphs.front()); e->graphs.erase(e->graphs.begin()); }
    flope_yolo::Captured c;
    memcpy(&c.key, &key, sizeof key); c.exec = exec; c.done = nullptr;
    if (hipEventCreateWithFlags(&c.done, hipEventDisableTiming) != hipSuccess) c.done = nullptr;
    e->graphs.push_back(c);
  }
  Y_TRY(e, hipGraphLaunch(exec, st));
  if (e->graphs.back().done) Y_TRY(e, hipEventRecord(e->graphs.back().done, st));   // most recently used = last
  return FLOPE_OK;
}

extern "C" int flope_yolo_read_tensor(flope_yolo_handle e, const char* name, float* dst_dev, int64_t* dims_out, void* stream) {
  if (!e || !name || !dims_out) return yfail(e, FLOPE_EINVAL, "flope_yolo_read_tensor: NULL argument");
  if (!e->loaded) return yfail(e, FLOPE_ESTATE, "flope_yolo_read_tensor before flope_yolo_load_weights");
  auto it = e->taps.find(name);
  if (it == e->taps.end()) return yfail(e, FLOPE_EINVAL, std::string("flope_yolo_read_tensor: unknown tensor ") + name);
  const Tap& t = it->second;
  dims_out[0] = t.C; dims_out[1] = t.H; dims_out[2] = t.W;
  if (!dst_dev) return FLOPE_OK;                   // size query
  Y_TRY(e, hipSetDevice(e->device));
  const int s = flope_yread_launch(t.ptr, t.is_f32, t.H, t.W, t.C, t.ld, e->dtype, dst_dev, stream);
  if (s) return yfail(e, FLOPE_EHIP, std::string("read_tensor: ") + hipGetErrorString((hipError_t)s));
  return FLOPE_OK;
}

extern "C" int flope_yolo_set_option(flope_yolo_handle e, const char* name, int value) {
  if (!e || !name) return yfail(e, FLOPE_EINVAL, "flope_yolo_set_option: NULL argument");
  if (!strcmp(name, "generic_attn")) { const int prev = e->opt_generic_attn; e->opt_generic_attn = value != 0; return prev; }
  if (!strcmp(name, "xcd") || !strcmp(name, "tile") || !strcmp(name, "splitk_max_m") || !strcmp(name, "wlds")) {   // process-wide A/B knobs; the batched schedule bakes them in
    const int prev = !strcmp(name, "xcd") ? flope_yconv_xcd_mode(value) : !strcmp(name, "tile") ? flope_yconv_tile_mode(value)
                   : !strcmp(name, "wlds") ? flope_yconv_wlds_mode(value) : flope_yconv_splitk_max_m(value);
    if (e->loaded) {
      if (int rc = build_schedules(e)) return rc;
      drop_graphs(e);
    }
    return prev;
  }
  if (!strcmp(name, "pool_lds")) return flope_ypool_lds_mode(value);
  if (!strcmp(name, "bneck")) {
    const int prev = e->opt_bneck; e->opt_bneck = value < 0 ? 0 : (value > 2 ? 2 : value);
    if (e->loaded && prev != e->opt_bneck) {
      if (int rc = build_schedules(e)) return rc;
      drop_graphs(e);
    }
    return prev;
  }
  if (!strcmp(name, "batch")) { const int prev = e->opt_batch; e->opt_batch = value != 0; return prev; }
  if (!strcmp(name, "chain")) {
    const int prev = e->opt_chain; e->opt_chain = value != 0;
    if (e->loaded && prev != e->opt_chain) {
      if (int rc = build_schedules(e)) return rc;
      drop_graphs(e);
    }
    return prev;
  }
  if (!strcmp(name, "f32mfma")) { const int prev = e->opt_f32mfma; e->opt_f32mfma = value != 0; return prev; }
  if (!strcmp(name, "graph")) { const int prev = e->opt_graph; e->opt_graph = value != 0; return prev; }
  return yfail(e, FLOPE_EINVAL, std::string("flope_yolo_set_option: unknown option ") + name);
}

// Developer aid: `iters` forwards with a HIP event pair around every launch of the graph; text table of the mean microseconds
// per launch (name, kind, geometry) into text_out.  Serialises nothing but the stream it is given.
extern "C" int flope_yolo_profile(flope_yolo_handle e, const uint8_t* frame_dev, int iters, char* text_out, int cap, void* stream) {
  if (!e || !frame_dev || !text_out || cap < 64 || iters < 1) return yfail(e, FLOPE_EINVAL, "flope_yolo_profile: bad argument");
  if (!e->loaded) return yfail(e, FLOPE_ESTATE, "flope_yolo_profile before flope_yolo_load_weights");
  Y_TRY(e, hipSetDevice(e->device));
  hipStream_t st = (hipStream_t)stream;
  const std::vector<Launch>& sched = e->sched[sched_index(e)];
  const size_t n = sched.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& x : ev) Y_TRY(e, hipEventCreate(&x));
  std::vector<double> us(n, 0.0);
  int rc = FLOPE_OK;
  for (int it = 0; it < iters + 1 && !rc; ++it) {
    YLetterP L = e->letter; L.frame = frame_dev;
    int s = e->dtype == FLOPE_DT_F32 ? flope_y32_letter_launch(&L, stream) : flope_yletter_launch(&L, e->dtype, stream);
    for (size_t i = 0; i < n && !s; ++i) {
      hipEventRecord(ev[i], st);
      s = launch_one(e, sched[i], st);
    }
    hipEventRecord(ev[n], st);
    if (s || hipStreamSynchronize(st) != hipSuccess) { rc = yfail(e, FLOPE_EHIP, "flope_yolo_profile: launch failed"); break; }
    if (it == 0) continue;                           // warm-up
    for (size_t i = 0; i < n; ++i) { float ms = 0.f; hipEventElapsedTime(&ms, ev[i], ev[i + 1]); us[i] += ms * 1e3; }
  }
  for (auto& x : ev) hipEventDestroy(x);
  if (rc) return rc;
  std::string t;
  char line[320];
  double total = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double u = us[i] / iters;
    total += u;
    if (sched[i].op >= 0 && e->ops[sched[i].op].kind == Op::CONV) {
      const Op& op = e->ops[sched[i].op];
      snprintf(line, sizeof line, "%3zu %7.2f L%-2d conv%d s%d %4dx%-4d cin %4d cout %4d nt %d mode %d %s\n", i, u, op.level, op.conv.k, op.conv.stride, op.conv.Ho,
               op.conv.Wo, op.conv.Cin, op.conv.Cout, op.nt, op.conv.out_mode, op.name.c_str());
    } else if (sched[i].op >= 0) {
      const Op& op = e->ops[sched[i].op];
      snprintf(line, sizeof line, "%3zu %7.2f L%-2d %s %s\n", i, u, op.level,
               op.kind == Op::DW ? "dw" : op.kind == Op::POOL ? "pool" : op.kind == Op::UP ? "up" : op.kind == Op::BNECK ? "bottleneck" : "attn", op.name.c_str());
    } else if (sched[i].chain_dev) {
      snprintf(line, sizeof line, "%3zu %7.2f L%-2d chain x%d (%d workgroups) %.200s\n", i, u, e->ops[sched[i].members[0]].level, sched[i].chain.n, sched[i].chain.tiles,
               launch_name(e, sched[i]).c_str());
    } else {
      snprintf(line, sizeof line, "%3zu %7.2f L%-2d multi x%d (%d workgroups) %.200s\n", i, u, e->ops[sched[i].members[0]].level, sched[i].multi.n, sched[i].multi.total,
               launch_name(e, sched[i]).c_str());
    }
    t += line;
  }
  snprintf(line, sizeof line, "total %.1f us over %zu launches (%zu ops)\n", total, n, e->ops.size());
  t += line;
  snprintf(text_out, (size_t)cap, "%s", t.c_str());
  return FLOPE_OK;
}

extern "C" int flope_yolo_graph_cache_size(flope_yolo_handle e) { return e ? (int)e->graphs.size() : 0; }
extern "C" double flope_yolo_flops(flope_yolo_handle e) { return e ? e->flops : 0.0; }
extern "C" int flope_yolo_launches(flope_yolo_handle e) { return e ? (int)e->sched[sched_index(e)].size() + 6 : 0; }   // + letterbox, decode, nms, 2 mask kernels, resize
