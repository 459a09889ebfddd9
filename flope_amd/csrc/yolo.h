// Launch-parameter structs of the YOLO11-seg detector kernels (yolo.hip) -- the front end of
// FastPosePredictor.get_bbox_mask (reference sunflower/predictor/fast_pose_predictor.py:36,44-57; the network
// itself lives in ultralytics 8.3.27, environment.yml:231).  gfx950 only.
//
// Activations are NHWC 16-bit *views*: a pointer (already offset to the view's first channel), the channel
// count of the view and the pixel stride `ld` of the buffer it lives in (in elements).  Concat / chunk / split
// of the reference graph are therefore free: producers write straight into channel slices of the consumer's
// buffer.  No spatial padding: border taps are redirected to a 16-byte zero page.
#pragma once
#include <stdint.h>

struct YConvP {
  const void* in;  int Hi, Wi, Cin, ldi;     // Cin % 8 == 0
  void* out;       int Ho, Wo, Cout, ldo;    // out_mode 1: float32 rows, ldo in floats
  const void* res; int ldr;                  // optional residual view, added AFTER the activation (nullptr = none)
  const void* w;                             // 16-bit, MFMA A-fragment order [channel block][k step][NT][lane][8] (Builder::pack)
  const float* bias;                         // [rows]
  const void* zero;                          // >= 16 bytes of zeros
  int k, stride;                             // 1 or 3 (pad = k / 2); 1 or 2
  int act;                                   // 1 = SiLU
  int M;                                     // Ho * Wo
  int cg;                                    // Cin / 8: 16-byte granules per tap
  unsigned cg_mg, cg_sh;                     // n / cg as multiply-shift
  int ksteps;                                // Kp / 32
  int out_mode;                              // 0: 16-bit view, 1: float32 view, 2: 2x2 stride-2 transposed conv scatter (16-bit)
  int dc;                                    // out_mode 2: real output channels (rows = 4 * dc, row = (dy*2+dx)*dc + co)
  int xcd;                                   // set by the launcher: workgroup order remapped so that an XCD owns an image band
  int tile, tiles_x;                         // set by the launcher: LDS-staged 8 x 16 output tiles; tiles per row
  int wlds, pad2_;                           // set by the launcher: tile path with the weight image staged in LDS as well
  unsigned pw_mg, pw_sh;                     // n / (patch width) as multiply-shift
  // float32 mode, exact-fp32 MFMA kernel (yolo_f32.hip y32m_conv_kernel): weights in A-fragment order of v_mfma_f32_16x16x4_f32,
  // [channel block of 16 nt rows][k16 step][channel tile][lane = kq * 16 + row][4 k] (k = 16 step + 4 kq + element), rows permuted
  // as in the 16-bit image; bias permuted the same way; k16 steps = ceil(k k Cin / 16)
  const void* w32m; const float* bias32m; int k16steps, nt32m;
  unsigned wo_mg, wo_sh;                     // n / Wo as multiply-shift (the pixel decode of the small-map kernels)
};

struct YDwP {            // depthwise 3x3, stride 1, pad 1 (+ folded BN) (+ SiLU) (+ add)
  const void* in; int H, W, C, ldi;
  void* out; int ldo;
  const void* add; int lda;                  // optional: out = dw(in) + add   (PSA: attention output + positional conv)
  const float* w;                            // [9][C]
  const float* bias;                         // [C]
  int act;
  int blk, blk_stride, blk_off;              // blk != 0: input channel of output channel c = (c / blk) * blk_stride + blk_off + c % blk
};

// Bottleneck = cv2(cv1(x)) (+ x): two 3x3 stride-1 Conv+BN+SiLU of <= 64 channels each, fused through LDS (ybneck_body).
// c1 / c2 are the launch parameters of the two convs exactly as the unfused path would use them (c1.out is never written).
struct YBneckP { YConvP c1, c2; int tiles_x, th; };      // th: output rows per workgroup (8, or 4 on maps <= 4096 pixels)

// ymulti_kernel: up to kYMultiMax independent conv / depthwise launches sharing one grid
constexpr int kYMultiMax = 8;
struct YMultiOp {
  int code;                                  // 0..11: conv, (NT index 0/1/2) * 4 + (3x3 ? 2 : 0) + (split-K ? 1 : 0); 12: depthwise;
                                             // 16 + (NT1 index) * 4 + (NT2 index): fused Bottleneck
  int nbx;                                   // conv: workgroups along pixels (local block b -> (b % nbx, b / nbx))
  int start;                                 // first workgroup of this op in the grid (a multiple of 8)
  int nblocks;                               // its workgroups; the grid pads every op to a multiple of 8
  union U { YConvP c; YDwP d; YBneckP b; } u;
};
struct YMultiP { int n, total, lds, pad_; YMultiOp op[kYMultiMax]; };

// ychain_kernel (r05): up to kYChainMax CONSECUTIVE 1x1 stride-1 convs on one small map, run back to back by one grid: a workgroup owns
// a pixel tile and pushes it through all of them (a 1x1 conv reads its own pixels only), its four waves taking the channel blocks of
// each conv side by side; what one conv wrote for the tile is visible to the next behind a workgroup barrier.  One launch instead of n.
constexpr int kYChainMax = 4;
struct YChainP { int n, tiles, pad0_, pad1_; int nt[kYChainMax]; YConvP op[kYChainMax]; };

struct YPoolP { const void* in; int H, W, C, ldi; void* out; int ldo; int n; };    // n (1..3) cascaded 5x5 s1 p2 max-pools, -inf border;
                                                                                   // result i -> channels [i C, (i+1) C) of out
struct YUpP { const void* in; int H, W, C, ldi; void* out; int ldo; };             // nearest 2x: out is [2H][2W]

struct YAttnP {          // ultralytics Attention: per head q(32) k(32) v(64) interleaved in the qkv map
  const void* qkv; int N, heads, ld;         // ld = heads * 128
  void* out; int ldo;                        // [N][heads * 64]
  float scale;
};

struct YLetterP {        // uint8 BGR frame -> letterboxed RGB / 255, 8-channel 16-bit NHWC (channels 3..7 zero)
  const uint8_t* frame; int H, W;
  void* out; int h, w;                       // letterboxed size
  int nh, nw, top, left;                     // resized content size and its offset
  double sx, sy;                             // cv::resize scale = 1 / ((double)n_dst / n_src)
};

struct YDecodeP {
  const float* pred; int A, no, nc;          // [A][no]: 64 DFL logits, nc class logits, 32 mask coefficients
  int lvl_a0[4], lvl_w[3], lvl_stride[3];    // anchors of level l: [lvl_a0[l], lvl_a0[l+1])
  float conf;
  float* cand_box;                           // [A][4] xyxy (letterbox pixels)
  float* cand_conf;                          // [A]
  int* cand_cls;                             // [A]
};

struct YNmsP {
  const float* cand_box; const float* cand_conf; const int* cand_cls;
  int A; float conf;                         // candidates = anchors with confidence > conf (the kNmsCap most confident ones)
  float iou; int max_det; float max_wh;
  // letterbox -> frame (ops.scale_boxes): x = (x - pad_x) / gain clipped to [0, W]
  float pad_x, pad_y, gain; int frame_w, frame_h;
  float* det;                                // [max_det][8]: frame xyxy, conf, cls, letterbox-box index (anchor), 0
  float* det_lb;                             // [max_det][4] letterbox xyxy (mask cropping)
  int* det_anchor;                           // [max_det]
  int* det_count;
};

struct YMaskP {
  const void* proto; int mh, mw;             // [mh][mw][32] 16-bit
  const float* pred; int no, nc;             // coefficient row of anchor a: pred + a * no + 64 + nc
  const float* det_lb; const int* det_anchor; const int* det_count; int max_det;
  int ih, iw;                                // letterboxed input size
  float* low;                                // [max_det][mh][mw] cropped low-resolution masks
  uint8_t* merged;                           // [ih][iw] 255 where any instance mask is set
};
