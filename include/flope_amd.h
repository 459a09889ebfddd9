/*
 * flope_amd.h -- C-ABI of the MI355X-native flower-pose hot path.
 *
 * The reference (wvu-irl/flope) has no FFI of its own: its seam is Python
 * duck-typing on an nn.Module and two predictor classes.  This
 * header is therefore the build-defined boundary the Python facade
 * (flope_amd/sunflower/...) binds with ctypes; every entry point names the
 * reference call it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, a negative FLOPE_E* code otherwise;
 *     flope_last_error() returns the message of the last failure on the handle
 *     (or of the last handle-less failure when h == NULL).
 *   - "dev" pointers are device (HBM) addresses owned by the caller
 *     (tensor.data_ptr()); "host" pointers are ordinary host memory.
 *   - kernels are enqueued on the hipStream_t passed as `void* stream`
 *     (0 = the null stream); nothing synchronises internally and nothing is
 *     allocated inside flope_forward / flope_procrustes / flope_crop_* /
 *     flope_depth_* (workspace is allocated by flope_create).
 *   - a handle is bound to one device and is not thread-safe.
 */
#ifndef FLOPE_AMD_H
#define FLOPE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flope_engine* flope_handle;

/* error codes */
#define FLOPE_OK            0
#define FLOPE_EINVAL       -1   /* bad argument / shape */
#define FLOPE_EHIP         -2   /* a HIP runtime call failed */
#define FLOPE_ESTATE       -3   /* call order (e.g. forward before load_weights) */
#define FLOPE_EWEIGHTS     -4   /* state_dict entry missing / wrong shape / not finite */

/* arithmetic type of the trunk (activations + conv weights as stored in HBM;
 * accumulation, bias, residual add and the whole head are always fp32) */
#define FLOPE_DT_BF16       0   /* MFMA v_mfma_f32_16x16x32_bf16 */
#define FLOPE_DT_F16        1   /* MFMA v_mfma_f32_16x16x32_f16  */
#define FLOPE_DT_F32        2   /* strict mode: plain fp32 direct convolution (no MFMA) */

/* layout / dtype of the crop batch handed to flope_forward */
#define FLOPE_IN_F32_NCHW   0   /* reference API: float32 [B,3,H,W] in [0,1] (posenet.py:31) */
#define FLOPE_IN_BF16_NHWC  1   /* bfloat16 [B,H,W,3] in [0,1]  (BASELINE cfg2) */
#define FLOPE_IN_F16_NHWC   2   /* float16  [B,H,W,3] in [0,1] */
#define FLOPE_IN_U8_NHWC    3   /* uint8    [B,H,W,3] 0..255, scaled by 1/255 on load */

/* activation taps for flope_read_stage (parity tests) */
#define FLOPE_STAGE_STEM    0   /* conv1+bn1+relu          [B,64,H/2,W/2] */
#define FLOPE_STAGE_POOL    1   /* maxpool                 [B,64,H/4,W/4] */
#define FLOPE_STAGE_LAYER(li, bi)  (2 + ((li) - 1) * 2 + (bi))   /* li 1..4, bi 0..1 */
#define FLOPE_STAGE_FEAT    10  /* global average pool     [B,512]  */
#define FLOPE_STAGE_HIDDEN  11  /* fc.0 + ReLU             [B,2048] */

/* ---- life cycle ----------------------------------------------------------
 * Replaces PoseResNet().to(device)   (sunflower/models/posenet.py:6-22,
 * fast_pose_predictor.py:31): builds the fixed launch plan and allocates every
 * activation buffer for crops of height x width up to max_batch.  Never fetches
 * ImageNet weights (the reference does, posenet.py:10). */
int flope_create(int device_id, int height, int width, int max_batch, int dtype,
                 int backbone_out_dim, flope_handle* out);
int flope_destroy(flope_handle h);
const char* flope_last_error(flope_handle h);

/* Replaces model.load_state_dict(torch.load(path, weights_only=True))
 * (scripts/test_posenet.py:51, fast_pose_predictor.py:32).  Entries are host
 * fp32, contiguous, named exactly as in the reference's 124-entry state_dict
 * (num_batches_tracked entries may be omitted).  The library folds eval-mode
 * BatchNorm into each conv, converts to the trunk dtype, repacks into the
 * MFMA tile order and uploads. */
int flope_load_weights(flope_handle h, int n, const char* const* names,
                       const float* const* host_ptrs, const int* ndims,
                       const int64_t* const* shapes);

/* ---- the hot path ----------------------------------------------------------
 * Replaces  r9 = model(image_batch); R = procrustes_to_rotmat(r9)
 * (fast_pose_predictor.py:126-127, scripts/test_posenet.py:142-144).
 * x_dev: crop batch in `in_format`; r9_dev: float32 [B,9] (may be NULL);
 * R_dev: float32 [B,9] row-major 3x3 rotations (may be NULL).  batch <= max_batch. */
int flope_forward(flope_handle h, const void* x_dev, int in_format, int batch,
                  float* r9_dev, float* R_dev, void* stream);

/* PoseResNet.extract_features (posenet.py:24-29): float32 [B,backbone_out_dim]. */
int flope_extract_features(flope_handle h, const void* x_dev, int in_format, int batch,
                           float* feat_dev, void* stream);

/* Stand-alone  roma.special_procrustes(M.reshape(-1,3,3))
 * (sunflower/utils/conversion.py:54-58): float32 [n,9] -> float32 [n,9]. */
int flope_procrustes(const float* M_dev, float* R_dev, int n, void* stream);

/* nullify_yaw_batch (sunflower/utils/mvg.py:240-251): R' = R * Rz(atan2(-R01,R00))^T,
 * float32 [n,9] -> float32 [n,9] (in place allowed). */
int flope_nullify_yaw(const float* R_dev, float* out_dev, int n, void* stream);

/* Pose assembly (fast_pose_predictor.py:131-144): optional yaw-nullification of R,
 * then Rt = [[R, xyz],[0,0,0,1]] as float32 [n,16]; xyz_dev float32 [n,3] (NULL -> 0). */
int flope_compose_pose(const float* R_dev, const float* xyz_dev, int n, int nullify_yaw,
                       float* Rt_dev, void* stream);

/* flope_forward + flope_compose_pose in one launch sequence: the head kernel that solves the
 * Procrustes problem also nullifies the yaw (if asked) and writes Rt = [[R', xyz],[0,0,0,1]]
 * as float32 [batch,16].  xyz_dev float32 [batch,3] or NULL (zeros); r9_dev / R_dev optional. */
int flope_forward_poses(flope_handle h, const void* x_dev, int in_format, int batch,
                        const float* xyz_dev, int nullify_yaw, float* r9_dev, float* R_dev,
                        float* Rt_dev, void* stream);

/* Crop-batch assembly (4 copies in the reference: fast_pose_predictor.py:108-123,
 * pose_predictor.py:138-153, scripts/test_posenet.py:124-140,
 * scripts/generate_metrics_utils.py:17-35): for each square box
 * [xmin,ymin,xmax,ymax] crop frame and mask, Lanczos-4 resize both to size x size,
 * out = img * (mask/255) / 255.  frame_dev uint8 [H,W,3], mask_dev uint8 [H,W],
 * boxes_dev int32 [n,4]; out_dev float32 [n,3,size,size] (FLOPE_IN_F32_NCHW) or
 * 16-bit [n,size,size,3] (FLOPE_IN_BF16_NHWC / FLOPE_IN_F16_NHWC). */
int flope_crop_resize_mask(const uint8_t* frame_dev, const uint8_t* mask_dev,
                           int frame_h, int frame_w, const int32_t* boxes_dev, int n,
                           int size, int out_format, void* out_dev, void* stream);

/* Test hook for the call above: the Lanczos-4 tables of one axis resized n_src -> n_dst, evaluated on the device by
 * the same code the crop kernel runs: s0_dev int32 [n_dst] (position of the first of the eight taps, unclamped),
 * coef_dev int16 [n_dst,8] (cv2's x2048 fixed-point weights).  Parity tests compare them bit for bit with the oracle. */
int flope_lanczos4_table(int n_src, int n_dst, int32_t* s0_dev, int16_t* coef_dev, void* stream);

/* Detector post-processing of `get_bbox_mask` (fast_pose_predictor.py:50-54): sum of the n
 * instance masks (float32 [n,h,w], any values) -> clip to [0,1] -> x255 -> uint8 -> bilinear
 * resize to the frame (cv2.resize default INTER_LINEAR, 8-bit fixed-point arithmetic) ->
 * out_dev uint8 [H,W].  n == 0 gives an all-zero mask.  scratch_dev: >= h*w bytes. */
int flope_merge_masks_resize(const float* masks_dev, int n, int h, int w, uint8_t* scratch_dev,
                             uint8_t* out_dev, int H, int W, void* stream);

/* Depth statistics + back-projection (image_manipulation.py:39-96, mvg.py:387-408):
 * valid = (near < d < far) & (mask > 128), eroded by the 10x10 ellipse; per box the
 * mean of valid depths, count >= 50 => reliable; xyz = K^-1 [u,v,1]^T * d/|K^-1[u,v,1]|.
 * depth_dev: depth_format 0 = uint16 [H,W] raw units, 1 = float32 [H,W]; metres = value /
 * depth_div (1000 at fast_pose_predictor.py:90, 10000 at pose_predictor.py:118, 1 for the
 * float-metres argument of get_depth_value itself), boxes int32 [n,4]
 * (un-squared boxes), K = {fx,fy,cx,cy}.  Outputs: depth_val float32 [n] (metres),
 * reliable int32 [n], xyz float32 [n,3].  scratch_dev: >= H*W + 16 + 512*n bytes
 * (valid mask, then 32 strip partials of 16 bytes per box). */
int flope_depth_lift(const void* depth_dev, int depth_format, const uint8_t* mask_dev,
                     int frame_h, int frame_w, float depth_div, float near_plane, float far_plane,
                     const int32_t* boxes_dev, int n, const float* K4_host,
                     uint8_t* scratch_dev, float* depth_val_dev, int32_t* reliable_dev,
                     float* xyz_dev, void* stream);

/* ---- introspection (parity tests / DESIGN.md numbers) ----------------------- */
/* Copy one internal activation of the LAST forward to float32: conv stages as
 * NCHW [B,C,h,w]; FEAT / HIDDEN as [B,n].  dims_out[4] receives the shape. */
int flope_read_stage(flope_handle h, int stage, int batch, float* dst_dev,
                     int64_t* dims_out, void* stream);
/* runtime knobs (A/B variants inside one build); returns previous value or <0 */
int flope_set_option(flope_handle h, const char* name, int value);
/* developer aid of diagnostic builds (-DFLOPE_STAG_DBG, option "dbg" = 64): in-kernel clock stamps that conv launch i of
 * the last forward left in the split-K workspace at byte offset i * 1048576 ({clk0, clk1, rt0, rt1} uint64 per workgroup and
 * wave group; per-double-step stamps 64 KB further with "dbg" = 128) -> host memory.  A production build leaves the workspace untouched. */
int flope_debug_read_ws(flope_handle h, void* dst_host, size_t offset, size_t bytes);
/* developer aid: the packed 16-bit epilogue helpers of csrc/common.h applied to n caller-supplied 32-bit words on the device
 * (which: 0 pk_out16<bf16>, 1 pk_out16<f16>, 2 pk_relu16<bf16>, 3 pk_relu16<f16>, 4 pk_max16_nonneg(w[i], w[i+1])) -- lets a test
 * run the device code of every conv / stem epilogue over all 65,536 patterns (tests/test_gpu_parity.py). */
int flope_debug_pk16(int which, int relu, const void* in_dev, void* out_dev, int n, void* stream);
/* algorithmic FLOPs of one forward for `batch` crops (2*MAC, convs + 2 FCs) */
double flope_forward_flops(flope_handle h, int batch);
/* number of kernel launches flope_forward enqueues */
int flope_forward_launches(flope_handle h);
/* Profile mode (flope_set_option(h, "profile", 1)): flope_forward records a HIP event on the
 * caller's stream before every launch and after the last.  flope_profile_read waits for the
 * last event and writes the GPU time (ms) of each launch of the most recent forward; returns
 * the number written or <0.  flope_launch_info: "layer|kernel" label and algorithmic FLOPs. */
int flope_profile_read(flope_handle h, float* ms_out, int cap);
/* option "profile" = 2: the last forward with its batch slices on their own streams, as in production, as a time line: event i was
 * recorded on slice slice_out[i]'s stream in front of that slice's next launch (behind its last one); ms_out[i] = milliseconds since the
 * fork (event 0).  Returns the number of events, < 0 on error.  Developer aid (tools/slice_timeline.py). */
int flope_profile_timeline(flope_handle h, float* ms_out, int* slice_out, int cap);
int flope_launch_info(flope_handle h, int idx, int batch, char* name, int name_cap, double* flops);
/* human-readable launch plan (one line per conv: tile config, patch/gather, LDS bytes) */
int flope_describe_plan(flope_handle h, char* buf, int buflen);
/* what a handle was created with (max_batch, FLOPE_DT_*, crop height / width, device ordinal); any pointer may be NULL */
int flope_engine_geometry(flope_handle h, int* max_batch, int* dtype, int* height, int* width, int* device_id);
/* library / build identification */
const char* flope_version(void);
/* A stream restricted to the compute units set in mask[words] (bit i of word i / 32 = CU i in the runtime's enumeration;
 * hipExtStreamCreateWithCUMask).  Used by the live loop to give the detector and the pose network disjoint parts of the chip
 * (FastPosePredictor.iter_flower_poses); any entry point of this library accepts such a stream. */
int flope_stream_create_cu_mask(int device_id, const uint32_t* mask, int words, void** out_stream);
int flope_stream_destroy(int device_id, void* stream);

/* ---- frame -> poses behind the detector (r05) ------------------------------------------------
 * Replaces the body of FastPosePredictor.get_flower_poses AFTER get_bbox_mask (fast_pose_predictor.py:55-56, 60-156): the box
 * loop (squarify_bb / bb_in_frame, mvg.py:324-351), get_depth_value + get_points3d (:88-106), the crop batch (:108-123), the
 * network, procrustes_to_rotmat, nullify_yaw_batch and the [N,4,4] assembly (:125-144), the depth-reliability filter (:97-102).
 * Inputs are the detector's device-resident outputs (flope_yolo_detect: det rows float32 [max_det,8] + count; or any detector's
 * xyxy rows in that layout), the uint8 BGR frame [H,W,3], the uint8 frame mask [H,W] and the depth image (format / divisor as for
 * flope_depth_lift).  Three calls per frame so that several frames can be in flight (one `slot` each):
 *   flope_frame_select   asynchronous: int16 boxes -> squarify -> in-frame filter on the device, in detection order
 *   flope_frame_enqueue  waits on the host for the NUMBER of surviving boxes (4 bytes: grids are sized by the host), then enqueues
 *                        everything else and the copy of the results to pinned host memory; returns that number (0: nothing to do)
 *   flope_frame_finish   waits for the results; poses_out float64 [n,16] row-major 4x4, flowers without reliable depth dropped;
 *                        returns n (0 = the reference's `None`)
 * flope_frame_to_poses = the three in sequence on slot 0.  max_boxes bounds the in-frame boxes of one frame (ultralytics' max_det = 300);
 * a frame with more is refused by flope_frame_enqueue.  More boxes than the engine's max_batch run as several forwards.
 * The PoseResNet engine must outlive the frame handle and must not run another forward concurrently. */
typedef struct flope_frame* flope_frame_handle;
int flope_frame_create(flope_handle pose_engine, int frame_h, int frame_w, int max_boxes, int slots, flope_frame_handle* out);
int flope_frame_destroy(flope_frame_handle f);
const char* flope_frame_last_error(flope_frame_handle f);
int flope_frame_select(flope_frame_handle f, int slot, const float* det_dev, const int32_t* count_dev, int max_det, void* stream);
int flope_frame_enqueue(flope_frame_handle f, int slot, const uint8_t* frame_dev, const uint8_t* mask_dev, const void* depth_dev,
                        int depth_format, float depth_div, const float* K4_host, float near_plane, float far_plane, void* stream);
int flope_frame_finish(flope_frame_handle f, int slot, double* poses_out, int cap);
int flope_frame_to_poses(flope_frame_handle f, const float* det_dev, const int32_t* count_dev, int max_det, const uint8_t* frame_dev,
                         const uint8_t* mask_dev, const void* depth_dev, int depth_format, float depth_div, const float* K4_host,
                         float near_plane, float far_plane, double* poses_out, int cap, void* stream);
/* test hook: the boxes flope_frame_select kept, as detected (good_host) and squared (sq_host), int32 [n,4] each; returns n */
int flope_frame_read_boxes(flope_frame_handle f, int slot, int32_t* good_host, int32_t* sq_host, int cap);

/* ---- TransformerEncoder (reference scripts/tf_encoder.py:5-27) -------------
 * Replaces `TransformerEncoder(input_dim, model_dim, out_dim, num_heads, num_layers, ff_dim,
 * dropout)` + `.load_state_dict()` + `forward(x)` in eval mode (dropout = identity):
 *   embedding Linear -> num_layers x post-norm nn.TransformerEncoderLayer (ReLU, batch_first,
 *   no mask, no positional encoding) -> out_layer Linear.
 * dtype FLOPE_DT_F32: fp32 everywhere (any dimensions).  FLOPE_DT_F16 / BF16: 16-bit activations,
 * fp32 accumulation; linears with N % 128 == 0 and K % 64 == 0 and attention with head_dim 64
 * (seq_len <= 512) run on MFMA, everything else on generic kernels.
 * max_tokens bounds batch*seq_len of any later forward.  Same ownership / error rules as above. */
typedef struct flope_tf_encoder* flope_tf_handle;
int flope_tf_create(int device_id, int input_dim, int model_dim, int out_dim, int num_heads,
                    int num_layers, int ff_dim, int max_tokens, int dtype, flope_tf_handle* out);
int flope_tf_destroy(flope_tf_handle h);
const char* flope_tf_last_error(flope_tf_handle h);
/* names as in the reference module's state_dict(): embedding.{weight,bias},
 * transformer_encoder.layers.<i>.{self_attn.in_proj_weight, self_attn.in_proj_bias,
 * self_attn.out_proj.{weight,bias}, linear1.*, linear2.*, norm1.*, norm2.*}, out_layer.* */
int flope_tf_load_weights(flope_tf_handle h, int n, const char* const* names,
                          const float* const* host_ptrs, const int* ndims,
                          const int64_t* const* shapes);
/* x_dev float32 [batch, seq_len, input_dim] -> y_dev float32 [batch, seq_len, out_dim] */
int flope_tf_forward(flope_tf_handle h, const float* x_dev, int batch, int seq_len, float* y_dev,
                     void* stream);
/* "generic" = 1 forces the generic kernels (A/B checks); returns previous value or <0 */
int flope_tf_set_option(flope_tf_handle h, const char* name, int value);
/* algorithmic FLOPs of one forward (2*MAC: linears + QK^T + PV) */
double flope_tf_forward_flops(flope_tf_handle h, int batch, int seq_len);

/* ---- YOLO11-seg detector front end ---------------------------------------------------
 * Replaces `self.yolo = YOLO(yolo_path)` (sunflower/predictor/fast_pose_predictor.py:36) and the
 * `results = self.yolo(image)` + mask / box post-processing of `get_bbox_mask` (:44-57).  The network and
 * its pre/post-processing are those of ultralytics 8.3.27 (environment.yml:231; not vendored by the reference):
 * LetterBox(imgsz, auto, stride 32) + BGR->RGB + /255 -> yolo11-seg graph (Conv+BN+SiLU, C3k2, SPPF, C2PSA,
 * upsample/concat neck, Segment head) -> DFL decode -> NMS -> coef.proto masks cropped, upsampled, > 0.
 * One handle = one device, one frame size.  dtype FLOPE_DT_F16 / FLOPE_DT_BF16 (16-bit maps, fp32 accumulation,
 * fp32 head outputs / decode / NMS), or FLOPE_DT_F32 = strict mode: float32 maps and plain float32 fused-multiply-add
 * convolutions over the same graph in program order (no MFMA, no 16-bit rounding; ~50x slower) -- the arithmetic the
 * reference runs ultralytics in (fast_pose_predictor.py:49), so that the INTEGER outputs of get_bbox_mask (int16 boxes,
 * uint8 mask, :52-56) can be compared for equality with a float32 pipeline.  Frame size / imgsz combinations whose
 * stride-32 map exceeds 2,560 tokens (imgsz above ~1600 for 16:9 frames) are rejected by flope_yolo_load_weights. */
typedef struct flope_yolo* flope_yolo_handle;
int flope_yolo_create(int device_id, int frame_h, int frame_w, int imgsz, int dtype, flope_yolo_handle* out);
int flope_yolo_destroy(flope_yolo_handle h);
const char* flope_yolo_last_error(flope_yolo_handle h);
/* letterboxed network input size for this frame size (multiples of 32) */
int flope_yolo_input_size(flope_yolo_handle h, int* in_h, int* in_w);
/* the ultralytics model's state_dict (`model.<i>. ...` names, host fp32): builds the graph from the key set and the
 * tensor shapes, folds BatchNorm(eps 1e-3), packs for MFMA, allocates every intermediate map.  Once per handle. */
int flope_yolo_load_weights(flope_yolo_handle h, int n, const char* const* names,
                            const float* const* host_ptrs, const int* ndims, const int64_t* const* shapes);
/* `results = self.yolo(image)` + get_bbox_mask's post-processing.  frame_dev: uint8 BGR [H,W,3] (a cv2 image).
 * conf / iou / max_det: ultralytics predict defaults are 0.25 / 0.7 / 300 (max_det <= 300).  NMS considers the 4,096 most
 * confident anchors above `conf` (ultralytics: 30,000) -- identical unless more than 4,096 anchors pass the threshold.
 * det_dev float32 [max_det,8]: rows = xyxy in frame pixels (ops.scale_boxes, clipped; the reference casts them to
 * int16), confidence, class, anchor index, 0 -- in NMS order; count_dev int32 [1]; mask_dev uint8 [H,W] = the summed /
 * clipped / x255 instance masks resized to the frame with cv2's 8-bit INTER_LINEAR (fast_pose_predictor.py:50-54). */
int flope_yolo_detect(flope_yolo_handle h, const uint8_t* frame_dev, float conf, float iou, int max_det,
                      float* det_dev, int32_t* count_dev, uint8_t* mask_dev, void* stream);
/* network only (parity tests): letterbox + every layer, no post-processing */
int flope_yolo_forward(flope_yolo_handle h, const uint8_t* frame_dev, void* stream);
/* copy one map of the LAST forward to float32 [C,H,W]: "input", graph outputs "0".."22" (yaml indices of Conv / C3k2 /
 * SPPF / C2PSA modules), "box0..2" / "cls0..2" / "coef0..2" (Segment head rows per level), "proto", "proto_up",
 * "mask_lb" (merged mask at the letterboxed size), "cand_box" [4,1,A] / "cand_conf" / "cand_cls" [1,1,A] (decoded box, best
 * confidence and class of every anchor) -- the last four after flope_yolo_detect.  dims_out[3] = {C,H,W}; dst_dev NULL = size
 * query only. */
int flope_yolo_read_tensor(flope_yolo_handle h, const char* name, float* dst_dev, int64_t* dims_out, void* stream);
/* runtime knobs (A/B variants inside one build; each returns the previous value or <0):
 *   "graph" (default 0): flope_yolo_detect captures its launch sequence into a hipGraph the first time it sees a
 *       (frame_dev, thresholds, output buffers) tuple and replays it afterwards -- keep those pointers stable across frames
 *       (measured as no gain on MI355X: the GPU-side chain of short kernels is the bound, not the host);
 *   "batch" (default 1): the independent launches of one dependency level of the graph (the Segment head's box / class /
 *       coefficient branches of a level, the Proto block beside them, parallel 1x1 convs inside C3k) share one grid; 0: one
 *       launch per op in the program order of the ultralytics yaml;
 *   "bneck" (default 1): Bottleneck pairs (3x3 -> 3x3, <= 64 channels) as one fused launch with the intermediate map in LDS;
 *       2: fused, but never in one grid with plain convs; 0: two conv launches;
 *   "generic_attn" (default 0): C2PSA attention on the generic fp32 kernel instead of the MFMA one;
 *   A/B knobs of the conv kernels, process-wide (every handle of the process; the schedule is rebuilt): "tile" (default 1:
 *       large maps stage an 8 x 16 tile's input patch in LDS; 0: fragments straight from global memory), "splitk_max_m"
 *       (default 8192: maps up to this many pixels split K over the four waves of a workgroup), "xcd" (default 0: 1 / 2 remap the
 *       workgroup order so that an XCD owns an image band -- measured as no gain), "pool_lds" (default 1: SPPF's pools in LDS), "wlds" (default 0: 1 stages the weight image of long-K
 *       3x3 tiles in LDS too -- measured slower, 0.78 vs 0.74 ms: 96 KB of LDS leave one workgroup per CU). */
int flope_yolo_set_option(flope_yolo_handle h, const char* name, int value);
/* developer aid: `iters` forwards with a HIP event pair around every launch of the graph; writes a text table (mean
 * microseconds per launch, kind, geometry, state_dict name) into text_out[cap] */
int flope_yolo_profile(flope_yolo_handle h, const uint8_t* frame_dev, int iters, char* text_out, int cap, void* stream);
double flope_yolo_flops(flope_yolo_handle h);      /* 2*MAC of one forward (convs + attention) */
int flope_yolo_launches(flope_yolo_handle h);      /* kernel launches per flope_yolo_detect */
int flope_yolo_graph_cache_size(flope_yolo_handle h);  /* captured launch sequences currently held ("graph" option; <= 8) */

#ifdef __cplusplus
}
#endif
#endif /* FLOPE_AMD_H */
