/*
 * include/lfd_hip.h -- C ABI of liblfd_hip.so: the MI355X (gfx950) implementation of the
 * LFD dense-prediction hot path (post-processing, losses, conv stack).
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it, no hidden sync, no
 *     allocation: outputs and workspace are caller-owned (`*_workspace_bytes()` query);
 *   - return value: 0 (LFD_OK) or a negative lfd_status code; no C++ exception crosses the ABI;
 *   - thread-safe and re-entrant per stream.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).
 */
#ifndef LFD_HIP_H_
#define LFD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: lfd_head_level_ptrs_t grew (w{1,2}_folded, tower1_out, w{1,2}_perm), lfd_conv3x3_c64_chain_nhwc_f16 removed,
 * lfd_p32_* (fp32-storage precision mode) added.
 * 3: lfd_pl_* added (the same precision mode on hi/lo fp16 planes, csrc/planes.hip).
 * Every loader checks lfd_hip_abi_version() against this number. */
#define LFD_HIP_ABI_VERSION 3
#define LFD_MAX_LEVELS 8

/* Conventions of every entry point below.
 *  - Plain pointers and sizes; device pointers unless a parameter says "host".  Outputs and workspaces are caller-allocated
 *    (`*_workspace_bytes()` queries); nothing is allocated, freed or retained by the library between calls.
 *  - Every launch goes to the `lfd_stream_t` (= hipStream_t, NULL = the default stream) the caller passes; no entry point
 *    synchronises the device or the stream unless its comment says so.  Calls are re-entrant per stream; two streams must not
 *    share a workspace.
 *  - NHWC fp16 tensors are contiguous and 16-byte aligned (the kernels move them as 16-byte vectors and LDS-DMA lines); the
 *    fused-block, downsample-block, stride-2 data-gradient, conv + statistics and head-output entry points check it.
 *  - The return value is an `lfd_status`: 0 = launched, negative = refused before anything touched the device (argument
 *    checks run on the host).  No C++ types or exceptions cross the ABI. */
typedef void* lfd_stream_t; /* hipStream_t */

#if defined(LFD_BUILDING)
#define LFD_API __attribute__((visibility("default")))
#else
#define LFD_API
#endif

enum lfd_status {
  LFD_OK = 0,
  LFD_ERR_INVALID_ARGUMENT = -1,
  LFD_ERR_WORKSPACE_TOO_SMALL = -2,
  LFD_ERR_LAUNCH_FAILED = -3,
  LFD_ERR_UNSUPPORTED = -4,
  LFD_ERR_NO_DEVICE = -5
};

enum lfd_dtype { LFD_F32 = 0, LFD_F16 = 1 };

LFD_API int lfd_hip_abi_version(void);
LFD_API const char* lfd_hip_status_string(int status);
/* "gfx950;<compiler>;<build date>" */
LFD_API const char* lfd_hip_build_info(void);

/* Tuning knobs: which kernel variant an entry point dispatches to where several compute the SAME result (bit for bit; the
 * hash-identity tests flip them) or an A/B timing needs a choice.  Process-global, explicit, thread-safe (relaxed atomics);
 * the library reads no environment variable.  (Rounds 1-3 read LFD_* variables once per process inside the library; the
 * Python host layer still translates those names into lfd_tuning_set calls when it loads the library, lfd_amd/_lib.py.)
 * lfd_tuning_set -> 0 / LFD_ERR_INVALID_ARGUMENT; lfd_tuning_get of an unknown key -> 0. */
typedef enum lfd_tune_key {
  LFD_TUNE_HEAD2 = 0,          /* 1: wave-per-32-pixel head passes (k_head2); 0: k_head (4 waves per 64-pixel tile)        default 1 */
  LFD_TUNE_H2_CHUNK = 1,       /* >= 2 (even): force this many 32-pixel groups per k_head2 work chunk; 0: planned by size   default 0 */
  LFD_TUNE_H2_AGPR = 2,        /* 1: output pass loads the folded tower filters straight into AccVGPRs                     default 1 */
  LFD_TUNE_H2_A1 = 3,          /* 1: output pass reads the tower-1 activations pass 2 stored instead of recomputing them   default 1 */
  LFD_TUNE_STEM2X = 4,         /* 1: fused 'faster' stem with both 32-channel slabs per wave (k_stem2x)                    default 1 */
  LFD_TUNE_X2_ALN = 5,         /* 1: aligned dword frame loads + funnel shift in k_stem2x                                  default 1 */
  LFD_TUNE_X2_STAGGER = 6,     /* 1: start-up stagger of k_stem2x workgroups (measured slower since round 2)               default 0 */
  LFD_TUNE_BLOCK_ROWS = 7,     /* -1: residual block kernel by map size; 0: k_block64 (8 x 16 tiles); 1: k_block64_rows    default -1 */
  LFD_TUNE_ROWS_WGS = 8,       /* > 0: workgroups per k_block64_rows launch; 0: one per CU                                 default 0 */
  LFD_TUNE_CONV128_SPLITK = 9, /* 1: split-K kernel for 128 -> 128 3x3 convs on maps of <= 16384 pixels                    default 1 */
  LFD_TUNE_CONV0_VALU = 10,    /* 1: the first stem conv of the training path on the VALU kernels instead of MFMA         default 0 */
  LFD_TUNE_PL_C3 = 11,         /* planes mode, 3x3 s1 64-channel convs: 2 = k_pl_c3p (K split over a wave pair per SIMD),
                                  1 = k_pl_c3 (one wave per SIMD, epilogue under the next contraction), 0 = generic      default 2 */
  LFD_TUNE_PL_HEAD_OUT_REGS = 12, /* 1: lfd_pl_head_levels mode 2 with the fp32 tile landing in registers (k_pl_head_out)           default 1 */
  LFD_TUNE_PL_HEAD_ROLES = 13,  /* 1: lfd_pl_head_levels mode 1 with producer / consumer waves (k_pl_head_b2)                              default 1 */
  LFD_TUNE_PL_STEM = 14,        /* lfd_pl_stem2x on aligned fp16 frames: 0 = tiles, one wave per SIMD (k_pl_stem2x); 1 = row stream with producer
                                   and consumer waves (k_pl_stem2xs)                                                          default 1 */
  LFD_TUNE_COUNT = 15
} lfd_tune_key_t;
LFD_API int lfd_tuning_set(int32_t key, int32_t value);
LFD_API int32_t lfd_tuning_get(int32_t key);

/* HOST-side members of the nms_ext surface, for CPU tensors / numpy arrays (host pointers, no stream): the reference's
 * module dispatches `nms` on the tensor's device (nms_ext.cpp:18-27 -> cpu/nms_cpu.cpp:7-66) and has `soft_nms`
 * (nms_cpu.cpp:76-206) and `nms_match` (:220-283) for CPU tensors only.  Not a fallback of the device path.
 *   lfd_nms_cpu_f32:       keep[<= n] original indices, score-descending (ties: input order); suppress when IoU > thr
 *   lfd_soft_nms_cpu_f32:  method 1 linear / 2 gaussian / else hard; out[<= n][6] = x1,y1,x2,y2,score,index(as float)
 *   lfd_nms_match_cpu_f32: members[n] grouped (kept box first, then the boxes it matched with IoU >= thr), group_sizes[<= n] */
LFD_API int lfd_nms_cpu_f32(const float* dets, int64_t n, float iou_thr, int64_t* keep, int64_t* num_keep);
LFD_API int lfd_soft_nms_cpu_f32(const float* dets, int64_t n, float iou_thr, int32_t method, float sigma, float min_score,
                                 float* out, int64_t* num_out);
LFD_API int lfd_nms_match_cpu_f32(const float* dets, int64_t n, float iou_thr, int32_t* members, int32_t* group_sizes,
                                  int64_t* num_groups);
/* the same three on float64 data: the reference dispatches on the tensor's dtype (AT_DISPATCH_FLOATING_TYPES,
 * nms_cpu.cpp:70,212,287) and evaluates double tensors -- a numpy array's default dtype -- in double */
LFD_API int lfd_nms_cpu_f64(const double* dets, int64_t n, float iou_thr, int64_t* keep, int64_t* num_keep);
LFD_API int lfd_soft_nms_cpu_f64(const double* dets, int64_t n, float iou_thr, int32_t method, float sigma, float min_score,
                                 double* out, int64_t* num_out);
LFD_API int lfd_nms_match_cpu_f64(const double* dets, int64_t n, float iou_thr, int32_t* members, int32_t* group_sizes,
                                  int64_t* num_groups);

/* ------------------------------------------------------------------------------------------
 * NMS.  Replaces nms_ext.nms(dets[n,5] f32, thr) -> LongTensor[k]
 *   (lfd/model/utils/build/nms/src/nms_ext.cpp:18-27,45-49; CUDA path
 *    src/cuda/nms_kernel.cu:24-68 kernel + :71-138 host sort / D2H mask / serial scan;
 *    CPU path src/cpu/nms_cpu.cpp:7-66).
 * dets = {x1,y1,x2,y2,score}; IoU = inter/(Sa+Sb-inter) in fp32, no +1, strict `>`;
 * keep[0..*num_keep) = ORIGINAL row indices of the kept boxes in score-descending order
 * (ties: lower original index first -- the reference leaves tie order unspecified).
 * Sort, 64x64-tile bitmask and the greedy scan all run on the device (no D2H of the mask).
 */
LFD_API size_t lfd_nms_workspace_bytes(int64_t n);
LFD_API int lfd_nms_f32(const float* dets, int64_t n, float iou_thr, int64_t* keep /*[n]*/,
                int32_t* num_keep /*[1]*/, void* workspace, size_t workspace_bytes,
                lfd_stream_t stream);

/* batched_nms(bboxes[k,4], scores[k], inds[k], nms_cfg, class_agnostic)
 *   (lfd/model/utils/nms.py:119-158): per-class NMS through the coordinate-offset trick
 *   off = label * (max(bboxes)+1) evaluated in fp32 exactly as the reference does (:148-150),
 *   IoU decided on the shifted boxes, offset subtracted from the kept rows (:156).
 * out_dets[r] = {x1,y1,x2,y2,score} of kept row r, keep[r] = its index into the inputs. */
LFD_API size_t lfd_batched_nms_workspace_bytes(int64_t k);
LFD_API int lfd_batched_nms_f32(const float* boxes /*[k,4]*/, const float* scores /*[k]*/,
                        const int64_t* labels /*[k]*/, int64_t k, float iou_thr,
                        int32_t class_agnostic, float* out_dets /*[k,5]*/,
                        int64_t* keep /*[k]*/, int32_t* num_keep /*[1]*/, void* workspace,
                        size_t workspace_bytes, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused per-location decode + score threshold + multi-class NMS for a whole batch.
 * Replaces LFD.get_results / _get_results_for_single_image / predict_for_single_image's
 * post-processing (lfd/model/lfd.py:397-509, :577-655), distance2bbox (:261-282) and
 * multiclass_nms (lfd/model/utils/nms.py:161-220), for every image of the batch in one call.
 */
typedef struct lfd_detect_desc {
  int32_t num_levels;
  int32_t level_h[LFD_MAX_LEVELS];       /* head_indexes_to_feature_map_sizes (lfd.py:532) */
  int32_t level_w[LFD_MAX_LEVELS];
  int32_t level_stride[LFD_MAX_LEVELS];  /* point_strides: x=j*stride, y=i*stride (lfd.py:93-98) */
  float level_range_lo[LFD_MAX_LEVELS];  /* regression_ranges[i] */
  float level_range_hi[LFD_MAX_LEVELS];
  int32_t num_classes;                   /* C (foreground classes) */
  int32_t num_cls_channels;              /* C for sigmoid scores, C+1 for softmax (CE loss) */
  int32_t score_mode;                    /* 0: sigmoid (lfd.py:454); 1: softmax, drop last (:450-452) */
  int32_t decode_mode;                   /* 0: sigmoid(reg)*max(range) (:483-486); 1: exp(reg) (:480-482);
                                            2: reg*range_hi ('independent', :468-478);
                                            3: reg is the distance itself (FCOS, fcos.py:392 on fcos_head.py:145-146) */
  int32_t class_agnostic;                /* nms_cfg['class_agnostic'] (nms.py:143) */
  int32_t max_candidates;                /* capacity K_cap per image of candidate / output rows */
  float score_thr;                       /* strict `>` (nms.py:204) */
  float iou_thr;                         /* strict `>` (nms_cpu.cpp:62) */
} lfd_detect_desc_t;

LFD_API size_t lfd_detect_workspace_bytes(const lfd_detect_desc_t* desc, int32_t batch);
/* cls [N,P,C'] and reg [N,P,4] in `in_dtype` (level-concatenated, row-major as LFD.forward
 * returns them, lfd.py:526-542).  img_meta [N,3] f32 = {clamp_width, clamp_height,
 * resize_scale} (meta['resized_width'|'resized_height'|'resize_scale'], lfd.py:443-444,499).
 * Outputs (rows beyond the count are untouched):
 *   out_dets   [N,K_cap,5] f32  x1,y1,x2,y2,score of kept boxes, score-descending
 *   out_labels [N,K_cap]   i32  0-based class
 *   out_cand   [N,K_cap]   i32  candidate ordinal (position in the reference's nonzero() order)
 *   out_point  [N,K_cap]   i32  flat point index p of the kept box
 *   out_counts [N,4]       i32  {num_candidates (clamped), num_kept, overflow_flag, true_candidates}
 */
LFD_API int lfd_detect_batched(const lfd_detect_desc_t* desc, int32_t batch, const void* cls,
                       const void* reg, int32_t in_dtype, const float* img_meta,
                       float* out_dets, int32_t* out_labels, int32_t* out_cand,
                       int32_t* out_point, int32_t* out_counts, void* workspace,
                       size_t workspace_bytes, lfd_stream_t stream);

/* The same step for the sibling meta-architectures that share the multiclass_nms boundary:
 *   FCOS._get_results_for_single_image  (lfd/model/fcos.py:356-412)
 *   LFDv2._get_results_for_single_image (lfd/model/lfdv2.py:593-669)
 * which use its two arguments LFD leaves at their defaults, plus a per-level pre-selection:
 *   centerness  [N,P] logits or NULL: every class score is multiplied by sigmoid(centerness) -- multiclass_nms's
 *               score_factors (nms.py:192-193; fcos.py:376,403-409), applied BEFORE the threshold as there;
 *   pre_nms_limit  > 0: a level with more points keeps its pre_nms_limit best, ranked by the point's largest final score
 *               (fcos.py:383-390 ranks after the centerness factor; lfdv2.py:620-627), before decode and threshold.
 *               torch.topk leaves the choice among EQUAL keys open; this implementation takes the lowest point index;
 *   post_nms_limit > 0: multiclass_nms max_num (nms.py:217-219) -- out_counts[n][1] = min(kept, post_nms_limit), rows
 *               beyond it are unspecified.
 * Candidate order (out_cand) is point-major within the image, not the reference's "top-k order within the level": it
 * only reaches the results through the order of exactly equal scores, which the reference leaves unspecified.
 * Everything else -- outputs, img_meta, thresholds -- as lfd_detect_batched; workspace: lfd_detect_ex_workspace_bytes. */
typedef struct lfd_detect_ext {
  int32_t pre_nms_limit;
  int32_t post_nms_limit;
} lfd_detect_ext_t;
LFD_API size_t lfd_detect_ex_workspace_bytes(const lfd_detect_desc_t* desc, int32_t batch);
LFD_API int lfd_detect_batched_ex(const lfd_detect_desc_t* desc, const lfd_detect_ext_t* ext, int32_t batch,
                                  const void* cls, const void* reg, const void* centerness, int32_t in_dtype,
                                  const float* img_meta, float* out_dets, int32_t* out_labels, int32_t* out_cand,
                                  int32_t* out_point, int32_t* out_counts, void* workspace, size_t workspace_bytes,
                                  lfd_stream_t stream);

/* Second half of lfd_detect_batched for a workspace whose candidate arrays were filled by a producer kernel
 * (lfd_head_forward_decode_f16: the head's last pass thresholds, decodes and appends by itself): sort + suppression mask
 * + scan, 3 launches.  Candidates may have arrived in any order -- ties are broken by (point, class), the reference's
 * nonzero() order -- so out_dets / out_labels / out_point / out_counts equal lfd_detect_batched's; out_cand is the
 * arrival slot (not reproducible).  With more than max_candidates candidates (overflow flag) the retained subset is
 * arbitrary where lfd_detect_batched keeps the first K_cap in point order.
 * The workspace's two counter arrays must be zero before the first producer launch (lfd_detect_workspace_reset, once
 * per allocation); lfd_detect_from_candidates re-arms them. */
LFD_API int lfd_detect_workspace_reset(const lfd_detect_desc_t* desc, int32_t batch, void* workspace,
                                       size_t workspace_bytes, lfd_stream_t stream);
LFD_API int lfd_detect_from_candidates(const lfd_detect_desc_t* desc, int32_t batch, float* out_dets, int32_t* out_labels,
                                       int32_t* out_cand, int32_t* out_point, int32_t* out_counts, void* workspace,
                                       size_t workspace_bytes, lfd_stream_t stream);

/* Decode only (no threshold/NMS): boxes [N,P,4] f32 and scores [N,P,C] f32 for parity tests
 * and for callers that want the reference's intermediate tensors (lfd.py:449-499). */
LFD_API int lfd_decode_all(const lfd_detect_desc_t* desc, int32_t batch, const void* cls, const void* reg,
                   int32_t in_dtype, const float* img_meta, float* out_boxes, float* out_scores,
                   lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sigmoid focal loss.  Replaces sigmoid_focal_loss_ext.forward / .backward
 *   (lfd/model/losses/build/sigmoid_focal_loss/src/sigmoid_focal_loss_ext.cpp:19-57,
 *    src/cuda/sigmoid_focal_loss_cuda.cu:24-59 fwd, :62-97 bwd).
 * logits [n,c], targets [n] int64 (c == background, <0 == ignore), losses / d_logits [n,c]. */
LFD_API int lfd_sigmoid_focal_loss_fwd(const void* logits, const int64_t* targets, int64_t n, int32_t c,
                               float gamma, float alpha, void* losses, int32_t dtype,
                               lfd_stream_t stream);
LFD_API int lfd_sigmoid_focal_loss_bwd(const void* logits, const int64_t* targets, const void* d_losses,
                               int64_t n, int32_t c, float gamma, float alpha, void* d_logits,
                               int32_t dtype, lfd_stream_t stream);
/* Fused forward + sum reduction (weight_reduce_loss 'sum/avg_factor',
 * lfd/model/losses/utils.py:28-54): *loss_sum += sum(losses) accumulated in fp64 partials,
 * deterministic (fixed-order two-stage reduction).  workspace: lfd_reduce_workspace_bytes(). */
LFD_API size_t lfd_reduce_workspace_bytes(void);
LFD_API int lfd_sigmoid_focal_loss_sum_f32(const float* logits, const int64_t* targets, int64_t n, int32_t c,
                                   float gamma, float alpha, float* loss_sum /*[1]*/, void* workspace,
                                   size_t workspace_bytes, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * IoU loss on aligned xyxy boxes.  Replaces bbox_overlaps(is_aligned=True) + iou_loss
 *   (lfd/model/losses/iou_loss.py:67-79,98-102,105-123): loss = -log(max(ov/max(a1+a2-ov,1e-6), eps)).
 * bwd: d_pred[n,4] = d_loss[n] * dloss/dpred (autograd of the reference expression). */
LFD_API int lfd_iou_loss_fwd_f32(const float* pred, const float* target, int64_t n, float eps, float* loss,
                         lfd_stream_t stream);
LFD_API int lfd_iou_loss_bwd_f32(const float* pred, const float* target, const float* d_loss, int64_t n,
                         float eps, float* d_pred, lfd_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Target assignment: LFD.annotation_to_target / _generate_target_for_single_image
 *   (lfd/model/lfd.py:109-259; on the CPU with [P, G] broadcasts in the reference).
 * Whole batch in one launch.  Points are generated on the fly (lfd.py:84-107: x = j*stride, y = i*stride,
 * level-major, row-major).  Ground truth: gt_boxes [sumG,4] (x, y, w, h), gt_labels [sumG], image i owns rows
 * [gt_offsets[i], gt_offsets[i+1]).  Outputs: cls_targets [n, P, num_classes] (score in (0,1], 0 = negative,
 * -1 = gray / ignored), reg_targets [n, P, 4] (l, t, r, b deltas; divided by the level's upper range when
 * `independent`).  assign_mode: 0 'longer', 1 'shorter', 2 'sqrt', 3 'dist' (lfd.py:206-215).
 * fp32, reference expression order, IEEE divide / sqrt: decisions and regression targets identical to the
 * reference, scores identical up to the last bit of the host's sqrt routine. */
typedef struct {
  int32_t n, num_levels;
  int32_t level_h[LFD_MAX_LEVELS], level_w[LFD_MAX_LEVELS], stride[LFD_MAX_LEVELS];
  int32_t reg_lo[LFD_MAX_LEVELS], reg_hi[LFD_MAX_LEVELS];     /* regression_ranges (lfd.py:43) */
  int32_t gray_lo[LFD_MAX_LEVELS], gray_hi[LFD_MAX_LEVELS];   /* gray ranges (lfd.py:49-50) */
  int32_t total_points, num_classes;
  int32_t assign_mode, independent;
} lfd_assign_desc_t;
LFD_API int lfd_assign_targets_f32(const lfd_assign_desc_t* d, const float* gt_boxes, const int64_t* gt_labels,
                           const int32_t* gt_offsets, float* cls_targets, float* reg_targets, lfd_stream_t stream);

/* CrossEntropyLoss core (lfd/model/losses/cross_entropy_loss.py:12-50 -> F.cross_entropy(pred, label,
 * reduction='none'); TT100K configs): loss[i] = logsumexp(logits[i,:]) - logits[i, labels[i]];
 * bwd: d_logits[i,j] = d_loss[i] * (softmax(logits[i,:])[j] - [j == labels[i]]). */
LFD_API int lfd_cross_entropy_fwd_f32(const float* logits, const int64_t* labels, int64_t m, int32_t channels, float* loss,
                              lfd_stream_t stream);
LFD_API int lfd_cross_entropy_bwd_f32(const float* logits, const int64_t* labels, const float* d_loss, int64_t m,
                              int32_t channels, float* d_logits, lfd_stream_t stream);

/* GIoU / DIoU / CIoU losses of aligned box pairs (lfd/model/losses/iou_loss.py:127-169, :172-223, :226-283; the other
 * members of LFD's IoU-type regression-loss family, lfd.py:64-66): kind 1 | 2 | 3; loss[n] and -- when
 * d_loss_d_pred != NULL -- the gradient [n,4] w.r.t. the predicted xyxy box, obtained by forward-mode
 * differentiation of the reference expression inside the kernel (max / min / clamp route like autograd). */
LFD_API int lfd_box_loss_f32(const float* pred, const float* target, int64_t n, int32_t kind, float eps, float* loss,
                     float* d_loss_d_pred, lfd_stream_t stream);
/* The remaining classification losses LFD accepts (lfd.py:52-56).  lfd_bce_with_logits_f32: elementwise
 * F.binary_cross_entropy_with_logits(logits, targets, reduction='none') over n values (bce_with_logits_loss.py:28-44) and
 * its derivative sigmoid(x) - t.  lfd_quality_focal_loss_f32 (gfocal_loss.py:11-52): per row the sum over classes of
 * BCE(x, t) |t - sigmoid(x)|^beta with t = scores[row] at class labels[row] (if it is a foreground label) and 0 elsewhere;
 * d_logits [rows, channels] (nullable) = its derivative. */
LFD_API int lfd_bce_with_logits_f32(const float* logits, const float* targets, int64_t n, float* loss, float* d_logits,
                            lfd_stream_t stream);
LFD_API int lfd_quality_focal_loss_f32(const float* logits, const int64_t* labels, const float* scores, int64_t rows,
                               int32_t channels, float beta, float* loss, float* d_logits, lfd_stream_t stream);
/* LFD's "independent" regression losses (lfd.py:61-66), elementwise over n values: kind 1 smooth-L1 with `beta`
 * (lfd/model/losses/smooth_l1_loss.py:11-22), 2 L1 (:25-30), 3 MSE (mse_loss.py:11-13); loss[n] and, when
 * d_loss_d_pred != NULL, the derivative w.r.t. pred. */
LFD_API int lfd_pointwise_loss_f32(const float* pred, const float* target, int64_t n, int32_t kind, float beta, float* loss,
                           float* d_loss_d_pred, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused LFD.get_loss (lfd/model/lfd.py:284-395): replaces the boolean gathers (green rows :309-315, positive
 * rows :319-321), the label construction (:328), FocalLoss / CrossEntropyLoss + IoULoss on the gathered rows
 * (:333-384, decode of prediction and target :360-377) and the three `.item()` syncs (:389-395) by
 *   sums     : one pass over all N*P rows -> 8 doubles {cls_sum, reg_sum, n_pos, sum of positive scores,
 *              n_green, 0, 0, 0} (fp64 block partials + fixed-order second stage: deterministic);
 *   finalize : out[8] floats {classification_loss, regression_loss, loss, n_pos, avg_factor_cls,
 *              avg_factor_reg, n_green(local), rank_scale}; `global_sums` are the sums all-reduced over the
 *              image-parallel ranks (== local_sums on one GPU) because the reference normalises by the
 *              global-batch n_pos (executor.py:198-200); rank_scale = world size when gradients are
 *              averaged over ranks afterwards (1 otherwise);
 *   bwd      : dense d loss / d pred_cls [N,P,channels], d loss / d pred_reg [N,P,4] for upstream
 *              gradients grad_out[3] of {classification_loss, regression_loss, loss}.
 * pred_cls has num_classes channels for cls_loss 0 (sigmoid focal) and num_classes + 1 for cls_loss 1
 * (cross entropy, background = last channel).  No positives -> regression_loss 0 with zero gradient
 * (lfd.py:386-387).  All tensors fp32, contiguous.
 */
typedef struct lfd_loss_desc {
  int32_t n, num_levels;
  int32_t level_h[LFD_MAX_LEVELS], level_w[LFD_MAX_LEVELS], stride[LFD_MAX_LEVELS];
  float range_max[LFD_MAX_LEVELS];  /* max(regression_ranges[i]) ('sigmoid' decode, lfd.py:368-373) */
  int32_t total_points, num_classes;
  int32_t cls_loss;                 /* 0 FocalLoss (sigmoid), 1 CrossEntropyLoss */
  int32_t decode_mode;              /* 0 sigmoid * range_max, 1 exp (lfd.py:366-374) */
  float gamma, alpha, iou_eps;
  float cls_loss_weight, reg_loss_weight;   /* loss_weight of the two loss modules */
  int32_t cls_weighted, reg_weighted;       /* enable_classification_weight / enable_regression_weight */
} lfd_loss_desc_t;
LFD_API size_t lfd_get_loss_workspace_bytes(void);
LFD_API int lfd_get_loss_sums_f32(const lfd_loss_desc_t* d, const float* pred_cls, const float* pred_reg,
                          const float* cls_targets, const float* reg_targets, void* workspace, size_t workspace_bytes,
                          double* sums, lfd_stream_t stream);
LFD_API int lfd_get_loss_finalize_f32(const lfd_loss_desc_t* d, const double* local_sums, const double* global_sums,
                              float rank_scale, float* out, lfd_stream_t stream);
LFD_API int lfd_get_loss_bwd_f32(const lfd_loss_desc_t* d, const float* pred_cls, const float* pred_reg,
                         const float* cls_targets, const float* reg_targets, const float* finalized,
                         const float* grad_out, float* grad_cls, float* grad_reg, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Parameter update of a training iteration over one flat fp32 buffer (all tensors of a param group
 * contiguous, 16-byte aligned).  Replaces OptimizerHook.after_train_iter's
 *   clip_grad.clip_grad_norm_(params, max_norm, norm_type=2)   (lfd/execution/hooks/optimizer_hook.py:21-24,30-33)
 *   optimizer.step() with torch.optim.SGD(lr, momentum, weight_decay) (optimizer_hook.py:36, WIDERFACE_LFD_S.py:216-226)
 * lfd_grad_norm_clip_coef_f32: norm_and_coef[0] = ||grads||_2 (fp64 accumulation, deterministic),
 *   norm_and_coef[1] = min(max_norm / (norm + 1e-6), 1).  extra_sumsq (nullable, device) is added to the
 *   sum of squares before the root (several buffers sharing one norm); sumsq_out (nullable) receives the total.
 * lfd_sgd_step_f32: g *= coef (when apply_clip; written back when write_clipped_grads, as
 *   clip_grad_norm_ mutates .grad); d = g + wd * p; buf = first_step ? d : momentum * buf + (1 - dampening) * d;
 *   d = nesterov ? d + momentum * buf : buf; p -= lr * d   (torch/optim/sgd.py _single_tensor_sgd).
 *   Overflow guard of the fp16 activation-gradient path (csrc/train.hip stores dz as fp16 x loss scale): when
 *   norm_and_coef != NULL and norm_and_coef[0] -- the total gradient norm -- is not finite (an inf gradient gives
 *   norm = inf and coef = 0, and 0 * inf = NaN would be written into the weights; a NaN gradient gives NaN), the whole
 *   update is skipped: parameters, momentum and gradients untouched; the caller reads the norm and lowers its loss scale
 *   (lfd_amd.train.DynamicLossScale).  norm_and_coef must be given when apply_clip != 0.
 */
LFD_API size_t lfd_grad_norm_workspace_bytes(void);
LFD_API int lfd_grad_norm_clip_coef_f32(const float* grads, int64_t n, float max_norm, const double* extra_sumsq,
                                void* workspace, size_t workspace_bytes, float* norm_and_coef, double* sumsq_out,
                                lfd_stream_t stream);
/* grads *= norm_and_coef[1] (the in-place scaling of clip_grad_norm_ when the update is not fused with it) */
LFD_API int lfd_scale_by_clip_coef_f32(float* grads, int64_t n, const float* norm_and_coef, lfd_stream_t stream);
LFD_API int lfd_sgd_step_f32(float* params, float* grads, float* momentum_buf, int64_t n, float lr, float momentum,
                     float dampening, float weight_decay, int32_t nesterov, int32_t first_step,
                     const float* norm_and_coef, int32_t apply_clip, int32_t write_clipped_grads, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Inference conv stack, NHWC fp16, fp32 accumulate on MFMA.  Replaces the nn.Conv2d +
 * nn.BatchNorm2d (folded) + ReLU (+ residual add) units of LFDResNet
 *   (lfd/model/backbone/lfd_resnet.py:96-154 FasterBlock, :21-93 FastBlock, :157-215 FastestBlock,
 *    :354-439 stem, :458-468 downsample; executed by cuDNN in the reference).
 * out = relu?( conv(in, w) + bias (+ residual) ), optionally followed in the same kernel by a
 * chained 1x1 conv: out = relu?( conv1x1(relu?(conv(in,w)+bias), tail_w) + tail_bias ).
 * Weights are pre-packed in MFMA fragment order (see lfd_pack_conv_weight_f16).
 */
typedef struct lfd_conv_desc {
  int32_t n, h, w;      /* input dims (NHWC) */
  int32_t cin, cout;    /* cin in {32,64,128}; cout multiple of 32 */
  int32_t ks, stride;   /* 1|3 (pad = ks/2), 1|2 */
  int32_t relu;
  int32_t tail_cout;    /* 0: no chained 1x1; else must equal cout */
  int32_t tail_relu;
} lfd_conv_desc_t;

LFD_API size_t lfd_conv_packed_weight_halfs(int32_t cin, int32_t cout, int32_t ks);
/* First conv of a residual block together with its identity branch (lfd_resnet.py:458-468): the
 * conv3x3 stride 2 (+BN+ReLU) -> out and the downsample conv1x1 stride 2 (+BN, no ReLU) -> ds_out are
 * produced by ONE launch from the same input tile (the 1x1 s2 reads the 3x3's centre tap). */
LFD_API int lfd_conv2d_downsample_nhwc_f16(const lfd_conv_desc_t* desc, const void* in, void* out,
                                           const void* w_packed, const float* bias,
                                           const void* ds_w_packed, const float* ds_bias, void* ds_out,
                                           const void* zeros, lfd_stream_t stream);

LFD_API int lfd_conv2d_nhwc_f16(const lfd_conv_desc_t* desc, const void* in, void* out,
                                const void* w_packed, const float* bias, const void* residual,
                                const void* tail_w_packed, const float* tail_bias,
                                const void* zeros /* 4096-byte line: [0,2048) zero (read), [2048,4096) trash (written) */, lfd_stream_t stream);

/* A whole residual block of the backbone without downsample branch (FasterBlock, lfd_resnet.py:96-154; every block of a
 * stage but the first) in ONE launch:   out = relu( conv3x3(relu(conv3x3(in, w1) + b1), w2) + b2 + in ),  64 -> 64 -> 64
 * channels, stride 1, BN folded into (w, b), NHWC fp16 [n, h, w, 64].  The intermediate map lives in LDS tile by tile
 * (csrc/block.hip: producer / consumer wave specialisation, both filters register-stationary); results are bit-identical
 * to two lfd_conv2d_nhwc_f16 launches.  w1_packed / w2_packed: lfd_conv_packed_weight_halfs(64, 64, 3) halfs each;
 * `in` must not alias `out`; zeros: the 4096-byte line of lfd_conv2d_nhwc_f16. */
LFD_API int lfd_fasterblock_fused_f16(int32_t n, int32_t h, int32_t w, const void* in, void* out, const void* w1_packed,
                                      const float* b1, const void* w2_packed, const float* b2, const void* zeros,
                                      lfd_stream_t stream);

/* The same block at 128 channels on the SMALL maps of the last backbone stage (17 x 30 at 1080p; WIDERFACE_LFD_S.py:97-146,
 * lfd_resnet.py:96-154) in ONE launch:   out = relu( conv3x3(relu(conv3x3(in, w1) + b1), w2) + b2 + in ),  128 -> 128 -> 128,
 * NHWC fp16 [n, h, w, 128].  A workgroup owns 4 x 8 output pixels and recomputes the 6 x 10 halo of the intermediate map in
 * LDS; both filters are streamed from L2 per workgroup, so this form is for maps of a few thousand pixels per launch (the
 * caller decides; csrc/block128.hip).  Results are bit-identical to two lfd_conv2d_nhwc_f16 launches on their split-K path
 * (n * h * w <= 16384).  w1_packed / w2_packed: lfd_conv_packed_weight_halfs(128, 128, 3) halfs each; `in` must not alias
 * `out`; all pointers 16-byte aligned. */
LFD_API int lfd_fasterblock128_fused_f16(int32_t n, int32_t h, int32_t w, const void* in, void* out, const void* w1_packed,
                                         const float* b1, const void* w2_packed, const float* b2, lfd_stream_t stream);

/* The FIRST block of a backbone stage -- the FasterBlock with a downsample branch (lfd_resnet.py:96-154, branch :458-468) --
 * in ONE launch:   y1 = relu(conv3x3_s2(in, w1) + b1);  ident = conv1x1_s2(in, wd) + bd;
 *                  out = relu(conv3x3_s1(y1, w2) + b2 + ident),      64 -> 64 channels, BN folded, NHWC fp16,
 * in [n, h, w, 64] -> out [n, (h-1)/2+1, (w-1)/2+1, 64].  y1 and ident live in LDS row rings (csrc/down.hip: a workgroup
 * streams down a 30-column strip; producer / consumer waves, three register-stationary filters); results are bit-identical
 * to lfd_conv2d_downsample_nhwc_f16 followed by lfd_conv2d_nhwc_f16 with `residual`.  w1_packed / w2_packed:
 * lfd_conv_packed_weight_halfs(64, 64, 3) halfs, wd_packed: lfd_conv_packed_weight_halfs(64, 64, 1); `in` must not alias
 * `out`; zeros: the 4096-byte line of lfd_conv2d_nhwc_f16. */
LFD_API int lfd_downblock_fused_f16(int32_t n, int32_t h, int32_t w, const void* in, void* out, const void* w1_packed,
                                    const float* b1, const void* wd_packed, const float* bd, const void* w2_packed,
                                    const float* b2, const void* zeros, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise operators of the sibling necks / heads, NHWC fp16, channels a multiple of 8:
 *   lfd_upsample_nearest_add_nhwc_f16   dst[n,H,W,c] += nearest(src[n,h,w,c]) -- the merge step of FPN / SimpleFPN
 *                                       (`lateral[i-1] += nn.Upsample(size, mode='nearest')(lateral[i])`,
 *                                       lfd/model/neck/fpn.py:133-135, simple_fpn.py:147-157); source index =
 *                                       min(floorf(dst_index * float(in)/out), in-1) as ATen computes it; fp32 add.
 *   lfd_relu_inplace_f16                nn.ReLU(inplace=True) in front of an extra level: it rewrites the PREVIOUS
 *                                       output level as well (fpn.py:66-79, simple_fpn.py:84-99 feed fpn_outputs[-1]).
 *   lfd_maxpool3x3s2_nhwc_f16           extra_type='pooling': nn.MaxPool2d(3, 2, 1) (fpn.py:74) -> [n, (h+1)/2, (w+1)/2, c].
 *   lfd_pack_level_outputs_f32          channels c0 .. c0+count of a level's fp32 output-conv map [n, hw, src_channels]
 *                                       -> rows point_offset .. +hw of the level-concatenated [n, total_points, count]
 *                                       tensor the meta-architecture returns (fcos.py:426-449, lfdv2.py:683-702),
 *                                       times `scale` (the head's per-level Scale), op 1: then expf
 *                                       (fcos_head.py:145-146). */
LFD_API int lfd_upsample_nearest_add_nhwc_f16(void* dst, const void* src, int32_t n, int32_t H, int32_t W, int32_t h,
                                              int32_t w, int32_t c, lfd_stream_t stream);
LFD_API int lfd_relu_inplace_f16(void* x, int64_t count, lfd_stream_t stream);
LFD_API int lfd_maxpool3x3s2_nhwc_f16(const void* in, void* out, int32_t n, int32_t h, int32_t w, int32_t c,
                                      lfd_stream_t stream);
LFD_API int lfd_pack_level_outputs_f32(const float* src, float* dst, int32_t n, int32_t hw, int32_t src_channels, int32_t c0,
                                       int32_t count, int32_t total_points, int32_t point_offset, float scale, int32_t op,
                                       lfd_stream_t stream);

/* Parity instrument (not on the product path): the same MFMA conv kernels with the fp32 accumulators (conv + bias, no
 * activation, no fp16 rounding) written to out_f32 [n, oh, ow, cout].  Used by the engine's G1 mode (SURVEY 8d gate G1:
 * fp32 inter-layer storage, operands split into fp16 hi + lo parts, three launches per conv).  desc->relu / tail_* ignored
 * (tail_cout must be 0). */
LFD_API int lfd_conv2d_nhwc_f16_acc32(const lfd_conv_desc_t* desc, const void* in, float* out_f32, const void* w_packed,
                                      const float* bias, const void* zeros, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 'fp32_storage' precision mode of the eval forward (LFD.precision = 'fp32_storage'): a shipped mode of the product for
 * callers that need the reference's fp32 outputs to <= 1e-4 on the raw logits (north_star: "cls/bbox tensors within
 * 1e-3"); replaces the same reference code as the fp16 entry points above -- nn.Conv2d + eval BatchNorm2d (+ residual)
 * + ReLU of lfd_resnet.py:354-501 / simple_neck.py:67-74, the head convs, GroupNorm + ReLU and Scale of
 * lfd_head.py:97-117,164-185.  Activations are fp32 NHWC; ONE launch per conv: operands are split exactly into fp16
 * hi + 2^-11 lo parts inside the kernel (activations by the tile loader, weights once per plan), three MFMAs per k-step,
 * fp32 accumulation, epilogue out = relu?( scale * (conv + bias (+ residual)) ) in fp32.
 *   in_format < 0: `in` is fp32 NHWC [n,h,w,cin], cin a multiple of 32; ks 1|3, stride 1|2, pad ks/2.
 *   in_format 0|1|2 (NCHW fp32 / NHWC fp16 / NHWC uint8 + simple_normalize): the first stem conv, 3x3 stride 2 on the
 *     3-channel frame (cin = 3); the 27 taps are one 32-wide k chunk gathered from the frame.
 *   w_packed: lfd_p32_conv_packed_weight_halfs() fp16 values, [cout/32 slabs][cin/32 chunks][ks*ks taps][2 k-steps]
 *     [hi | 2^11 lo][64 lanes][8]; lane = 32 * khalf + cout_local, element j = channel 32 chunk + 16 kstep + 8 khalf + j
 *     (first conv: k = (dy*3 + dx)*3 + c, zero above 27); bias [32 * slabs] fp32 (zero padded).
 *   out: channel c of pixel (oy, ox) of image i at out[i * out_image_stride + (oy*OW + ox) * out_pixel_stride + c]
 *     (0 = dense: out_pixel_stride = cout, out_image_stride = OH*OW*cout) -- the head's output convs write straight into
 *     the level-concatenated [N,P,C'] / [N,P,4] tensors (lfd.py:526-542); residual: dense fp32 [n,OH,OW,cout] or NULL;
 *     scale: device pointer to ONE float (lfd_head.py Scale, applied after the bias) or NULL. */
typedef struct lfd_p32_conv_desc {
  int32_t n, h, w, cin, cout, ks, stride, relu;
  int32_t in_format;
  int32_t out_pixel_stride;
  int64_t out_image_stride;
} lfd_p32_conv_desc_t;
LFD_API size_t lfd_p32_conv_packed_weight_halfs(int32_t cin, int32_t cout, int32_t ks);
LFD_API int lfd_p32_conv2d_nhwc_f32(const lfd_p32_conv_desc_t* desc, const void* in, float* out, const void* w_packed,
                                    const float* bias, const float* residual, const float* scale, lfd_stream_t stream);
/* the same conv chained with a 1x1 conv 64 -> 64 in ONE launch: out = relu?( conv1x1( relu?(conv(in) + bias) ) + tail_bias )
 * -- the stem pairs conv3x3 s2 + BN + ReLU -> conv1x1 + BN + ReLU (lfd_resnet.py:356-413); the 64-channel fp32 intermediate
 * stays in LDS.  desc->cout must be 64, the conv 3x3 stride 2 (in_format < 0) or the first stem conv (in_format 0|1|2);
 * tail_w_packed = lfd_p32_conv_packed_weight_halfs(64, 64, 1) halfs in the same layout. */
LFD_API int lfd_p32_conv2d_tail_nhwc_f32(const lfd_p32_conv_desc_t* desc, const void* in, float* out, const void* w_packed,
                                         const float* bias, const void* tail_w_packed, const float* tail_bias, int32_t tail_relu,
                                         lfd_stream_t stream);
/* x [n, hw, c] fp32 <- relu?( GroupNorm(groups)(x) * gamma + beta ) in place (nn.GroupNorm semantics: biased variance
 * over hw x c/groups elements per image and group; statistics summed in fp64 in a fixed order); c % 4 == 0,
 * 256 % (c/4) == 0, (c/groups) % 4 == 0, groups <= 64. */
LFD_API size_t lfd_p32_groupnorm_workspace_bytes(int32_t n, int32_t groups);
LFD_API int lfd_p32_groupnorm_relu_f32(float* x, int32_t n, int64_t hw, int32_t c, int32_t groups, const float* gamma,
                                       const float* beta, float eps, int32_t relu, void* workspace, size_t workspace_bytes,
                                       lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same precision mode on "hi/lo planes" (csrc/planes.hip, csrc/planes_impl.h; lfd_amd/engine_p2.py) -- the fast form of
 * the mode inside north_star's 1e-3 (lfd/model/lfd.py:511-542 to <= 1e-4 on the raw logits).
 * A plane tensor stores an fp32-valued NHWC activation x as two contiguous NHWC fp16 tensors: hi = fp16(x) at the base pointer,
 * lo = fp16(2^11 (x - hi)) `*_plane_halfs` halfs behind it (x = hi + 2^-11 lo to ~22 bits; the bytes of fp32).  Both planes are
 * 16-byte aligned, plane strides multiples of 8 halfs.  Convolutions are weights-stationary LDS-DMA kernels issuing three
 * fp16 MFMAs per k-step (w_hi x_hi | w_hi x_lo + w_lo x_hi), fp32 accumulation, fp32 epilogue, split on the way out.
 * Packed weights of every entry point: [2 = hi | 2^11 lo][fragments of lfd_conv_packed_weight_halfs order], rows zero-padded
 * to a multiple of 32 (engine_p2.pack_planes_weight); biases fp32 padded likewise.
 * RANGE: a plane pair has the PRECISION of ~22 bits but the RANGE of fp16 -- |x| <= 65504; the split does not saturate, a larger
 * value stores hi = +-inf and reads back as NaN (the fp32-tensor form of the mode, lfd_p32_*, has fp32's range and is what to use
 * for a network whose activations exceed it; BatchNorm / GroupNorm-normalised LFD activations stay below ~1e2).  The fixed-point
 * GroupNorm sums (2^-24 units in int64) hold a group's sum of squares up to ~5.5e11.  Where the text below says "statistics in
 * fp64": a thread adds the <= 64 values it copies out of ONE tile in fp32, tiles are combined in fp64, workgroups in fixed point.
 *
 * lfd_pl_stem_pair: frame (in_format as lfd_stem_conv_f16) -> conv3x3 s2 (3 -> C) + BN + ReLU -> conv1x1 (C -> C) + BN + ReLU
 *   -> planes [n, (h+1)/2, (w+1)/2, C], C = 32 | 64 (lfd_resnet.py:356-374, :376-395).  w1: [2][C/32][2][64][8] in
 *   engine.pack_stem_weight order per plane.  uint8 / fp32 frames keep their low parts (simple_normalize in fp32).
 * lfd_pl_stem2x: the whole 'faster' stem (lfd_resnet.py:376-413), frame -> planes [n, ceil(ceil(h/2)/2), ceil(ceil(w/2)/2), 64]
 *   in one launch: the second pair's persistent workgroups compute their input tile from the frame in LDS, the first pair's
 *   output (the largest tensor of the network) never exists in HBM.  w1: conv0 AND its bias, [2][2 slabs][2 k-steps][64 lanes][8]
 *   in the k-slot order of the fused gather (engine_p2.pack_planes_stem2x_weight: per frame row the 10 halfs [pixel left of the
 *   patch, e0..e8], e = 3 dx + c, as aligned dwords; the bias on the slot of a constant one); w2: the first 1x1 with K in the
 *   order the conv0 accumulators hold the channels (engine_p2.pack_planes_stem2x_tail_weight), b2 its bias [64]; w3 / w4:
 *   lfd_pl_conv2d order (3x3 s2 64 -> 64, 1x1 64 -> 64), b3 / b4 [128] zero padded.
 *   fp16 NHWC frames with w % 8 == 0, uint8 NHWC frames with w % 16 == 0 and fp32 NCHW frames with w % 4 == 0, on a 16-byte aligned
 *   base, run k_pl_stem2xs (round 6: a row stream down strips of 16 output columns, producer waves and consumer waves, the frame
 *   patch by 16-byte LDS-DMA; uint8 values through a 256-entry table of split simple_normalize values, fp32 values split at the
 *   gather; planes_stem2xs.hip; the tuning knob PL_STEM set to 0 selects the tile kernel for them too); other widths / alignments run
 *   the tile kernel k_pl_stem2x with loads.  Both kernels take the same packed filters; their results differ by the order of fp32 sums (<= 2e-6 at 8 x 1080p).
 * lfd_pl_conv2d: planes [n,h,w,cin] -> conv ks x ks / stride (+ bias, ReLU) with ONE of
 *     tail_cout > 0   : a chained 1x1 cout -> cout (+ tail_bias, tail_relu) in the same launch (the intermediate stays in LDS):
 *                       the second stem pair (3x3 s2 -> 1x1, lfd_resnet.py:396-413), the neck conv -> first tower conv;
 *     ds_w_packed     : second output ds_out = conv1x1 stride 2 (in) + ds_bias, no ReLU -- the identity branch of a stage's
 *                       first block (lfd_resnet.py:458-468) from the centre tap of a 3x3 stride-2 conv;
 *     residual        : y += residual (planes, same shape as out) before the ReLU (lfd_resnet.py:151-152);
 *   out_mode 0: planes out.  1: planes out + GroupNorm sums: gn_sums[((replica * n + image) * cout/8 + group) * 2 + {0,1}] +=
 *   {sum, sum of squares} of the stored values of that 8-channel group as 64-bit two's-complement fixed point (2^-24
 *   units), replica = workgroup % LFD_PL_GN_REPLICAS (LFD_PL_GN_REPLICAS * n * cout/8 * 2 words; consumers add the replicas)
 *   -- order-independent integer atomics, so the statistics are bit-reproducible; the caller zeroes gn_sums per forward.  2: fp32
 *   outputs without planes: channel c < f_c0 -> f_out0[image * f_image_stride0 + pixel * f_c0 + c], f_c0 <= c < f_c0 + f_c1
 *   -> f_out1[image * f_image_stride1 + pixel * f_c1 + c - f_c0] * (*scale1) (lfd_head.py:176-183: cls / reg convs + Scale
 *   straight into the level-concatenated [N,P,C'] / [N,P,4] tensors, lfd.py:526-542).
 *   gn_in_sums != NULL (1x1 convs on 128 channels): the input planes are the PRE-normalisation output of the conv that
 *   accumulated gn_in_sums; the landed tile is normalised (GroupNorm(16, 128), gn_in_gamma / gn_in_beta / desc->gn_in_eps,
 *   mean / rstd in fp64 from the sums) + ReLU'd in LDS before the contraction -- the tower's conv -> GroupNorm -> ReLU
 *   (lfd_head.py:97-117) without a pass over the tensor.
 *   Supported shapes: the layers of every named configuration (see the dispatch in csrc/planes.hip); LFD_ERR_UNSUPPORTED
 *   otherwise -- the host falls back to lfd_p32_*.  zeros: the 4 KB line of lfd_conv2d_nhwc_f16.
 * lfd_pl_conv2d_levels: the same 1x1 conv (desc: n, cin, cout, relu, tail_cout / tail_relu, out_mode, f_c0 / f_c1,
 *   f_image_stride0 / 1, gn_in_eps; ks = stride = 1; desc->h / w / *_plane_halfs unused) over num_levels <= LFD_MAX_LEVELS
 *   feature maps in ONE launch -- simple_neck.py:67-74 + lfd_head.py:164-185 apply the same conv stack to every pyramid
 *   level, each level with its own filters; levels[i] carries what lfd_pl_conv2d takes per call.  Results are identical to
 *   num_levels lfd_pl_conv2d calls: same arithmetic per tile; the GroupNorm sums are order-independent integers (which
 *   REPLICA a workgroup adds to depends on the grid -- the statistic is the sum over the replicas).
 * lfd_pl_groupnorm_relu: x (planes [n, hw, c]) <- relu?(GroupNorm(c/8 groups)(x) * gamma + beta) in place, mean / rstd in
 *   fp64 from gn_sums (lfd_head.py:97-117 conv -> GroupNorm -> ReLU). */
#define LFD_PL_GN_REPLICAS 8   /* gn_sums holds this many replicas of [n][cout/8][2]: a producer spreads its atomics, consumers add */
typedef struct lfd_pl_conv_desc {
  int32_t n, h, w, cin, cout, ks, stride, relu;
  int32_t tail_cout, tail_relu;
  int32_t out_mode;
  int32_t f_c0, f_c1;
  float gn_in_eps;
  int64_t in_plane_halfs, out_plane_halfs, res_plane_halfs, ds_plane_halfs;
  int64_t f_image_stride0, f_image_stride1;
} lfd_pl_conv_desc_t;
LFD_API int lfd_pl_stem_pair(const void* in, int32_t in_format, int32_t n, int32_t h, int32_t w, int32_t channels,
                             const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                             int64_t out_plane_halfs, lfd_stream_t stream);
LFD_API int lfd_pl_stem2x(const void* in, int32_t in_format, int32_t n, int32_t h, int32_t w, const void* w1_packed,
                          const void* w2_packed, const float* b2, const void* w3_packed, const float* b3,
                          const void* w4_packed, const float* b4, void* out, int64_t out_plane_halfs, const void* zeros,
                          lfd_stream_t stream);
LFD_API int lfd_pl_conv2d(const lfd_pl_conv_desc_t* desc, const void* in, void* out, const void* w_packed, const float* bias,
                          const void* residual, const void* tail_w_packed, const float* tail_bias, const void* ds_w_packed,
                          const float* ds_bias, void* ds_out, void* gn_sums, float* f_out0, float* f_out1, const float* scale1,
                          const void* gn_in_sums, const float* gn_in_gamma, const float* gn_in_beta, const void* zeros,
                          lfd_stream_t stream);
typedef struct lfd_pl_level {
  const void* in;                 /* planes [n, h, w, cin] */
  void* out;                      /* planes [n, h, w, cout] (out_mode 0 | 1) */
  const void* w_packed;
  const float* bias;
  const void* tail_w_packed;      /* desc->tail_cout > 0 */
  const float* tail_bias;
  void* gn_sums;                  /* out_mode 1 */
  const void* gn_in_sums;         /* all levels or none */
  const float* gn_in_gamma;
  const float* gn_in_beta;
  float* f_out0;                  /* out_mode 2 */
  float* f_out1;
  const float* scale1;
  int32_t h, w;
  int64_t in_plane_halfs, out_plane_halfs;
} lfd_pl_level_t;
LFD_API int lfd_pl_conv2d_levels(const lfd_pl_conv_desc_t* desc, const lfd_pl_level_t* levels, int32_t num_levels,
                                 const void* zeros, lfd_stream_t stream);
LFD_API int lfd_pl_groupnorm_relu(void* x, int64_t plane_halfs, int32_t n, int64_t hw, int32_t c, const void* gn_sums,
                                  const float* gamma, const float* beta, float eps, int32_t relu, lfd_stream_t stream);

/* lfd_pl_head_levels (round 6, csrc/planes_head.hip): the neck + head 1x1 convs of the planes mode (simple_neck.py:67-74,
 * lfd_head.py:88-139,164-185, lfd.py:526-542) over FLAT pixel lists -- a pyramid level of an image is `pixels` = h * w pixels,
 * tiles of 64, all levels of a launch in one persistent grid, every level with its own filters.  The pre-GroupNorm outputs of
 * the tower convs are private to these launches and stored as plain fp32 [n][pixels][128] (the bytes of a plane pair).
 *   mode 0: in = planes [n][pixels][cin] (cin 64 | 128, the backbone tap) -> conv w0 (cin -> 128) + b0 (+ ReLU if relu0) ->
 *           conv w1 (128 -> 128) + b1 -> out fp32, gn_sums += {sum, sum of squares} of out per (image, 8-channel group)
 *           in the fixed-point replica layout of lfd_pl_conv2d (caller zeroes gn_sums per forward)
 *   mode 1: in = fp32 [n][pixels][128] + its producer's gn_in_sums / gamma / beta (GroupNorm(16, 128), eps) -> ReLU ->
 *           conv w0 (128 -> 128) + b0 -> out fp32, gn_sums as above
 *   mode 2: in as mode 1 -> conv w0 (128 -> f_c0 + f_c1 <= 64 channels, packed to 32 | 64 rows) + b0 -> fp32 outputs as
 *           lfd_pl_conv2d out_mode 2 (f_out0 / f_out1 at the level's point offset, image strides f_image_stride0 / 1, scale1)
 *   mode 3: in = planes [n][pixels][128] (the stored output of a neck conv: heads with separate cls / reg towers) -> conv w0
 *           (128 -> 128) + b0 -> out fp32, gn_sums as mode 0
 * Packed weights: engine_p2.pack_planes_weight order ([2 = hi | 2^11 lo][cout / 32][cin / 16][64 lanes] x 8 halfs). */
typedef struct lfd_pl_head_desc {
  int32_t mode, n, cin, relu0;
  int32_t f_c0, f_c1;
  float gn_in_eps;
  int32_t pad_;
  int64_t f_image_stride0, f_image_stride1;
} lfd_pl_head_desc_t;
typedef struct lfd_pl_head_level {
  const void* in;
  void* out;                      /* fp32 [n][pixels][128] (modes 0, 1) */
  const void* w0;
  const float* b0;
  const void* w1;                 /* mode 0 */
  const float* b1;
  void* gn_sums;                  /* modes 0, 1 */
  const void* gn_in_sums;         /* modes 1, 2 */
  const float* gn_in_gamma;
  const float* gn_in_beta;
  float* f_out0;                  /* mode 2 */
  float* f_out1;
  const float* scale1;
  int64_t in_plane_halfs;         /* mode 0 */
  int32_t pixels, pad_;
} lfd_pl_head_level_t;
LFD_API int lfd_pl_head_levels(const lfd_pl_head_desc_t* desc, const lfd_pl_head_level_t* levels, int32_t num_levels,
                               const void* zeros, lfd_stream_t stream);

/* First stem unit: conv3x3 s2 (3 -> C) + BN + ReLU chained with conv1x1 (C -> C) + BN + ReLU
 * (lfd_resnet.py:356-374 'fast' stem; first half of the 'faster' stem :376-395).
 * in_format: 0 = NCHW fp32 (the tensor LFD.forward receives, lfd.py:511), 1 = NHWC fp16,
 * 2 = NHWC uint8 with simple_normalize (x/255-0.5)/0.5 fused (augmentation_pipeline.py:31-36).
 * w2_packed == NULL: no chained 1x1.  out: NHWC fp16 [n, (h+1)/2, (w+1)/2, channels]. */
LFD_API int lfd_stem_conv_f16(const void* in, int32_t in_format, int32_t n, int32_t h, int32_t w,
                              int32_t channels, const void* w1_packed, const float* b1,
                              const void* w2_packed, const float* b2, void* out, lfd_stream_t stream);

/* The whole 'faster' stem (lfd_resnet.py:376-413) in one kernel: conv3x3 s2 (3->C), conv1x1, conv3x3 s2
 * (C->C), conv1x1, each + BN + ReLU.  The stride-2 intermediate (the largest activation of the
 * network) is produced tile-by-tile into LDS and never written to HBM (csrc/stem_fused.hip).
 * out: NHWC fp16 [n, ceil(ceil(h/2)/2), ceil(ceil(w/2)/2), channels]; channels in {32, 64}. */
LFD_API int lfd_stem_faster_fused_f16(const void* in, int32_t in_format, int32_t n, int32_t h, int32_t w,
                                      int32_t channels, const void* w1_packed, const float* b1,
                                      const void* w2_packed, const float* b2, const void* w3_packed,
                                      const float* b3, const void* w4_packed, const float* b4, void* out,
                                      lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode conv stack.  The reference trains through PyTorch autograd over
 * nn.Conv2d(bias=False) + nn.BatchNorm2d + ReLU (+ residual add) units (lfd_resnet.py:96-154 blocks,
 * :354-439 stem, :458-468 downsample; `loss.backward()` at optimizer_hook.py:28).  Here activations and
 * activation gradients are NHWC fp16 (gradients multiplied by a power-of-two loss scale; `inv_scale`
 * = 1/scale is applied where fp32 parameter gradients are written), statistics and parameter
 * gradients fp32.  A conv unit's forward is lfd_conv2d_nhwc_f16 (bias NULL-equivalent: zero bias, no
 * ReLU) -> lfd_bn_train_stats_f16 -> lfd_bn_train_apply_f16; its backward lfd_bn_train_bwd_f16 ->
 * lfd_conv_wgrad_nhwc_f16 + lfd_conv2d_nhwc_f16 on the transposed, tap-flipped weights (stride 2:
 * after lfd_zero_insert2_nhwc_f16).  channels: power of two in [8, 256].  Every reduction is per-block
 * partials + a fixed-order final stage (deterministic).  `workspace`: lfd_train_workspace_bytes().
 * Parameter-gradient outputs (dgamma, dbeta, dw) are `+=` when `accumulate` != 0 -- they can be the
 * zeroed .grad buffers themselves (autograd's AccumulateGrad semantics without a temporary + add per tensor).
 */
LFD_API size_t lfd_train_workspace_bytes(void);
/* nn.Conv2d.weight [cout, cin, ks, ks] fp32 -> the fp16 MFMA fragment order lfd_conv2d_nhwc_f16 consumes
 * (lfd_conv_packed_weight_halfs).  mode 0: the forward conv, rows >= rows_valid zero-filled (cout must already be the
 * padded row count, multiple of 32); mode 1: the data-gradient conv (channel roles swapped, taps flipped). */
LFD_API int lfd_pack_conv_weight_train_f16(const float* weight_oihw, int32_t cout, int32_t cin, int32_t ks, int32_t mode,
                                   int32_t rows_valid, void* packed, lfd_stream_t stream);
/* The same for many convs in one launch (the weights change every iteration: all forward packs at the start of the
 * forward, all data-gradient packs at the start of the backward).  `jobs_device`: table in device memory, sorted by
 * first_vec (= running sum of the jobs' output vectors of 8 halfs); fields as in lfd_pack_conv_weight_train_f16. */
typedef struct lfd_pack_job {
  const float* w;
  void* out;
  int32_t cout, cin, ks, mode, rows_valid, first_vec;
} lfd_pack_job_t;
LFD_API int lfd_pack_conv_weights_train_f16(const lfd_pack_job_t* jobs_device, int32_t njobs, int32_t total_vecs,
                                    lfd_stream_t stream);
/* batch statistics of y [pixels, channels]: stats[0..C) = mean, stats[C..2C) = 1/sqrt(biased var + eps)
 * (F.batch_norm training=True); running_mean / running_var (nullable) are updated with `momentum` and
 * the unbiased variance like nn.BatchNorm2d. */
LFD_API int lfd_bn_train_stats_f16(const void* y, int64_t pixels, int32_t channels, float eps, float momentum,
                           float* running_mean, float* running_var, void* workspace, size_t workspace_bytes,
                           float* stats, lfd_stream_t stream);
/* The conv in front of a train-mode BatchNorm and the batch statistics of its output in one pass (replaces
 * lfd_conv2d_nhwc_f16 -> lfd_bn_train_stats_f16 for a unit nn.Conv2d(bias=False) -> nn.BatchNorm2d, lfd_resnet.py:96-154,
 * :354-439, :458-468 in train mode): the conv kernel sums the fp16 values it stores and their squares per channel on the
 * way out, one row of partials per workgroup in `workspace`; the fp64 final pass is lfd_bn_train_stats_f16's.  `d->relu`
 * and `d->tail_cout` must be 0.  Shapes without such a kernel run the two calls it replaces.  `out`, `stats`,
 * running_mean / running_var as in the two calls. */
LFD_API int lfd_conv2d_bn_stats_nhwc_f16(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed,
                                 const float* bias, const void* zeros, float eps, float momentum, float* running_mean,
                                 float* running_var, void* workspace, size_t workspace_bytes, float* stats,
                                 lfd_stream_t stream);
/* A 1x1 stride-1 conv unit whose INPUT is the activation of a train-mode BatchNorm + ReLU unit WITHOUT a residual
 * (lfd_resnet.py:376-413: the second conv of each stem pair): `y_in` is that unit's PRE-normalisation conv output and
 * (in_stats, in_gamma, in_beta) its batch statistics (lfd_bn_train_stats_f16 layout) and affine; the kernel normalises +
 * ReLUs its activation fragments on their way into the MFMA with lfd_bn_train_apply_f16's arithmetic (bit-identical to
 * applying first), so the producer's z tensor is never written or read.  Everything else as lfd_conv2d_bn_stats_nhwc_f16;
 * d->cin must be 64, d->ks = d->stride = 1.  The matching weight gradient: lfd_conv1x1_wgrad_partials_of_bn_relu_f16. */
LFD_API int lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16(const lfd_conv_desc_t* d, const void* y_in, const float* in_stats,
                                             const float* in_gamma, const float* in_beta, void* out, const void* w_packed,
                                             const float* bias, const void* zeros, float eps, float momentum,
                                             float* running_mean, float* running_var, void* workspace,
                                             size_t workspace_bytes, float* stats, lfd_stream_t stream);
/* z = relu?( gamma * (y - mean) * rstd + beta (+ residual) ) */
LFD_API int lfd_bn_train_apply_f16(const void* y, int64_t pixels, int32_t channels, const float* stats, const float* gamma,
                           const float* beta, const void* residual, int32_t relu, void* z, lfd_stream_t stream);
/* backward of the above: g = dz * [ReLU passed] (relu == 0: no ReLU; z given: mask = [z > 0]; z NULL: the mask is
 * recomputed from y as [gamma * xhat + beta > 0], only valid without a residual input), dgamma = inv_scale * sum g * xhat,
 * dbeta = inv_scale * sum g, dy = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)); g_out (nullable)
 * receives g, the gradient of the residual branch. */
LFD_API int lfd_bn_train_bwd_f16(const void* dz, const void* y, const void* z, int32_t relu, int64_t pixels, int32_t channels,
                         const float* stats, const float* gamma, const float* beta, float inv_scale,
                         int32_t accumulate, void* workspace,
                         size_t workspace_bytes, float* dgamma, float* dbeta, void* dy, void* g_out,
                         lfd_stream_t stream);
/* GroupNorm of the head towers in training (nn.GroupNorm(groups, channels) + ReLU, lfd_head.py:97-105), groups of
 * exactly 8 channels (128 / 16): stats[img][0][g] = mean, [1][g] = rstd over hw x 8 elements; apply; backward
 * (dgamma / dbeta are `+=` when accumulate != 0: the towers are shared by all pyramid levels, lfd_head.py:67-82). */
LFD_API int lfd_gn_train_stats_f16(const void* y, int32_t n, int64_t hw, int32_t channels, int32_t groups, float eps,
                           void* workspace, size_t workspace_bytes, float* stats, lfd_stream_t stream);
LFD_API int lfd_gn_train_apply_f16(const void* y, int32_t n, int64_t hw, int32_t channels, int32_t groups, const float* stats,
                           const float* gamma, const float* beta, int32_t relu, void* z, lfd_stream_t stream);
/* lfd_gn_train_stats_f16 + lfd_gn_train_apply_f16 in two launches instead of three: the apply pass adds the per-block partial
 * sums of its image itself (same fp64 arithmetic; `stats` is an OUTPUT here, written for the backward pass). */
LFD_API int lfd_gn_train_stats_apply_f16(const void* y, int32_t n, int64_t hw, int32_t channels, int32_t groups, float eps,
                                 const float* gamma, const float* beta, int32_t relu, void* workspace, size_t workspace_bytes,
                                 float* stats, void* z, lfd_stream_t stream);
LFD_API int lfd_gn_train_bwd_f16(const void* dz, const void* y, const void* z, int32_t n, int64_t hw, int32_t channels,
                         int32_t groups, const float* stats, const float* gamma, float inv_scale, int32_t accumulate,
                         void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, void* dy,
                         lfd_stream_t stream);
/* out[n, 2i, 2j, :] = in[n, i, j, :], zero elsewhere; ho in {2*hi-1, 2*hi}, wo likewise */
LFD_API int lfd_zero_insert2_nhwc_f16(const void* in, int32_t n, int32_t hi, int32_t wi, int32_t channels, int32_t ho,
                              int32_t wo, void* out, lfd_stream_t stream);
/* Data gradient of a 3x3 stride-2 pad-1 convolution, 64 -> 64 channels, WITHOUT the zero-inserted tensor:
 *   dx [n, h, w, 64] = conv3x3_s1(zero_insert2(dy), w)  (+ residual)   computed per output parity (1 / 2 / 2 / 4 taps),
 * dy [n, (h-1)/2+1, (w-1)/2+1, 64]; w_packed = the data-gradient pack of the conv's weight (lfd_pack_conv_weight_train_f16 with
 * mode 1: roles swapped, taps flipped -- the fragments lfd_conv2d_nhwc_f16 would consume on the zero-inserted tensor);
 * residual (nullable): the gradient already collected for the same activation, added before the fp16 rounding.  Bit-identical
 * to lfd_zero_insert2_nhwc_f16 + lfd_conv2d_nhwc_f16 (csrc/dgrad_s2.hip). */
LFD_API int lfd_conv3x3s2_dgrad_nhwc_f16(int32_t n, int32_t h, int32_t w, const void* dy, void* dx, const void* w_packed,
                                         const void* residual, lfd_stream_t stream);
/* dW [cout, cin, ks, ks] fp32 (OIHW, the nn.Conv2d.weight layout) = inv_scale * sum over pixels of
 * dy (x) x for a conv with pad ks/2; x [n,h,w,cin], dy [n,ho,wo,cout]; cin, cout multiples of 8, <= 128. */
LFD_API int lfd_conv_wgrad_nhwc_f16(const void* x, const void* dy, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout,
                            int32_t ks, int32_t stride, float inv_scale, int32_t accumulate, void* workspace,
                            size_t workspace_bytes, float* dw, lfd_stream_t stream);
/* The LEVEL-CONCATENATED form of the head's training passes (round 4).  The reference's head applies the SAME towers to every
 * pyramid level, one tensor per level (lfd_head.py:164-185); the training schedule keeps the levels of an image back to back in
 * one tensor [n, P, c] (P = points of all levels, the layout LFD.forward returns, lfd.py:526-542), so that a shared 1x1 conv, its
 * weight gradient and its data gradient are ONE launch over all levels instead of one per level:
 *   lfd_bn_train_apply_into_f16   a level's neck unit writes its normalised output into rows [point0, point0 + hw) of every image;
 *   lfd_bn_train_bwd_from_f16     ... and reads its output gradient from there (no residual input; ReLU mask recomputed from y);
 *   lfd_gn_train_*_seg_f16        GroupNorm with statistics per (image, segment): nseg segments of seg_hw[] pixels per image,
 *                                 stats [n * nseg][2][groups] (virtual image = img * nseg + segment);
 *   lfd_head_out_*_concat_f16     the glue kernels below with y / dy = the concatenated conv output [n, P, 64]. */
LFD_API int lfd_bn_train_apply_into_f16(const void* y, int32_t n, int64_t hw, int32_t channels, const float* stats, const float* gamma,
                                const float* beta, int32_t relu, void* z_concat, int64_t points_total, int64_t point0,
                                lfd_stream_t stream);
/* The forward counterpart: the convs of several independent units leave their statistics rows behind
 * (lfd_conv2d_bn_partials_nhwc_f16: lfd_conv2d_bn_stats_nhwc_f16 without its final pass, rows in the caller's buffer of at
 * least 512 x 2 x cout floats, *nrows = how many), then ONE call finishes all of them: the per-channel finals (mean, rstd,
 * running statistics) of every level in one launch, the apply passes into the level-concatenated tensor in another.  Values
 * bit-identical to lfd_conv2d_bn_stats_nhwc_f16 + lfd_bn_train_apply_into_f16 per level. */
LFD_API int lfd_conv2d_bn_partials_nhwc_f16(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed, const float* bias,
                                    const void* zeros, float* rows, size_t rows_bytes, int32_t* nrows, lfd_stream_t stream);
typedef struct {
  const void* y;           /* [n, hw, channels] fp16 */
  const float* rows;       /* statistics rows of the conv that produced y */
  float* running_mean;     /* nullable (both or none) */
  float* running_var;
  float* stats;            /* out: [2][channels] */
  const float* gamma;
  const float* beta;
  int64_t hw, point0;
  int32_t channels, nrows;
  float eps, momentum;
} lfd_bn_fwd_level_t;
LFD_API int lfd_bn_train_finish_into_levels_f16(const lfd_bn_fwd_level_t* levels, int32_t nlevels, int32_t n, int32_t relu,
                                        void* z_concat, int64_t points_total, lfd_stream_t stream);
/* lfd_bn_train_bwd_from_f16 for SEVERAL levels (units of the same batch, independent of each other: the neck units of the pyramid
 * levels) in three launches instead of three per level; values bit-identical to the per-level calls.  `levels`: host array. */
typedef struct {
  const void* y;          /* the level's pre-normalisation output [n, hw, channels] fp16 */
  void* dy;               /* out: [n, hw, channels] fp16 */
  const float* stats;     /* [2][channels] */
  const float* gamma;
  const float* beta;
  float* dgamma;          /* [channels], (+)= */
  float* dbeta;
  int64_t hw, point0;     /* pixels per image; first point of the level inside an image of dz_concat */
  int32_t channels;
  int32_t reserved_;
} lfd_bn_bwd_level_t;
LFD_API int lfd_bn_train_bwd_from_levels_f16(const void* dz_concat, int64_t points_total, const lfd_bn_bwd_level_t* levels,
                                     int32_t nlevels, int32_t relu, int32_t n, float inv_scale, int32_t accumulate,
                                     void* workspace, size_t workspace_bytes, lfd_stream_t stream);
LFD_API int lfd_bn_train_bwd_from_f16(const void* dz_concat, int64_t points_total, int64_t point0, const void* y, int32_t relu, int32_t n,
                              int64_t hw, int32_t channels, const float* stats, const float* gamma, const float* beta,
                              float inv_scale, int32_t accumulate, void* workspace, size_t workspace_bytes, float* dgamma,
                              float* dbeta, void* dy, lfd_stream_t stream);
LFD_API int lfd_gn_train_stats_apply_seg_f16(const void* y, int32_t n, int32_t nseg, const int64_t* seg_hw, int32_t channels,
                                     int32_t groups, float eps, const float* gamma, const float* beta, int32_t relu,
                                     void* workspace, size_t workspace_bytes, float* stats, void* z, lfd_stream_t stream);
LFD_API int lfd_gn_train_bwd_seg_f16(const void* dz, const void* y, const void* z, int32_t n, int32_t nseg, const int64_t* seg_hw,
                             int32_t channels, int32_t groups, const float* stats, const float* gamma, float inv_scale,
                             int32_t accumulate, void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, void* dy,
                             lfd_stream_t stream);
/* The same weight gradient in two stages for a schedule that defers the final sums (round 4): every conv's k_wgrad writes
 * its per-workgroup partial sums into ITS OWN buffer (`partials`: lfd_conv_wgrad_partial_rows() x blocks x ks*ks x 64 x 64
 * floats, blocks = ceil(cout/64) * ceil(cin/64)), and ONE lfd_wgrad_final_batched_f32 launch after the last of them turns all
 * buffers into dW tensors -- the 49 dependent final launches of a WIDERFACE_LFD_S iteration become one, and the partial
 * launches may run on another stream than the data-gradient chain.  Job table in device memory: head jobs (first_block >= 0,
 * = running sum of the head jobs' nblk * taps * 32 blocks) own output blocks; `next` chains further partial sets of the same dW
 * (a conv shared by the pyramid levels, lfd_head.py:67-82; chained jobs have first_block = -1 and the head job's nblk / taps),
 * summed after the head's rows in chain order, fp64, one rounding.  Rows [co_lo, co_hi) of the conv go to dw rows
 * [0, co_hi - co_lo): the padded per-level output conv writes its classification and regression rows into two tensors
 * through two head jobs over the same partials.  dW (+)= inv_scale * sum. */
typedef struct lfd_wgrad_job {
  const float* partials;
  float* dw;
  int32_t nwg, nblk, cin, cout, taps;
  int32_t co_lo, co_hi;
  int32_t first_block, next, accumulate;
  float inv_scale;
} lfd_wgrad_job_t;
LFD_API int32_t lfd_conv_wgrad_partial_rows(int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ks, int32_t stride);
LFD_API int lfd_conv_wgrad_partials_nhwc_f16(const void* x, const void* dy, int32_t n, int32_t h, int32_t w, int32_t cin,
                                             int32_t cout, int32_t ks, int32_t stride, float* partials, size_t partials_bytes,
                                             lfd_stream_t stream);
/* lfd_conv_wgrad_partials_nhwc_f16 for a 1x1 stride-1 conv whose operand x = relu(BatchNorm(y_in)) was never stored (see
 * lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16): the tile loader re-forms it from y_in.  Same partial rows / final pass. */
LFD_API int lfd_conv1x1_wgrad_partials_of_bn_relu_f16(const void* y_in, const float* in_stats, const float* in_gamma,
                                              const float* in_beta, const void* dy, int32_t n, int32_t h, int32_t w,
                                              int32_t cin, int32_t cout, float* partials, size_t partials_bytes,
                                              lfd_stream_t stream);
LFD_API int lfd_wgrad_final_batched_f32(const lfd_wgrad_job_t* jobs_device, int32_t njobs, int32_t total_blocks,
                                        lfd_stream_t stream);
/* dst[i] (+)= sum over r < nrows of src[r * row_stride + i], i < count, fp64 in row order; one block per job: the private
 * per-level copies of small shared parameter gradients (GroupNorm weight / bias of shared towers, output-conv biases)
 * after the pyramid levels ran on their own streams. */
typedef struct lfd_rowsum_job {
  const float* src;
  float* dst;
  int32_t nrows, row_stride, count, accumulate;
} lfd_rowsum_job_t;
LFD_API int lfd_rows_sum_batched_f32(const lfd_rowsum_job_t* jobs_device, int32_t njobs, lfd_stream_t stream);
/* The glue around the head's per-level OUTPUT convs in a training iteration (csrc/head_out.hip).  The reference's head ends,
 * per level, in a classification and a regression conv with fp32 outputs, the regression one through a learnable per-level
 * Scale (lfd_head.py:157-185); LFD.forward concatenates the levels along the point axis (lfd.py:526-542).  The training
 * engine runs a level's convs as ONE 1x1 conv padded to 64 output rows; `y` [n, hw, 64] fp16 is its output, a segment is
 * the row range [row0, row0 + channels) of one of the convs.
 *   lfd_head_out_split_f16: out[img, point0 + p, j] = float(y[img, p, row0 + j]) (* *scale) for every segment -- `out` is the
 *     level-concatenated [n, points_total, channels] fp32 tensor.
 *   lfd_head_out_grad_f16: from grad (same layout as out) writes dy [n, hw, 64] fp16 = grad (* *scale) * loss_scale (other rows
 *     zero) and ACCUMULATES dbias[j] += sum grad (* *scale), *dscale += sum grad * float(y) (both nullable) through per-block
 *     partials in `workspace` (>= 512 KB) and one fixed-order fp64 final launch: what autograd computes for
 *     conv bias / Scale, as 2 launches instead of ~20 per level. */
typedef struct lfd_head_out_seg {
  float* out;          /* split: destination */
  const float* grad;   /* grad: source */
  float* dbias;        /* grad: [channels], += (nullable) */
  const float* scale;  /* device scalar (nullable: no Scale) */
  float* dscale;       /* grad: device scalar, += (nullable) */
  int32_t channels, row0;
} lfd_head_out_seg_t;
LFD_API int lfd_head_out_split_f16(const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                           const lfd_head_out_seg_t* segs, int32_t nsegs, lfd_stream_t stream);
LFD_API int lfd_head_out_grad_f16(const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                          const lfd_head_out_seg_t* segs, int32_t nsegs, float loss_scale, void* dy, void* workspace,
                          size_t workspace_bytes, lfd_stream_t stream);
LFD_API int lfd_head_out_split_concat_f16(const void* y_concat, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                                  const lfd_head_out_seg_t* segs, int32_t nsegs, lfd_stream_t stream);
LFD_API int lfd_head_out_grad_concat_f16(const void* y_concat, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                                 const lfd_head_out_seg_t* segs, int32_t nsegs, float loss_scale, void* dy_concat, void* workspace,
                                 size_t workspace_bytes, lfd_stream_t stream);

/* ... and ALL pyramid levels of one (shared) output conv in one launch each (round 4): the per-level calls of the two `_concat`
 * entry points above, with the levels as blockIdx.y -- every level keeps its own block count and the final stage adds the levels
 * in level order, so outputs, dy, dbias and dscale are those of the per-level call sequence, bit for bit.  `levels`: host array. */
typedef struct {
  int32_t hw;                      /* pixels per image of the level */
  int32_t nsegs;
  int64_t point0;                  /* first point of the level inside an image of the concatenated tensors */
  lfd_head_out_seg_t segs[2];
} lfd_head_out_level_t;
LFD_API int lfd_head_out_split_levels_f16(const void* y_concat, int32_t n, int64_t points_total, const lfd_head_out_level_t* levels,
                                  int32_t nlevels, lfd_stream_t stream);
LFD_API int lfd_head_out_grad_levels_f16(const void* y_concat, int32_t n, int64_t points_total, const lfd_head_out_level_t* levels,
                                 int32_t nlevels, float loss_scale, void* dy_concat, void* workspace, size_t workspace_bytes,
                                 lfd_stream_t stream);

/* first stem conv (3 -> channels, 3x3 stride 2 pad 1, lfd_resnet.py:358,:378) on the NCHW fp32 image batch:
 * forward -> y NHWC fp16 (pre-norm), and its weight gradient (OIHW fp32); channels in {32, 64} */
LFD_API int lfd_stem_conv0_train_fwd(const float* x_nchw, int32_t n, int32_t h, int32_t w, int32_t channels,
                             const float* weight_oihw, void* y, lfd_stream_t stream);
/* the same conv with the batch statistics of its train-mode BatchNorm (lfd_resnet.py:358-359) taken from the values on
 * their way to memory (64 channels; 32 channels: conv, then lfd_bn_train_stats_f16); stats / running_* / workspace as in
 * lfd_bn_train_stats_f16 */
LFD_API int lfd_stem_conv0_train_fwd_bn_stats(const float* x_nchw, int32_t n, int32_t h, int32_t w, int32_t channels,
                                      const float* weight_oihw, void* y, float eps, float momentum, float* running_mean,
                                      float* running_var, void* workspace, size_t workspace_bytes, float* stats,
                                      lfd_stream_t stream);
LFD_API int lfd_stem_conv0_wgrad(const float* x_nchw, const void* dy, int32_t n, int32_t h, int32_t w, int32_t channels,
                         float inv_scale, int32_t accumulate, void* workspace, size_t workspace_bytes, float* dw, lfd_stream_t stream);
/* The backward of the FIRST conv unit (Conv2d(3, 64, 3, 2, 1, bias=False) -> train-mode BatchNorm2d -> ReLU, lfd_resnet.py:358-366)
 * without materialising dL/dy: BatchNorm's sums (dgamma, dbeta +=) and the conv's weight gradient computed straight from dz = dL/dz
 * of the unit's output and its stored pre-norm output y -- the unit has no data gradient, so its dy has no other consumer.  The
 * values are those of lfd_bn_train_bwd_f16 (ReLU mask recomputed from y) followed by lfd_stem_conv0_wgrad, bit for bit. */
LFD_API int lfd_stem_conv0_bn_bwd_wgrad(const float* x_nchw, const void* dz, const void* y, int32_t n, int32_t h, int32_t w,
                                int32_t channels, const float* stats, const float* gamma, const float* beta, float inv_scale,
                                int32_t accumulate, void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, float* dw,
                                lfd_stream_t stream);

/* BatchNorm's backward sums in the epilogue of the data-gradient conv that produces dz (round 4; the stem pairs,
 * lfd_resnet.py:376-413): a unit Conv -> train-mode BatchNorm2d -> ReLU (no residual) whose activation feeds exactly one 1x1
 * stride-1 conv gets its dz = dL/d(activation) from that conv's data gradient -- a 1x1 conv over dy with the transposed
 * weights.  lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16 runs that conv (desc: cin = cout = 64, ks = stride = 1, relu = 0; bias:
 * 64 zeros) and, while copying dz out, adds sum g and sum g * xhat per channel (g = dz * [gamma * xhat + beta > 0], the
 * arithmetic of lfd_bn_train_bwd_f16's first pass on the fp16 values it stores) into one row per workgroup in `workspace`;
 * *sum_rows (host) receives the row count.  The unit's backward then skips its own pass over (dz, y):
 *   lfd_bn_train_bwd_rows_f16           the final + apply stages of lfd_bn_train_bwd_f16 (ReLU mask recomputed from y, no g output)
 *   lfd_stem_conv0_bn_bwd_wgrad_rows    lfd_stem_conv0_bn_bwd_wgrad without its sums pass (sum_rows = 0: with it)
 * with the SAME workspace and no other workspace user in between.  Values equal the unfused calls up to the order of the
 * fp32 partial sums. */
LFD_API int lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16(const lfd_conv_desc_t* d, const void* dy, void* dz, const void* w_packed,
                                           const float* bias, const void* zeros, const void* y_unit, const float* unit_stats,
                                           const float* unit_gamma, const float* unit_beta, void* workspace,
                                           size_t workspace_bytes, int32_t* sum_rows, lfd_stream_t stream);
LFD_API int lfd_bn_train_bwd_rows_f16(const void* dz, const void* y, int64_t pixels, int32_t channels, const float* stats,
                              const float* gamma, const float* beta, float inv_scale, int32_t accumulate, int32_t sum_rows,
                              void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, void* dy, lfd_stream_t stream);
LFD_API int lfd_stem_conv0_bn_bwd_wgrad_rows(const float* x_nchw, const void* dz, const void* y, int32_t n, int32_t h, int32_t w,
                                     int32_t channels, const float* stats, const float* gamma, const float* beta, float inv_scale,
                                     int32_t accumulate, int32_t sum_rows, void* workspace, size_t workspace_bytes, float* dgamma,
                                     float* dbeta, float* dw, lfd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Neck + head of ALL pyramid levels.  Replaces SimpleNeck.forward (simple_neck.py:67-74),
 * LFDHead.forward (lfd_head.py:164-185: GroupNorm towers + cls/reg convs + Scale) and the
 * NCHW -> [N,P,C] permute/concat of LFD.forward (lfd.py:526-542): writes the fp32 cls / reg rows of
 * every level directly at its point offset of the level-concatenated outputs.
 * GroupNorm statistics are obtained by recomputing the chain (pass 1, 2) -- see csrc/head.hip.
 *   pass 1: per-tile stats of tower conv1 -> `partial`;  lfd_groupnorm_finalize -> ab1
 *   pass 2: per-tile stats of tower conv2 -> `partial`;  lfd_groupnorm_finalize -> ab2
 *   pass 3: outputs.
 * For a BatchNorm / no-norm head, skip passes 1-2 and pass the folded (scale, shift) as ab1/ab2.
 * One call = one tower (merged cls+reg tower, or the cls / reg tower of an unmerged head). */
typedef struct lfd_head_desc {
  int32_t n;                                  /* images */
  int32_t num_levels;
  int32_t level_hw[LFD_MAX_LEVELS];           /* pixels (h*w) per level */
  int32_t level_cin[LFD_MAX_LEVELS];          /* backbone tap channels per level: 64 | 128 */
  int32_t level_point_offset[LFD_MAX_LEVELS]; /* first point of each level in [0, P) */
  int32_t head_channels;                      /* 128 */
  int32_t num_groups;                         /* GroupNorm groups (16) */
  int32_t total_points;                       /* P */
  int32_t cls_channels;                       /* C' (row length of out_cls) */
  int32_t final_reg_rows;                     /* final conv rows [0, reg_rows) -> reg: 0 or 4 */
  int32_t final_cls_rows;                     /* ... followed by cls_rows rows -> cls (reg+cls <= 64) */
} lfd_head_desc_t;

typedef struct lfd_head_level_ptrs {
  const void* x;           /* [n, hw, cin] fp16 backbone tap (NHWC) */
  const void* wn_packed;   /* neck conv1x1 (BN folded), packed */
  const float* bn;         /* neck bias [128] */
  const void* w1_packed;   /* tower conv 1, packed */
  const void* w2_packed;   /* tower conv 2, packed */
  const void* wf_packed;   /* final conv (rows padded to a multiple of 32), packed */
  const float* bf;         /* final bias */
  const float* scale;      /* per-level Scale (device scalar) or NULL */
  /* optional scratch, LFD_HEAD_FOLDED_HALFS halfs per image each ([n] images): tower conv 1 / 2 with the GroupNorm scale
   * of (level, image) folded into the rows and the shift as a bias fragment, in the kernel's register layout -- written
   * by lfd_groupnorm_finalize_fold(which = 1 / 2), read by passes 2-3 / pass 3.  NULL: every work chunk folds on its own. */
  void* w1_folded;
  void* w2_folded;
  /* optional scratch, n * ceil(hw / 32) * LFD_HEAD_TOWER1_GROUP_HALFS halfs: ReLU(GN1(conv1)) of every pixel in fp16, as the MFMA
   * fragments pass 2 feeds its conv2 with.  Pass 2 writes it; when every level has it (and both folded filters) pass 3 /
   * lfd_head_forward_decode_f16 load it instead of recomputing neck + conv1: same bits, 45 instead of 101 MFMAs per 32 pixels. */
  void* tower1_out;
  /* optional: tower conv 1 / 2 in the kernel's K-permuted register layout (element e of lane (m, hk) of fragment (ct, k) =
   * W[32 ct + m][16 k + 8 (e >> 2) + 4 hk + (e & 3)], LFD_HEAD_FOLDED_HALFS - 4 * 64 * 8 halfs: no bias fragments), for the
   * passes that use the filter unscaled (conv1 in pass 1, conv2 in pass 2): plain 16-byte loads instead of a permute per
   * work chunk.  NULL: permuted from w1_packed / w2_packed on the fly. */
  const void* w1_perm;
  const void* w2_perm;
} lfd_head_level_ptrs_t;
#define LFD_HEAD_FOLDED_HALFS (4 * 9 * 64 * 8)
#define LFD_HEAD_TOWER1_GROUP_HALFS (8 * 64 * 8)

LFD_API size_t lfd_head_partial_floats(const lfd_head_desc_t* desc);
LFD_API int lfd_head_forward_f16(const lfd_head_desc_t* desc, int32_t pass,
                                 const lfd_head_level_ptrs_t* levels /*[num_levels], host array*/,
                                 const float* ab1 /*[L][n][128][2]*/, const float* ab2,
                                 float* partial, float* out_cls, float* out_reg, const void* zeros,
                                 lfd_stream_t stream);
/* Pass 3 that also does the front half of lfd_detect_batched (lfd_head.py:164-185 + lfd.py:449-499 in one launch,
 * SURVEY 8b lfd_head_final_decode): sigma(cls) > score_thr -> decode -> append {box, score, label 0, point} to the
 * candidate arrays of `det_workspace`; lfd_detect_from_candidates finishes the step.  out_cls / out_reg may be NULL
 * (the fp32 [N,P,C'+4] logits then never reach HBM).  Supported: one foreground class with sigmoid scores, a merged
 * tower (final_reg_rows 4 + final_cls_rows 1), GroupNorm group size >= 8, det levels == head levels;
 * anything else returns LFD_ERR_UNSUPPORTED (use pass 3 + lfd_detect_batched). */
LFD_API int lfd_head_forward_decode_f16(const lfd_head_desc_t* desc, const lfd_head_level_ptrs_t* levels,
                                        const float* ab1, const float* ab2, float* out_cls, float* out_reg,
                                        const void* zeros, const lfd_detect_desc_t* det, const float* img_meta,
                                        void* det_workspace, size_t det_workspace_bytes, lfd_stream_t stream);
/* gamma / beta: host arrays of num_levels device pointers ([128] each). ab: [L][n][128][2]. */
LFD_API int lfd_groupnorm_finalize(const lfd_head_desc_t* desc, const float* partial,
                                   const float* const* gamma, const float* const* beta, float eps,
                                   float* ab, lfd_stream_t stream);
/* the same, and -- for every level whose w{which}_folded is not NULL -- the folded copy of tower conv `which` (1 | 2)
 * for each image (lfd_head_level_ptrs_t).  Same rounding as the in-kernel fold: fp16(fp32(w) * scale). */
LFD_API int lfd_groupnorm_finalize_fold(const lfd_head_desc_t* desc, const float* partial,
                                        const float* const* gamma, const float* const* beta, float eps, float* ab,
                                        const lfd_head_level_ptrs_t* levels, int32_t which, lfd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LFD_HIP_H_ */
