// csrc/decode_impl.h -- the per-location decode arithmetic (reference lfd/model/lfd.py:449-499, 261-282) shared by the
// post-processing kernels (postproc.hip) and the head pass that appends candidates itself (head.hip,
// lfd_head_forward_decode_f16): one definition, so both paths round identically.  Both translation units are compiled
// with -ffp-contract=off.
#pragma once
#include "common.h"

__device__ __forceinline__ float lfd_sigmoidf_ref(float x) { return 1.f / (1.f + expf(-x)); }

// regression outputs r0..r3 of the point at (px, py) -> clamped, rescaled box.  m = max(range) of the level for
// decode_mode 0 ('sigmoid', lfd.py:483-486), range[1] for mode 2 ('independent', :468-478); mode 1 = exp (:480-482);
// mode 3 = the regression outputs ARE the distances (FCOS: the head already applied Scale + exp, fcos_head.py:145-146,
// fcos.py:392 distance2bbox on the raw tensor).
__device__ __forceinline__ float4 lfd_decode_core(int decode_mode, float r0, float r1, float r2, float r3, float px, float py,
                                                  float m, float W, float H, float sc) {
  float d0, d1, d2, d3;
  if (decode_mode == 0) {
    d0 = lfd_sigmoidf_ref(r0) * m; d1 = lfd_sigmoidf_ref(r1) * m;
    d2 = lfd_sigmoidf_ref(r2) * m; d3 = lfd_sigmoidf_ref(r3) * m;
  } else if (decode_mode == 1) {
    d0 = expf(r0); d1 = expf(r1); d2 = expf(r2); d3 = expf(r3);
  } else if (decode_mode == 3) {
    d0 = r0; d1 = r1; d2 = r2; d3 = r3;
  } else {
    d0 = r0 * m; d1 = r1 * m; d2 = r2 * m; d3 = r3 * m;
  }
  float x1 = fminf(fmaxf(px - d0, 0.f), W);
  float y1 = fminf(fmaxf(py - d1, 0.f), H);
  float x2 = fminf(fmaxf(px + d2, 0.f), W);
  float y2 = fminf(fmaxf(py + d3, 0.f), H);
  return make_float4(x1 / sc, y1 / sc, x2 / sc, y2 / sc);
}

// Where a producer kernel appends the candidates of image n (slot = atomic counter, any order: the sort key of an
// appended segment is (score, point, label), so the order of arrival does not reach the results).
struct LfdAppendTarget {
  float4* cand_box;    // [N, cap]
  float* cand_score;
  int* cand_label;
  int* cand_point;
  uint32_t* maxord;    // [N] ordered-uint max coordinate (the reference's class-offset step); reset by the scan kernel
  int* total;          // [N] candidates seen so far (may exceed cap); reset by the scan kernel
  int cap;
  int decode_mode;
  float score_thr;
  float logit_lo;      // conservative bound: sigma(x) > score_thr implies x > logit_lo (prefilter only, never decides)
  const float* meta;   // [N,3] clampW, clampH, resize_scale
  int w[LFD_MAX_LEVELS], stride[LFD_MAX_LEVELS];
  float m[LFD_MAX_LEVELS];   // the level's range constant for decode_mode (rmax | rhi)
};

// host (postproc.hip): carve `workspace` exactly as lfd_detect_from_candidates will and describe the append target
int lfd_detect_bind_append(const lfd_detect_desc_t* desc, int32_t batch, const float* img_meta, void* workspace,
                           size_t workspace_bytes, LfdAppendTarget* out);
