// csrc/targets.hip -- training-side kernels next to the focal / IoU losses:
//   * lfd_assign_targets_f32: LFD.annotation_to_target / _generate_target_for_single_image
//     (reference lfd/model/lfd.py:109-259) for a whole batch in one launch, one thread per (image, point).
//     The reference builds [P, G] broadcasts on the CPU, sorts every row by score and scatters twice; per point
//     that is equivalent to one pass over the image's boxes:
//       cls[p, c] = -1                      if a GRAY box of class c covers p          (lfd.py:241-243, written last)
//                 = max score of the GREEN boxes of class c covering p               (lfd.py:236-238: ascending
//                                                                                        sort, the largest is written last)
//                 = 0                       otherwise
//       reg[p]    = delta of the box selected by `(sorted_score * green).max(dim=1)`   (lfd.py:246-249): the green box
//                   with the largest score (first in the stable ascending order among equals = lowest index), or --
//                   when no box is green -- position 0 of the sorted order: the box with the smallest score.
//     Every expression is evaluated in fp32 in the reference's order (-ffp-contract=off, IEEE divide / sqrt): the
//     targets are bit-identical to an IEEE evaluation of the reference (and to the reference-generated golden
//     fixtures); torch's own CPU sqrt is a <= 1 ulp routine in some builds, so scores may differ from a live torch
//     run in the last bit -- never the green / gray / negative decisions or the selected box.
//   * lfd_cross_entropy_{fwd,bwd}_f32: F.cross_entropy(pred, label, reduction='none') and its gradient
//     (reference lfd/model/losses/cross_entropy_loss.py:12-50, TT100K configs).
#include "common.h"

namespace {

struct AssignArgs {
  lfd_assign_desc_t d;
  const float* gt_boxes;      // [sumG, 4] x, y, w, h
  const int64_t* gt_labels;   // [sumG]
  const int32_t* gt_offsets;  // [n + 1]
  float* cls_t;               // [n, P, C]
  float* reg_t;               // [n, P, 4]
};

__device__ __forceinline__ float axis_score(float d, float half_stride) {
  float v = d / half_stride;                 // lfd.py:190,193
  v = v * (v >= 1.f ? 1.f : 0.f) + (v < 1.f ? 1.f : 0.f);   // :191,194
  return sqrtf(1.f / v);                     // :192,195 (IEEE divide / sqrt: -fhip-fp32-correctly-rounded-divide-sqrt;
                                             //  the __f*_rn intrinsics map to the native approximations)
}

__global__ __launch_bounds__(256) void k_assign(AssignArgs a) {
  const int P = a.d.total_points, C = a.d.num_classes;
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int g0 = a.gt_offsets[n], g1 = a.gt_offsets[n + 1];
  __shared__ float s_box[128 * 4];
  __shared__ int s_lab[128];
  // level / coordinates of this point (generate_point_coordinates, lfd.py:84-107: x = j*stride, y = i*stride)
  int l = 0, q = p;
  bool valid = p < P;
  if (valid) {
    while (l < a.d.num_levels - 1 && q >= a.d.level_h[l] * a.d.level_w[l]) { q -= a.d.level_h[l] * a.d.level_w[l]; ++l; }
  }
  const int stride = a.d.stride[l];
  const float px = valid ? (float)((q % a.d.level_w[l]) * stride) : 0.f;
  const float py = valid ? (float)((q / a.d.level_w[l]) * stride) : 0.f;
  const float hs = (float)stride / 2.f;
  const float rlo = (float)a.d.reg_lo[l], rhi = (float)a.d.reg_hi[l], glo = (float)a.d.gray_lo[l], ghi = (float)a.d.gray_hi[l];
  float* crow = a.cls_t + ((size_t)n * P + p) * C;
  if (valid)
    for (int c = 0; c < C; ++c) crow[c] = 0.f;

  float best_s = 0.f, min_s = 0.f;
  float bd[4] = {0.f, 0.f, 0.f, 0.f}, md[4] = {0.f, 0.f, 0.f, 0.f};
  bool have_best = false, have_min = false;
  for (int base = g0; base < g1; base += 128) {
    const int cnt = (g1 - base) < 128 ? (g1 - base) : 128;
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 4; i += blockDim.x) s_box[i] = a.gt_boxes[(size_t)base * 4 + i];
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) s_lab[i] = (int)a.gt_labels[base + i];
    __syncthreads();
    if (!valid) continue;
    for (int g = 0; g < cnt; ++g) {
      const float bx = s_box[4 * g], by = s_box[4 * g + 1], bw = s_box[4 * g + 2], bh = s_box[4 * g + 3];
      const float cx = bx + bw / 2.f, cy = by + bh / 2.f;                          // :184-185
      const float score = axis_score(fabsf(px - cx), hs) * axis_score(fabsf(py - cy), hs);   // :188-196
      float d0 = px - bx, d1 = py - by;                                            // :199-200
      float d2 = (bx + bw - 1.f) - px, d3 = (by + bh - 1.f) - py;                  // :201-202
      float measure;
      switch (a.d.assign_mode) {                                                   // :206-215
        case 0: measure = fmaxf(bw, bh); break;
        case 1: measure = fminf(bw, bh); break;
        case 2: measure = sqrtf(bw * bh); break;
        default: measure = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)); break;
      }
      if (a.d.independent) { d0 = d0 / rhi; d1 = d1 / rhi; d2 = d2 / rhi; d3 = d3 / rhi; }   // :217-218
      const bool hit = fminf(fminf(d0, d1), fminf(d2, d3)) >= 0.f;                // :221
      const bool green = (rlo <= measure) && (measure <= rhi) && hit;             // :220,222
      const bool gray = (((glo <= measure) && (measure < rlo)) || ((rhi < measure) && (measure <= ghi))) && hit;   // :224-226
      const int c = s_lab[g];
      if (gray) {
        crow[c] = -1.f;
      } else if (green) {
        const float cur = crow[c];
        if (cur != -1.f && score > cur) crow[c] = score;
      }
      if (green && !gray && (!have_best || score > best_s)) { have_best = true; best_s = score; bd[0] = d0; bd[1] = d1; bd[2] = d2; bd[3] = d3; }
      if (!have_min || score < min_s) { have_min = true; min_s = score; md[0] = d0; md[1] = d1; md[2] = d2; md[3] = d3; }
    }
  }
  if (valid) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    // a green box whose score is exactly 0 cannot occur (scores are > 0), so "no green" == "filtered max is 0"
    if (have_best) r = make_float4(bd[0], bd[1], bd[2], bd[3]);
    else if (have_min) r = make_float4(md[0], md[1], md[2], md[3]);
    *reinterpret_cast<float4*>(a.reg_t + ((size_t)n * P + p) * 4) = r;
  }
}

__global__ __launch_bounds__(256) void k_ce_fwd(const float* x, const int64_t* lab, int64_t m, int c, float* loss) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float* r = x + i * c;
  float mx = r[0];
  for (int j = 1; j < c; ++j) mx = fmaxf(mx, r[j]);
  float s = 0.f;
  for (int j = 0; j < c; ++j) s += expf(r[j] - mx);
  const int64_t t = lab[i];
  loss[i] = (t >= 0 && t < c) ? -((r[t] - mx) - logf(s)) : 0.f;
}

__global__ __launch_bounds__(256) void k_ce_bwd(const float* x, const int64_t* lab, const float* dl, int64_t m, int c,
                                                float* dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float* r = x + i * c;
  float mx = r[0];
  for (int j = 1; j < c; ++j) mx = fmaxf(mx, r[j]);
  float s = 0.f;
  for (int j = 0; j < c; ++j) s += expf(r[j] - mx);
  const int64_t t = lab[i];
  const float g = dl[i], inv = 1.f / s;
  for (int j = 0; j < c; ++j) dx[i * c + j] = (t >= 0 && t < c) ? g * (expf(r[j] - mx) * inv - (j == t ? 1.f : 0.f)) : 0.f;
}

}  // namespace

extern "C" {

int lfd_assign_targets_f32(const lfd_assign_desc_t* d, const float* gt_boxes, const int64_t* gt_labels,
                           const int32_t* gt_offsets, float* cls_targets, float* reg_targets, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !gt_offsets || !cls_targets || !reg_targets) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->num_levels < 1 || d->num_levels > LFD_MAX_LEVELS || d->num_classes < 1 || d->total_points < 0)
    return LFD_ERR_INVALID_ARGUMENT;
  if (d->assign_mode < 0 || d->assign_mode > 3) return LFD_ERR_INVALID_ARGUMENT;
  long long pts = 0;
  for (int i = 0; i < d->num_levels; ++i) {
    if (d->level_h[i] < 0 || d->level_w[i] < 0 || d->stride[i] < 1) return LFD_ERR_INVALID_ARGUMENT;
    pts += (long long)d->level_h[i] * d->level_w[i];
  }
  if (pts != d->total_points) return LFD_ERR_INVALID_ARGUMENT;
  if (d->total_points == 0) return LFD_OK;
  AssignArgs a{*d, gt_boxes, gt_labels, gt_offsets, cls_targets, reg_targets};
  hipLaunchKernelGGL(k_assign, dim3((d->total_points + 255) / 256, d->n), dim3(256), 0, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_cross_entropy_fwd_f32(const float* logits, const int64_t* labels, int64_t m, int32_t channels, float* loss,
                              lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (m < 0 || channels < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (m == 0) return LFD_OK;
  if (!logits || !labels || !loss) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_ce_fwd, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, logits, labels, m, channels, loss);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_cross_entropy_bwd_f32(const float* logits, const int64_t* labels, const float* d_loss, int64_t m,
                              int32_t channels, float* d_logits, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (m < 0 || channels < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (m == 0) return LFD_OK;
  if (!logits || !labels || !d_loss || !d_logits) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_ce_bwd, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, logits, labels, d_loss, m, channels, d_logits);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
