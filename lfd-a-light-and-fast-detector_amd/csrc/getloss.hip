// csrc/getloss.hip -- LFD.get_loss (reference lfd/model/lfd.py:284-395) as three launches, no host sync and no
// boolean gathers: every (image, point) row is classified in registers (gray: any class target < 0, lfd.py:309-315;
// positive: max class score >= 0.001, lfd.py:319-321; label = argmax or the background index C, lfd.py:328) and
// contributes its focal / cross-entropy term and -- for positives -- the IoU loss of the decoded prediction
// against the decoded regression target (lfd.py:360-384) to fp64 block partials.  A fixed-order second stage makes
// the sums deterministic; `finalize` divides by the normalisers (n_pos + 1 / n_pos, or the summed positive scores
// when the enable_*_weight switches are on, lfd.py:333-340,378-384), which under image-parallel training are the
// GLOBAL sums (the caller all-reduces the 8-double sums vector between `sums` and `finalize`).
// The backward kernel writes dense [N,P,channels] / [N,P,4] gradients directly (zeros for gray / non-positive rows).
// HBM-bound: reads (channels + 4 + C + 4) floats per row once.
#include "common.h"
#include "loss_elems.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kSums = 8;  // cls_sum, reg_sum, n_pos, w_sum, n_green, (3 spare)

struct RowInfo {
  bool gray, pos;
  int label;
  float mx;
};

// classification of one row from its class targets (lfd.py:309-328); `first max` like torch.max for ties,
// which only occur among zeros (one class score per point is scattered by the target assignment)
__device__ __forceinline__ RowInfo classify(const float* __restrict__ ct, int C) {
  float mn = ct[0], mx = ct[0];
  int mi = 0;
  for (int j = 1; j < C; ++j) {
    const float v = ct[j];
    mn = fminf(mn, v);
    if (v > mx) { mx = v; mi = j; }
  }
  RowInfo r;
  r.gray = !(mn >= 0.f);
  r.pos = mx >= 0.001f;
  r.label = r.pos ? mi : C;
  r.mx = mx;
  return r;
}

struct Pt { float x, y, rmax; };

__device__ __forceinline__ Pt point_of(const lfd_loss_desc_t& d, int p) {
  int l = 0, base = 0;
  for (; l < d.num_levels - 1; ++l) {
    const int cnt = d.level_h[l] * d.level_w[l];
    if (p < base + cnt) break;
    base += cnt;
  }
  const int q = p - base, w = d.level_w[l];
  const int i = q / w, j = q - i * w;
  Pt r;
  r.x = (float)(j * d.stride[l]);  // generate_point_coordinates lfd.py:84-107 (no half-stride offset)
  r.y = (float)(i * d.stride[l]);
  r.rmax = d.range_max[l];
  return r;
}

// predicted distances (lfd.py:366-374): 'exp' -> exp(reg), else sigmoid(reg) * max(range)
__device__ __forceinline__ float dist_of(float r, float rmax, int mode) {
  if (mode == 1) return expf(r);
  return (1.f / (1.f + expf(-r))) * rmax;
}

__device__ __forceinline__ float4 box_of(Pt pt, float d0, float d1, float d2, float d3) {  // distance2bbox :261-282
  return make_float4(pt.x - d0, pt.y - d1, pt.x + d2, pt.y + d3);
}

__device__ __forceinline__ float cls_row_loss(const lfd_loss_desc_t& d, const float* __restrict__ x, int label) {
  if (d.cls_loss == 0) {
    float s = 0.f;
    for (int j = 0; j < d.num_classes; ++j) s += focal_fwd_elem(x[j], label, j, d.gamma, d.alpha);
    return s;
  }
  const int c = d.num_classes + 1;
  float mx = x[0];
  for (int j = 1; j < c; ++j) mx = fmaxf(mx, x[j]);
  float s = 0.f;
  for (int j = 0; j < c; ++j) s += expf(x[j] - mx);
  return -((x[label] - mx) - logf(s));
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__global__ __launch_bounds__(kThreads) void k_loss_partial(lfd_loss_desc_t d, const float* __restrict__ pc,
                                                          const float* __restrict__ pr,
                                                          const float* __restrict__ ct,
                                                          const float* __restrict__ rt, double* partials) {
  __shared__ double sm[kThreads / 64][5];
  const int C = d.num_classes, ch = d.cls_loss ? C + 1 : C, P = d.total_points;
  const int64_t rows = (int64_t)d.n * P;
  double a_cls = 0.0, a_reg = 0.0, a_w = 0.0;
  int a_pos = 0, a_green = 0;
  for (int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x; row < rows; row += (int64_t)gridDim.x * kThreads) {
    const RowInfo ri = classify(ct + row * C, C);
    if (ri.gray) continue;
    ++a_green;
    a_cls += (double)cls_row_loss(d, pc + row * ch, ri.label);
    if (!ri.pos) continue;
    ++a_pos;
    a_w += (double)ri.mx;
    const Pt pt = point_of(d, (int)(row % P));
    const float4 r = *reinterpret_cast<const float4*>(pr + row * 4);
    const float4 t = *reinterpret_cast<const float4*>(rt + row * 4);
    const float4 pb = box_of(pt, dist_of(r.x, pt.rmax, d.decode_mode), dist_of(r.y, pt.rmax, d.decode_mode),
                             dist_of(r.z, pt.rmax, d.decode_mode), dist_of(r.w, pt.rmax, d.decode_mode));
    const float4 tb = box_of(pt, t.x, t.y, t.z, t.w);
    float l = iou_loss_elem(pb, tb, d.iou_eps);
    if (d.reg_weighted) l *= ri.mx;
    a_reg += (double)l;
  }
  double v[5] = {a_cls, a_reg, (double)a_pos, a_w, (double)a_green};
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = wave_sum_d(v[k]);
  if (lfd_lane() == 0)
    for (int k = 0; k < 5; ++k) sm[threadIdx.x >> 6][k] = v[k];
  __syncthreads();
  if (threadIdx.x < 5) {
    double s = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) s += sm[i][threadIdx.x];
    partials[(size_t)blockIdx.x * kSums + threadIdx.x] = s;
  }
}

// reduction of the block partials: wave k sums component k (lanes stride over the blocks, fixed butterfly)
__global__ __launch_bounds__(64 * kSums) void k_loss_sums(const double* partials, int nblocks, double* sums) {
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  if (k < 5)
    for (int i = lane; i < nblocks; i += 64) s += partials[(size_t)i * kSums + k];
  s = wave_sum_d(s);
  if (lane == 0) sums[k] = s;
}

// out: [0] classification loss, [1] regression loss, [2] their sum, [3] global n_pos, [4] avg_factor cls,
//      [5] avg_factor reg, [6] local n_green, [7] rank scale
__global__ void k_loss_finalize(lfd_loss_desc_t d, const double* local, const double* global, float scale,
                                float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float n_pos = (float)global[2];
  const float avg_c = d.cls_weighted ? (float)global[3] : n_pos + 1.f;   // lfd.py:333-340
  const float avg_r = d.reg_weighted ? (float)global[3] : n_pos;         // lfd.py:378-384
  const float lc = d.cls_loss_weight * ((float)local[0] / avg_c) * scale;
  const float lr = n_pos > 0.f ? d.reg_loss_weight * ((float)local[1] / avg_r) * scale : 0.f;  // :386-387
  out[0] = lc;
  out[1] = lr;
  out[2] = lc + lr;
  out[3] = n_pos;
  out[4] = avg_c;
  out[5] = avg_r;
  out[6] = (float)local[4];
  out[7] = scale;
}

__global__ __launch_bounds__(kThreads) void k_loss_bwd(lfd_loss_desc_t d, const float* __restrict__ pc,
                                                      const float* __restrict__ pr, const float* __restrict__ ct,
                                                      const float* __restrict__ rt, const float* __restrict__ fin,
                                                      const float* __restrict__ gout, float* __restrict__ gcls,
                                                      float* __restrict__ greg) {
  const int C = d.num_classes, ch = d.cls_loss ? C + 1 : C, P = d.total_points;
  const int64_t rows = (int64_t)d.n * P;
  // d(out[0]) and d(out[1]) both also flow through out[2]
  const float g_c = (gout[0] + gout[2]) * d.cls_loss_weight * fin[7] / fin[4];
  const float g_r = fin[3] > 0.f ? (gout[1] + gout[2]) * d.reg_loss_weight * fin[7] / fin[5] : 0.f;
  for (int64_t row = (int64_t)blockIdx.x * kThreads + threadIdx.x; row < rows; row += (int64_t)gridDim.x * kThreads) {
    const RowInfo ri = classify(ct + row * C, C);
    const float* x = pc + row * ch;
    float* gx = gcls + row * ch;
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ri.gray) {
      for (int j = 0; j < ch; ++j) gx[j] = 0.f;
    } else if (d.cls_loss == 0) {
      for (int j = 0; j < C; ++j) gx[j] = focal_bwd_elem(x[j], ri.label, j, d.gamma, d.alpha, g_c);
    } else {
      float mx = x[0];
      for (int j = 1; j < ch; ++j) mx = fmaxf(mx, x[j]);
      float s = 0.f;
      for (int j = 0; j < ch; ++j) s += expf(x[j] - mx);
      const float inv = 1.f / s;
      for (int j = 0; j < ch; ++j) gx[j] = g_c * (expf(x[j] - mx) * inv - (j == ri.label ? 1.f : 0.f));
    }
    if (!ri.gray && ri.pos && g_r != 0.f) {
      const Pt pt = point_of(d, (int)(row % P));
      const float4 r = *reinterpret_cast<const float4*>(pr + row * 4);
      const float4 t = *reinterpret_cast<const float4*>(rt + row * 4);
      const float rv[4] = {r.x, r.y, r.z, r.w};
      float dv[4], dd[4];
      for (int k = 0; k < 4; ++k) dv[k] = dist_of(rv[k], pt.rmax, d.decode_mode);
      const float4 pb = box_of(pt, dv[0], dv[1], dv[2], dv[3]);
      const float4 tb = box_of(pt, t.x, t.y, t.z, t.w);
      const float4 gb = iou_loss_grad_elem(pb, tb, d.iou_eps, d.reg_weighted ? g_r * ri.mx : g_r);
      const float gd[4] = {-gb.x, -gb.y, gb.z, gb.w};  // x1 = px - d0, y1 = py - d1, x2 = px + d2, y2 = py + d3
      for (int k = 0; k < 4; ++k) {
        if (d.decode_mode == 1) {
          dd[k] = gd[k] * dv[k];                        // d exp(r) = exp(r)
        } else {
          const float sg = 1.f / (1.f + expf(-rv[k]));
          dd[k] = gd[k] * pt.rmax * ((1.f - sg) * sg);  // d sigmoid
        }
      }
      gr = make_float4(dd[0], dd[1], dd[2], dd[3]);
    }
    *reinterpret_cast<float4*>(greg + row * 4) = gr;
  }
}

inline unsigned grid_for(int64_t rows) {
  int64_t b = (rows + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int check_desc(const lfd_loss_desc_t* d) {
  if (!d) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 0 || d->num_levels < 1 || d->num_levels > LFD_MAX_LEVELS || d->num_classes < 1 || d->total_points < 0)
    return LFD_ERR_INVALID_ARGUMENT;
  if (d->cls_loss < 0 || d->cls_loss > 1 || d->decode_mode < 0 || d->decode_mode > 1) return LFD_ERR_INVALID_ARGUMENT;
  long long pts = 0;
  for (int i = 0; i < d->num_levels; ++i) {
    if (d->level_h[i] < 0 || d->level_w[i] < 0 || d->stride[i] < 1) return LFD_ERR_INVALID_ARGUMENT;
    pts += (long long)d->level_h[i] * d->level_w[i];
  }
  if (pts != d->total_points) return LFD_ERR_INVALID_ARGUMENT;
  return LFD_OK;
}

}  // namespace

extern "C" {

size_t lfd_get_loss_workspace_bytes(void) { return sizeof(double) * kSums * kMaxBlocks; }

int lfd_get_loss_sums_f32(const lfd_loss_desc_t* d, const float* pred_cls, const float* pred_reg,
                          const float* cls_targets, const float* reg_targets, void* workspace, size_t workspace_bytes,
                          double* sums, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc = check_desc(d);
  if (rc != LFD_OK) return rc;
  if (!sums) return LFD_ERR_INVALID_ARGUMENT;
  const int64_t rows = (int64_t)d->n * d->total_points;
  if (rows == 0) {
    if (hipMemsetAsync(sums, 0, sizeof(double) * kSums, st) != hipSuccess) return LFD_ERR_LAUNCH_FAILED;
    return LFD_OK;
  }
  if (!pred_cls || !pred_reg || !cls_targets || !reg_targets || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_get_loss_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const unsigned g = grid_for(rows);
  hipLaunchKernelGGL(k_loss_partial, dim3(g), dim3(kThreads), 0, st, *d, pred_cls, pred_reg, cls_targets,
                     reg_targets, (double*)workspace);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_loss_sums, dim3(1), dim3(64 * kSums), 0, st, (const double*)workspace, (int)g, sums);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_get_loss_finalize_f32(const lfd_loss_desc_t* d, const double* local_sums, const double* global_sums,
                              float rank_scale, float* out, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc = check_desc(d);
  if (rc != LFD_OK) return rc;
  if (!local_sums || !global_sums || !out) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, *d, local_sums, global_sums, rank_scale, out);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_get_loss_bwd_f32(const lfd_loss_desc_t* d, const float* pred_cls, const float* pred_reg,
                         const float* cls_targets, const float* reg_targets, const float* finalized,
                         const float* grad_out, float* grad_cls, float* grad_reg, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rc = check_desc(d);
  if (rc != LFD_OK) return rc;
  const int64_t rows = (int64_t)d->n * d->total_points;
  if (rows == 0) return LFD_OK;
  if (!pred_cls || !pred_reg || !cls_targets || !reg_targets || !finalized || !grad_out || !grad_cls || !grad_reg)
    return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_loss_bwd, dim3(grid_for(rows)), dim3(kThreads), 0, st, *d, pred_cls, pred_reg, cls_targets,
                     reg_targets, finalized, grad_out, grad_cls, grad_reg);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
