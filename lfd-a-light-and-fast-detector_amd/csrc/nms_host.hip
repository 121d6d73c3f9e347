// csrc/nms_host.hip -- HOST-side members of the nms_ext surface (no kernels in this file).
//
// The reference's pybind module serves CPU tensors too: `nms` dispatches on the tensor's device (nms_ext.cpp:18-27 ->
// cpu/nms_cpu.cpp:7-66), `soft_nms` and `nms_match` exist ONLY for CPU tensors (nms_ext.cpp:29-43 -> nms_cpu.cpp:76-206,
// :220-283) and are not on any shipped config's path.  A drop-in has to keep that surface, so the three CPU
// routines are provided here as plain C functions on HOST pointers; they are what runs for CPU tensors / numpy arrays -- they
// are NOT a fallback of the device path (lfd_nms_f32 & co. never call them, and a GPU tensor never reaches them).
// Arithmetic: the tensor's own dtype like the reference (AT_DISPATCH_FLOATING_TYPES, nms_cpu.cpp:70,212,287 -- float32 and
// float64 entry points; float64 is the default dtype of a numpy array), IoU = inter / (area_a + area_b - inter) evaluated
// left to right, no +1, as the reference writes it (this translation unit is compiled with -ffp-contract=off like the
// device NMS).  The thresholds / sigma / min_score stay `float` arguments as in the reference's signatures and are promoted
// in the comparisons.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>
#include "common.h"

namespace {
template <typename T>
inline T iou_of(const T* a, T area_a, const T* b, T area_b) {
  const T xx1 = std::max(a[0], b[0]), yy1 = std::max(a[1], b[1]);
  const T xx2 = std::min(a[2], b[2]), yy2 = std::min(a[3], b[3]);
  const T w = std::max((T)0, xx2 - xx1), h = std::max((T)0, yy2 - yy1);
  const T inter = w * h;
  return inter / (area_a + area_b - inter);
}
// score-descending order; equal scores keep their input order (the contract of the device path; the reference's
// sort is unstable, nms_cpu.cpp:23)
template <typename T>
std::vector<int64_t> order_of(const T* dets, int64_t n) {
  std::vector<int64_t> o(n);
  std::iota(o.begin(), o.end(), 0);
  std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return dets[a * 5 + 4] > dets[b * 5 + 4]; });
  return o;
}

// nms_cpu_kernel (nms_cpu.cpp:7-66): keep[k] = original indices of the kept boxes, score-descending; suppress when IoU > thr.
template <typename T>
int nms_cpu(const T* dets, int64_t n, float iou_thr, int64_t* keep, int64_t* num_keep) {
  if (n < 0 || !num_keep || (n > 0 && (!dets || !keep))) return LFD_ERR_INVALID_ARGUMENT;
  std::vector<T> area(n);
  for (int64_t i = 0; i < n; ++i) area[i] = (dets[i * 5 + 2] - dets[i * 5 + 0]) * (dets[i * 5 + 3] - dets[i * 5 + 1]);
  const std::vector<int64_t> order = order_of<T>(dets, n);
  std::vector<unsigned char> sup(n, 0);
  int64_t k = 0;
  for (int64_t a = 0; a < n; ++a) {
    const int64_t i = order[a];
    if (sup[i]) continue;
    keep[k++] = i;
    for (int64_t b = a + 1; b < n; ++b) {
      const int64_t j = order[b];
      if (sup[j]) continue;
      if (iou_of<T>(dets + i * 5, area[i], dets + j * 5, area[j]) > iou_thr) sup[j] = 1;
    }
  }
  *num_keep = k;
  return LFD_OK;
}

// soft_nms_cpu_kernel (nms_cpu.cpp:76-206): selection-sort style Soft-NMS.  method 1 = linear (weight 1 - IoU above thr),
// 2 = gaussian (exp(-IoU^2 / sigma)), anything else = hard NMS; boxes whose score falls below min_score are dropped by swapping
// the last box in.  out[k][6] = x1, y1, x2, y2, new score, original index (as float, like the reference's result tensor).
template <typename T>
int soft_nms_cpu(const T* dets, int64_t n, float iou_thr, int32_t method, float sigma, float min_score, T* out, int64_t* num_out) {
  if (n < 0 || !num_out || (n > 0 && (!dets || !out))) return LFD_ERR_INVALID_ARGUMENT;
  std::vector<T> x1(n), y1(n), x2(n), y2(n), sc(n), ar(n), id(n);
  for (int64_t i = 0; i < n; ++i) {
    x1[i] = dets[i * 5 + 0]; y1[i] = dets[i * 5 + 1]; x2[i] = dets[i * 5 + 2]; y2[i] = dets[i * 5 + 3]; sc[i] = dets[i * 5 + 4];
    ar[i] = (x2[i] - x1[i]) * (y2[i] - y1[i]);
    id[i] = (T)i;
  }
  int64_t nd = n;
  for (int64_t i = 0; i < nd; ++i) {
    int64_t mp = i;
    T ms = sc[i];
    for (int64_t p = i + 1; p < nd; ++p)
      if (ms < sc[p]) { ms = sc[p]; mp = p; }
    std::swap(x1[i], x1[mp]); std::swap(y1[i], y1[mp]); std::swap(x2[i], x2[mp]); std::swap(y2[i], y2[mp]);
    std::swap(sc[i], sc[mp]); std::swap(ar[i], ar[mp]); std::swap(id[i], id[mp]);
    const T bi[4] = {x1[i], y1[i], x2[i], y2[i]};
    for (int64_t p = i + 1; p < nd; ++p) {
      const T bp[4] = {x1[p], y1[p], x2[p], y2[p]};
      const T ovr = iou_of<T>(bi, ar[i], bp, ar[p]);
      T weight = 1;
      if (method == 1) { if (ovr > iou_thr) weight = 1 - ovr; }
      else if (method == 2) weight = std::exp(-(ovr * ovr) / sigma);
      else weight = ovr > iou_thr ? (T)0 : (T)1;
      sc[p] = weight * sc[p];
      if (sc[p] < min_score) {
        x1[p] = x1[nd - 1]; y1[p] = y1[nd - 1]; x2[p] = x2[nd - 1]; y2[p] = y2[nd - 1];
        sc[p] = sc[nd - 1]; ar[p] = ar[nd - 1]; id[p] = id[nd - 1];
        --nd; --p;
      }
    }
  }
  for (int64_t i = 0; i < nd; ++i) {
    out[i * 6 + 0] = x1[i]; out[i * 6 + 1] = y1[i]; out[i * 6 + 2] = x2[i]; out[i * 6 + 3] = y2[i]; out[i * 6 + 4] = sc[i];
    out[i * 6 + 5] = id[i];
  }
  *num_out = nd;
  return LFD_OK;
}

// nms_match_cpu_kernel (nms_cpu.cpp:220-283): greedy NMS that records what every kept box suppressed (IoU >= thr, note the
// >=).  members = all n indices grouped: group g starts with its kept box followed by the boxes it matched, group_sizes[g]
// entries each; *num_groups groups.
template <typename T>
int nms_match_cpu(const T* dets, int64_t n, float iou_thr, int32_t* members, int32_t* group_sizes, int64_t* num_groups) {
  if (n < 0 || !num_groups || (n > 0 && (!dets || !members || !group_sizes))) return LFD_ERR_INVALID_ARGUMENT;
  std::vector<T> area(n);
  for (int64_t i = 0; i < n; ++i) area[i] = (dets[i * 5 + 2] - dets[i * 5 + 0]) * (dets[i * 5 + 3] - dets[i * 5 + 1]);
  const std::vector<int64_t> order = order_of<T>(dets, n);
  std::vector<unsigned char> sup(n, 0);
  int64_t g = 0, m = 0;
  for (int64_t a = 0; a < n; ++a) {
    const int64_t i = order[a];
    if (sup[i]) continue;
    const int64_t start = m;
    members[m++] = (int32_t)i;
    for (int64_t b = a + 1; b < n; ++b) {
      const int64_t j = order[b];
      if (sup[j]) continue;
      if (iou_of<T>(dets + i * 5, area[i], dets + j * 5, area[j]) >= iou_thr) { sup[j] = 1; members[m++] = (int32_t)j; }
    }
    group_sizes[g++] = (int32_t)(m - start);
  }
  *num_groups = g;
  return LFD_OK;
}

}  // namespace

extern "C" {

int lfd_nms_cpu_f32(const float* dets, int64_t n, float iou_thr, int64_t* keep, int64_t* num_keep) {
  return nms_cpu<float>(dets, n, iou_thr, keep, num_keep);
}
int lfd_nms_cpu_f64(const double* dets, int64_t n, float iou_thr, int64_t* keep, int64_t* num_keep) {
  return nms_cpu<double>(dets, n, iou_thr, keep, num_keep);
}
int lfd_soft_nms_cpu_f32(const float* dets, int64_t n, float iou_thr, int32_t method, float sigma, float min_score, float* out,
                         int64_t* num_out) {
  return soft_nms_cpu<float>(dets, n, iou_thr, method, sigma, min_score, out, num_out);
}
int lfd_soft_nms_cpu_f64(const double* dets, int64_t n, float iou_thr, int32_t method, float sigma, float min_score, double* out,
                         int64_t* num_out) {
  return soft_nms_cpu<double>(dets, n, iou_thr, method, sigma, min_score, out, num_out);
}
int lfd_nms_match_cpu_f32(const float* dets, int64_t n, float iou_thr, int32_t* members, int32_t* group_sizes, int64_t* num_groups) {
  return nms_match_cpu<float>(dets, n, iou_thr, members, group_sizes, num_groups);
}
int lfd_nms_match_cpu_f64(const double* dets, int64_t n, float iou_thr, int32_t* members, int32_t* group_sizes, int64_t* num_groups) {
  return nms_match_cpu<double>(dets, n, iou_thr, members, group_sizes, num_groups);
}

}  // extern "C"
