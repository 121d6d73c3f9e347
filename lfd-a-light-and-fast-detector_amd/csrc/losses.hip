// csrc/losses.hip -- sigmoid focal loss (fwd / bwd / fused sum) and aligned IoU loss (fwd / bwd).
//
// Replaces sigmoid_focal_loss_ext.{forward,backward} (reference
// lfd/model/losses/build/sigmoid_focal_loss/src/cuda/sigmoid_focal_loss_cuda.cu:24-59, :62-97) and the
// ATen op chain of bbox_overlaps(is_aligned=True)+iou_loss (reference lfd/model/losses/iou_loss.py:67-123).
// Elementwise and HBM-bound: 16-byte vector accesses where the row length allows, grid-stride,
// wave64 shuffle reductions + fixed-order second stage for the fused sum (deterministic).
#include <float.h>
#include "common.h"
#include "loss_elems.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8 blocks: grid-stride beyond (guide G11)

template <typename T> __device__ __forceinline__ float ldf(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

template <typename T>
__global__ __launch_bounds__(kThreads) void k_focal_fwd(const T* logits, const int64_t* targets, int64_t total,
                                                        int c, float gamma, float alpha, T* losses) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / c;
    const int d = (int)(i - n * c);
    stf(losses, i, focal_fwd_elem(ldf(logits, i), (int)targets[n], d, gamma, alpha));
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_focal_bwd(const T* logits, const int64_t* targets, const T* dl,
                                                        int64_t total, int c, float gamma, float alpha, T* out) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / c;
    const int d = (int)(i - n * c);
    stf(out, i, focal_bwd_elem(ldf(logits, i), (int)targets[n], d, gamma, alpha, ldf(dl, i)));
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// stage 1: per-block partial sums (fp64), stage 2 (last launch): fixed-order sum of the partials
__global__ __launch_bounds__(kThreads) void k_focal_sum_partial(const float* logits, const int64_t* targets,
                                                                int64_t total, int c, float gamma, float alpha,
                                                                double* partials) {
  __shared__ double sm[kThreads / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int64_t n = i / c;
    const int d = (int)(i - n * c);
    acc += (double)focal_fwd_elem(logits[i], (int)targets[n], d, gamma, alpha);
  }
  acc = wave_sum(acc);
  if (lfd_lane() == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) s += sm[i];
    partials[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kThreads) void k_sum_final(const double* partials, int n, float* out) {
  __shared__ double sm[kThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) acc += partials[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sm[0];
}

// ---- IoU loss (iou_loss.py:67-79 aligned overlaps, :98-102 union clamp, :121-123 -log(clamp))
__global__ __launch_bounds__(kThreads) void k_iou_fwd(const float4* pred, const float4* target, int64_t n,
                                                      float eps, float* loss) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    loss[i] = iou_loss_elem(pred[i], target[i], eps);
  }
}

__global__ __launch_bounds__(kThreads) void k_iou_bwd(const float4* pred, const float4* target, const float* dl,
                                                      int64_t n, float eps, float4* dpred) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    dpred[i] = iou_loss_grad_elem(pred[i], target[i], eps, dl[i]);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" {

int lfd_sigmoid_focal_loss_fwd(const void* logits, const int64_t* targets, int64_t n, int32_t c, float gamma,
                               float alpha, void* losses, int32_t dtype, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || c < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!logits || !targets || !losses) return LFD_ERR_INVALID_ARGUMENT;
  const int64_t total = n * c;
  if (dtype == LFD_F32)
    hipLaunchKernelGGL(k_focal_fwd<float>, dim3(grid_for(total)), dim3(kThreads), 0, st, (const float*)logits,
                       targets, total, c, gamma, alpha, (float*)losses);
  else if (dtype == LFD_F16)
    hipLaunchKernelGGL(k_focal_fwd<__half>, dim3(grid_for(total)), dim3(kThreads), 0, st, (const __half*)logits,
                       targets, total, c, gamma, alpha, (__half*)losses);
  else
    return LFD_ERR_INVALID_ARGUMENT;
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_sigmoid_focal_loss_bwd(const void* logits, const int64_t* targets, const void* d_losses, int64_t n,
                               int32_t c, float gamma, float alpha, void* d_logits, int32_t dtype,
                               lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || c < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!logits || !targets || !d_losses || !d_logits) return LFD_ERR_INVALID_ARGUMENT;
  const int64_t total = n * c;
  if (dtype == LFD_F32)
    hipLaunchKernelGGL(k_focal_bwd<float>, dim3(grid_for(total)), dim3(kThreads), 0, st, (const float*)logits,
                       targets, (const float*)d_losses, total, c, gamma, alpha, (float*)d_logits);
  else if (dtype == LFD_F16)
    hipLaunchKernelGGL(k_focal_bwd<__half>, dim3(grid_for(total)), dim3(kThreads), 0, st, (const __half*)logits,
                       targets, (const __half*)d_losses, total, c, gamma, alpha, (__half*)d_logits);
  else
    return LFD_ERR_INVALID_ARGUMENT;
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

size_t lfd_reduce_workspace_bytes(void) { return sizeof(double) * kMaxBlocks + 256; }

int lfd_sigmoid_focal_loss_sum_f32(const float* logits, const int64_t* targets, int64_t n, int32_t c, float gamma,
                                   float alpha, float* loss_sum, void* workspace, size_t workspace_bytes,
                                   lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || c < 1 || !loss_sum) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) {
    if (hipMemsetAsync(loss_sum, 0, sizeof(float), st) != hipSuccess) return LFD_ERR_LAUNCH_FAILED;
    return LFD_OK;
  }
  if (!logits || !targets || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_reduce_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  double* partials = reinterpret_cast<double*>(workspace);
  const int64_t total = n * c;
  const unsigned g = grid_for(total);
  hipLaunchKernelGGL(k_focal_sum_partial, dim3(g), dim3(kThreads), 0, st, logits, targets, total, c, gamma, alpha,
                     partials);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kThreads), 0, st, partials, (int)g, loss_sum);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_iou_loss_fwd_f32(const float* pred, const float* target, int64_t n, float eps, float* loss,
                         lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!pred || !target || !loss) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_iou_fwd, dim3(grid_for(n)), dim3(kThreads), 0, st, (const float4*)pred,
                     (const float4*)target, n, eps, loss);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_iou_loss_bwd_f32(const float* pred, const float* target, const float* d_loss, int64_t n, float eps,
                         float* d_pred, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!pred || !target || !d_loss || !d_pred) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_iou_bwd, dim3(grid_for(n)), dim3(kThreads), 0, st, (const float4*)pred,
                     (const float4*)target, d_loss, n, eps, (float4*)d_pred);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
