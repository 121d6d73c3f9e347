// csrc/losses.hip -- sigmoid focal loss (fwd / bwd / fused sum) and aligned IoU loss (fwd / bwd).
//
// Replaces sigmoid_focal_loss_ext.{forward,backward} (reference
// lfd/model/losses/build/sigmoid_focal_loss/src/cuda/sigmoid_focal_loss_cuda.cu:24-59, :62-97) and the
// ATen op chain of bbox_overlaps(is_aligned=True)+iou_loss (reference lfd/model/losses/iou_loss.py:67-123).
// Elementwise and HBM-bound: 16-byte vector accesses where the row length allows, grid-stride,
// wave64 shuffle reductions + fixed-order second stage for the fused sum (deterministic).
#include <float.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8 blocks: grid-stride beyond (guide G11)

template <typename T> __device__ __forceinline__ float ldf(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

// one element of SigmoidFocalLossForward (sigmoid_focal_loss_cuda.cu:31-57), fp32 math
__device__ __forceinline__ float focal_fwd_elem(float x, int t, int d, float gamma, float alpha) {
  const float c1 = (float)(t == d);
  const float c2 = (float)((t >= 0) & (t != d));
  const float zn = 1.0f - alpha, zp = alpha;
  const float p = 1.f / (1.f + expf(-x));
  const float term1 = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
  const float ge = (float)(x >= 0.f);
  const float term2 = powf(p, gamma) * (-1.f * x * ge - logf(1.f + expf(x - 2.f * x * ge)));
  float l = 0.f;
  l += -c1 * term1 * zp;
  l += -c2 * term2 * zn;
  return l;
}

// one element of SigmoidFocalLossBackward (sigmoid_focal_loss_cuda.cu:69-95)
__device__ __forceinline__ float focal_bwd_elem(float x, int t, int d, float gamma, float alpha, float g) {
  const float c1 = (float)(t == d);
  const float c2 = (float)((t >= 0) & (t != d));
  const float zn = 1.0f - alpha, zp = alpha;
  const float p = 1.f / (1.f + expf(-x));
  const float term1 = powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(fmaxf(p, FLT_MIN))));
  const float ge = (float)(x >= 0.f);
  const float term2 =
      powf(p, gamma) * ((-1.f * x * ge - logf(1.f + expf(x - 2.f * x * ge))) * (1.f - p) * gamma - p);
  float r = 0.f;
  r += -c1 * term1 * zp;
  r += -c2 * term2 * zn;
  return r * g;
}

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
    const float4 a = pred[i], b = target[i];
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float ov = w * h;
    const float a1 = (a.z - a.x) * (a.w - a.y);
    const float a2 = (b.z - b.x) * (b.w - b.y);
    const float un = fmaxf(a1 + a2 - ov, 1e-6f);
    const float iou = fmaxf(ov / un, eps);
    loss[i] = -logf(iou);
  }
}

// d loss / d pred, following autograd through the same expression graph:
//   loss = -log(q), q = max(ov/un, eps); un = max(a1+a2-ov, 1e-6); ov = w*h with clamps at 0;
//   torch.max / torch.min route the gradient to the larger / smaller operand (ties: split evenly
//   in ATen; ties have measure zero for float boxes and are resolved towards `pred` here).
__global__ __launch_bounds__(kThreads) void k_iou_bwd(const float4* pred, const float4* target, const float* dl,
                                                      int64_t n, float eps, float4* dpred) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const float4 a = pred[i], b = target[i];
    const float ltx = fmaxf(a.x, b.x), lty = fmaxf(a.y, b.y);
    const float rbx = fminf(a.z, b.z), rby = fminf(a.w, b.w);
    const float wr = rbx - ltx, hr = rby - lty;
    const float w = fmaxf(wr, 0.f), h = fmaxf(hr, 0.f);
    const float ov = w * h;
    const float pw = a.z - a.x, ph = a.w - a.y;
    const float a1 = pw * ph;
    const float a2 = (b.z - b.x) * (b.w - b.y);
    const float ur = a1 + a2 - ov;
    const float un = fmaxf(ur, 1e-6f);
    const float q = ov / un;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q > eps) {  // clamp(min=eps) passes gradient only above eps
      const float dq = -dl[i] / q;
      const float dov_direct = dq / un;
      const float dun = (ur > 1e-6f) ? (-dq * ov / (un * un)) : 0.f;
      const float dov = dov_direct - dun;  // un depends on -ov
      const float da1 = dun;
      // ov = w*h
      const float dw = (wr > 0.f) ? dov * h : 0.f;
      const float dh = (hr > 0.f) ? dov * w : 0.f;
      // w = min(a.z,b.z) - max(a.x,b.x)
      if (a.z <= b.z) g.z += dw;
      if (a.x >= b.x) g.x -= dw;
      if (a.w <= b.w) g.w += dh;
      if (a.y >= b.y) g.y -= dh;
      // a1 = (a.z-a.x)*(a.w-a.y)
      g.z += da1 * ph; g.x -= da1 * ph;
      g.w += da1 * pw; g.y -= da1 * pw;
    }
    dpred[i] = g;
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
