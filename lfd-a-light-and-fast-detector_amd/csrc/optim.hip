// csrc/optim.hip -- the parameter update of a training iteration over ONE flat fp32 buffer:
// clip_grad_norm_(max_norm, L2) + SGD(momentum, dampening, weight_decay, nesterov).
// Replaces OptimizerHook.after_train_iter's clip + step (reference lfd/execution/hooks/optimizer_hook.py:21-36:
// torch.nn.utils.clip_grad.clip_grad_norm_ + torch.optim.SGD.step, configured at WIDERFACE_LFD_S.py:216-226), which
// in the reference are ~3 ATen launches per parameter tensor (55 tensors for WF-S).  All parameters / gradients /
// momentum buffers of a param group live contiguously (lfd_amd/optim.py), so the norm is two launches and the update
// one, 16-byte vector accesses, HBM-bound (WF-S: 1.57 M parameters = 6.3 MB per array).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__global__ __launch_bounds__(kThreads) void k_sumsq_partial(const float* __restrict__ g, int64_t n, double* partials) {
  __shared__ double sm[kThreads / 64];
  double acc = 0.0;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    const float4 v = g4[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc += (double)v * v;
  }
  acc = wave_sum_d(acc);
  if (lfd_lane() == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) s += sm[i];
    partials[blockIdx.x] = s;
  }
}

// out[0] = total L2 norm, out[1] = min(max_norm / (norm + 1e-6), 1)  (clip_grad.py: clip_coef clamped to 1)
// `extra_sumsq` (nullable): sum of squares of further param groups, so that several flat buffers share one norm
__global__ __launch_bounds__(kThreads) void k_norm_final(const double* partials, int nblocks, const double* extra_sumsq,
                                                         float max_norm, float* out, double* sumsq_out) {
  __shared__ double sm[kThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kThreads) acc += partials[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double ss = sm[0];
    if (extra_sumsq) ss += extra_sumsq[0];
    if (sumsq_out) sumsq_out[0] = ss;
    const float norm = (float)sqrt(ss);
    const float coef = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = coef > 1.0f ? 1.0f : coef;  // NaN propagates like torch.clamp(max=1)
  }
}

struct SgdArgs {
  float* p;
  float* g;
  float* buf;
  int64_t n;
  float lr, momentum, dampening, weight_decay;
  int nesterov, first, write_grad, clip;
  const float* coef;  // nullable: device pointer to {total norm, clip coefficient}
};

__device__ __forceinline__ void sgd_elem(const SgdArgs& a, float coef, float& p, float& g, float& b) {
  g = g * coef;                                          // clip_grad_norm_: g.mul_(clip_coef_clamped)
  float d = a.weight_decay != 0.f ? g + a.weight_decay * p : g;   // grad.add(param, alpha=wd)
  if (a.momentum != 0.f) {
    b = a.first ? d : b * a.momentum + (1.f - a.dampening) * d;   // buf.mul_(m).add_(grad, alpha=1-dampening)
    d = a.nesterov ? d + a.momentum * b : b;
  }
  p = p + (-a.lr) * d;                                   // param.add_(grad, alpha=-lr)
}

__global__ __launch_bounds__(kThreads) void k_sgd(SgdArgs a) {
  const bool clip = a.clip != 0;
  const float coef = clip ? a.coef[1] : 1.0f;
  // overflow guard of the fp16 activation-gradient path (train.hip stores dz as fp16 with a loss scale).  The gate is the
  // NORM, not the coefficient: an inf gradient gives norm = inf and coef = max_norm / inf = 0 -- a perfectly valid-looking
  // coefficient -- and g * coef = inf * 0 = NaN would go into the weights and the momentum; a NaN gradient gives NaN.  Skip
  // the whole update instead (parameters, momentum, gradients untouched); the caller sees the norm in norm_and_coef[0].
  if (a.coef) {
    const float norm = a.coef[0];
    if (!(norm >= 0.0f && norm <= 3.402823466e38f)) return;
  }
  const int64_t n4 = a.n >> 2;
  float4* p4 = reinterpret_cast<float4*>(a.p);
  float4* g4 = reinterpret_cast<float4*>(a.g);
  float4* b4 = reinterpret_cast<float4*>(a.buf);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 p = p4[i], g = g4[i];
    float4 b = (a.momentum != 0.f && !a.first) ? b4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    sgd_elem(a, coef, p.x, g.x, b.x);
    sgd_elem(a, coef, p.y, g.y, b.y);
    sgd_elem(a, coef, p.z, g.z, b.z);
    sgd_elem(a, coef, p.w, g.w, b.w);
    p4[i] = p;
    if (a.momentum != 0.f) b4[i] = b;
    if (clip && a.write_grad) g4[i] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float p = a.p[i], g = a.g[i], b = (a.momentum != 0.f && !a.first) ? a.buf[i] : 0.f;
    sgd_elem(a, coef, p, g, b);
    a.p[i] = p;
    if (a.momentum != 0.f) a.buf[i] = b;
    if (clip && a.write_grad) a.g[i] = g;
  }
}

__global__ __launch_bounds__(kThreads) void k_scale(float* __restrict__ x, int64_t n, const float* coef) {
  const float c = coef[1];
  const int64_t n4 = n >> 2;
  float4* x4 = reinterpret_cast<float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 v = x4[i];
    v.x *= c; v.y *= c; v.z *= c; v.w *= c;
    x4[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[(n4 << 2) + threadIdx.x] *= c;
}

inline unsigned grid_for(int64_t n) {
  int64_t b = ((n >> 2) + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" {

size_t lfd_grad_norm_workspace_bytes(void) { return sizeof(double) * kMaxBlocks; }

int lfd_grad_norm_clip_coef_f32(const float* grads, int64_t n, float max_norm, const double* extra_sumsq,
                                void* workspace, size_t workspace_bytes, float* norm_and_coef, double* sumsq_out,
                                lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || !norm_and_coef || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  if (n > 0 && !grads) return LFD_ERR_INVALID_ARGUMENT;
  if ((reinterpret_cast<uintptr_t>(grads) & 15) != 0) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_grad_norm_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const unsigned g = grid_for(n);
  hipLaunchKernelGGL(k_sumsq_partial, dim3(g), dim3(kThreads), 0, st, grads, n, (double*)workspace);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(kThreads), 0, st, (const double*)workspace, (int)g, extra_sumsq,
                     max_norm, norm_and_coef, sumsq_out);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_scale_by_clip_coef_f32(float* grads, int64_t n, const float* norm_and_coef, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || !norm_and_coef) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!grads || (reinterpret_cast<uintptr_t>(grads) & 15) != 0) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(kThreads), 0, st, grads, n, norm_and_coef);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_sgd_step_f32(float* params, float* grads, float* momentum_buf, int64_t n, float lr, float momentum,
                     float dampening, float weight_decay, int32_t nesterov, int32_t first_step,
                     const float* norm_and_coef, int32_t apply_clip, int32_t write_clipped_grads, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!params || !grads || (momentum != 0.f && !momentum_buf)) return LFD_ERR_INVALID_ARGUMENT;
  if (nesterov && (momentum <= 0.f || dampening != 0.f)) return LFD_ERR_INVALID_ARGUMENT;  // torch.optim.SGD ctor check
  if (apply_clip && !norm_and_coef) return LFD_ERR_INVALID_ARGUMENT;
  if (((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
        reinterpret_cast<uintptr_t>(momentum_buf)) & 15) != 0)
    return LFD_ERR_INVALID_ARGUMENT;
  SgdArgs a{params, grads, momentum_buf, n, lr, momentum, dampening, weight_decay, nesterov, first_step,
            write_clipped_grads, apply_clip, norm_and_coef};
  hipLaunchKernelGGL(k_sgd, dim3(grid_for(n)), dim3(kThreads), 0, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
