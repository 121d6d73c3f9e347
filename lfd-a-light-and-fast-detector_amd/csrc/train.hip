// csrc/train.hip -- training-mode kernels of the LFDResNet conv stack (the reference runs
// `loss.backward()` (lfd/execution/hooks/optimizer_hook.py:28) through cuDNN conv + ATen BatchNorm autograd for the
// nn.Conv2d / nn.BatchNorm2d / ReLU (+ residual add) units of lfd/model/backbone/lfd_resnet.py:96-154, :354-439, :458-468).
//
// Activations and activation gradients are NHWC fp16 (gradients carry a power-of-two loss scale), parameters,
// BatchNorm statistics and parameter gradients fp32.  Per conv unit:
//   forward :  y = conv(x, W)            lfd_conv2d_nhwc_f16 (conv.hip, MFMA) / lfd_stem_conv0_train_fwd (3 input channels)
//              batch statistics of y     lfd_bn_train_stats_f16      (+ running_mean / running_var update)
//              z = relu?(bn(y) (+ res))  lfd_bn_train_apply_f16
//   backward:  g = dz * [z > 0];  dgamma, dbeta;  dy = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))
//                                        lfd_bn_train_bwd_f16        (2 reductions + 1 apply, deterministic)
//              dW = x (*) dy             lfd_conv_wgrad_nhwc_f16     (MFMA, k = pixels, LDS transpose reads)
//              dx = conv(dy, W^T flipped)  the forward conv kernel again (stride 2: after lfd_zero_insert2_nhwc_f16)
// All reductions use per-block partials + a fixed-order final stage (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kMaxC = 256;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

union Vec16 {
  uint4 u;
  h8 h;
};

__device__ __forceinline__ h8 ld8(const __half* p, int64_t vec) {
  Vec16 v;
  v.u = reinterpret_cast<const uint4*>(p)[vec];
  return v.h;
}
__device__ __forceinline__ void st8(__half* p, int64_t vec, h8 h) {
  Vec16 v;
  v.h = h;
  reinterpret_cast<uint4*>(p)[vec] = v.u;
}

// nn.Conv2d weight [cout][cin][ks][ks] fp32 -> MFMA fragment order of conv.hip, fp16:
//   out[ct][kstep][lane][e],  lane = 32*half + co_l,  kstep = (r*ks + s)*(cin/16) + q,  ci = 16q + 8*half + e.
// mode 1 packs the data-gradient conv instead: W'[co'][ci'][r][s] = W[ci'][co'][ks-1-r][ks-1-s] (co' over the forward
// conv's input channels).  Rows >= rows_valid of the LOGICAL conv are zero (output convs padded to 64 rows).
__device__ __forceinline__ void pack_vec(const float* __restrict__ w, int cout, int cin, int ks, int mode, int rows_valid,
                                         __half* __restrict__ out, int v) {
  const int li = mode ? cout : cin;                              // logical conv: input channels
  const int nq = li / 16, nk = ks * ks * nq;
  const int lane = v & 63, kstep = (v >> 6) % nk, ct = (v >> 6) / nk;
  const int half = lane >> 5, co = ct * 32 + (lane & 31);
  const int q = kstep % nq, tap = kstep / nq, r = tap / ks, sx = tap - r * ks;
  h8 o;
  for (int e = 0; e < 8; ++e) {
    const int ci = 16 * q + 8 * half + e;
    float f = 0.f;
    if (co < rows_valid) {
      if (mode == 0) f = w[(((size_t)co * cin + ci) * ks + r) * ks + sx];
      else f = w[(((size_t)ci * cin + co) * ks + (ks - 1 - r)) * ks + (ks - 1 - sx)];
    }
    o[e] = (_Float16)f;
  }
  st8(out, v, o);
}

__global__ __launch_bounds__(kThreads) void k_pack_weight(const float* __restrict__ w, int cout, int cin, int ks,
                                                         int mode, int rows_valid, __half* __restrict__ out) {
  const int lc = mode ? cin : cout, li = mode ? cout : cin;     // logical conv: lc output rows, li input channels
  const int total = (lc / 32) * ks * ks * (li / 16) * 64;
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v < total) pack_vec(w, cout, cin, ks, mode, rows_valid, out, v);
}

// all packs of a pass in ONE launch: job table in device memory (first_vec ascending), binary search per thread
__global__ __launch_bounds__(kThreads) void k_pack_weights(const lfd_pack_job_t* __restrict__ jobs, int njobs, int total) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= total) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_vec <= v) lo = mid; else hi = mid - 1;
  }
  const lfd_pack_job_t j = jobs[lo];
  pack_vec(j.w, j.cout, j.cin, j.ks, j.mode, j.rows_valid, (__half*)j.out, v - j.first_vec);
}

inline unsigned grid_for_vecs(int64_t vecs) {
  int64_t b = (vecs + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// fixed-pattern butterfly: every lane ends with the same, order-independent-of-scheduling sum
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

inline bool channels_ok(int c) { return c >= 8 && c <= kMaxC && (c & (c - 1)) == 0; }

// ---------------------------------------------------------------------------------------------------------
// Per-channel reductions over an NHWC fp16 tensor [m][c]: every thread owns one 8-channel group (the grid stride is
// a multiple of c/8), accumulates NQ quantities per channel in fp32, the block combines the threads of a group through
// LDS and writes partial[block][NQ][c]; `k_channel_final` sums the partials in fp64 in block order.
// ---------------------------------------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void block_channel_reduce(float (&acc)[NQ][8], int c, float* partial_out, int bx = -1) {
  if (bx < 0) bx = blockIdx.x;
  __shared__ float red[kThreads][NQ * 8 + 1];
  for (int q = 0; q < NQ; ++q)
    for (int e = 0; e < 8; ++e) red[threadIdx.x][q * 8 + e] = acc[q][e];
  __syncthreads();
  const int groups = c >> 3;
  for (int o = threadIdx.x; o < NQ * c; o += kThreads) {
    const int q = o / c, ch = o - q * c;
    const int cg = ch >> 3, e = ch & 7;
    float s = 0.f;
    for (int t = cg; t < kThreads; t += groups) s += red[t][q * 8 + e];
    partial_out[(size_t)bx * NQ * c + o] = s;
  }
}

__global__ __launch_bounds__(kThreads) void k_bn_stats_partial(const __half* __restrict__ y, int64_t vecs, int c,
                                                              float* partials) {
  float acc[2][8];
  for (int e = 0; e < 8; ++e) acc[0][e] = acc[1][e] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < vecs; v += (int64_t)gridDim.x * kThreads) {
    const h8 h = ld8(y, v);
    for (int e = 0; e < 8; ++e) {
      const float f = (float)h[e];
      acc[0][e] += f;
      acc[1][e] += f * f;
    }
  }
  block_channel_reduce<2>(acc, c, partials);
}

// stats[0][c] = mean, stats[1][c] = 1/sqrt(var + eps) (biased variance, as F.batch_norm normalises in training);
// running_mean / running_var updated with `momentum` and the unbiased variance (torch BatchNorm semantics)
__device__ __forceinline__ void bn_stats_final_body(const float* partials, int nblocks, int c, double m, float eps, float momentum,
                                                    float* running_mean, float* running_var, float* stats, int ch) {
  // one wave per channel: lanes stride over the block partials
  double s = 0.0, ss = 0.0;
  // (four partial rows requested before the first add: the loop is a chain of dependent round trips otherwise; adds in row order)
  int b = threadIdx.x;
  for (; b + 192 < nblocks; b += 256) {
    float u[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u[k] = partials[(size_t)(b + 64 * k) * 2 * c + ch];
      v[k] = partials[(size_t)(b + 64 * k) * 2 * c + c + ch];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += (double)u[k]; ss += (double)v[k]; }
  }
  for (; b < nblocks; b += 64) {
    s += (double)partials[(size_t)b * 2 * c + ch];
    ss += (double)partials[(size_t)b * 2 * c + c + ch];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (threadIdx.x != 0) return;
  const double mean = s / m;
  double var = ss / m - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[ch] = (float)mean;
  stats[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = m > 1.0 ? var * m / (m - 1.0) : var;
    running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mean);
    running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unb);
  }
}
__global__ __launch_bounds__(64) void k_bn_stats_final(const float* partials, int nblocks, int c, double m,
                                                      float eps, float momentum, float* running_mean,
                                                      float* running_var, float* stats) {
  bn_stats_final_body(partials, nblocks, c, m, eps, momentum, running_mean, running_var, stats, blockIdx.x);
}

// A per-level tensor [n, hw, c] that lives inside a LEVEL-CONCATENATED one [n, P, c] (the head activations of the training
// schedule): vector v of the level = pixel v / groups of image (pixel / hw) -> row img * P + p0 + pixel % hw of the concatenated
// tensor.  32-bit divisions: the tensors this is used on (necks at <= 80 x 80) have far fewer than 2^31 vectors.
struct RowMap {
  int hw_vecs;          // vectors per image of the level (hw * groups); 0: identity
  int64_t img_vecs;     // vectors per image of the concatenated tensor (P * groups)
  int64_t off_vecs;     // first vector of the level inside an image (p0 * groups)
};
__device__ __forceinline__ int64_t mapped(const RowMap& m, int64_t v) {
  if (!m.hw_vecs) return v;
  const unsigned img = (unsigned)v / (unsigned)m.hw_vecs, r = (unsigned)v - img * (unsigned)m.hw_vecs;
  return (int64_t)img * m.img_vecs + m.off_vecs + r;
}

// (All streaming passes below request their thread's FIRST vectors before the per-channel parameters: the parameter loads and the
//  data loads were two dependent round trips -- parameters, s_waitcnt vmcnt(0), then the loop's first load -- and on the small
//  maps, where a thread owns one or two vectors, that chain is most of the ~5 us a launch takes.)
__device__ __forceinline__ void bn_apply_body(const __half* __restrict__ y, int64_t vecs, int c, const float* __restrict__ stats,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const __half* __restrict__ res, int relu, __half* __restrict__ z, RowMap zmap,
                                              int bx, int nbx) {
  const int groups = c >> 3;
  const int64_t v0 = (int64_t)bx * kThreads + threadIdx.x, stride = (int64_t)nbx * kThreads;
  h8 h0, r0;
  if (v0 < vecs) {
    h0 = ld8(y, v0);
    if (res) r0 = ld8(res, v0);
  }
  const int cg = (int)(v0 & (groups - 1));      // channel counts are powers of two (channels_ok): no 64-bit division
  float a[8], b[8];
  for (int e = 0; e < 8; ++e) {
    const int ch = cg * 8 + e;
    a[e] = gamma[ch] * stats[c + ch];
    b[e] = beta[ch] - stats[ch] * a[e];
  }
  auto body = [&](int64_t v, const h8& h, const h8& r) {
    h8 o;
    for (int e = 0; e < 8; ++e) {
      float f = (float)h[e] * a[e] + b[e];
      if (res) f += (float)r[e];
      if (relu) f = fmaxf(f, 0.f);
      o[e] = (_Float16)f;
    }
    st8(z, mapped(zmap, v), o);
  };
  if (v0 >= vecs) return;
  body(v0, h0, r0);
  for (int64_t v = v0 + stride; v < vecs; v += stride) {
    const h8 h = ld8(y, v);
    h8 r;
    if (res) r = ld8(res, v);
    body(v, h, r);
  }
}
__global__ __launch_bounds__(kThreads) void k_bn_apply(const __half* __restrict__ y, int64_t vecs, int c,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const __half* __restrict__ res, int relu,
                                                      __half* __restrict__ z, RowMap zmap) {
  bn_apply_body(y, vecs, c, stats, gamma, beta, res, relu, z, zmap, blockIdx.x, gridDim.x);
}

// The forward tail of SEVERAL BatchNorm units in two launches (the neck units of the pyramid levels, see BnBwdJobs below): the
// one-wave-per-channel final of every level, then the apply pass of every level into the level-concatenated tensor.
struct BnFwdJob {
  const __half* y;
  int64_t vecs;
  int c, blocks, nrows;
  const float* rows;
  double m;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* stats;
  const float* gamma;
  const float* beta;
  RowMap zmap;
};
struct BnFwdJobs {
  BnFwdJob j[LFD_MAX_LEVELS];
  int n, relu;
  __half* z;
};
__global__ __launch_bounds__(64) void k_bn_stats_final_jobs(BnFwdJobs J) {
  const BnFwdJob& b = J.j[blockIdx.y];
  if ((int)blockIdx.x < b.c)
    bn_stats_final_body(b.rows, b.nrows, b.c, b.m, b.eps, b.momentum, b.running_mean, b.running_var, b.stats, blockIdx.x);
}
__global__ __launch_bounds__(kThreads) void k_bn_apply_jobs(BnFwdJobs J) {
  const BnFwdJob& b = J.j[blockIdx.y];
  if ((int)blockIdx.x < b.blocks)
    bn_apply_body(b.y, b.vecs, b.c, b.stats, b.gamma, b.beta, nullptr, J.relu, J.z, b.zmap, blockIdx.x, b.blocks);
}

// sums of g and g * xhat, g = dz * [ReLU passed]: mask from the stored output z when given (units with a residual
// input), else -- relu_y -- recomputed from y as [gamma * xhat + beta > 0] (saves reading z), else no ReLU
__device__ __forceinline__ void bn_bwd_partial_body(const __half* __restrict__ dz, const __half* __restrict__ y,
                                                    const __half* __restrict__ z, int64_t vecs, int c,
                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int relu_y, float* partials, RowMap dmap,
                                                    int bx, int nbx) {
  const int groups = c >> 3;
  const int64_t v0 = (int64_t)bx * kThreads + threadIdx.x, stride = (int64_t)nbx * kThreads;
  h8 d0, y0, z0;
  if (v0 < vecs) {
    d0 = ld8(dz, mapped(dmap, v0));
    y0 = ld8(y, v0);
    if (z) z0 = ld8(z, v0);
  }
  const int cg = (int)(v0 & (groups - 1));      // channel counts are powers of two (channels_ok): no 64-bit division
  float mean[8], rstd[8], acc[2][8], ga[8], be[8];
  for (int e = 0; e < 8; ++e) {
    mean[e] = stats[cg * 8 + e];
    rstd[e] = stats[c + cg * 8 + e];
    ga[e] = relu_y ? gamma[cg * 8 + e] : 0.f;
    be[e] = relu_y ? beta[cg * 8 + e] : 0.f;
    acc[0][e] = acc[1][e] = 0.f;
  }
  auto body = [&](const h8& d, const h8& yy, const h8& zz) {
    for (int e = 0; e < 8; ++e) {
      float g = (float)d[e];
      const float xh = ((float)yy[e] - mean[e]) * rstd[e];
      if (z && !((float)zz[e] > 0.f)) g = 0.f;
      if (relu_y && !(ga[e] * xh + be[e] > 0.f)) g = 0.f;
      acc[0][e] += g;
      acc[1][e] += g * xh;
    }
  };
  if (v0 < vecs) {
    body(d0, y0, z0);
    for (int64_t v = v0 + stride; v < vecs; v += stride) {
      const h8 d = ld8(dz, mapped(dmap, v)), yy = ld8(y, v);
      h8 zz;
      if (z) zz = ld8(z, v);
      body(d, yy, zz);
    }
  }
  block_channel_reduce<2>(acc, c, partials, bx);
}
__global__ __launch_bounds__(kThreads) void k_bn_bwd_partial(const __half* __restrict__ dz, const __half* __restrict__ y,
                                                            const __half* __restrict__ z, int64_t vecs, int c,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu_y, float* partials,
                                                            RowMap dmap) {
  bn_bwd_partial_body(dz, y, z, vecs, c, stats, gamma, beta, relu_y, partials, dmap, blockIdx.x, gridDim.x);
}

// sums[0][c] = sum g (= dbeta * scale), sums[1][c] = sum g*xhat (= dgamma * scale); parameter gradients unscaled
__device__ __forceinline__ void bn_bwd_final_body(const float* partials, int nblocks, int c, float inv_scale, int accumulate,
                                                  float* sums, float* dgamma, float* dbeta, int ch) {
  double s = 0.0, sx = 0.0;
  int b = threadIdx.x;
  for (; b + 192 < nblocks; b += 256) {        // four rows in flight, adds in row order (see k_bn_stats_final)
    float u[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u[k] = partials[(size_t)(b + 64 * k) * 2 * c + ch];
      v[k] = partials[(size_t)(b + 64 * k) * 2 * c + c + ch];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += (double)u[k]; sx += (double)v[k]; }
  }
  for (; b < nblocks; b += 64) {
    s += (double)partials[(size_t)b * 2 * c + ch];
    sx += (double)partials[(size_t)b * 2 * c + c + ch];
  }
  s = wave_sum_d(s);
  sx = wave_sum_d(sx);
  if (threadIdx.x != 0) return;
  sums[ch] = (float)s;
  sums[c + ch] = (float)sx;
  if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)(s * (double)inv_scale);
  if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)(sx * (double)inv_scale);
}
__global__ __launch_bounds__(64) void k_bn_bwd_final(const float* partials, int nblocks, int c, float inv_scale,
                                                    int accumulate, float* sums, float* dgamma, float* dbeta) {
  bn_bwd_final_body(partials, nblocks, c, inv_scale, accumulate, sums, dgamma, dbeta, blockIdx.x);
}

__device__ __forceinline__ void bn_bwd_apply_body(const __half* __restrict__ dz, const __half* __restrict__ y,
                                                  const __half* __restrict__ z, int64_t vecs, int c,
                                                  const float* __restrict__ stats, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int relu_y, const float* __restrict__ sums,
                                                  float inv_m, __half* __restrict__ dy, __half* __restrict__ g_out, RowMap dmap,
                                                  int bx, int nbx) {
  const int groups = c >> 3;
  const int64_t v0 = (int64_t)bx * kThreads + threadIdx.x, stride = (int64_t)nbx * kThreads;
  h8 d0, y0, z0;
  if (v0 < vecs) {
    d0 = ld8(dz, mapped(dmap, v0));
    y0 = ld8(y, v0);
    if (z) z0 = ld8(z, v0);
  }
  const int cg = (int)(v0 & (groups - 1));      // channel counts are powers of two (channels_ok): no 64-bit division
  float mean[8], rstd[8], a[8], mg[8], mgx[8], ga[8], be[8];
  for (int e = 0; e < 8; ++e) {
    const int ch = cg * 8 + e;
    mean[e] = stats[ch];
    rstd[e] = stats[c + ch];
    ga[e] = gamma[ch];
    be[e] = relu_y ? beta[ch] : 0.f;
    a[e] = gamma[ch] * rstd[e];
    mg[e] = sums[ch] * inv_m;
    mgx[e] = sums[c + ch] * inv_m;
  }
  auto body = [&](int64_t v, const h8& d, const h8& yy, const h8& zz) {
    h8 o, go;
    for (int e = 0; e < 8; ++e) {
      float g = (float)d[e];
      const float xh = ((float)yy[e] - mean[e]) * rstd[e];
      if (z && !((float)zz[e] > 0.f)) g = 0.f;
      if (relu_y && !(ga[e] * xh + be[e] > 0.f)) g = 0.f;
      o[e] = (_Float16)(a[e] * (g - mg[e] - xh * mgx[e]));
      go[e] = (_Float16)g;
    }
    st8(dy, v, o);
    if (g_out) st8(g_out, v, go);
  };
  if (v0 >= vecs) return;
  body(v0, d0, y0, z0);
  for (int64_t v = v0 + stride; v < vecs; v += stride) {
    const h8 d = ld8(dz, mapped(dmap, v)), yy = ld8(y, v);
    h8 zz;
    if (z) zz = ld8(z, v);
    body(v, d, yy, zz);
  }
}
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(const __half* __restrict__ dz, const __half* __restrict__ y,
                                                          const __half* __restrict__ z, int64_t vecs, int c,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int relu_y,
                                                          const float* __restrict__ sums, float inv_m,
                                                          __half* __restrict__ dy, __half* __restrict__ g_out, RowMap dmap) {
  bn_bwd_apply_body(dz, y, z, vecs, c, stats, gamma, beta, relu_y, sums, inv_m, dy, g_out, dmap, blockIdx.x, gridDim.x);
}

// The backward of SEVERAL BatchNorm units in three launches (round 4: the neck units of the pyramid levels, mutually independent,
// each three ~5 us launches on its own): jobs = blockIdx.y, every job with the block count its own call would use -- the values
// are those of the per-unit call sequence, bit for bit.
struct BnBwdJob {
  const __half* dz;
  const __half* y;
  int64_t vecs;
  int c, blocks;
  const float* stats;
  const float* gamma;
  const float* beta;
  float* rows;
  float* sums;
  float* dgamma;
  float* dbeta;
  __half* dy;
  float inv_m;
  RowMap dmap;
};
struct BnBwdJobs {
  BnBwdJob j[LFD_MAX_LEVELS];
  int n, relu_y, accumulate;
  float inv_scale;
};
__global__ __launch_bounds__(kThreads) void k_bn_bwd_partial_jobs(BnBwdJobs J) {
  const BnBwdJob& b = J.j[blockIdx.y];
  if ((int)blockIdx.x < b.blocks)
    bn_bwd_partial_body(b.dz, b.y, nullptr, b.vecs, b.c, b.stats, b.gamma, b.beta, J.relu_y, b.rows, b.dmap, blockIdx.x, b.blocks);
}
__global__ __launch_bounds__(64) void k_bn_bwd_final_jobs(BnBwdJobs J) {
  const BnBwdJob& b = J.j[blockIdx.y];
  if ((int)blockIdx.x < b.c) bn_bwd_final_body(b.rows, b.blocks, b.c, J.inv_scale, J.accumulate, b.sums, b.dgamma, b.dbeta, blockIdx.x);
}
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply_jobs(BnBwdJobs J) {
  const BnBwdJob& b = J.j[blockIdx.y];
  if ((int)blockIdx.x < b.blocks)
    bn_bwd_apply_body(b.dz, b.y, nullptr, b.vecs, b.c, b.stats, b.gamma, b.beta, J.relu_y, b.sums, b.inv_m, b.dy, nullptr, b.dmap,
                      blockIdx.x, b.blocks);
}

// out[n, 2i, 2j, :] = in[n, i, j, :], zero elsewhere (the gradient of a stride-2 subsampling)
__global__ __launch_bounds__(kThreads) void k_zero_insert2(const __half* __restrict__ in, int n, int hi, int wi,
                                                          int c, int ho, int wo, __half* __restrict__ out) {
  const int groups = c >> 3;
  const int64_t vecs = (int64_t)n * ho * wo * groups;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < vecs; v += (int64_t)gridDim.x * kThreads) {
    // 32-bit index arithmetic (the host refuses 2^31 vectors; channel counts are powers of two): lesson 38
    const unsigned vu = (unsigned)v;
    const int cg = (int)(vu & (unsigned)(groups - 1));
    unsigned p = vu / (unsigned)groups;
    const unsigned pw = p / (unsigned)wo;
    const int x = (int)(p - pw * (unsigned)wo);
    const unsigned ph = pw / (unsigned)ho;
    const int yy = (int)(pw - ph * (unsigned)ho);
    const int img = (int)ph;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (!(x & 1) && !(yy & 1) && (yy >> 1) < hi && (x >> 1) < wi)
      val = reinterpret_cast<const uint4*>(in)[(((int64_t)img * hi + (yy >> 1)) * wi + (x >> 1)) * groups + cg];
    reinterpret_cast<uint4*>(out)[v] = val;
  }
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm in training (the head towers: nn.GroupNorm(16, 128) behind every 1x1 conv, lfd_head.py:97-105): statistics
// per (image, group) over h*w pixels x 8 channels -- a group is exactly one 16-byte vector of an NHWC pixel.
// grid = (blocks per image, images); a thread owns one group (the stride is a multiple of the group count).
// ---------------------------------------------------------------------------------------------------------
// A "virtual image" of the GroupNorm kernels = one (image, segment) pair: blockIdx.y = img * nseg + seg.  Plain tensors have one
// segment per image (the whole image); the LEVEL-CONCATENATED head activations [N, P, C] of the training schedule (round 4: the
// shared towers run ONCE over all pyramid levels) have one segment per level -- statistics per (image, level) as in the
// reference, where every level is a tensor of its own (lfd_head.py:164-185).
struct GnGeom {
  int nseg;
  int64_t img_vecs;                    // vectors (8 channels) per image over all segments
  int64_t off[LFD_MAX_LEVELS];         // first vector of a segment inside its image
  int64_t vecs[LFD_MAX_LEVELS];        // vectors of a segment
};
__device__ __forceinline__ void gn_segment(const GnGeom& G, size_t* base, int64_t* vecs) {
  const int seg = (int)(blockIdx.y % G.nseg), img = (int)(blockIdx.y / G.nseg);
  *base = (size_t)img * G.img_vecs + G.off[seg];
  *vecs = G.vecs[seg];
}

__global__ __launch_bounds__(kThreads) void k_gn_stats_partial(const __half* __restrict__ y, GnGeom G, int g,
                                                              float* partials) {
  __shared__ float red[kThreads][2];
  size_t base_;
  int64_t vecs_per_img;
  gn_segment(G, &base_, &vecs_per_img);
  const __half* yi = y + base_ * 8;
  float s = 0.f, ss = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < vecs_per_img; v += (int64_t)gridDim.x * kThreads) {
    const h8 h = ld8(yi, v);
    for (int e = 0; e < 8; ++e) {
      const float f = (float)h[e];
      s += f;
      ss += f * f;
    }
  }
  red[threadIdx.x][0] = s;
  red[threadIdx.x][1] = ss;
  __syncthreads();
  if (threadIdx.x < 2 * g) {
    const int q = threadIdx.x / g, cg = threadIdx.x - q * g;
    float a = 0.f;
    for (int t = cg; t < kThreads; t += g) a += red[t][q];
    partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * g + threadIdx.x] = a;
  }
}

// stats[img][0][g] = mean, stats[img][1][g] = rstd.  One wave per image: lane = part * g + group, the 64 / g parts stride over the
// block partials (independent loads in flight instead of one dependent chain of `nblocks` L2 round trips: 15 -> ~3 us for the
// 64 partial blocks of a large map), then a fixed-order butterfly over the parts -- deterministic.
__global__ __launch_bounds__(64) void k_gn_stats_final(const float* partials, int nblocks, int g, GnGeom G, float eps,
                                                      float* stats) {
  const double m = (double)(G.vecs[blockIdx.x % G.nseg] / g) * 8.0;       // pixels of the segment x 8 channels per group
  const int lane = threadIdx.x, cg = lane % g, part = lane / g, nparts = 64 / g;     // g is a power of two <= 32
  double s = 0.0, ss = 0.0;
#pragma unroll 4
  for (int b = part; b < nblocks; b += nparts) {
    const float* p = partials + ((size_t)blockIdx.x * nblocks + b) * 2 * g;
    s += (double)p[cg];
    ss += (double)p[g + cg];
  }
  for (int off = g; off < 64; off <<= 1) {
    s += __shfl_xor(s, off, 64);
    ss += __shfl_xor(ss, off, 64);
  }
  if (part != 0) return;
  const double mean = s / m;
  double var = ss / m - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[(size_t)blockIdx.x * 2 * g + cg] = (float)mean;
  stats[(size_t)blockIdx.x * 2 * g + g + cg] = (float)(1.0 / sqrt(var + (double)eps));
}

// FOLD: `stats` is written HERE -- every workgroup of an image adds that image's k_gn_stats_partial rows itself (<= 64 rows
// of 2g floats, fp64, row order; the arithmetic of k_gn_stats_final), workgroup x == 0 stores them for the backward pass: the
// apply pass does not wait for a k_gn_stats_final launch (10 dependent ~4-6 us launches of a WIDERFACE_LFD_S iteration)
template <bool FOLD>
__global__ __launch_bounds__(kThreads) void k_gn_apply(const __half* __restrict__ y, GnGeom G, int g,
                                                      float* __restrict__ stats, const float* __restrict__ partials,
                                                      int nblocks, float eps,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int relu,
                                                      __half* __restrict__ z) {
  __shared__ float sst[FOLD ? 64 : 1];
  size_t base;
  int64_t vecs_per_img;
  gn_segment(G, &base, &vecs_per_img);
  const double m = (double)(vecs_per_img / g) * 8.0;       // pixels of the segment x 8 channels per group
  const int64_t v0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
  h8 h0;
  if (v0 < vecs_per_img) h0 = ld8(y, base + v0);        // (requested before the statistics: one round trip, not two)
  if constexpr (FOLD) {
    if (threadIdx.x < g) {
      const int cg_ = threadIdx.x;
      double s = 0.0, ss = 0.0;
#pragma unroll 4
      for (int b = 0; b < nblocks; ++b) {
        const float* p = partials + ((size_t)blockIdx.y * nblocks + b) * 2 * g;
        s += (double)p[cg_];
        ss += (double)p[g + cg_];
      }
      const double mean = s / m;
      double var = ss / m - mean * mean;
      if (var < 0.0) var = 0.0;
      const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
      sst[cg_] = mf;
      sst[g + cg_] = rf;
      if (blockIdx.x == 0) {
        stats[(size_t)blockIdx.y * 2 * g + cg_] = mf;
        stats[(size_t)blockIdx.y * 2 * g + g + cg_] = rf;
      }
    }
    __syncthreads();
  }
  const int cg = (int)((blockIdx.x * kThreads + threadIdx.x) & (g - 1));   // g: a power of two (gn_ok)
  const float mean = FOLD ? sst[cg] : stats[(size_t)blockIdx.y * 2 * g + cg];
  const float rstd = FOLD ? sst[g + cg] : stats[(size_t)blockIdx.y * 2 * g + g + cg];
  float a[8], b[8];
  for (int e = 0; e < 8; ++e) {
    a[e] = gamma[cg * 8 + e] * rstd;
    b[e] = beta[cg * 8 + e] - mean * a[e];
  }
  auto body = [&](int64_t v, const h8& h) {
    h8 o;
    for (int e = 0; e < 8; ++e) {
      float f = (float)h[e] * a[e] + b[e];
      if (relu) f = fmaxf(f, 0.f);
      o[e] = (_Float16)f;
    }
    st8(z, base + v, o);
  };
  if (v0 >= vecs_per_img) return;
  body(v0, h0);
  for (int64_t v = v0 + stride; v < vecs_per_img; v += stride) body(v, ld8(y, base + v));
}

// per (image, group): sum g*gamma, sum g*gamma*xhat;  per (image, channel): sum g*xhat, sum g
__global__ __launch_bounds__(kThreads) void k_gn_bwd_partial(const __half* __restrict__ dz,
                                                            const __half* __restrict__ y,
                                                            const __half* __restrict__ z, GnGeom G, int g,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, float* pgroup,
                                                            float* pchan) {
  __shared__ float red[kThreads][18];
  size_t base;
  int64_t vecs_per_img;
  gn_segment(G, &base, &vecs_per_img);
  const int64_t v0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
  h8 d0, y0, z0;
  if (v0 < vecs_per_img) {
    d0 = ld8(dz, base + v0);
    y0 = ld8(y, base + v0);
    if (z) z0 = ld8(z, base + v0);
  }
  const int cg = (int)((blockIdx.x * kThreads + threadIdx.x) & (g - 1));   // g: a power of two (gn_ok)
  const float mean = stats[(size_t)blockIdx.y * 2 * g + cg], rstd = stats[(size_t)blockIdx.y * 2 * g + g + cg];
  float gam[8], acc[18];
  for (int e = 0; e < 8; ++e) gam[e] = gamma[cg * 8 + e];
  for (int i = 0; i < 18; ++i) acc[i] = 0.f;
  auto body = [&](const h8& d, const h8& yy, const h8& zz) {
    for (int e = 0; e < 8; ++e) {
      float gr = (float)d[e];
      if (z && !((float)zz[e] > 0.f)) gr = 0.f;
      const float xh = ((float)yy[e] - mean) * rstd;
      acc[0] += gr * gam[e];
      acc[1] += gr * gam[e] * xh;
      acc[2 + e] += gr * xh;
      acc[10 + e] += gr;
    }
  };
  if (v0 < vecs_per_img) {
    body(d0, y0, z0);
    for (int64_t v = v0 + stride; v < vecs_per_img; v += stride) {
      const h8 d = ld8(dz, base + v), yy = ld8(y, base + v);
      h8 zz;
      if (z) zz = ld8(z, base + v);
      body(d, yy, zz);
    }
  }
  for (int i = 0; i < 18; ++i) red[threadIdx.x][i] = acc[i];
  __syncthreads();
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int c = g * 8;
  for (int o = threadIdx.x; o < 2 * g + 2 * c; o += kThreads) {
    int col, grp;
    if (o < 2 * g) {
      col = o / g;             // 0: sum g*gamma, 1: sum g*gamma*xhat
      grp = o - col * g;
    } else {
      const int q = (o - 2 * g) / c, ch = (o - 2 * g) - q * c;   // q 0: dgamma, 1: dbeta
      grp = ch >> 3;
      col = 2 + q * 8 + (ch & 7);
    }
    float a = 0.f;
    for (int t = grp; t < kThreads; t += g) a += red[t][col];
    if (o < 2 * g) pgroup[blk * 2 * g + o] = a;
    else pchan[blk * 2 * c + (o - 2 * g)] = a;
  }
}

// gsums[img][2][g] (per image), and dgamma / dbeta over all images (+= when accumulate: heads shared by the levels)
__global__ __launch_bounds__(64) void k_gn_bwd_final(const float* pgroup, const float* pchan, int n, int nblocks,
                                                    int g, float inv_scale, int accumulate, float* gsums,
                                                    float* dgamma, float* dbeta) {
  const int c = g * 8;
  const int o = blockIdx.x;              // one wave per output: [0, n*2g) group sums, then 2c channel sums
  double s = 0.0;
  if (o < n * 2 * g) {
    const int img = o / (2 * g), j = o - img * 2 * g;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += (double)pgroup[((size_t)img * nblocks + b) * 2 * g + j];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) gsums[o] = (float)s;
  } else {
    const int j = o - n * 2 * g;
    // (eight rows requested before the first add: the level-concatenated head has n * nblocks ~ 4000 rows, a chain of 62
    //  dependent L2 round trips per lane otherwise -- 27 us per launch; adds in row order within a lane)
    const int rows = n * nblocks;
    int b = threadIdx.x;
    for (; b + 7 * 64 < rows; b += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pchan[(size_t)(b + 64 * u) * 2 * c + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < rows; b += 64) s += (double)pchan[(size_t)b * 2 * c + j];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
      float* dst = j < c ? dgamma + j : dbeta + (j - c);
      const float v = (float)(s * (double)inv_scale);
      *dst = accumulate ? *dst + v : v;
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_gn_bwd_apply(const __half* __restrict__ dz,
                                                          const __half* __restrict__ y,
                                                          const __half* __restrict__ z, GnGeom G, int g,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ gsums,
                                                          __half* __restrict__ dy) {
  size_t base;
  int64_t vecs_per_img;
  gn_segment(G, &base, &vecs_per_img);
  const float inv_m = (float)(1.0 / ((double)(vecs_per_img / g) * 8.0));
  const int64_t v0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, stride = (int64_t)gridDim.x * kThreads;
  h8 d0, y0, z0;
  if (v0 < vecs_per_img) {
    d0 = ld8(dz, base + v0);
    y0 = ld8(y, base + v0);
    if (z) z0 = ld8(z, base + v0);
  }
  const int cg = (int)((blockIdx.x * kThreads + threadIdx.x) & (g - 1));   // g: a power of two (gn_ok)
  const float mean = stats[(size_t)blockIdx.y * 2 * g + cg], rstd = stats[(size_t)blockIdx.y * 2 * g + g + cg];
  const float m1 = gsums[(size_t)blockIdx.y * 2 * g + cg] * inv_m, m2 = gsums[(size_t)blockIdx.y * 2 * g + g + cg] * inv_m;
  float gam[8];
  for (int e = 0; e < 8; ++e) gam[e] = gamma[cg * 8 + e];
  auto body = [&](int64_t v, const h8& d, const h8& yy, const h8& zz) {
    h8 o;
    for (int e = 0; e < 8; ++e) {
      float gr = (float)d[e];
      if (z && !((float)zz[e] > 0.f)) gr = 0.f;
      const float xh = ((float)yy[e] - mean) * rstd;
      o[e] = (_Float16)(rstd * (gr * gam[e] - m1 - xh * m2));
    }
    st8(dy, base + v, o);
  };
  if (v0 >= vecs_per_img) return;
  body(v0, d0, y0, z0);
  for (int64_t v = v0 + stride; v < vecs_per_img; v += stride) {
    const h8 d = ld8(dz, base + v), yy = ld8(y, base + v);
    h8 zz;
    if (z) zz = ld8(z, base + v);
    body(v, d, yy, zz);
  }
}

// ---------------------------------------------------------------------------------------------------------
// First stem conv in training: conv3x3 stride 2 pad 1, 3 -> c channels, NCHW fp32 image -> NHWC fp16 pre-norm output
// (lfd_resnet.py:358 / :378 `nn.Conv2d(input_channels, stem_channels, 3, 2, 1, bias=False)`).  27 taps on the VALU in
// fp32 (0.5 % of the network's FLOPs); one thread = one output pixel x 8 channels, weights [27][c] in LDS.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_conv0_fwd(const float* __restrict__ x, int n, int h, int w, int c,
                                                       const float* __restrict__ wt, __half* __restrict__ y) {
  __shared__ float sw[27 * 64];
  for (int i = threadIdx.x; i < 27 * c; i += kThreads) {
    const int co = i / 27, t = i - co * 27;  // OIHW: wt[co][ci][ky][kx], t = ci*9 + ky*3 + kx
    sw[t * c + co] = wt[i];
  }
  __syncthreads();
  const int ho = (h + 1) / 2, wo = (w + 1) / 2, groups = c >> 3;
  const int64_t vecs = (int64_t)n * ho * wo * groups;
  for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < vecs; v += (int64_t)gridDim.x * kThreads) {
    const int cg = (int)(v % groups);
    int64_t p = v / groups;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho);
    const int img = (int)(p / ho);
    float acc[8];
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ci = 0; ci < 3; ++ci)
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = 2 * ox + kx - 1;
          float xv = 0.f;
          if (iy >= 0 && iy < h && ix >= 0 && ix < w) xv = x[(((int64_t)img * 3 + ci) * h + iy) * w + ix];
          const float* wr = sw + (ci * 9 + ky * 3 + kx) * c + cg * 8;
          for (int e = 0; e < 8; ++e) acc[e] += xv * wr[e];
        }
      }
    h8 o;
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
    st8(y, v, o);
  }
}

// dW[co][t] = sum over output pixels of dy[p][co] * x[tap t of p]; thread = (co, pixel sub-slice), all lanes of a wave
// read the same 27 input values (one broadcast transaction each).  partial[block][27][c]
__global__ __launch_bounds__(kThreads) void k_conv0_wgrad_partial(const float* __restrict__ x,
                                                                 const __half* __restrict__ dy, int n, int h, int w,
                                                                 int c, int pixels_per_block, float* partials) {
  __shared__ float red[kThreads][28];
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int64_t total = (int64_t)n * ho * wo;
  const int subs = kThreads / c;               // c = 32: 8 sub-slices, c = 64: 4
  const int co = threadIdx.x % c, sub = threadIdx.x / c;
  float acc[27];
  for (int t = 0; t < 27; ++t) acc[t] = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * pixels_per_block;
  int64_t p1 = p0 + pixels_per_block;
  if (p1 > total) p1 = total;
  for (int64_t p = p0 + sub; p < p1; p += subs) {
    const int ox = (int)(p % wo);
    const int64_t q = p / wo;
    const int oy = (int)(q % ho);
    const int img = (int)(q / ho);
    const float g = __half2float(dy[p * c + co]);
    for (int ci = 0; ci < 3; ++ci)
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = 2 * ox + kx - 1;
          float xv = 0.f;
          if (iy >= 0 && iy < h && ix >= 0 && ix < w) xv = x[(((int64_t)img * 3 + ci) * h + iy) * w + ix];
          acc[ci * 9 + ky * 3 + kx] += g * xv;
        }
      }
  }
  for (int t = 0; t < 27; ++t) red[threadIdx.x][t] = acc[t];
  __syncthreads();
  for (int o = threadIdx.x; o < 27 * c; o += kThreads) {
    const int t = o / c, ch = o - t * c;
    float s = 0.f;
    for (int k = 0; k < subs; ++k) s += red[k * c + ch][t];
    partials[(size_t)blockIdx.x * 27 * c + o] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same two first-conv kernels on MFMA (default): the 27 taps (ci, ky, kx) are the contraction index padded to 32,
//   forward :  y[px][co]  = sum_t W[co][t] * patch[px][t]      A = W (rows = co), B = patches (cols = 32 pixels)
//   wgrad   :  dW[co][t]  = sum_px dy[px][co] * patch[px][t]   A = dy^T (rows = co, k = pixels; transposing LDS read),
//                                                              B = patches (cols = 32 taps, k = 8 consecutive pixels)
// patches are gathered straight from the NCHW fp32 image (L1/L2-resident: every input value is used by ~7 taps) and
// rounded to fp16; accumulation fp32.
// ---------------------------------------------------------------------------------------------------------
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4v lds_s4v;

__device__ __forceinline__ h8 tr_frag(const char* base, uint32_t off0, uint32_t off1) {
  union { s4v s[2]; h8 h; } u;
  u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(base + off0));
  u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(base + off1));
  return u.h;
}

typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));      // three consecutive floats at any 4-byte boundary

struct TapTab {
  int off[16];     // ci*h*w + (ky-1)*w + (kx-1), relative to (2*oy, 2*ox)
  int kyx[16];     // (ky << 8) | kx, or -1 for the padding taps >= 27
};

__device__ __forceinline__ void tap_of(int t, int h, int w, int& off, int& kyx) {
  if (t >= 27) { off = 0; kyx = -1; return; }
  const int ci = t / 9, r = t - ci * 9, ky = r / 3, kx = r - ky * 3;
  off = ci * h * w + (ky - 1) * w + (kx - 1);
  kyx = (ky << 8) | kx;
}

__global__ __launch_bounds__(kThreads, 3) void k_conv0_fwd_mfma(const float* __restrict__ x, int n, int h, int w, int c,
                                                            const float* __restrict__ wt, __half* __restrict__ y,
                                                            float* stat_partials) {
  __shared__ __attribute__((aligned(16))) uint4 s_stage[kThreads / 64][256];     // per wave: 32 pixels x 128 B
  // stat_partials (c == 64 only): row [blockIdx.x][2][64] of per-channel sums of the stored fp16 outputs and of their squares,
  // for the train-mode BatchNorm behind this conv (lfd_resnet.py:358-359) -- every lane copies chunk l & 7 of every line it
  // stores, so 8 + 8 sums per lane last the whole walk (the same scheme as the STATS instantiations of conv_impl.h)
  float st_s[8], st_q[8];
  for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
  const int l = threadIdx.x & 63, hk = l >> 5;
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int64_t total = (int64_t)n * ho * wo;
  const int nct = c / 32;
  // A fragments: W[ct*32 + (l&31)][16*ks + 8*hk + j]
  h8 wa[2][2];
  for (int ct = 0; ct < 2; ++ct)
    for (int ks = 0; ks < 2; ++ks)
      for (int j = 0; j < 8; ++j) {
        const int t = 16 * ks + 8 * hk + j, co = ct * 32 + (l & 31);
        wa[ct][ks][j] = (_Float16)((t < 27 && co < c) ? wt[co * 27 + t] : 0.f);
      }
  TapTab tab;
  for (int s = 0; s < 16; ++s) tap_of(16 * (s >> 3) + 8 * hk + (s & 7), h, w, tab.off[s], tab.kyx[s]);
  const int64_t wave0 = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * 32;
  const int64_t stride = (int64_t)gridDim.x * (kThreads / 64) * 32;
  for (int64_t p0 = wave0; p0 < total; p0 += stride) {
    const int64_t p = p0 + (l & 31);
    const bool pv = p < total;
    // 32-bit index arithmetic (the host refuses maps of 2^31 elements; lesson 38)
    const unsigned pc = pv ? (unsigned)p : 0u;
    const unsigned qq = pc / (unsigned)wo;
    const int ox = (int)(pc - qq * (unsigned)wo);
    const unsigned im = qq / (unsigned)ho;
    const int oy = (int)(qq - im * (unsigned)ho);
    const int img = (int)im;
    const float* xb = x + ((int64_t)img * 3 * h + 2 * oy) * w + 2 * ox;
    h8 b[2];
    // WIDE path (groups without a lane on the left / right image border: 8 of 10 at 640 columns): the three kx taps of an image
    // row are 12 consecutive bytes -- six 12-byte loads per lane cover its 16 taps (rows r = ci * 3 + ky: half 0 of the wave
    // needs rows 0 1 2 5 6 7, half 1 rows 2 3 4 5 8) instead of sixteen 4-byte gathers.  The vector memory path spends ~one clock
    // per lane ADDRESS on these stride-2 accesses (the kernel sat at ~52 clocks per load instruction and CU whatever its
    // occupancy or VALU count, lesson 42): with a third of the load instructions a third wave per SIMD pays too
    // (__launch_bounds__(256, 3)): 197 -> 179 (wide loads) -> 158 us (+ occupancy).  Same values, same tap order.
    const bool lr_inner = pv && ox > 0 && 2 * ox + 1 < w;
    if (__builtin_amdgcn_ballot_w64(pv && !lr_inner) == 0) {
      f3u T[6];
#pragma unroll
      for (int sl = 0; sl < 6; ++sl) {
        constexpr int R0[6] = {0, 1, 2, 5, 6, 7}, R1[6] = {2, 3, 4, 5, 8, 8};
        const int r = hk ? R1[sl] : R0[sl];
        const int ci = r / 3, ky = r - ci * 3;
        const int iy = 2 * oy + ky - 1;
        const bool rv = pv && iy >= 0 && iy < h;
        const float* src = rv ? xb + (ci * h + ky - 1) * w - 1 : x;      // (unconditional load from a safe address + select)
        const f3u v = *reinterpret_cast<const f3u*>(src);
        T[sl] = rv ? v : f3u{0.f, 0.f, 0.f};
      }
      // tap s of half 0 / half 1 = T[slot][kx]; taps 27..31 (half 1, s >= 11) are padding
      constexpr int S0[16] = {0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 4, 4, 4, 5, 5, 5}, K0[16] = {0, 1, 2, 0, 1, 2, 0, 1, 1, 2, 0, 1, 2, 0, 1, 2};
      constexpr int S1[16] = {0, 1, 1, 1, 2, 2, 2, 3, 4, 4, 4, 4, 4, 4, 4, 4}, K1[16] = {2, 0, 1, 2, 0, 1, 2, 0, 0, 1, 2, 0, 0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float v0 = T[S0[s]][K0[s]];
        const float v1 = s < 11 ? T[S1[s]][K1[s]] : 0.f;
        b[s >> 3][s & 7] = (_Float16)(hk ? v1 : v0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int ky = tab.kyx[s] >> 8, kx = tab.kyx[s] & 255;
        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
        const bool ok = pv && tab.kyx[s] >= 0 && iy >= 0 && iy < h && ix >= 0 && ix < w;
        // unconditional load from a safe address + select: `ok ? xb[off] : 0` compiles to a branch around every load, and the
        // 16 loads of a pixel then complete one after the other instead of together
        const float raw = xb[ok ? tab.off[s] : 0];
        b[s >> 3][s & 7] = (_Float16)(ok ? raw : 0.f);
      }
    }
    if (c == 64) {
      // 64 output channels: the wave holds the whole 128-byte line of each of its 32 pixels.  Straight from the accumulator
      // layout that is 16 eight-byte pieces per pixel from two lanes (eight store instructions touching 32 lines each:
      // 272 us for the 419 MB of a 32 x 320 x 320 map); staged through LDS -- [pixel][8 chunks of 16 B], chunk XOR (pixel & 7) --
      // consecutive lanes write consecutive chunks: four fully coalesced 16-byte stores.  LDS operations of one wave execute
      // in order: no barrier.
      char* stg = reinterpret_cast<char*>(s_stage[threadIdx.x >> 6]);
      const int px = l & 31;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ct][0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ct][1], b[1], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // rows 8g + 4hk + 0..3 of this pixel: 4 consecutive channels
          h4 o;
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)acc[4 * g + e];
          *reinterpret_cast<h4*>(stg + px * 128 + (((ct * 4 + g) ^ (px & 7)) << 4) + 8 * hk) = o;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = j * 64 + l;
        const int q = idx >> 3, ck = idx & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + q * 128 + ((ck ^ (q & 7)) << 4));
        const bool ok = p0 + q < total;
        if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + (p0 + q) * 128 + ck * 16) = v;
        if (stat_partials) {
          const uint32_t wd[4] = {ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(wd[k] & 0xffffu));
            const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(wd[k] >> 16));
            st_s[2 * k] += lo;  st_q[2 * k] += lo * lo;
            st_s[2 * k + 1] += hi;  st_q[2 * k + 1] += hi * hi;
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      if (ct >= nct) break;
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ct][0], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ct][1], b[1], acc, 0, 0, 0);
      if (pv) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // rows 8g + 4hk + 0..3 of this pixel: 4 consecutive channels
          h4 o;
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)acc[4 * g + e];
          *reinterpret_cast<h4*>(y + p * c + ct * 32 + 8 * g + 4 * hk) = o;
        }
      }
    }
  }
  if (stat_partials) {
    // lanes l, l + 8, ... hold the same chunk: butterfly, then the four waves through LDS in wave order
    for (int e = 0; e < 8; ++e)
      for (int d = 32; d >= 8; d >>= 1) {
        st_s[e] += __shfl_xor(st_s[e], d);
        st_q[e] += __shfl_xor(st_q[e], d);
      }
    __syncthreads();
    float* red = reinterpret_cast<float*>(&s_stage[0][0]);        // [4 waves][2][64]
    const int wv = threadIdx.x >> 6;
    if (l < 8)
      for (int e = 0; e < 8; ++e) {
        red[(wv * 2 + 0) * 64 + l * 8 + e] = st_s[e];
        red[(wv * 2 + 1) * 64 + l * 8 + e] = st_q[e];
      }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int q = threadIdx.x >> 6, ch = threadIdx.x & 63;
      stat_partials[(size_t)blockIdx.x * 128 + threadIdx.x] =
          ((red[(0 * 2 + q) * 64 + ch] + red[(1 * 2 + q) * 64 + ch]) + red[(2 * 2 + q) * 64 + ch]) + red[(3 * 2 + q) * 64 + ch];
    }
  }
}

// partial[block][co 64][t 32]
// BN (round 4): `dy` is dL/dz of the first unit -- the gradient w.r.t. the OUTPUT of its BatchNorm + ReLU -- and the kernel applies
// BatchNorm's backward to every 16-byte chunk on its way into LDS (the arithmetic of k_bn_bwd_apply, bit for bit: mask recomputed
// from y, dy = a (g - mean(g) - xhat mean(g xhat)) rounded to fp16).  The first conv has no data gradient, so its dy has exactly
// this one consumer: the separate apply pass (read dz + y, write dy: 1.26 GB at 32 x 640 x 640, 245 us) and this kernel's read of
// dy disappear, for one more tensor read here.
struct Conv0BnArgs {
  const __half* y;          // pre-norm output of the first conv
  const float* stats;       // [2][c] mean, rstd
  const float* gamma;
  const float* beta;
  const float* sums;        // [2][c] sum g, sum g * xhat (k_bn_bwd_final)
  float inv_m;
};

template <bool BN>
__global__ __launch_bounds__(kThreads) void k_conv0_wgrad_mfma(const float* __restrict__ x,
                                                              const __half* __restrict__ dy, int n, int h, int w,
                                                              int c, float* partials, Conv0BnArgs bn) {
  // FOUR k-steps (16-pixel segments) per wave and loop trip, visited in the order s, s + S, s + 2 S, s + 3 S of the rolled loop (the
  // same sums, bit for bit): all their loads -- 8 x 16 B of dy and 32 gathered floats per lane -- are requested before the first LDS
  // write.  One k-step per trip was a chain load -> LDS -> transposing read -> 2 MFMAs with one round trip to HBM each: 265 us for
  // the 576 MB of a 32 x 640 x 640 batch (2.2 TB/s).
  constexpr int U = 4;
  __shared__ __attribute__((aligned(16))) char smem_w[4 * U * 16 * 144 > 4 * 64 * 32 * 4 ? 4 * U * 16 * 144 : 4 * 64 * 32 * 4];
  char (*sdy)[U][16 * 144] = reinterpret_cast<char (*)[U][16 * 144]>(smem_w);
  float (*red)[64 * 32] = reinterpret_cast<float (*)[64 * 32]>(smem_w);        // after the loop (behind a barrier)
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hk = l >> 5;
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int segs = (wo + 15) / 16;                       // k-step = 16 consecutive output pixels of one row
  const int64_t total = (int64_t)n * ho * segs;
  const int t = l & 31;                                   // this lane's tap (B column)
  int toff, tkyx;
  tap_of(t, h, w, toff, tkyx);
  const int ky = tkyx >> 8, kx = tkyx & 255;
  f16v acc[2];
  for (int ct = 0; ct < 2; ++ct)
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int i16 = l & 15, grp = (l >> 4) & 1;
  uint32_t a_off[2];
  for (int r = 0; r < 2; ++r) a_off[r] = (8 * hk + 4 * r + (i16 >> 2)) * 144 + (16 * grp + 4 * (i16 & 3)) * 2;
  // BN: per-channel constants of this lane's 16-byte chunk (chunk l & 7 in both of its loads: 64 k does not change the low bits)
  float p_mean[8], p_rstd[8], p_a[8], p_mg[8], p_mgx[8], p_ga[8], p_be[8];
  if constexpr (BN) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = (l & 7) * 8 + e;
      const bool cv = ch < c;
      p_mean[e] = cv ? bn.stats[ch] : 0.f;
      p_rstd[e] = cv ? bn.stats[c + ch] : 0.f;
      p_ga[e] = cv ? bn.gamma[ch] : 0.f;
      p_be[e] = cv ? bn.beta[ch] : 0.f;
      p_a[e] = p_ga[e] * p_rstd[e];
      p_mg[e] = cv ? bn.sums[ch] * bn.inv_m : 0.f;
      p_mgx[e] = cv ? bn.sums[c + ch] * bn.inv_m : 0.f;
    }
  }
  const int64_t S = (int64_t)gridDim.x * 4;
  for (int64_t s0 = (int64_t)blockIdx.x * 4 + wave; s0 < total; s0 += U * S) {
    uint4 dv[U][2];
    uint4 yv[BN ? U : 1][2];
    bool okd[BN ? U : 1][2];
    float raw[U][8];
    bool okj[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = s0 + u * S;
      const bool sv = s < total;
      const unsigned sc = sv ? (unsigned)s : 0u;          // 32-bit index arithmetic, see k_conv0_fwd_mfma
      const unsigned qq = sc / (unsigned)segs;
      const int seg = (int)(sc - qq * (unsigned)segs);
      const unsigned im = qq / (unsigned)ho;
      const int oy = (int)(qq - im * (unsigned)ho);
      const int img = (int)im;
      const int ox0 = seg * 16;
      // dy[16 px][64 ch] (zero beyond the row / channel range): 128 16-byte chunks, 2 per lane
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = l + 64 * k;
        const int px = i >> 3, c8 = i & 7;
        const bool in = sv && ox0 + px < wo && c8 * 8 < c;
        const int64_t eo = in ? (((int64_t)img * ho + oy) * wo + ox0 + px) * c + c8 * 8 : 0;
        dv[u][k] = *reinterpret_cast<const uint4*>(dy + eo);
        if constexpr (BN) yv[u][k] = *reinterpret_cast<const uint4*>(bn.y + eo);
        if (!in) dv[u][k] = make_uint4(0, 0, 0, 0);      // (g = 0 and, BN: the chunk is zeroed again after the transform)
        if constexpr (BN) okd[u][k] = in;
      }
      // B: patch[ox0 + 8hk + j][t], j = 0..7
      const int iy = 2 * oy + ky - 1;
      const float* xr = x + ((int64_t)img * 3 * h + 2 * oy) * w + toff;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ox = ox0 + 8 * hk + j, ix = 2 * ox + kx - 1;
        okj[u][j] = sv && tkyx >= 0 && ox < wo && iy >= 0 && iy < h && ix >= 0 && ix < w;
        raw[u][j] = *(okj[u][j] ? xr + 2 * ox : x);      // unconditional load + select (see k_conv0_fwd_mfma)
      }
    }
    if constexpr (BN) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          union { uint4 q; _Float16 hh[8]; } d, yy, o;
          d.q = dv[u][k]; yy.q = yv[u][k];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float g = (float)d.hh[e];
            const float xh = ((float)yy.hh[e] - p_mean[e]) * p_rstd[e];
            if (!(p_ga[e] * xh + p_be[e] > 0.f)) g = 0.f;
            o.hh[e] = (_Float16)(p_a[e] * (g - p_mg[e] - xh * p_mgx[e]));
          }
          dv[u][k] = okd[u][k] ? o.q : make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      char* my = sdy[wave][u];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = l + 64 * k;
        *reinterpret_cast<uint4*>(my + (i >> 3) * 144 + (i & 7) * 16) = dv[u][k];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      h8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = (_Float16)(okj[u][j] ? raw[u][j] : 0.f);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const h8 af = tr_frag(sdy[wave][u], a_off[0] + ct * 64, a_off[1] + ct * 64);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b, acc[ct], 0, 0, 0);
      }
    }
  }
  __syncthreads();     // red aliases the staging tiles of waves that may still have been contracting
  // combine the 4 waves in fixed order; D layout: lane = column (tap), reg r -> row (co) 8*(r>>2) + 4*hk + (r&3)
  for (int ct = 0; ct < 2; ++ct)
    for (int r = 0; r < 16; ++r) red[wave][(ct * 32 + 8 * (r >> 2) + 4 * hk + (r & 3)) * 32 + t] = acc[ct][r];
  __syncthreads();
  for (int o = threadIdx.x; o < 64 * 32; o += kThreads)
    partials[(size_t)blockIdx.x * 2048 + o] = red[0][o] + red[1][o] + red[2][o] + red[3][o];
}

// generic sum of partial[block][count] -> out[perm(i)] * inv_scale, one wave per output; perm: 0 identity,
// 1: i = t*c + co -> co*taps + t (OIHW of the first conv)
__global__ __launch_bounds__(64) void k_sum_partials(const float* partials, int nblocks, int count, float inv_scale,
                                                    int perm, int c, int taps, int accumulate, float* out) {
  const int i = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += (double)partials[(size_t)b * count + i];
  s = wave_sum_d(s);
  if (threadIdx.x != 0) return;
  int o = i;
  if (perm == 1) {
    const int t = i / c, co = i - t * c;
    o = co * taps + t;
  } else if (perm == 2) {          // i = co*32 + t (taps padded to 32) -> co*taps + t
    const int co = i >> 5, t = i & 31;
    if (t >= taps || co >= c) return;
    o = co * taps + t;
  }
  out[o] = (accumulate ? out[o] : 0.f) + (float)(s * (double)inv_scale);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of a conv (ks 1|3, stride 1|2, pad ks/2) on MFMA:
//   dW[co][tap][ci] = sum over output pixels p of dy[p][co] * x[pixel of tap at p][ci]
// is a GEMM whose contraction index is the PIXEL, while both operands are channel-contiguous (NHWC).  The tiles
// (8 x 16 output pixels of dy, the matching input halo of x; 64 channels each, 144-byte pixel pitch) are staged in LDS
// as they lie in memory and the MFMA operands (row = channel, 8 consecutive k = pixels) are fetched with gfx950's
// transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block, lane i receives
// channel i of the 4 pixels); the per-lane pixel address makes the stride-2 gather free.
// One workgroup = one 64(cout) x 64(cin) block of dW for all taps, persistent over pixel tiles with the accumulators
// (ks*ks 32x32 tiles per wave) resident; wave = (cout half, cin half).  partial[wg][tap][co 64][ci 64] fp32, summed in
// block order by k_wgrad_final (deterministic).
// ---------------------------------------------------------------------------------------------------------
template <int KS, int S>
struct WgradCfg {
  static constexpr int TH = 8, TW = 16, PAD = KS / 2;
  static constexpr int XH = (TH - 1) * S + KS, XW = (TW - 1) * S + KS;
  static constexpr int PITCH = 144;
  static constexpr int DY_BYTES = TH * TW * PITCH;
  static constexpr int X_BYTES = XH * XW * PITCH;
  static constexpr int LDS_BYTES = DY_BYTES + X_BYTES;
};

struct WgradArgs {
  const __half* x;
  const __half* dy;
  int n, h, w, ho, wo, cin, cout;
  int tiles_y, tiles_x;
  float* partials;
  int vtaps;      // 9: this launch of the 1x1 kernel is ONE TAP (blockIdx.z) of a 3x3 conv -- x is read at the tap's offset
  // XPRO: `x` is the PRE-normalisation output y of the BatchNorm + ReLU unit whose activation this conv consumed (a unit whose
  // z tensor is never stored: the forward conv normalised its operand on the way into the MFMA, conv_impl.h PRO); the tile
  // loader re-forms z = fp16(max(y * a + b, 0)) with k_bn_apply's arithmetic
  const float* xpro_stats;   // [2][cin] mean | rstd
  const float* xpro_gamma;
  const float* xpro_beta;
};

template <int KS, int S, bool XPRO = false>
__global__ __launch_bounds__(kThreads) void k_wgrad(WgradArgs a) {
  static_assert(!XPRO || (KS == 1 && S == 1), "XPRO: 1x1 stride-1 convs");
  using C = WgradCfg<KS, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sdy = smem;
  char* sx = smem + C::DY_BYTES;
  const int nib = (a.cin + 63) / 64;
  const int cb = blockIdx.y / nib, ib = blockIdx.y - cb * nib;
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  const int ch = wave >> 1, ih = wave & 1;
  const bool active = (cb * 64 + ch * 32 < a.cout) && (ib * 64 + ih * 32 < a.cin);
  // tap-split form (small maps, KS == 1 instantiations only): dW[tap] = sum over pixels of dy (x) x shifted by the tap
  const int vtap = (KS == 1 && a.vtaps == 9) ? (int)blockIdx.z : 0;
  const int offy = (KS == 1 && a.vtaps == 9) ? vtap / 3 - 1 : 0, offx = (KS == 1 && a.vtaps == 9) ? vtap % 3 - 1 : 0;
  f16v acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // lane constants of the transposing reads: read r in {0,1} covers k = 8*(l>>5) + 4r + ((l&15)>>2)
  const int i16 = l & 15, grp = (l >> 4) & 1, kh = l >> 5;
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int kp = 8 * kh + 4 * r + (i16 >> 2);
    a_off[r] = kp * C::PITCH + (ch * 32 + 16 * grp + 4 * (i16 & 3)) * 2;
    b_off[r] = kp * S * C::PITCH + (ih * 32 + 16 * grp + 4 * (i16 & 3)) * 2;
  }
  const int tiles = a.n * a.tiles_y * a.tiles_x;
  // Staging in two halves: global -> registers (tile_load) and registers -> LDS (tile_store).  Stride-1 variants keep the
  // NEXT tile's loads in flight while the current tile is contracted (4 + 6 uint4 per thread); the stride-2 halo tile is
  // 18 uint4 per thread -- too many registers next to the 144 accumulators -- and stays synchronous.
  constexpr int NDY = (C::TH * C::TW * 8 + kThreads - 1) / kThreads;
  constexpr int NX = (C::XH * C::XW * 8 + kThreads - 1) / kThreads;
  constexpr bool PF = (S == 1);
  uint4 rdy[NDY], rx[NX];
  float xa[XPRO ? 8 : 1], xb[XPRO ? 8 : 1];      // a thread stages the same 8 channels (tid & 7) of every x pixel it loads
  if constexpr (XPRO) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = ib * 64 + (tid & 7) * 8 + e;
      const bool okc = c < a.cin;
      xa[e] = okc ? a.xpro_gamma[c] * a.xpro_stats[a.cin + c] : 0.f;
      xb[e] = okc ? a.xpro_beta[c] - a.xpro_stats[c] * xa[e] : 0.f;
    }
  }
  auto tile_load = [&](int t) {
    const int tx = t % a.tiles_x;
    const int q = t / a.tiles_x;
    const int ty = q % a.tiles_y;
    const int img = q / a.tiles_y;
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i = tid + j * kThreads;
      const int c8 = i & 7, p = i >> 3;
      const int oy = ty * C::TH + p / C::TW, ox = tx * C::TW + p % C::TW;
      const int co = cb * 64 + c8 * 8;
      const bool ok = i < C::TH * C::TW * 8 && oy < a.ho && ox < a.wo && co < a.cout;
      const int64_t off = ok ? ((((int64_t)img * a.ho + oy) * a.wo + ox) * a.cout + co) : 0;
      const uint4 v = *reinterpret_cast<const uint4*>(a.dy + off);       // unconditional (clamped) load: no branch
      rdy[j] = ok ? v : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + j * kThreads;
      const int c8 = i & 7, p = i >> 3;
      const int iy = ty * C::TH * S - C::PAD + p / C::XW + offy, ix = tx * C::TW * S - C::PAD + p % C::XW + offx;
      const int ci = ib * 64 + c8 * 8;
      const bool ok = i < C::XH * C::XW * 8 && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w && ci < a.cin;
      const int64_t off = ok ? ((((int64_t)img * a.h + iy) * a.w + ix) * a.cin + ci) : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + off);
      if constexpr (XPRO) {
        v = __builtin_bit_cast(uint4, lfd_affine_relu_f16x8(__builtin_bit_cast(h8, v), xa, xb));
      }
      rx[j] = ok ? v : make_uint4(0, 0, 0, 0);
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int i = tid + j * kThreads;
      if (i < C::TH * C::TW * 8) *reinterpret_cast<uint4*>(sdy + (i >> 3) * C::PITCH + (i & 7) * 16) = rdy[j];
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + j * kThreads;
      if (i < C::XH * C::XW * 8) *reinterpret_cast<uint4*>(sx + (i >> 3) * C::PITCH + (i & 7) * 16) = rx[j];
    }
  };
  int t = blockIdx.x;
  if (PF && t < tiles) tile_load(t);
  for (; t < tiles; t += gridDim.x) {
    if (!PF) tile_load(t);
    tile_store();
    __syncthreads();
    if (PF && t + (int)gridDim.x < tiles) tile_load(t + gridDim.x);     // in flight during the contraction
    if (active) {
#pragma unroll 1
      for (int py = 0; py < C::TH; ++py) {
        const uint32_t arow = py * C::TW * C::PITCH;
        const h8 af = tr_frag(sdy, arow + a_off[0], arow + a_off[1]);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const uint32_t brow = ((py * S + ky) * C::XW + kx) * C::PITCH;
            const h8 bf = tr_frag(sx, brow + b_off[0], brow + b_off[1]);
            acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[ky * KS + kx], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  // ---- partial[wg][block][tap][co 64][ci 64]: D layout lane = column (ci), reg r -> row (co) 8*(r>>2) + 4*(l>>5) + (r&3)
  float* out = (KS == 1 && a.vtaps == 9)
                   ? a.partials + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 9 + vtap) * (64 * 64)
                   : a.partials + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (KS * KS * 64 * 64);
#pragma unroll
  for (int t = 0; t < KS * KS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = ch * 32 + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
      const int ci = ih * 32 + (l & 31);
      out[(t * 64 + co) * 64 + ci] = active ? acc[t][r] : 0.f;
    }
}

// dW (fp32, OIHW [cout][cin][ks][ks]) = inv_scale * sum over workgroups of the partials.  A block sums 128 consecutive
// outputs: 32 lanes x one 16-byte load per partial row and slice (512 contiguous bytes per request instead of 128), 8
// slices of the workgroup loop in parallel, then the slices in fixed order -- the same summation order as ever, so the
// results did not change when the loads were widened (14.5 -> ~8 us per call on the 37 MB of a 3x3 64 -> 64 layer's partials).
__global__ __launch_bounds__(kThreads) void k_wgrad_final(const float* partials, int nwg, int nblk, int cin, int cout,
                                                         int taps, float inv_scale, int accumulate, float* dw) {
  __shared__ double red[8][128];
  const int per_blk = taps * 64 * 64;
  const int lane32 = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int i = blockIdx.x * 128 + lane32 * 4;           // 128 consecutive outputs per block, 4 per thread
  const int blk = i / per_blk, j = i - blk * per_blk;    // per_blk is a multiple of 128: a block never straddles
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  // eight partial rows requested before the first is added (one dependent HBM round trip per row otherwise: 32 of them for
  // 256 workgroups); the adds stay in row order
  const float* src = partials + (size_t)blk * per_blk + j;
  const size_t row = (size_t)nblk * per_blk;
  int g = slice;
  for (; g + 56 < nwg; g += 64) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(g + 8 * u) * row);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
  }
  for (; g < nwg; g += 8) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)g * row);
    s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
  }
  red[slice][lane32 * 4 + 0] = s0; red[slice][lane32 * 4 + 1] = s1;
  red[slice][lane32 * 4 + 2] = s2; red[slice][lane32 * 4 + 3] = s3;
  __syncthreads();
  if (threadIdx.x >= 128) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
  const int jj = j - lane32 * 4 + (int)threadIdx.x;      // this thread's output inside the block's 128
  const int t = jj / 4096, co_l = (jj >> 6) & 63, ci_l = jj & 63;
  const int nib = (cin + 63) / 64;
  const int co = (blk / nib) * 64 + co_l, ci = (blk % nib) * 64 + ci_l;
  if (co >= cout || ci >= cin) return;
  float* dst = dw + ((size_t)co * cin + ci) * taps + t;
  *dst = (accumulate ? *dst : 0.f) + (float)(s * (double)inv_scale);
}

// The final stage of MANY weight gradients in one launch (round 4): the training schedule runs every k_wgrad with its own
// partial buffer and sums all of them after the last one -- 49 dependent ~8 us launches of an iteration become one.  A
// block of 128 consecutive outputs finds its job in the table (first_block: running sum of the head jobs' blocks); jobs
// chained through `next` are further partial sets of the SAME dW (a conv shared by the pyramid levels: lfd_head.py:67-82),
// summed after the head job's rows in chain order -- slices walk the rows as in k_wgrad_final, one rounding at the end.
__global__ __launch_bounds__(kThreads) void k_wgrad_final_batched(const lfd_wgrad_job_t* __restrict__ jobs, int njobs) {
  __shared__ double red[8][128];
  __shared__ int sjob;
  if (threadIdx.x == 0) {
    int f = -1;
    for (int j = 0; j < njobs; ++j) {
      const int fb = jobs[j].first_block;
      if (fb >= 0 && (int)blockIdx.x >= fb && (int)blockIdx.x < fb + jobs[j].nblk * jobs[j].taps * 32) f = j;
    }
    sjob = f;
  }
  __syncthreads();
  const int hj = sjob;
  if (hj < 0) return;
  const lfd_wgrad_job_t J = jobs[hj];
  const int per_blk = J.taps * 64 * 64;
  const int lane32 = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - J.first_block) * 128 + lane32 * 4;
  const int blk = i / per_blk, j = i - blk * per_blk;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int k = hj; k >= 0; k = jobs[k].next) {
    const float* src = jobs[k].partials + (size_t)blk * per_blk + j;
    const int nwg = jobs[k].nwg;
    const size_t row = (size_t)J.nblk * per_blk;
    int g = slice;
    for (; g + 56 < nwg; g += 64) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(g + 8 * u) * row);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
    }
    for (; g < nwg; g += 8) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)g * row);
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
  }
  red[slice][lane32 * 4 + 0] = s0; red[slice][lane32 * 4 + 1] = s1;
  red[slice][lane32 * 4 + 2] = s2; red[slice][lane32 * 4 + 3] = s3;
  __syncthreads();
  if (threadIdx.x >= 128) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
  const int jj = j - lane32 * 4 + (int)threadIdx.x;
  const int t = jj / 4096, co_l = (jj >> 6) & 63, ci_l = jj & 63;
  const int nib = (J.cin + 63) / 64;
  const int co = (blk / nib) * 64 + co_l, ci = (blk % nib) * 64 + ci_l;
  if (co < J.co_lo || co >= J.co_hi || ci >= J.cin) return;
  float* dst = J.dw + ((size_t)(co - J.co_lo) * J.cin + ci) * J.taps + t;
  *dst = (J.accumulate ? *dst : 0.f) + (float)(s * (double)J.inv_scale);
}

// dst[i] (+)= sum over rows of src[row][i], fp64, rows in order: the per-level private copies of small shared gradients
// (GroupNorm weight / bias of the shared towers, the output convs' biases) once the levels ran on their own streams
__global__ __launch_bounds__(kThreads) void k_rows_sum_batched(const lfd_rowsum_job_t* __restrict__ jobs) {
  const lfd_rowsum_job_t J = jobs[blockIdx.x];
  for (int i = threadIdx.x; i < J.count; i += kThreads) {
    double s = 0.0;
    for (int r = 0; r < J.nrows; ++r) s += (double)J.src[(size_t)r * J.row_stride + i];
    J.dst[i] = (J.accumulate ? J.dst[i] : 0.f) + (float)s;
  }
}

template <int KS, int S, bool XPRO = false>
int launch_wgrad(const WgradArgs& a0, int nwg, int nblk, hipStream_t st) {
  using C = WgradCfg<KS, S>;
  static unsigned long long attr_set_mask = 0;
  const int attr_set_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_set_mask, attr_set_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<KS, S, XPRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_set_mask, attr_set_dev);
  }
  WgradArgs a = a0;
  a.tiles_y = (a.ho + C::TH - 1) / C::TH;
  a.tiles_x = (a.wo + C::TW - 1) / C::TW;
  hipLaunchKernelGGL((k_wgrad<KS, S, XPRO>), dim3(nwg, nblk, a.vtaps == 9 ? 9 : 1), dim3(kThreads), C::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

constexpr int kWgradMaxWg = 256;   // sizes the partial buffer: 256 x 4 blocks x 9 taps x 64 x 64 floats
constexpr int kWgradWgCap = 1024;  // most persistent workgroups per (cout, cin) block

}  // namespace

// csrc/conv_stats.hip: the conv kernels that leave BatchNorm's partial sums behind hand their rows to the same final pass
int lfd_bn_stats_final_launch(const float* partials, int nblocks, int channels, double pixels, float eps, float momentum,
                              float* running_mean, float* running_var, float* stats, hipStream_t st) {
  if (nblocks < 1 || nblocks > kMaxBlocks || !channels_ok(channels)) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_bn_stats_final, dim3(channels), dim3(64), 0, st, partials, nblocks, channels, pixels, eps, momentum,
                     running_mean, running_var, stats);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

extern "C" {

size_t lfd_train_workspace_bytes(void) {
  // the largest user: wgrad partials, 256 workgroups x 4 blocks (128 x 128 channels) x 9 taps x 64 x 64 floats
  return (size_t)kWgradMaxWg * 4 * 9 * 64 * 64 * sizeof(float) + 4096;
}

int lfd_pack_conv_weight_train_f16(const float* weight_oihw, int32_t cout, int32_t cin, int32_t ks, int32_t mode,
                                   int32_t rows_valid, void* packed, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!weight_oihw || !packed || (ks != 1 && ks != 3) || (mode != 0 && mode != 1)) return LFD_ERR_INVALID_ARGUMENT;
  const int lc = mode ? cin : cout, li = mode ? cout : cin;
  if (lc < 32 || (lc & 31) || li < 16 || (li & 15)) return LFD_ERR_INVALID_ARGUMENT;
  if (mode == 1 && rows_valid != lc) return LFD_ERR_INVALID_ARGUMENT;   // padding only for forward packs
  if (mode == 0 && (rows_valid < 1 || rows_valid > lc)) return LFD_ERR_INVALID_ARGUMENT;
  const int total = (lc / 32) * ks * ks * (li / 16) * 64;
  hipLaunchKernelGGL(k_pack_weight, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, weight_oihw, cout, cin,
                     ks, mode, rows_valid, (__half*)packed);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_pack_conv_weights_train_f16(const lfd_pack_job_t* jobs_device, int32_t njobs, int32_t total_vecs,
                                    lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (njobs < 0 || total_vecs < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (njobs == 0 || total_vecs == 0) return LFD_OK;
  if (!jobs_device) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_pack_weights, dim3((total_vecs + kThreads - 1) / kThreads), dim3(kThreads), 0, st, jobs_device, njobs,
                     total_vecs);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_bn_train_stats_f16(const void* y, int64_t pixels, int32_t channels, float eps, float momentum,
                           float* running_mean, float* running_var, void* workspace, size_t workspace_bytes,
                           float* stats, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!y || !stats || !workspace || pixels < 1 || !channels_ok(channels)) return LFD_ERR_INVALID_ARGUMENT;
  if ((running_mean == nullptr) != (running_var == nullptr)) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int64_t vecs = pixels * (channels / 8);
  const unsigned g = grid_for_vecs(vecs);
  float* partials = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(k_bn_stats_partial, dim3(g), dim3(kThreads), 0, st, (const __half*)y, vecs, channels, partials);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_stats_final, dim3(channels), dim3(64), 0, st, partials, (int)g, channels, (double)pixels, eps,
                     momentum, running_mean, running_var, stats);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

static bool row_map(RowMap* m, int64_t n, int64_t hw, int64_t points_total, int64_t point0, int groups) {
  if (n < 1 || hw < 1 || point0 < 0 || point0 + hw > points_total || n * hw * groups >= ((int64_t)1 << 31)) return false;
  m->hw_vecs = (int)(hw * groups); m->img_vecs = points_total * groups; m->off_vecs = point0 * groups;
  return true;
}

int lfd_bn_train_apply_f16(const void* y, int64_t pixels, int32_t channels, const float* stats, const float* gamma,
                           const float* beta, const void* residual, int32_t relu, void* z, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!y || !stats || !gamma || !beta || !z || pixels < 1 || !channels_ok(channels)) return LFD_ERR_INVALID_ARGUMENT;
  const int64_t vecs = pixels * (channels / 8);
  hipLaunchKernelGGL(k_bn_apply, dim3(grid_for_vecs(vecs)), dim3(kThreads), 0, st, (const __half*)y, vecs, channels,
                     stats, gamma, beta, (const __half*)residual, relu, (__half*)z, RowMap{});
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_bn_train_apply_into_f16(const void* y, int32_t n, int64_t hw, int32_t channels, const float* stats, const float* gamma,
                                const float* beta, int32_t relu, void* z_concat, int64_t points_total, int64_t point0,
                                lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RowMap m{};
  if (!y || !stats || !gamma || !beta || !z_concat || !channels_ok(channels) || !row_map(&m, n, hw, points_total, point0, channels / 8))
    return LFD_ERR_INVALID_ARGUMENT;
  const int64_t vecs = (int64_t)n * hw * (channels / 8);
  hipLaunchKernelGGL(k_bn_apply, dim3(grid_for_vecs(vecs)), dim3(kThreads), 0, st, (const __half*)y, vecs, channels,
                     stats, gamma, beta, (const __half*)nullptr, relu, (__half*)z_concat, m);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

static int bn_bwd(const void* dz, const RowMap& dmap, const void* y, const void* z, int32_t relu, int64_t pixels, int32_t channels,
                  const float* stats, const float* gamma, const float* beta, float inv_scale, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, float* dgamma, float* dbeta, void* dy, void* g_out, hipStream_t st, int sum_rows = 0) {
  if (!dz || !y || !stats || !gamma || !dy || !workspace || pixels < 1 || !channels_ok(channels))
    return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int relu_y = (relu && !z) ? 1 : 0;       // no stored output given: the ReLU mask is recomputed from y
  if (relu_y && !beta) return LFD_ERR_INVALID_ARGUMENT;
  if (!relu) z = nullptr;
  const int64_t vecs = pixels * (channels / 8);
  const unsigned g = grid_for_vecs(vecs);
  float* partials = reinterpret_cast<float*>(workspace);
  float* sums = partials + (size_t)kMaxBlocks * 2 * kMaxC;
  // (Round 4 tried folding k_bn_bwd_final into the apply pass for small maps -- the sums pass on 64 workgroups, every apply
  // workgroup re-adding the 64 rows: the 21 saved launches (6.3 us each) were paid back by the slower 64-workgroup sums pass
  // (+6 us each) and the re-add (+4 us each): 7.06 against 6.98 ms per iteration.  Three launches it stays.)
  if (sum_rows < 0 || sum_rows > kMaxBlocks) return LFD_ERR_INVALID_ARGUMENT;
  if (!sum_rows) {      // (else: the rows are already there -- lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16 left them)
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(g), dim3(kThreads), 0, st, (const __half*)dz, (const __half*)y,
                       (const __half*)z, vecs, channels, stats, gamma, beta, relu_y, partials, dmap);
    LFD_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_bn_bwd_final, dim3(channels), dim3(64), 0, st, partials, sum_rows ? sum_rows : (int)g, channels, inv_scale,
                     accumulate, sums, dgamma, dbeta);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(g), dim3(kThreads), 0, st, (const __half*)dz, (const __half*)y,
                     (const __half*)z, vecs, channels, stats, gamma, beta, relu_y, sums, (float)(1.0 / (double)pixels),
                     (__half*)dy, (__half*)g_out, dmap);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_bn_train_bwd_f16(const void* dz, const void* y, const void* z, int32_t relu, int64_t pixels, int32_t channels,
                         const float* stats, const float* gamma, const float* beta, float inv_scale,
                         int32_t accumulate, void* workspace,
                         size_t workspace_bytes, float* dgamma, float* dbeta, void* dy, void* g_out,
                         lfd_stream_t stream) {
  return bn_bwd(dz, RowMap{}, y, z, relu, pixels, channels, stats, gamma, beta, inv_scale, accumulate, workspace, workspace_bytes,
                dgamma, dbeta, dy, g_out, reinterpret_cast<hipStream_t>(stream));
}

int lfd_bn_train_bwd_rows_f16(const void* dz, const void* y, int64_t pixels, int32_t channels, const float* stats,
                              const float* gamma, const float* beta, float inv_scale, int32_t accumulate, int32_t sum_rows,
                              void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, void* dy, lfd_stream_t stream) {
  if (sum_rows < 1) return LFD_ERR_INVALID_ARGUMENT;
  return bn_bwd(dz, RowMap{}, y, nullptr, 1, pixels, channels, stats, gamma, beta, inv_scale, accumulate, workspace, workspace_bytes,
                dgamma, dbeta, dy, nullptr, reinterpret_cast<hipStream_t>(stream), sum_rows);
}

int lfd_bn_train_bwd_from_f16(const void* dz_concat, int64_t points_total, int64_t point0, const void* y, int32_t relu, int32_t n,
                              int64_t hw, int32_t channels, const float* stats, const float* gamma, const float* beta,
                              float inv_scale, int32_t accumulate, void* workspace, size_t workspace_bytes, float* dgamma,
                              float* dbeta, void* dy, lfd_stream_t stream) {
  RowMap m{};
  if (!channels_ok(channels) || !row_map(&m, n, hw, points_total, point0, channels / 8)) return LFD_ERR_INVALID_ARGUMENT;
  return bn_bwd(dz_concat, m, y, nullptr, relu, (int64_t)n * hw, channels, stats, gamma, beta, inv_scale, accumulate, workspace,
                workspace_bytes, dgamma, dbeta, dy, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int lfd_bn_train_finish_into_levels_f16(const lfd_bn_fwd_level_t* levels, int32_t nlevels, int32_t n, int32_t relu, void* z_concat,
                                        int64_t points_total, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!levels || nlevels < 1 || nlevels > LFD_MAX_LEVELS || !z_concat || n < 1) return LFD_ERR_INVALID_ARGUMENT;
  BnFwdJobs J{};
  J.n = nlevels; J.relu = relu ? 1 : 0; J.z = (__half*)z_concat;
  int max_blocks = 1, max_c = 8;
  for (int l = 0; l < nlevels; ++l) {
    const lfd_bn_fwd_level_t& L = levels[l];
    BnFwdJob& b = J.j[l];
    if (!L.y || !L.rows || !L.stats || !L.gamma || !L.beta || L.hw < 1 || L.nrows < 1 || L.nrows > kMaxBlocks * 2 ||
        !channels_ok(L.channels) || (L.running_mean == nullptr) != (L.running_var == nullptr))
      return LFD_ERR_INVALID_ARGUMENT;
    if (!row_map(&b.zmap, n, L.hw, points_total, L.point0, L.channels / 8)) return LFD_ERR_INVALID_ARGUMENT;
    const int64_t pixels = (int64_t)n * L.hw;
    b.y = (const __half*)L.y; b.vecs = pixels * (L.channels / 8); b.c = L.channels; b.blocks = (int)grid_for_vecs(b.vecs);
    b.rows = L.rows; b.nrows = L.nrows; b.m = (double)pixels; b.eps = L.eps; b.momentum = L.momentum;
    b.running_mean = L.running_mean; b.running_var = L.running_var; b.stats = L.stats; b.gamma = L.gamma; b.beta = L.beta;
    if (b.blocks > max_blocks) max_blocks = b.blocks;
    if (b.c > max_c) max_c = b.c;
  }
  hipLaunchKernelGGL(k_bn_stats_final_jobs, dim3(max_c, nlevels), dim3(64), 0, st, J);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_apply_jobs, dim3(max_blocks, nlevels), dim3(kThreads), 0, st, J);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_bn_train_bwd_from_levels_f16(const void* dz_concat, int64_t points_total, const lfd_bn_bwd_level_t* levels, int32_t nlevels,
                                     int32_t relu, int32_t n, float inv_scale, int32_t accumulate, void* workspace,
                                     size_t workspace_bytes, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!dz_concat || !levels || nlevels < 1 || nlevels > LFD_MAX_LEVELS || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  BnBwdJobs J{};
  J.n = nlevels; J.relu_y = relu ? 1 : 0; J.accumulate = accumulate; J.inv_scale = inv_scale;
  float* ws = reinterpret_cast<float*>(workspace);
  const size_t per_job = (size_t)kMaxBlocks * 2 * kMaxC + 2 * kMaxC;      // rows | sums, as one lfd_bn_train_bwd_f16 call lays them out
  int max_blocks = 1, max_c = 8;
  for (int l = 0; l < nlevels; ++l) {
    const lfd_bn_bwd_level_t& L = levels[l];
    BnBwdJob& b = J.j[l];
    if (!L.y || !L.stats || !L.gamma || !L.dy || L.hw < 1 || !channels_ok(L.channels) || (relu && !L.beta)) return LFD_ERR_INVALID_ARGUMENT;
    if (!row_map(&b.dmap, n, L.hw, points_total, L.point0, L.channels / 8)) return LFD_ERR_INVALID_ARGUMENT;
    const int64_t pixels = (int64_t)n * L.hw;
    b.dz = (const __half*)dz_concat; b.y = (const __half*)L.y; b.vecs = pixels * (L.channels / 8); b.c = L.channels;
    b.blocks = (int)grid_for_vecs(b.vecs);
    b.stats = L.stats; b.gamma = L.gamma; b.beta = L.beta; b.dgamma = L.dgamma; b.dbeta = L.dbeta; b.dy = (__half*)L.dy;
    b.rows = ws + (size_t)l * per_job; b.sums = b.rows + (size_t)kMaxBlocks * 2 * kMaxC;
    b.inv_m = (float)(1.0 / (double)pixels);
    if (b.blocks > max_blocks) max_blocks = b.blocks;
    if (b.c > max_c) max_c = b.c;
  }
  if ((size_t)nlevels * per_job * sizeof(float) > workspace_bytes) return LFD_ERR_WORKSPACE_TOO_SMALL;
  hipLaunchKernelGGL(k_bn_bwd_partial_jobs, dim3(max_blocks, nlevels), dim3(kThreads), 0, st, J);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_bwd_final_jobs, dim3(max_c, nlevels), dim3(64), 0, st, J);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_bwd_apply_jobs, dim3(max_blocks, nlevels), dim3(kThreads), 0, st, J);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

static inline unsigned gn_blocks(int64_t vecs_per_img, int nvirt) {
  int64_t b = (vecs_per_img + kThreads - 1) / kThreads;
  int64_t cap = 4096 / (nvirt < 1 ? 1 : nvirt);       // (images x segments) x blocks <= 4096 workgroups
  if (cap < 1) cap = 1;
  if (cap > 64) cap = 64;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}
static inline bool gn_ok(int n, int64_t hw, int c, int g) {
  return n >= 1 && n <= 1024 && hw >= 1 && g >= 1 && g <= 32 && (g & (g - 1)) == 0 && c == 8 * g;
}
// geometry of n images x nseg segments of seg_hw[] pixels; -> the largest segment's vectors (0: invalid)
static int64_t gn_geometry(GnGeom* G, int n, int nseg, const int64_t* seg_hw, int channels, int groups) {
  if (nseg < 1 || nseg > LFD_MAX_LEVELS || !seg_hw || n < 1 || n * nseg > 4096) return 0;
  int64_t tot = 0, mx = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!gn_ok(n, seg_hw[i], channels, groups)) return 0;
    G->off[i] = tot * groups;
    G->vecs[i] = seg_hw[i] * groups;
    tot += seg_hw[i];
    if (G->vecs[i] > mx) mx = G->vecs[i];
  }
  G->nseg = nseg;
  G->img_vecs = tot * groups;
  return mx;
}
// workspace layout of the GroupNorm passes: [4096][2g <= 64] group partials | [4096][2c <= 512] channel partials | group sums
static constexpr size_t kGnGroupFloats = (size_t)4096 * 2 * 32, kGnChanFloats = (size_t)4096 * 2 * 256;

static int gn_stats_apply(const void* y, int n, int nseg, const int64_t* seg_hw, int channels, int groups, float eps,
                          const float* gamma, const float* beta, int relu, void* workspace, size_t workspace_bytes, float* stats,
                          void* z, bool fold, hipStream_t st) {
  GnGeom G{};
  const int64_t mx = gn_geometry(&G, n, nseg, seg_hw, channels, groups);
  if (!mx || !y || !stats || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int nv = n * nseg;
  const unsigned b = gn_blocks(mx, nv);
  float* partials = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(k_gn_stats_partial, dim3(b, nv), dim3(kThreads), 0, st, (const __half*)y, G, groups, partials);
  LFD_CHECK_LAUNCH();
  if (fold) {
    if (!gamma || !beta || !z) return LFD_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_gn_apply<true>, dim3(b, nv), dim3(kThreads), 0, st, (const __half*)y, G, groups, stats, partials, (int)b,
                       eps, gamma, beta, relu, (__half*)z);
  } else {
    hipLaunchKernelGGL(k_gn_stats_final, dim3(nv), dim3(64), 0, st, partials, (int)b, groups, G, eps, stats);
  }
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_gn_train_stats_f16(const void* y, int32_t n, int64_t hw, int32_t channels, int32_t groups, float eps,
                           void* workspace, size_t workspace_bytes, float* stats, lfd_stream_t stream) {
  return gn_stats_apply(y, n, 1, &hw, channels, groups, eps, nullptr, nullptr, 0, workspace, workspace_bytes, stats, nullptr, false,
                        reinterpret_cast<hipStream_t>(stream));
}

int lfd_gn_train_apply_f16(const void* y, int32_t n, int64_t hw, int32_t channels, int32_t groups, const float* stats,
                           const float* gamma, const float* beta, int32_t relu, void* z, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GnGeom G{};
  const int64_t mx = gn_geometry(&G, n, 1, &hw, channels, groups);
  if (!mx || !y || !stats || !gamma || !beta || !z) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_gn_apply<false>, dim3(gn_blocks(mx, n), n), dim3(kThreads), 0, st, (const __half*)y, G, groups,
                     const_cast<float*>(stats), nullptr, 0, 0.f, gamma, beta, relu, (__half*)z);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_gn_train_stats_apply_f16(const void* y, int32_t n, int64_t hw, int32_t channels, int32_t groups, float eps,
                                 const float* gamma, const float* beta, int32_t relu, void* workspace, size_t workspace_bytes,
                                 float* stats, void* z, lfd_stream_t stream) {
  return gn_stats_apply(y, n, 1, &hw, channels, groups, eps, gamma, beta, relu, workspace, workspace_bytes, stats, z, true,
                        reinterpret_cast<hipStream_t>(stream));
}

int lfd_gn_train_stats_apply_seg_f16(const void* y, int32_t n, int32_t nseg, const int64_t* seg_hw, int32_t channels,
                                     int32_t groups, float eps, const float* gamma, const float* beta, int32_t relu,
                                     void* workspace, size_t workspace_bytes, float* stats, void* z, lfd_stream_t stream) {
  return gn_stats_apply(y, n, nseg, seg_hw, channels, groups, eps, gamma, beta, relu, workspace, workspace_bytes, stats, z, true,
                        reinterpret_cast<hipStream_t>(stream));
}

static int gn_bwd(const void* dz, const void* y, const void* z, int n, int nseg, const int64_t* seg_hw, int channels, int groups,
                  const float* stats, const float* gamma, float inv_scale, int accumulate, void* workspace, size_t workspace_bytes,
                  float* dgamma, float* dbeta, void* dy, hipStream_t st) {
  GnGeom G{};
  const int64_t mx = gn_geometry(&G, n, nseg, seg_hw, channels, groups);
  if (!mx || !dz || !y || !stats || !gamma || !dgamma || !dbeta || !dy || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int nv = n * nseg;
  const unsigned b = gn_blocks(mx, nv);
  float* pgroup = reinterpret_cast<float*>(workspace);
  float* pchan = pgroup + kGnGroupFloats;
  float* gsums = pchan + kGnChanFloats;
  hipLaunchKernelGGL(k_gn_bwd_partial, dim3(b, nv), dim3(kThreads), 0, st, (const __half*)dz, (const __half*)y,
                     (const __half*)z, G, groups, stats, gamma, pgroup, pchan);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_gn_bwd_final, dim3(nv * 2 * groups + 2 * channels), dim3(64), 0, st, pgroup, pchan, nv, (int)b, groups,
                     inv_scale, accumulate, gsums, dgamma, dbeta);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_gn_bwd_apply, dim3(b, nv), dim3(kThreads), 0, st, (const __half*)dz, (const __half*)y,
                     (const __half*)z, G, groups, stats, gamma, gsums, (__half*)dy);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_gn_train_bwd_f16(const void* dz, const void* y, const void* z, int32_t n, int64_t hw, int32_t channels,
                         int32_t groups, const float* stats, const float* gamma, float inv_scale, int32_t accumulate,
                         void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, void* dy,
                         lfd_stream_t stream) {
  return gn_bwd(dz, y, z, n, 1, &hw, channels, groups, stats, gamma, inv_scale, accumulate, workspace, workspace_bytes, dgamma,
                dbeta, dy, reinterpret_cast<hipStream_t>(stream));
}

int lfd_gn_train_bwd_seg_f16(const void* dz, const void* y, const void* z, int32_t n, int32_t nseg, const int64_t* seg_hw,
                             int32_t channels, int32_t groups, const float* stats, const float* gamma, float inv_scale,
                             int32_t accumulate, void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, void* dy,
                             lfd_stream_t stream) {
  return gn_bwd(dz, y, z, n, nseg, seg_hw, channels, groups, stats, gamma, inv_scale, accumulate, workspace, workspace_bytes,
                dgamma, dbeta, dy, reinterpret_cast<hipStream_t>(stream));
}

int lfd_zero_insert2_nhwc_f16(const void* in, int32_t n, int32_t hi, int32_t wi, int32_t channels, int32_t ho,
                              int32_t wo, void* out, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || n < 1 || hi < 1 || wi < 1 || ho < 2 * hi - 1 || wo < 2 * wi - 1 || ho > 2 * hi || wo > 2 * wi ||
      !channels_ok(channels))
    return LFD_ERR_INVALID_ARGUMENT;
  const int64_t vecs = (int64_t)n * ho * wo * (channels / 8);
  if (vecs >= ((int64_t)1 << 31)) return LFD_ERR_UNSUPPORTED;      // the kernel indexes vectors in 32 bits
  hipLaunchKernelGGL(k_zero_insert2, dim3(grid_for_vecs(vecs)), dim3(kThreads), 0, st, (const __half*)in, n, hi, wi,
                     channels, ho, wo, (__half*)out);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// Workgroups and partial rows of a weight gradient.  Every workgroup row leaves taps x 64 x 64 floats per (cout, cin) block
// behind for the final sum, so rows are bytes: round 3 used as many rows as its workspace held (up to 1024) and an iteration
// of WIDERFACE_LFD_S wrote and re-read 1.29 GB of partials -- 38 MB for a stage-3 conv whose operands are 1.6 MB.  Round 4:
//  * 1x1: rows x blocks ~ 1024 / 512 / 256 workgroups by operand size (the kernel prefetches: one or two per CU suffice);
//  * 3x3 on SMALL maps (operands <= 16 MB: stages 1-3 at 640x640): TAP-SPLIT -- the 1x1 kernel runs one tap per
//    blockIdx.z with x read at the tap's offset (dW[tap] = sum dy (x) x shifted), 288-360 workgroups = splits x blocks x 9,
//    ~5 MB of partials whatever the layer; the 9 workgroups of a split share an XCD (splits x blocks is a multiple of 8), so
//    the nine reads of an operand tile meet in one L2;
//  * 3x3 on large maps: all taps per workgroup as before (operands read once), 256 rows (stride 2: up to 512 on the
//    largest map, where 75 MB of partials are 14 % of the operands).
static int wgrad_geometry(int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ks, int32_t stride, int* nwg_out,
                          int* nblk_out, int* ho_out, int* wo_out, int* tap_split_out) {
  if (n < 1 || h < 1 || w < 1) return LFD_ERR_INVALID_ARGUMENT;
  if ((ks != 1 && ks != 3) || (stride != 1 && stride != 2)) return LFD_ERR_INVALID_ARGUMENT;
  if (cin < 8 || cout < 8 || (cin & 7) || (cout & 7) || cin > 128 || cout > 128) return LFD_ERR_INVALID_ARGUMENT;
  const int pad = ks / 2;
  const int ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
  const int nblk = ((cout + 63) / 64) * ((cin + 63) / 64);
  const int tiles = n * ((ho + 7) / 8) * ((wo + 15) / 16);
  const int64_t operand_bytes = ((int64_t)n * h * w * cin + (int64_t)n * ho * wo * cout) * 2;
  int nwg, tap_split = 0;
  if (ks == 3 && operand_bytes <= (16 << 20)) {
    tap_split = 1;
    nwg = nblk == 1 ? 40 : (nblk == 2 ? 16 : 8);
  } else if (ks == 3) {
    nwg = 256;
    if (stride == 2) { nwg = tiles / 6; if (nwg < 256) nwg = 256; if (nwg > 512) nwg = 512; }
  } else {
    const int target = operand_bytes >= (128 << 20) ? 1024 : (operand_bytes >= (32 << 20) ? 512 : 256);
    nwg = target / nblk;
  }
  if (nwg > kWgradWgCap) nwg = kWgradWgCap;
  if (nwg > tiles) nwg = tiles;
  *nwg_out = nwg; *nblk_out = nblk; *ho_out = ho; *wo_out = wo; *tap_split_out = tap_split;
  return LFD_OK;
}

static int wgrad_launch(const void* x, const void* dy, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ks,
                        int32_t stride, float* partials, int nwg, int nblk, int ho, int wo, int tap_split, hipStream_t st,
                        const float* x_stats = nullptr, const float* x_gamma = nullptr, const float* x_beta = nullptr) {
  WgradArgs a{};
  a.xpro_stats = x_stats; a.xpro_gamma = x_gamma; a.xpro_beta = x_beta;
  if (x_stats) {
    if (ks != 1 || stride != 1 || tap_split) return LFD_ERR_UNSUPPORTED;
  }
  a.x = (const __half*)x;
  a.dy = (const __half*)dy;
  a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
  a.ho = ho; a.wo = wo;
  a.partials = partials;
  a.vtaps = tap_split ? 9 : 0;
  if (tap_split) return stride == 1 ? launch_wgrad<1, 1>(a, nwg, nblk, st) : launch_wgrad<1, 2>(a, nwg, nblk, st);
  if (ks == 3 && stride == 1) return launch_wgrad<3, 1>(a, nwg, nblk, st);
  if (ks == 3) return launch_wgrad<3, 2>(a, nwg, nblk, st);
  if (stride == 1) return x_stats ? launch_wgrad<1, 1, true>(a, nwg, nblk, st) : launch_wgrad<1, 1>(a, nwg, nblk, st);
  return launch_wgrad<1, 2>(a, nwg, nblk, st);
}

int lfd_conv_wgrad_nhwc_f16(const void* x, const void* dy, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout,
                            int32_t ks, int32_t stride, float inv_scale, int32_t accumulate, void* workspace,
                            size_t workspace_bytes, float* dw, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!x || !dy || !dw || !workspace) return LFD_ERR_INVALID_ARGUMENT;
  int nwg, nblk, ho, wo, ts;
  int rc = wgrad_geometry(n, h, w, cin, cout, ks, stride, &nwg, &nblk, &ho, &wo, &ts);
  if (rc != LFD_OK) return rc;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  float* partials = reinterpret_cast<float*>(workspace);
  rc = wgrad_launch(x, dy, n, h, w, cin, cout, ks, stride, partials, nwg, nblk, ho, wo, ts, st);
  if (rc != LFD_OK) return rc;
  const int taps = ks * ks, total = nblk * taps * 64 * 64;
  hipLaunchKernelGGL(k_wgrad_final, dim3(total / 128), dim3(kThreads), 0, st, partials, nwg, nblk,
                     cin, cout, taps, inv_scale, accumulate, dw);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int32_t lfd_conv_wgrad_partial_rows(int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t ks, int32_t stride) {
  int nwg, nblk, ho, wo, ts;
  const int rc = wgrad_geometry(n, h, w, cin, cout, ks, stride, &nwg, &nblk, &ho, &wo, &ts);
  return rc != LFD_OK ? rc : nwg;
}

int lfd_conv_wgrad_partials_nhwc_f16(const void* x, const void* dy, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                     int32_t ks, int32_t stride, float* partials, size_t partials_bytes, lfd_stream_t stream) {
  if (!x || !dy || !partials) return LFD_ERR_INVALID_ARGUMENT;
  int nwg, nblk, ho, wo, ts;
  const int rc = wgrad_geometry(n, h, w, cin, cout, ks, stride, &nwg, &nblk, &ho, &wo, &ts);
  if (rc != LFD_OK) return rc;
  if (partials_bytes < (size_t)nwg * nblk * ks * ks * 64 * 64 * sizeof(float)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  return wgrad_launch(x, dy, n, h, w, cin, cout, ks, stride, partials, nwg, nblk, ho, wo, ts, reinterpret_cast<hipStream_t>(stream));
}

int lfd_conv1x1_wgrad_partials_of_bn_relu_f16(const void* y_in, const float* in_stats, const float* in_gamma,
                                              const float* in_beta, const void* dy, int32_t n, int32_t h, int32_t w,
                                              int32_t cin, int32_t cout, float* partials, size_t partials_bytes,
                                              lfd_stream_t stream) {
  if (!y_in || !in_stats || !in_gamma || !in_beta || !dy || !partials) return LFD_ERR_INVALID_ARGUMENT;
  int nwg, nblk, ho, wo, ts;
  const int rc = wgrad_geometry(n, h, w, cin, cout, 1, 1, &nwg, &nblk, &ho, &wo, &ts);
  if (rc != LFD_OK) return rc;
  if (partials_bytes < (size_t)nwg * nblk * 64 * 64 * sizeof(float)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  return wgrad_launch(y_in, dy, n, h, w, cin, cout, 1, 1, partials, nwg, nblk, ho, wo, ts, reinterpret_cast<hipStream_t>(stream),
                      in_stats, in_gamma, in_beta);
}

int lfd_wgrad_final_batched_f32(const lfd_wgrad_job_t* jobs_device, int32_t njobs, int32_t total_blocks, lfd_stream_t stream) {
  if (njobs < 0 || total_blocks < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (njobs == 0 || total_blocks == 0) return LFD_OK;
  if (!jobs_device || njobs > 4096) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_wgrad_final_batched, dim3(total_blocks), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream),
                     jobs_device, njobs);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_rows_sum_batched_f32(const lfd_rowsum_job_t* jobs_device, int32_t njobs, lfd_stream_t stream) {
  if (njobs < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (njobs == 0) return LFD_OK;
  if (!jobs_device) return LFD_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_rows_sum_batched, dim3(njobs), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), jobs_device);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

static int conv0_fwd(const float* x_nchw, int32_t n, int32_t h, int32_t w, int32_t channels, const float* weight_oihw, void* y,
                     float* stat_partials, unsigned* blocks_out, hipStream_t st);

int lfd_stem_conv0_train_fwd(const float* x_nchw, int32_t n, int32_t h, int32_t w, int32_t channels,
                             const float* weight_oihw, void* y, lfd_stream_t stream) {
  return conv0_fwd(x_nchw, n, h, w, channels, weight_oihw, y, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int lfd_stem_conv0_train_fwd_bn_stats(const float* x_nchw, int32_t n, int32_t h, int32_t w, int32_t channels,
                                      const float* weight_oihw, void* y, float eps, float momentum, float* running_mean,
                                      float* running_var, void* workspace, size_t workspace_bytes, float* stats,
                                      lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!workspace || !stats || (running_mean == nullptr) != (running_var == nullptr)) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int64_t pixels = (int64_t)n * ((h + 1) / 2) * ((w + 1) / 2);
  const int use_valu = lfd_tune(LFD_TUNE_CONV0_VALU);
  if (channels != 64 || use_valu) {      // the 32-channel stem (XS) and the VALU A/B kernel: conv, then the statistics pass
    const int rc = conv0_fwd(x_nchw, n, h, w, channels, weight_oihw, y, nullptr, nullptr, st);
    if (rc != LFD_OK) return rc;
    return lfd_bn_train_stats_f16(y, pixels, channels, eps, momentum, running_mean, running_var, workspace, workspace_bytes, stats,
                                  stream);
  }
  unsigned blocks = 0;
  float* partials = reinterpret_cast<float*>(workspace);
  const int rc = conv0_fwd(x_nchw, n, h, w, channels, weight_oihw, y, partials, &blocks, st);
  if (rc != LFD_OK) return rc;
  hipLaunchKernelGGL(k_bn_stats_final, dim3(channels), dim3(64), 0, st, partials, (int)blocks, channels, (double)pixels, eps,
                     momentum, running_mean, running_var, stats);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

static int conv0_fwd(const float* x_nchw, int32_t n, int32_t h, int32_t w, int32_t channels, const float* weight_oihw, void* y,
                     float* stat_partials, unsigned* blocks_out, hipStream_t st) {
  if (!x_nchw || !weight_oihw || !y || n < 1 || h < 1 || w < 1 || (channels != 32 && channels != 64))
    return LFD_ERR_INVALID_ARGUMENT;
  const int use_valu = lfd_tune(LFD_TUNE_CONV0_VALU);
  if ((int64_t)n * 3 * h * w >= ((int64_t)1 << 31)) return LFD_ERR_UNSUPPORTED;   // 32-bit pixel / element indices in the kernels
  if (!use_valu) {
    const int64_t groups = ((int64_t)n * ((h + 1) / 2) * ((w + 1) / 2) + 127) / 128;   // 4 waves x 32 pixels per block pass
    const unsigned blocks = (unsigned)(groups < 2048 ? groups : 2048);   // (2048 rows x 2 x 64 floats = half the partials area)
    if (blocks_out) *blocks_out = blocks;
    hipLaunchKernelGGL(k_conv0_fwd_mfma, dim3(blocks), dim3(kThreads), 0, st, x_nchw, n, h, w, channels, weight_oihw, (__half*)y,
                       stat_partials);
    LFD_CHECK_LAUNCH();
    return LFD_OK;
  }
  const int64_t vecs = (int64_t)n * ((h + 1) / 2) * ((w + 1) / 2) * (channels / 8);
  int64_t b = (vecs + kThreads - 1) / kThreads;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(k_conv0_fwd, dim3((unsigned)b), dim3(kThreads), 0, st, x_nchw, n, h, w, channels, weight_oihw,
                     (__half*)y);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_stem_conv0_wgrad(const float* x_nchw, const void* dy, int32_t n, int32_t h, int32_t w, int32_t channels,
                         float inv_scale, int32_t accumulate, void* workspace, size_t workspace_bytes, float* dw, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!x_nchw || !dy || !dw || !workspace || n < 1 || h < 1 || w < 1 || (channels != 32 && channels != 64))
    return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  float* partials = reinterpret_cast<float*>(workspace);
  const int use_valu = lfd_tune(LFD_TUNE_CONV0_VALU);
  if (!use_valu) {
    const int64_t ksteps = (int64_t)n * ((h + 1) / 2) * (((w + 1) / 2 + 15) / 16);
    if (ksteps >= ((int64_t)1 << 31) || (int64_t)n * 3 * h * w >= ((int64_t)1 << 31)) return LFD_ERR_UNSUPPORTED;     // 32-bit indices in the kernel
    const int nb = (int)(ksteps / 4 < 1 ? 1 : (ksteps / 4 > 1024 ? 1024 : ksteps / 4));
    hipLaunchKernelGGL(k_conv0_wgrad_mfma<false>, dim3(nb), dim3(kThreads), 0, st, x_nchw, (const __half*)dy, n, h, w, channels,
                       partials, Conv0BnArgs{});
    LFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_partials, dim3(2048), dim3(64), 0, st, partials, nb, 2048, inv_scale, 2, channels, 27,
                       accumulate, dw);
    LFD_CHECK_LAUNCH();
    return LFD_OK;
  }
  const int64_t total = (int64_t)n * ((h + 1) / 2) * ((w + 1) / 2);
  int blocks = 2048;
  int ppb = (int)((total + blocks - 1) / blocks);
  if (ppb < 64) ppb = 64;
  blocks = (int)((total + ppb - 1) / ppb);
  hipLaunchKernelGGL(k_conv0_wgrad_partial, dim3(blocks), dim3(kThreads), 0, st, x_nchw, (const __half*)dy, n, h, w,
                     channels, ppb, partials);
  LFD_CHECK_LAUNCH();
  const int count = 27 * channels;
  hipLaunchKernelGGL(k_sum_partials, dim3(count), dim3(64), 0, st, partials, blocks,
                     count, inv_scale, 1, channels, 27, accumulate, dw);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// BatchNorm backward of the FIRST unit without its apply pass (round 4): the sums (k_bn_bwd_partial + k_bn_bwd_final: dgamma, dbeta,
// and the two per-channel sums in the workspace) and then the first conv's weight gradient straight from dz and y
// (k_conv0_wgrad_mfma<true>) -- the unit has no data gradient, so nobody else needs its dy.
int lfd_stem_conv0_bn_bwd_wgrad_rows(const float* x_nchw, const void* dz, const void* y, int32_t n, int32_t h, int32_t w,
                                     int32_t channels, const float* stats, const float* gamma, const float* beta, float inv_scale,
                                     int32_t accumulate, int32_t sum_rows, void* workspace, size_t workspace_bytes, float* dgamma,
                                     float* dbeta, float* dw, lfd_stream_t stream);

int lfd_stem_conv0_bn_bwd_wgrad(const float* x_nchw, const void* dz, const void* y, int32_t n, int32_t h, int32_t w, int32_t channels,
                                const float* stats, const float* gamma, const float* beta, float inv_scale, int32_t accumulate,
                                void* workspace, size_t workspace_bytes, float* dgamma, float* dbeta, float* dw, lfd_stream_t stream) {
  return lfd_stem_conv0_bn_bwd_wgrad_rows(x_nchw, dz, y, n, h, w, channels, stats, gamma, beta, inv_scale, accumulate, 0, workspace,
                                          workspace_bytes, dgamma, dbeta, dw, stream);
}

int lfd_stem_conv0_bn_bwd_wgrad_rows(const float* x_nchw, const void* dz, const void* y, int32_t n, int32_t h, int32_t w,
                                     int32_t channels, const float* stats, const float* gamma, const float* beta, float inv_scale,
                                     int32_t accumulate, int32_t sum_rows, void* workspace, size_t workspace_bytes, float* dgamma,
                                     float* dbeta, float* dw, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (sum_rows < 0 || sum_rows > kMaxBlocks) return LFD_ERR_INVALID_ARGUMENT;
  if (!x_nchw || !dz || !y || !stats || !gamma || !beta || !dw || !workspace || n < 1 || h < 1 || w < 1 || channels != 64)
    return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int64_t pixels = (int64_t)n * ho * wo, vecs = pixels * (channels / 8);
  const unsigned g = grid_for_vecs(vecs);
  float* partials = reinterpret_cast<float*>(workspace);
  float* sums = partials + (size_t)kMaxBlocks * 2 * kMaxC;
  float* wpart = sums + 2 * kMaxC;                      // the weight gradient's own partial rows: [<= 1024][2048]
  if (!sum_rows) {
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(g), dim3(kThreads), 0, st, (const __half*)dz, (const __half*)y, (const __half*)nullptr,
                       vecs, channels, stats, gamma, beta, 1, partials, RowMap{});
    LFD_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_bn_bwd_final, dim3(channels), dim3(64), 0, st, partials, sum_rows ? sum_rows : (int)g, channels, inv_scale,
                     accumulate, sums, dgamma, dbeta);
  LFD_CHECK_LAUNCH();
  const int64_t ksteps = (int64_t)n * ho * ((wo + 15) / 16);
  const int nb = (int)(ksteps / 4 < 1 ? 1 : (ksteps / 4 > 1024 ? 1024 : ksteps / 4));
  Conv0BnArgs bn{(const __half*)y, stats, gamma, beta, sums, (float)(1.0 / (double)pixels)};
  hipLaunchKernelGGL(k_conv0_wgrad_mfma<true>, dim3(nb), dim3(kThreads), 0, st, x_nchw, (const __half*)dz, n, h, w, channels, wpart, bn);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_sum_partials, dim3(2048), dim3(64), 0, st, wpart, nb, 2048, inv_scale, 2, channels, 27, accumulate, dw);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
