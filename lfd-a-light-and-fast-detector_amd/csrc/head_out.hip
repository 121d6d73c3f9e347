// csrc/head_out.hip -- the glue around the head's OUTPUT convs in a training iteration, one launch where autograd ran ~25.
//
// The reference's head ends, per pyramid level, in a classification conv and a regression conv whose outputs leave as
// fp32, the regression one through a learnable per-level `Scale` (lfd_head.py:157-185), and LFD.forward concatenates the
// levels along the point axis (lfd.py:526-542).  The training engine runs the level's convs as ONE padded 1x1 conv
// (64 output rows: class rows, 4 regression rows, zeros) on the MFMA kernel; this file is what sits on both sides of it:
//   forward :  y [n, hw, 64] fp16  ->  cls[:, p0:p0+hw, :] / reg[:, p0:p0+hw, :] fp32 (x scale) of the concatenated tensors
//   backward:  dcls / dreg fp32 (slices of the concatenated gradients)  ->  dy [n, hw, 64] fp16 (x scale x loss scale, zero
//              rows included), and  dbias += sum d,  dscale += sum dreg * raw  through per-block partials + one fixed-order
//              fp64 final (deterministic, no atomics -- like every other reduction of train.hip).
// As PyTorch ops this was, per level and iteration: 2 slices -> float, 1 multiply, then 3 multiplies, 3 sums, 5 adds, a
// zero fill, 2 half conversions and 2 strided copies -- 130 of the iteration's 155 PyTorch launches for five levels.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 64;          // output rows of the padded conv
constexpr int kMaxBlocks = 1024;    // k_out_grad is a chain of gathers per trip: many short walks, not few long ones

struct Seg {
  float* out;           // forward: [n, points_total, channels] fp32
  const float* grad;    // backward: same layout
  float* dbias;         // [channels], +=
  const float* scale;   // device scalar or null
  float* dscale;        // device scalar, += (null: none)
  int channels, row0;
};

struct Args {
  const __half* y;      // [n, hw, 64]
  __half* dy;           // [n, hw, 64]
  int n, hw;
  int64_t points_total, point0;
  int64_t y_total, y_point0;   // y / dy row of (img, p) = img * y_total + y_point0 + p  (a level's own tensor: hw, 0; a level inside
                               // the level-concatenated conv output of the training schedule: points_total, point0)
  Seg seg[2];
  int nsegs;
  float loss_scale;
  float* partials;      // [blocks][2][64]
};

__device__ __forceinline__ void out_split_body(const Args& a, int bx, int nbx) {
  // one thread per (pixel, segment channel): tiny tensors, 2-byte gathers from the 128-byte line of the pixel
  const int64_t pixels = (int64_t)a.n * a.hw;
  for (int s = 0; s < a.nsegs; ++s) {
    const Seg g = a.seg[s];
    const float mul = g.scale ? *g.scale : 1.f;
    const int64_t total = pixels * g.channels;
    for (int64_t i = (int64_t)bx * kThreads + threadIdx.x; i < total; i += (int64_t)nbx * kThreads) {
      // 32-bit index arithmetic (fill() refuses n * hw * 64 >= 2^31): as 64-bit divisions these lines were ~400 instructions per element
      const unsigned iu = (unsigned)i;
      const unsigned pxu = iu / (unsigned)g.channels;
      const int j = (int)(iu - pxu * (unsigned)g.channels);
      const unsigned imgu = pxu / (unsigned)a.hw;
      const int64_t img = imgu, p = pxu - imgu * (unsigned)a.hw;
      const float v = __half2float(a.y[(img * a.y_total + a.y_point0 + p) * kRows + g.row0 + j]);
      g.out[(img * a.points_total + a.point0 + p) * g.channels + j] = g.scale ? v * mul : v;
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_out_split(Args a) { out_split_body(a, blockIdx.x, gridDim.x); }

// all pyramid levels of one output conv in ONE launch (blockIdx.y = level; a level uses its own block count, so the values --
// and for the gradient the partial rows and their sums -- are those of the per-level launches, bit for bit)
struct LevelsArgs {
  Args lv[LFD_MAX_LEVELS];
  int nblocks[LFD_MAX_LEVELS];
  int nlev;
};
__global__ __launch_bounds__(kThreads) void k_out_split_levels(LevelsArgs L) {
  const int l = blockIdx.y;
  if ((int)blockIdx.x < L.nblocks[l]) out_split_body(L.lv[l], blockIdx.x, L.nblocks[l]);
}

// thread = (pixel, 16-byte chunk of its dy line); the chunk index is the same in every trip of the grid-stride loop, so
// 8 + 8 sums per thread last the walk
__device__ __forceinline__ void out_grad_body(const Args& a, int bx, int nbx) {
  __shared__ float red[kThreads][17];
  const int64_t vecs = (int64_t)a.n * a.hw * (kRows / 8);
  const int ck = threadIdx.x & 7;
  // which segment (if any) and which of its channels each of this thread's 8 rows is
  int sidx[8], sch[8];
  for (int e = 0; e < 8; ++e) {
    const int r = ck * 8 + e;
    sidx[e] = -1; sch[e] = 0;
    for (int s = 0; s < a.nsegs; ++s)
      if (r >= a.seg[s].row0 && r < a.seg[s].row0 + a.seg[s].channels) { sidx[e] = s; sch[e] = r - a.seg[s].row0; }
  }
  float mul[2] = {1.f, 1.f};
  for (int s = 0; s < a.nsegs; ++s)
    if (a.seg[s].scale) mul[s] = *a.seg[s].scale;
  float acc_d[8], acc_r[8];
  for (int e = 0; e < 8; ++e) acc_d[e] = acc_r[e] = 0.f;
  for (int64_t v = (int64_t)bx * kThreads + threadIdx.x; v < vecs; v += (int64_t)nbx * kThreads) {
    const unsigned pxu = (unsigned)(v >> 3), imgu = pxu / (unsigned)a.hw;       // 32-bit, see k_out_split
    const int64_t img = imgu, p = pxu - imgu * (unsigned)a.hw;
    const int64_t yrow = img * a.y_total + a.y_point0 + p;
    union { uint4 u; _Float16 h[8]; } o;
    o.u = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int s = sidx[e];
      if (s < 0) continue;
      const Seg& g = a.seg[s];
      float d = g.grad[(img * a.points_total + a.point0 + p) * g.channels + sch[e]];
      if (g.dscale) acc_r[e] += d * __half2float(a.y[yrow * kRows + ck * 8 + e]);     // dL/dscale: sum of dreg * raw
      if (g.scale) d = d * mul[s];
      acc_d[e] += d;                                                                // dL/dbias
      o.h[e] = (_Float16)(d * a.loss_scale);
    }
    reinterpret_cast<uint4*>(a.dy)[yrow * 8 + ck] = o.u;
  }
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = acc_d[e]; red[threadIdx.x][8 + e] = acc_r[e]; }
  __syncthreads();
  if (threadIdx.x < 2 * kRows) {
    const int q = threadIdx.x >> 6, r = threadIdx.x & 63;
    float s = 0.f;
    for (int t = r >> 3; t < kThreads; t += 8) s += red[t][q * 8 + (r & 7)];
    a.partials[((size_t)bx * 2 + q) * kRows + r] = s;
  }
}
__global__ __launch_bounds__(kThreads) void k_out_grad(Args a) { out_grad_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(kThreads) void k_out_grad_levels(LevelsArgs L) {
  const int l = blockIdx.y;
  if ((int)blockIdx.x < L.nblocks[l]) out_grad_body(L.lv[l], blockIdx.x, L.nblocks[l]);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// sum over the blocks' partials of quantity q, row r: lanes stride over the rows of partials, four requests in flight
// (one thread walking 256 rows one dependent load after the other was 45 us)
__device__ __forceinline__ double column_sum(const float* partials, int nblocks, int q, int r) {
  double s = 0.0;
  int b = threadIdx.x & 63;
  for (; b + 192 < nblocks; b += 256) {
    float u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = partials[((size_t)(b + 64 * k) * 2 + q) * kRows + r];
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (double)u[k];
  }
  for (; b < nblocks; b += 64) s += (double)partials[((size_t)b * 2 + q) * kRows + r];
  return wave_sum(s);
}

// block = output row r; wave 0: dbias of the row; wave 1 of a Scale segment's first row: dscale over the segment's rows
__device__ __forceinline__ void out_grad_final_body(const Args& a, int nblocks) {
  const int q = threadIdx.x >> 6, r = blockIdx.x;
  for (int k = 0; k < a.nsegs; ++k) {
    const Seg& g = a.seg[k];
    if (q == 0) {
      if (!g.dbias || r < g.row0 || r >= g.row0 + g.channels) continue;
      const double s = column_sum(a.partials, nblocks, 0, r);
      if ((threadIdx.x & 63) == 0) g.dbias[r - g.row0] += (float)s;
    } else {
      if (!g.dscale || r != g.row0) continue;
      double t = 0.0;
      for (int j = 0; j < g.channels; ++j) t += column_sum(a.partials, nblocks, 1, g.row0 + j);
      if ((threadIdx.x & 63) == 0) *g.dscale += (float)t;
    }
  }
}

__global__ __launch_bounds__(128) void k_out_grad_final(Args a, int nblocks) { out_grad_final_body(a, nblocks); }
// the levels one after the other in level order: the += into the shared biases happens in the order of the per-level launches
__global__ __launch_bounds__(128) void k_out_grad_final_levels(LevelsArgs L) {
  for (int l = 0; l < L.nlev; ++l) out_grad_final_body(L.lv[l], L.nblocks[l]);
}

bool fill(Args& a, const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0, const lfd_head_out_seg_t* segs,
          int32_t nsegs) {
  if (!y || !segs || n < 1 || hw < 1 || nsegs < 1 || nsegs > 2 || point0 < 0 || point0 + hw > points_total) return false;
  if ((int64_t)n * hw * kRows >= ((int64_t)1 << 31)) return false;       // the kernels index pixels and elements in 32 bits
  a.y = (const __half*)y; a.n = n; a.hw = hw; a.points_total = points_total; a.point0 = point0; a.nsegs = nsegs;
  a.y_total = hw; a.y_point0 = 0;
  for (int s = 0; s < nsegs; ++s) {
    const lfd_head_out_seg_t& g = segs[s];
    if (g.channels < 1 || g.row0 < 0 || g.row0 + g.channels > kRows) return false;
    if (s == 1 && !(segs[0].row0 + segs[0].channels <= g.row0 || g.row0 + g.channels <= segs[0].row0)) return false;
    a.seg[s] = Seg{g.out, g.grad, g.dbias, g.scale, g.dscale, g.channels, g.row0};
  }
  return true;
}

}  // namespace

extern "C" {

static int out_split(const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0, const lfd_head_out_seg_t* segs,
                     int32_t nsegs, bool concat, lfd_stream_t stream) {
  Args a{};
  if (!fill(a, y, n, hw, points_total, point0, segs, nsegs)) return LFD_ERR_INVALID_ARGUMENT;
  if (concat) { a.y_total = points_total; a.y_point0 = point0; }
  int maxc = 0;
  for (int s = 0; s < nsegs; ++s) {
    if (!a.seg[s].out) return LFD_ERR_INVALID_ARGUMENT;
    if (a.seg[s].channels > maxc) maxc = a.seg[s].channels;
  }
  int64_t b = ((int64_t)n * hw * maxc + kThreads - 1) / kThreads;
  if (b > 1024) b = 1024;
  hipLaunchKernelGGL(k_out_split, dim3((unsigned)b), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_head_out_split_f16(const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                           const lfd_head_out_seg_t* segs, int32_t nsegs, lfd_stream_t stream) {
  return out_split(y, n, hw, points_total, point0, segs, nsegs, false, stream);
}

int lfd_head_out_split_concat_f16(const void* y_concat, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                                  const lfd_head_out_seg_t* segs, int32_t nsegs, lfd_stream_t stream) {
  return out_split(y_concat, n, hw, points_total, point0, segs, nsegs, true, stream);
}

static int out_grad(const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0, const lfd_head_out_seg_t* segs,
                    int32_t nsegs, float loss_scale, void* dy, void* workspace, size_t workspace_bytes, bool concat,
                    lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  Args a{};
  if (!fill(a, y, n, hw, points_total, point0, segs, nsegs) || !dy || !workspace || !lfd_aligned16(dy)) return LFD_ERR_INVALID_ARGUMENT;
  if (concat) { a.y_total = points_total; a.y_point0 = point0; }
  if (workspace_bytes < (size_t)kMaxBlocks * 2 * kRows * sizeof(float)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  for (int s = 0; s < nsegs; ++s)
    if (!a.seg[s].grad || (a.seg[s].dscale && !a.seg[s].scale)) return LFD_ERR_INVALID_ARGUMENT;
  a.dy = (__half*)dy; a.loss_scale = loss_scale; a.partials = reinterpret_cast<float*>(workspace);
  int64_t b = ((int64_t)n * hw * (kRows / 8) + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  hipLaunchKernelGGL(k_out_grad, dim3((unsigned)b), dim3(kThreads), 0, st, a);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_out_grad_final, dim3(kRows), dim3(128), 0, st, a, (int)b);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_head_out_grad_f16(const void* y, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                          const lfd_head_out_seg_t* segs, int32_t nsegs, float loss_scale, void* dy, void* workspace,
                          size_t workspace_bytes, lfd_stream_t stream) {
  return out_grad(y, n, hw, points_total, point0, segs, nsegs, loss_scale, dy, workspace, workspace_bytes, false, stream);
}

int lfd_head_out_grad_concat_f16(const void* y_concat, int32_t n, int32_t hw, int64_t points_total, int64_t point0,
                                 const lfd_head_out_seg_t* segs, int32_t nsegs, float loss_scale, void* dy_concat, void* workspace,
                                 size_t workspace_bytes, lfd_stream_t stream) {
  return out_grad(y_concat, n, hw, points_total, point0, segs, nsegs, loss_scale, dy_concat, workspace, workspace_bytes, true, stream);
}

static int fill_levels(LevelsArgs& L, const void* y_concat, int32_t n, int64_t points_total, const lfd_head_out_level_t* levels,
                       int32_t nlevels) {
  if (!levels || nlevels < 1 || nlevels > LFD_MAX_LEVELS) return LFD_ERR_INVALID_ARGUMENT;
  L.nlev = nlevels;
  for (int l = 0; l < nlevels; ++l) {
    if (!fill(L.lv[l], y_concat, n, levels[l].hw, points_total, levels[l].point0, levels[l].segs, levels[l].nsegs))
      return LFD_ERR_INVALID_ARGUMENT;
    L.lv[l].y_total = points_total; L.lv[l].y_point0 = levels[l].point0;
  }
  return LFD_OK;
}

int lfd_head_out_split_levels_f16(const void* y_concat, int32_t n, int64_t points_total, const lfd_head_out_level_t* levels,
                                  int32_t nlevels, lfd_stream_t stream) {
  LevelsArgs L{};
  const int rc = fill_levels(L, y_concat, n, points_total, levels, nlevels);
  if (rc != LFD_OK) return rc;
  int mx = 1;
  for (int l = 0; l < nlevels; ++l) {
    int maxc = 0;
    for (int s = 0; s < L.lv[l].nsegs; ++s) {
      if (!L.lv[l].seg[s].out) return LFD_ERR_INVALID_ARGUMENT;
      if (L.lv[l].seg[s].channels > maxc) maxc = L.lv[l].seg[s].channels;
    }
    int64_t b = ((int64_t)n * L.lv[l].hw * maxc + kThreads - 1) / kThreads;       // the block count of lfd_head_out_split_f16
    if (b > 1024) b = 1024;
    L.nblocks[l] = (int)b;
    if (b > mx) mx = (int)b;
  }
  hipLaunchKernelGGL(k_out_split_levels, dim3((unsigned)mx, (unsigned)nlevels), dim3(kThreads), 0, reinterpret_cast<hipStream_t>(stream), L);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_head_out_grad_levels_f16(const void* y_concat, int32_t n, int64_t points_total, const lfd_head_out_level_t* levels,
                                 int32_t nlevels, float loss_scale, void* dy_concat, void* workspace, size_t workspace_bytes,
                                 lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LevelsArgs L{};
  const int rc = fill_levels(L, y_concat, n, points_total, levels, nlevels);
  if (rc != LFD_OK) return rc;
  if (!dy_concat || !workspace || !lfd_aligned16(dy_concat)) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < (size_t)nlevels * kMaxBlocks * 2 * kRows * sizeof(float)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  int mx = 1;
  for (int l = 0; l < nlevels; ++l) {
    Args& a = L.lv[l];
    for (int s = 0; s < a.nsegs; ++s)
      if (!a.seg[s].grad || (a.seg[s].dscale && !a.seg[s].scale)) return LFD_ERR_INVALID_ARGUMENT;
    a.dy = (__half*)dy_concat; a.loss_scale = loss_scale;
    a.partials = reinterpret_cast<float*>(workspace) + (size_t)l * kMaxBlocks * 2 * kRows;
    int64_t b = ((int64_t)n * a.hw * (kRows / 8) + kThreads - 1) / kThreads;       // the block count of lfd_head_out_grad_f16
    if (b > kMaxBlocks) b = kMaxBlocks;
    L.nblocks[l] = (int)b;
    if (b > mx) mx = (int)b;
  }
  hipLaunchKernelGGL(k_out_grad_levels, dim3((unsigned)mx, (unsigned)nlevels), dim3(kThreads), 0, st, L);
  LFD_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_out_grad_final_levels, dim3(kRows), dim3(128), 0, st, L);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
