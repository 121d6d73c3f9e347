// csrc/boxloss.hip -- GIoU / DIoU / CIoU box-regression losses (reference lfd/model/losses/iou_loss.py:127-283, the
// other members of LFD's "union" regression-loss family next to IoULoss, lfd.py:64-66), forward AND gradient in one pass.
//
// The gradient is not hand-derived: the loss expression is evaluated on dual numbers (value + the four partial
// derivatives w.r.t. the predicted x1, y1, x2, y2), i.e. forward-mode differentiation inside the kernel.  max / min /
// clamp route the derivative like torch's autograd (to the selected operand; exact ties have measure zero for float
// boxes and go to the first operand), so the result is what `loss.sum().backward()` produces on the reference's
// expression graph, with one read of the boxes and no saved intermediates.  Elementwise, HBM-bound.
#include <math.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

struct Dual {
  float v;
  float d[4];
};

__device__ __forceinline__ Dual cst(float v) { return Dual{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual var(float v, int i) {
  Dual r = cst(v);
  r.d[i] = 1.f;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) {
  return Dual{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}};
}
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) {
  return Dual{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}};
}
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v / b.v;
  const float inv = 1.f / b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, float c) { Dual r = a; r.v += c; return r; }
__device__ __forceinline__ Dual operator*(const Dual& a, float c) {
  return Dual{a.v * c, {a.d[0] * c, a.d[1] * c, a.d[2] * c, a.d[3] * c}};
}
__device__ __forceinline__ Dual dmax(const Dual& a, const Dual& b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ Dual dmin(const Dual& a, const Dual& b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ Dual clamp0(const Dual& a) { return a.v > 0.f ? a : cst(0.f); }   // .clamp(min=0)
__device__ __forceinline__ Dual datan(const Dual& a) {
  const float g = 1.f / (1.f + a.v * a.v);
  return Dual{atanf(a.v), {a.d[0] * g, a.d[1] * g, a.d[2] * g, a.d[3] * g}};
}

// kind: 1 GIoU (iou_loss.py:127-169), 2 DIoU (:172-223), 3 CIoU (:226-283)
__device__ __forceinline__ Dual box_loss(float4 p, float4 t, int kind, float eps) {
  const Dual x1 = var(p.x, 0), y1 = var(p.y, 1), x2 = var(p.z, 2), y2 = var(p.w, 3);
  const Dual tx1 = cst(t.x), ty1 = cst(t.y), tx2 = cst(t.z), ty2 = cst(t.w);
  const Dual w = clamp0(dmin(x2, tx2) - dmax(x1, tx1)), h = clamp0(dmin(y2, ty2) - dmax(y1, ty1));
  const Dual overlap = w * h;
  const Dual ap = (x2 - x1) * (y2 - y1);
  const Dual ag = cst((t.z - t.x) * (t.w - t.y));
  const Dual uni = ap + ag - overlap + eps;
  const Dual iou = overlap / uni;
  const Dual ew = clamp0(dmax(x2, tx2) - dmin(x1, tx1)), eh = clamp0(dmax(y2, ty2) - dmin(y1, ty1));
  if (kind == 1) {
    const Dual earea = ew * eh + eps;
    const Dual giou = iou - (earea - uni) / earea;
    return cst(1.f) - giou;
  }
  const Dual c2 = ew * ew + eh * eh + eps;
  const Dual dx = (tx1 + tx2) - (x1 + x2), dy = (ty1 + ty2) - (y1 + y2);
  const Dual rho2 = (dx * dx) * 0.25f + (dy * dy) * 0.25f;
  if (kind == 2) return cst(1.f) - (iou - rho2 / c2);
  const Dual w1 = x2 - x1, h1 = (y2 - y1) + eps;
  const float w2 = t.z - t.x, h2 = (t.w - t.y) + eps;
  const float factor = 4.f / (float)(M_PI * M_PI);
  const Dual da = cst(atanf(w2 / h2)) - datan(w1 / h1);
  const Dual v = (da * da) * factor;
  const Dual ciou = iou - (rho2 / c2 + (v * v) / (cst(1.f) - iou + v));
  return cst(1.f) - ciou;
}

__global__ __launch_bounds__(kThreads) void k_box_loss(const float4* __restrict__ pred, const float4* __restrict__ target,
                                                      int64_t n, int kind, float eps, float* __restrict__ loss,
                                                      float4* __restrict__ dpred) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const Dual l = box_loss(pred[i], target[i], kind, eps);
    loss[i] = l.v;
    if (dpred) dpred[i] = make_float4(l.d[0], l.d[1], l.d[2], l.d[3]);
  }
}

// Elementwise regression losses of LFD's "independent" family (lfd.py:61-66): smooth-L1 (smooth_l1_loss.py:11-22:
// 0.5 d^2 / beta below beta, d - 0.5 beta above), L1 (:25-30) and MSE (mse_loss.py:11-13), loss and d loss / d pred
// in one pass.  kind: 1 smooth-L1, 2 L1, 3 MSE.  |x| has derivative sign(x) with sign(0) = 0, like torch.abs.
__global__ __launch_bounds__(kThreads) void k_pointwise_loss(const float* __restrict__ pred,
                                                            const float* __restrict__ target, int64_t n, int kind,
                                                            float beta, float* __restrict__ loss,
                                                            float* __restrict__ dpred) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const float x = pred[i] - target[i];
    const float d = fabsf(x);
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    float l, g;
    if (kind == 1) {
      if (d < beta) { l = 0.5f * d * d / beta; g = sg * (d / beta); }
      else { l = d - 0.5f * beta; g = sg; }
    } else if (kind == 2) {
      l = d; g = sg;
    } else {
      l = x * x; g = 2.f * x;
    }
    loss[i] = l;
    if (dpred) dpred[i] = g;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The two remaining classification losses LFD accepts (lfd.py:52-56): binary cross-entropy with logits against float
// targets (bce_with_logits_loss.py:28-44 -> F.binary_cross_entropy_with_logits, reduction 'none') and Quality Focal Loss
// (gfocal_loss.py:11-52).  B(x, t) = max(x, 0) - x t + log(1 + exp(-|x|)),  dB/dx = sigmoid(x) - t.
// QFL row n, class c:  t = score[n] if c == label[n] (a foreground label) else 0;  l = B(x, t) |t - s|^beta, s = sigmoid(x);
//   dl/dx = (s - t) |t - s|^beta - B beta |t - s|^(beta-1) sign(t - s) s (1 - s);  the row loss is the sum over classes.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_logits(float x, float t) {
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(kThreads) void k_bce_logits(const float* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                        float* __restrict__ loss, float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const float xv = x[i], tv = t[i];
    loss[i] = bce_logits(xv, tv);
    if (dx) dx[i] = 1.f / (1.f + expf(-xv)) - tv;
  }
}

__global__ __launch_bounds__(kThreads) void k_qfl(const float* __restrict__ x, const int64_t* __restrict__ label,
                                                 const float* __restrict__ score, int64_t rows, int c, float beta,
                                                 float* __restrict__ loss, float* __restrict__ dx) {
  for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kThreads) {
    const int64_t lab = label[r];
    const float sc = score[r];
    float sum = 0.f;
    for (int j = 0; j < c; ++j) {
      const float xv = x[r * c + j];
      const float t = (lab == j) ? sc : 0.f;            // labels outside [0, c) are background: no positive class
      const float s = 1.f / (1.f + expf(-xv));
      const float u = t - s, m = fabsf(u);
      const float b = bce_logits(xv, t);
      const float mb = powf(m, beta);
      sum += b * mb;
      if (dx) {
        const float sg = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
        dx[r * c + j] = (s - t) * mb - b * beta * powf(m, beta - 1.f) * sg * s * (1.f - s);
      }
    }
    loss[r] = sum;
  }
}

}  // namespace

extern "C" {

int lfd_bce_with_logits_f32(const float* logits, const float* targets, int64_t n, float* loss, float* d_logits,
                            lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!logits || !targets || !loss) return LFD_ERR_INVALID_ARGUMENT;
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  hipLaunchKernelGGL(k_bce_logits, dim3((unsigned)b), dim3(kThreads), 0, st, logits, targets, n, loss, d_logits);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_quality_focal_loss_f32(const float* logits, const int64_t* labels, const float* scores, int64_t rows,
                               int32_t channels, float beta, float* loss, float* d_logits, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (rows < 0 || channels < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (rows == 0) return LFD_OK;
  if (!logits || !labels || !scores || !loss) return LFD_ERR_INVALID_ARGUMENT;
  int64_t b = (rows + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  hipLaunchKernelGGL(k_qfl, dim3((unsigned)b), dim3(kThreads), 0, st, logits, labels, scores, rows, channels, beta, loss,
                     d_logits);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_pointwise_loss_f32(const float* pred, const float* target, int64_t n, int32_t kind, float beta, float* loss,
                           float* d_loss_d_pred, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || kind < 1 || kind > 3 || (kind == 1 && !(beta > 0.f))) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!pred || !target || !loss) return LFD_ERR_INVALID_ARGUMENT;
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  hipLaunchKernelGGL(k_pointwise_loss, dim3((unsigned)b), dim3(kThreads), 0, st, pred, target, n, kind, beta, loss,
                     d_loss_d_pred);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_box_loss_f32(const float* pred, const float* target, int64_t n, int32_t kind, float eps, float* loss,
                     float* d_loss_d_pred, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || kind < 1 || kind > 3) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) return LFD_OK;
  if (!pred || !target || !loss) return LFD_ERR_INVALID_ARGUMENT;
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  hipLaunchKernelGGL(k_box_loss, dim3((unsigned)b), dim3(kThreads), 0, st, (const float4*)pred, (const float4*)target, n,
                     kind, eps, loss, (float4*)d_loss_d_pred);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
