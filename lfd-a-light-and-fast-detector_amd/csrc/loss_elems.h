// csrc/loss_elems.h -- per-element device functions shared by the stand-alone loss kernels (losses.hip,
// targets.hip) and the fused get_loss kernels (getloss.hip), so both paths produce the same values.
#pragma once
#include <float.h>
#include "common.h"

// one element of SigmoidFocalLossForward (sigmoid_focal_loss_cuda.cu:31-57), fp32 math
static __device__ __forceinline__ float focal_fwd_elem(float x, int t, int d, float gamma, float alpha) {
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
static __device__ __forceinline__ float focal_bwd_elem(float x, int t, int d, float gamma, float alpha, float g) {
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


// aligned IoU loss of one box pair (iou_loss.py:67-79 overlaps, :98-102 union clamp, :121-123 -log(clamp))
static __device__ __forceinline__ float iou_loss_elem(float4 a, float4 b, float eps) {
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float ov = w * h;
  const float a1 = (a.z - a.x) * (a.w - a.y);
  const float a2 = (b.z - b.x) * (b.w - b.y);
  const float un = fmaxf(a1 + a2 - ov, 1e-6f);
  const float iou = fmaxf(ov / un, eps);
  return -logf(iou);
}

// d loss / d pred for one pair, following autograd through the same expression graph:
//   loss = -log(q), q = max(ov/un, eps); un = max(a1+a2-ov, 1e-6); ov = w*h with clamps at 0;
//   torch.max / torch.min route the gradient to the larger / smaller operand (ties: split evenly
//   in ATen; ties have measure zero for float boxes and are resolved towards `pred` here).
static __device__ __forceinline__ float4 iou_loss_grad_elem(float4 a, float4 b, float eps, float dl) {
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
    const float dq = -dl / q;
    const float dov_direct = dq / un;
    const float dun = (ur > 1e-6f) ? (-dq * ov / (un * un)) : 0.f;
    const float dov = dov_direct - dun;  // un depends on -ov
    const float da1 = dun;
    const float dw = (wr > 0.f) ? dov * h : 0.f;
    const float dh = (hr > 0.f) ? dov * w : 0.f;
    if (a.z <= b.z) g.z += dw;
    if (a.x >= b.x) g.x -= dw;
    if (a.w <= b.w) g.w += dh;
    if (a.y >= b.y) g.y -= dh;
    g.z += da1 * ph; g.x -= da1 * ph;
    g.w += da1 * pw; g.y -= da1 * pw;
  }
  return g;
}
