// csrc/sibling.hip -- the few element-wise NHWC fp16 operators the sibling necks / heads need next to the conv kernels
// (SURVEY 8 f4): FPN / SimpleFPN top-down (or bottom-up) merge, the in-place ReLU in front of an extra level, the
// 'pooling' extra level, and the packing of a level's fp32 output-conv maps into the meta-architecture's
// level-concatenated [N,P,C] tensors.  All HBM-bound streaming kernels: 16-byte accesses, one pixel-chunk per lane.
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// dst[n,y,x,:] += src[n, sy(y), sx(x), :], nearest neighbour exactly as ATen's upsample_nearest2d computes the source
// index for nn.Upsample(size=...) (scale = float(in) / out, floorf(dst * scale), clamped): fpn.py:133-135,
// simple_fpn.py:150-157
__global__ __launch_bounds__(256) void k_upsample_add(_Float16* dst, const _Float16* src, int N, int H, int W, int h,
                                                      int w, int c8, float sy, float sx) {
  const int64_t total = (int64_t)N * H * W * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % c8);
    int64_t t = i / c8;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    int yy = (int)floorf((float)y * sy), xx = (int)floorf((float)x * sx);
    yy = yy < h - 1 ? yy : h - 1;
    xx = xx < w - 1 ? xx : w - 1;
    const h8 a = reinterpret_cast<const h8*>(dst)[i];
    const h8 b = reinterpret_cast<const h8*>(src)[(((int64_t)n * h + yy) * w + xx) * c8 + q];
    h8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (_Float16)((float)a[k] + (float)b[k]);   // fp32 add, one rounding
    reinterpret_cast<h8*>(dst)[i] = r;
  }
}

__global__ __launch_bounds__(256) void k_relu_inplace(_Float16* x, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    h8 v = reinterpret_cast<h8*>(x)[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] > (_Float16)0 ? v[k] : (_Float16)0;
    reinterpret_cast<h8*>(x)[i] = v;
  }
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (fpn.py:74, simple_fpn.py:92): padding never wins
__global__ __launch_bounds__(256) void k_maxpool3s2(const _Float16* in, _Float16* out, int N, int H, int W, int OH, int OW,
                                                    int c8) {
  const int64_t total = (int64_t)N * OH * OW * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % c8);
    int64_t t = i / c8;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    h8 m;
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -(_Float16)INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      const int y = oy * 2 - 1 + dy;
      if (y < 0 || y >= H) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int x = ox * 2 - 1 + dx;
        if (x < 0 || x >= W) continue;
        const h8 v = reinterpret_cast<const h8*>(in)[(((int64_t)n * H + y) * W + x) * c8 + q];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
      }
    }
    reinterpret_cast<h8*>(out)[i] = m;
  }
}

// src [N, hw, cs] fp32 (an output conv's padded channel block) -> dst [N, P, cnt] fp32 rows p0 .. p0+hw of every image:
// v = src[.., c0 + j] * scale, then op (0: none, 1: expf -- fcos_head.py:145-146 `scales[i](x).float().exp()`).
__global__ __launch_bounds__(256) void k_pack_level(const float* src, float* dst, int N, int hw, int cs, int c0, int cnt,
                                                    int P, int p0, float scale, int op) {
  const int64_t total = (int64_t)N * hw * cnt;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % cnt);
    int64_t t = i / cnt;
    const int p = (int)(t % hw);
    const int n = (int)(t / hw);
    float v = src[((int64_t)n * hw + p) * cs + c0 + j] * scale;
    if (op == 1) v = expf(v);
    dst[((int64_t)n * P + p0 + p) * cnt + j] = v;
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" {

int lfd_upsample_nearest_add_nhwc_f16(void* dst, const void* src, int32_t n, int32_t H, int32_t W, int32_t h, int32_t w,
                                      int32_t c, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!dst || !src || n < 1 || H < 1 || W < 1 || h < 1 || w < 1 || c < 8) return LFD_ERR_INVALID_ARGUMENT;
  if (c % 8) return LFD_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)n * H * W * (c / 8);
  hipLaunchKernelGGL(k_upsample_add, dim3(grid_for(total)), dim3(256), 0, st, (_Float16*)dst, (const _Float16*)src, n, H, W,
                     h, w, c / 8, (float)h / (float)H, (float)w / (float)W);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_relu_inplace_f16(void* x, int64_t count, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!x || count < 0) return LFD_ERR_INVALID_ARGUMENT;
  if (count % 8) return LFD_ERR_UNSUPPORTED;
  if (count == 0) return LFD_OK;
  hipLaunchKernelGGL(k_relu_inplace, dim3(grid_for(count / 8)), dim3(256), 0, st, (_Float16*)x, count / 8);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_maxpool3x3s2_nhwc_f16(const void* in, void* out, int32_t n, int32_t h, int32_t w, int32_t c, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || n < 1 || h < 1 || w < 1 || c < 8) return LFD_ERR_INVALID_ARGUMENT;
  if (c % 8) return LFD_ERR_UNSUPPORTED;
  const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * oh * ow * (c / 8);
  hipLaunchKernelGGL(k_maxpool3s2, dim3(grid_for(total)), dim3(256), 0, st, (const _Float16*)in, (_Float16*)out, n, h, w, oh,
                     ow, c / 8);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int lfd_pack_level_outputs_f32(const float* src, float* dst, int32_t n, int32_t hw, int32_t src_channels, int32_t c0,
                               int32_t count, int32_t total_points, int32_t point_offset, float scale, int32_t op,
                               lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!src || !dst || n < 1 || hw < 0 || count < 1 || c0 < 0 || c0 + count > src_channels || point_offset < 0 ||
      point_offset + hw > total_points || (op != 0 && op != 1))
    return LFD_ERR_INVALID_ARGUMENT;
  if (hw == 0) return LFD_OK;
  const int64_t total = (int64_t)n * hw * count;
  hipLaunchKernelGGL(k_pack_level, dim3(grid_for(total)), dim3(256), 0, st, src, dst, n, hw, src_channels, c0, count,
                     total_points, point_offset, scale, op);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
