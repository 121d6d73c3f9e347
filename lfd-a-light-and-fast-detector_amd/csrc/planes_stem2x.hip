// csrc/planes_stem2x.hip -- lfd_pl_stem2x: the whole 'faster' stem (lfd_resnet.py:376-413: conv3x3 s2 3 -> 64, conv1x1,
// conv3x3 s2 64 -> 64, conv1x1, each + BN + ReLU) on hi/lo planes in ONE launch.  As two launches (lfd_pl_stem_pair +
// lfd_pl_conv2d with a chained 1x1) the pair-1 output -- N x H/2 x W/2 x 64 x 2 planes, 1.06 GB at bs 8 / 1080p, the largest
// tensor of the network -- was written (VALU-bound producer + 130 us of stores) and read back (HBM-bound consumer): 665 of the
// 1790 us forward.  Here the persistent workgroups of the second pair's weights-stationary kernel (planes_impl.h, 288 filter
// registers per wave) COMPUTE their 9 x 33-pixel input tile from a 19 x 68-pixel frame patch instead of fetching it
// (pl_block<..., PFMT>: frame_dma / frame_fill / produce); csrc/stem_fused.hip is the fp16 form of the same idea.
#include "planes_impl.h"

using namespace pl;

int lfd_pl_stem2xs_launch(pl::PlArgs a, const pl::PlProd& p, int in_format, hipStream_t st);   // planes_stem2xs.hip

namespace {

template <int FMT>
__global__ __launch_bounds__(256, 1) void k_pl_stem2x(PlArgs a, PlProd p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PlLevels none;
  none.n = 0;
  pl_block<64, 3, 2, 2, true, true, false, false, 0, 0, false, false, FMT>(a, none, smem, p);
}

template <int FMT>
int launch_stem2x(PlArgs a, const PlProd& p, hipStream_t st) {
  using C = PCfg<64, 3, 2, 2, true, false, 0, true>;
  a.tiles_x = (a.OW + C::TW - 1) / C::TW;
  a.tiles_y = (a.OH + C::TH - 1) / C::TH;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  constexpr int LDSB = C::LDS_BYTES + ProdCfg::BYTES;
  auto kern = k_pl_stem2x<FMT>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  int blocks = 256;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  hipLaunchKernelGGL(kern, dim3(blocks, 1), dim3(256), LDSB, st, a, p);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // namespace

#ifdef LFD_PL_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_pl_stem2x_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pl::g_pl_dbg), sizeof(unsigned long long) * 128);
}
#endif

extern "C" int lfd_pl_stem2x(const void* in, int32_t in_format, int32_t n, int32_t h, int32_t w, const void* w1_packed,
                             const void* w2_packed, const float* b2, const void* w3_packed, const float* b3,
                             const void* w4_packed, const float* b4, void* out, int64_t out_plane_halfs, const void* zeros,
                             lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !w1_packed || !w2_packed || !b2 || !w3_packed || !b3 || !w4_packed || !b4 || !out || !zeros || n < 1 || h < 1 ||
      w < 1)
    return LFD_ERR_INVALID_ARGUMENT;
  if (!lfd_aligned16(out) || (out_plane_halfs & 7)) return LFD_ERR_INVALID_ARGUMENT;
  PlProd p{};
  p.frame = in; p.FH = h; p.FW = w;
  p.dma_ok = (in_format == IN_NHWC_F16 && (w & 7) == 0 && ((uintptr_t)in & 15) == 0) ? 1 : 0;
  p.w1 = (const half8*)w1_packed; p.w2 = (const half8*)w2_packed; p.b2 = b2;
  PlArgs a{};
  a.out = (_Float16*)out; a.out_plane = out_plane_halfs;
  a.w = (const half8*)w3_packed; a.w_plane = 2L * 36 * 64; a.bias = b3;
  a.w2 = (const half8*)w4_packed; a.w2_plane = 2L * 4 * 64; a.bias2 = b4;
  a.zeros = (const _Float16*)zeros;
  a.N = n; a.H = (h + 1) / 2; a.W = (w + 1) / 2;          // the mid tensor (pair 1's output) the consumer's geometry refers to
  a.OH = (a.H + 1) / 2; a.OW = (a.W + 1) / 2;
  a.cout = 64; a.cout2 = 64; a.relu = 1; a.relu2 = 1;
  if (lfd_tune(LFD_TUNE_PL_STEM) == 1) {
    if (p.dma_ok) return lfd_pl_stem2xs_launch(a, p, IN_NHWC_F16, st);
    if (in_format == IN_NHWC_U8 && (w & 15) == 0 && ((uintptr_t)in & 15) == 0) return lfd_pl_stem2xs_launch(a, p, IN_NHWC_U8, st);
    if (in_format == IN_NCHW_F32 && (w & 3) == 0 && ((uintptr_t)in & 15) == 0) return lfd_pl_stem2xs_launch(a, p, IN_NCHW_F32, st);
  }
  switch (in_format) {
    case IN_NCHW_F32: return launch_stem2x<IN_NCHW_F32>(a, p, st);
    case IN_NHWC_F16: return launch_stem2x<IN_NHWC_F16>(a, p, st);
    case IN_NHWC_U8: return launch_stem2x<IN_NHWC_U8>(a, p, st);
    default: return LFD_ERR_INVALID_ARGUMENT;
  }
}
