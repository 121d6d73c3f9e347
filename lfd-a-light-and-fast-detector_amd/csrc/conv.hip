// csrc/conv.hip -- C ABI entry points of the NHWC fp16 implicit-GEMM convolution (kernel templates: conv_impl.h).
#include "conv_impl.h"
#include <cstdlib>

// csrc/conv_small.hip: split-K, one output slab per workgroup -- the 128 -> 128 3x3 convs of the small last-stage maps
int lfd_conv128_splitk_launch(const _Float16* in, _Float16* out, const void* w_packed, const float* bias, const _Float16* res,
                              int n, int h, int w, int relu, hipStream_t st);

#ifdef LFD_CONV_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_conv_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_dbg), sizeof(unsigned long long) * 128);
}
#endif

extern "C" {

// Packed-weight sizes: [cout/32][NK][64 lanes] half8  (NK = ks*ks*cin/16).
size_t lfd_conv_packed_weight_halfs(int32_t cin, int32_t cout, int32_t ks) {
  return (size_t)(cout / 32) * (size_t)(ks * ks * cin / 16) * 64 * 8;
}

static int conv_dispatch(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed,
                         const float* bias, const void* residual, const void* tail_w_packed,
                         const float* tail_bias, const void* ds_w_packed, const float* ds_bias, void* ds_out,
                         const void* zeros, lfd_stream_t stream);

int lfd_conv2d_nhwc_f16(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed,
                        const float* bias, const void* residual, const void* tail_w_packed,
                        const float* tail_bias, const void* zeros, lfd_stream_t stream) {
  return conv_dispatch(d, in, out, w_packed, bias, residual, tail_w_packed, tail_bias, nullptr, nullptr, nullptr, zeros,
                       stream);
}

int lfd_conv2d_downsample_nhwc_f16(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed,
                                   const float* bias, const void* ds_w_packed, const float* ds_bias, void* ds_out,
                                   const void* zeros, lfd_stream_t stream) {
  if (!ds_w_packed || !ds_bias || !ds_out || !d || d->ks != 3 || d->stride != 2 || d->tail_cout) return LFD_ERR_INVALID_ARGUMENT;
  return conv_dispatch(d, in, out, w_packed, bias, nullptr, nullptr, nullptr, ds_w_packed, ds_bias, ds_out, zeros, stream);
}

static int conv_dispatch(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed,
                         const float* bias, const void* residual, const void* tail_w_packed,
                         const float* tail_bias, const void* ds_w_packed, const float* ds_bias, void* ds_out,
                         const void* zeros, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !in || !out || !w_packed || !bias || !zeros) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->h < 1 || d->w < 1) return LFD_ERR_INVALID_ARGUMENT;
  if ((d->ks != 1 && d->ks != 3) || (d->stride != 1 && d->stride != 2)) return LFD_ERR_UNSUPPORTED;
  if (d->cout % 32 || d->cin % 16) return LFD_ERR_UNSUPPORTED;
  ConvArgs a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out; a.w = (const half8*)w_packed; a.bias = bias;
  a.res = (const _Float16*)residual; a.res_px = d->cout; a.w2 = (const half8*)tail_w_packed; a.bias2 = tail_bias;
  a.zeros = (const _Float16*)zeros;
  a.wds = (const half8*)ds_w_packed; a.bds = ds_bias; a.out_ds = (_Float16*)ds_out;
  a.N = d->n; a.H = d->h; a.W = d->w;
  const int pad = d->ks / 2;
  a.OH = (d->h + 2 * pad - d->ks) / d->stride + 1;
  a.OW = (d->w + 2 * pad - d->ks) / d->stride + 1;
  a.cout = d->cout; a.cout2 = d->tail_cout; a.relu = d->relu; a.relu2 = d->tail_relu;
  const bool tail = d->tail_cout > 0;
  if (tail && (!tail_w_packed || !tail_bias || d->tail_cout != d->cout || residual)) return LFD_ERR_UNSUPPORTED;
  const int key = d->cin * 10000 + d->ks * 1000 + d->stride * 100 + (d->cout / 32) * 10 + (tail ? 1 : 0);
  switch (key) {
    // ---- 64-channel backbone body
    case 64 * 10000 + 3100 + 20: return launch_conv<64, 3, 1, 2, true, false>(a, st);
    case 64 * 10000 + 3200 + 20: return launch_conv<64, 3, 2, 2, true, false>(a, st);
    case 64 * 10000 + 3200 + 21: return launch_conv<64, 3, 2, 2, true, true>(a, st);
    case 64 * 10000 + 3200 + 40: return launch_conv<64, 3, 2, 4, true, false>(a, st);
    case 64 * 10000 + 3100 + 40: return launch_conv<64, 3, 1, 4, true, false>(a, st);   // 3x3 head tower on a 64-channel neck (sibling heads)
    case 64 * 10000 + 3100 + 10: return launch_conv<64, 3, 1, 1, true, false>(a, st);   // data gradient of 32->64 s2 (XS)
    case 64 * 10000 + 1100 + 10: return launch_conv<64, 1, 1, 1, true, false>(a, st);   // data gradient of the 32->64 downsample
    case 64 * 10000 + 1100 + 20: return launch_conv<64, 1, 1, 2, true, false>(a, st);
    case 64 * 10000 + 1100 + 40: return launch_conv<64, 1, 1, 4, true, false>(a, st);
    case 64 * 10000 + 1200 + 20: return launch_conv<64, 1, 2, 2, true, false>(a, st);
    case 64 * 10000 + 1200 + 40: return launch_conv<64, 1, 2, 4, true, false>(a, st);
    // ---- 128-channel stages (tiny maps): weights streamed from L2 per k-step
    // (weights streamed from L2 per k-step: right for the tiny maps of every shipped configuration.  On the LARGE maps of the
    //  sibling heads -- 128-channel 3x3 towers at stride 4-8 -- a 4-row x 64-cout tile, <128,3,1,2>, halves the filter bytes
    //  streamed per pixel but measured slower, FCOS forward 3.25 vs 2.97 ms: the two waves sharing a filter slab do not hit
    //  in each other's fetches.  A register-stationary 128-channel variant is what those maps would want.)
    case 128 * 10000 + 3100 + 40: {
      // small maps (the 17 x 30 / 23 x 40 last stages): 4 x the workgroups, a quarter of the filter and of the k-steps each
      // (conv_small.hip); large maps (sibling heads) keep the streamed-weight kernel.  LFD_CONV128_SPLITK=0: A/B switch.
      const bool splitk = lfd_tune(LFD_TUNE_CONV128_SPLITK) != 0;
      if (splitk && (long)a.N * a.OH * a.OW <= 16384)
        return lfd_conv128_splitk_launch(a.in, a.out, w_packed, bias, a.res, a.N, a.H, a.W, a.relu, st);
      return launch_conv<128, 3, 1, 4, false, false>(a, st);
    }
    case 128 * 10000 + 3200 + 40: return launch_conv<128, 3, 2, 4, false, false>(a, st);
    case 128 * 10000 + 3100 + 20: return launch_conv<128, 3, 1, 2, false, false>(a, st);  // data gradient of 64->128 s2
    case 128 * 10000 + 3200 + 20: return launch_conv<128, 3, 2, 2, false, false>(a, st);  // FPN extra level on a 128-channel input, 64 outputs
    case 128 * 10000 + 1100 + 20: return launch_conv<128, 1, 1, 2, true, false>(a, st);   // data gradient of the 64->128 downsample
    case 128 * 10000 + 1100 + 40: return launch_conv<128, 1, 1, 4, true, false>(a, st);
    case 128 * 10000 + 1200 + 40: return launch_conv<128, 1, 2, 4, true, false>(a, st);
    // ---- 32-channel stem of the XS model
    case 32 * 10000 + 3200 + 11: return launch_conv<32, 3, 2, 1, true, true>(a, st);
    case 32 * 10000 + 3200 + 10: return launch_conv<32, 3, 2, 1, true, false>(a, st);
    case 32 * 10000 + 3200 + 20: return launch_conv<32, 3, 2, 2, true, false>(a, st);
    case 32 * 10000 + 3100 + 10: return launch_conv<32, 3, 1, 1, true, false>(a, st);
    case 32 * 10000 + 1100 + 10: return launch_conv<32, 1, 1, 1, true, false>(a, st);
    case 32 * 10000 + 1200 + 20: return launch_conv<32, 1, 2, 2, true, false>(a, st);
    case 32 * 10000 + 1100 + 40: return launch_conv<32, 1, 1, 4, true, false>(a, st);   // data gradient of the head's output convs (<= 32 padded channels -> 128)
    default: return LFD_ERR_UNSUPPORTED;
  }
}

}  // extern "C"
