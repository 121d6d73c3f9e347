// csrc/conv_acc32.hip -- parity instrument: the MFMA conv kernels of conv_impl.h with the fp32 accumulators written out
// un-rounded (ACC32 = true).  The engine's G1 mode (lfd_amd/engine_g1.py, SURVEY 8d "Parity gates" G1) runs every conv of
// the network through these on operands split into fp16 hi + lo parts, so that the SAME tiling / addressing / MFMA
// contraction code is exercised while inter-layer storage stays fp32:
//     conv(x, w) ~= acc32(x_hi, w_hi) + acc32(x_lo, w_hi) + acc32(x_hi, w_lo)        (x_lo * w_lo ~ 2^-22, dropped)
// Not on the LFD product path (never called by engine.py); kept in the library so that the test runs the shipped code.
// The generic layer engine of the sibling meta-architectures (lfd_amd/engine_sibling.py) uses it for the heads' output
// convs, whose logits leave the network as fp32.
#include "conv_impl.h"

extern "C" int lfd_conv2d_nhwc_f16_acc32(const lfd_conv_desc_t* d, const void* in, float* out_f32, const void* w_packed,
                                         const float* bias, const void* zeros, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !in || !out_f32 || !w_packed || !bias || !zeros) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->h < 1 || d->w < 1 || d->tail_cout) return LFD_ERR_INVALID_ARGUMENT;
  if ((d->ks != 1 && d->ks != 3) || (d->stride != 1 && d->stride != 2)) return LFD_ERR_UNSUPPORTED;
  if (d->cout % 32 || d->cin % 16) return LFD_ERR_UNSUPPORTED;
  ConvArgs a{};
  a.in = (const _Float16*)in; a.out32 = out_f32; a.w = (const half8*)w_packed; a.bias = bias;
  a.zeros = (const _Float16*)zeros;
  a.N = d->n; a.H = d->h; a.W = d->w;
  const int pad = d->ks / 2;
  a.OH = (d->h + 2 * pad - d->ks) / d->stride + 1;
  a.OW = (d->w + 2 * pad - d->ks) / d->stride + 1;
  a.cout = d->cout;
#define ACC32_CASE(CIN, KS, S, NCT, WREG) \
  case CIN * 10000 + KS * 1000 + S * 100 + NCT * 10: return launch_conv_<CIN, KS, S, NCT, WREG, false, false, false, true>(a, st)
  switch (d->cin * 10000 + d->ks * 1000 + d->stride * 100 + (d->cout / 32) * 10) {
    // the (cin, ks, stride, cout) combinations the six shipped configurations contain (3-channel input padded to 32)
    ACC32_CASE(32, 3, 2, 1, true);  ACC32_CASE(32, 3, 2, 2, true);  ACC32_CASE(32, 1, 1, 1, true);  ACC32_CASE(32, 1, 2, 2, true);
    ACC32_CASE(64, 3, 1, 2, true);  ACC32_CASE(64, 3, 2, 2, true);  ACC32_CASE(64, 3, 2, 4, true);
    ACC32_CASE(64, 1, 1, 2, true);  ACC32_CASE(64, 1, 1, 4, true);  ACC32_CASE(64, 1, 2, 2, true);  ACC32_CASE(64, 1, 2, 4, true);
    ACC32_CASE(128, 3, 1, 4, false); ACC32_CASE(128, 3, 2, 4, false);
    ACC32_CASE(128, 1, 1, 2, true); ACC32_CASE(128, 1, 1, 4, true); ACC32_CASE(128, 1, 2, 4, true);
    // output convs of the sibling heads (FCOSHead's 3x3 cls / centerness / reg, LFDHead on 64 channels): fp32 logits
    ACC32_CASE(128, 3, 1, 2, false); ACC32_CASE(64, 3, 1, 1, true); ACC32_CASE(64, 1, 1, 1, true);
    default: return LFD_ERR_UNSUPPORTED;
  }
#undef ACC32_CASE
}
