// csrc/conv_stats.hip -- train-mode conv + BatchNorm batch statistics in one pass over the output.
//
// The reference's unit is nn.Conv2d(bias=False) -> nn.BatchNorm2d in train mode (lfd_resnet.py:96-154 blocks, :354-439
// stem, :458-468 downsample).  The conv kernels of conv_impl.h, instantiated with STATS, sum every channel of the fp16
// values they store (and their squares) while copying them out, one row of partials per workgroup; k_bn_stats_final
// (train.hip) adds the rows in fp64 and updates the running statistics.  Shapes without a STATS instantiation (the tiny
// last-stage maps, whose split-K kernel has no single owner of an output pixel) run the conv and the two statistics
// launches of lfd_bn_train_stats_f16 -- the result is the same up to the order of the fp32 partial sums.
#include "conv_impl.h"

int lfd_bn_stats_final_launch(const float* partials, int nblocks, int channels, double pixels, float eps, float momentum,
                              float* running_mean, float* running_var, float* stats, hipStream_t st);

namespace {

template <int CIN, int KS, int S, int NCT, bool WREG>
int launch_stats(const ConvArgs& a, hipStream_t st, int* blocks) {
  if constexpr (KS == 1 && S == 1 && CIN == 64) {
    if (a.pro_stats) return launch_conv_<CIN, KS, S, NCT, WREG, false, false, false, false, true, true>(a, st, blocks);
  }
  if (a.pro_stats) return LFD_ERR_UNSUPPORTED;
  return launch_conv_<CIN, KS, S, NCT, WREG, false, false, false, false, true>(a, st, blocks);
}

}  // namespace

extern "C" {

static int conv_bn_stats(const lfd_conv_desc_t* d, const void* in, const float* in_stats, const float* in_gamma,
                         const float* in_beta, void* out, const void* w_packed,
                         const float* bias, const void* zeros, float eps, float momentum, float* running_mean,
                         float* running_var, void* workspace, size_t workspace_bytes, float* stats,
                         lfd_stream_t stream, int32_t* rows_only = nullptr) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !in || !out || !w_packed || !bias || !zeros || !workspace || (!stats && !rows_only)) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->h < 1 || d->w < 1 || d->tail_cout || d->relu) return LFD_ERR_INVALID_ARGUMENT;
  if (!lfd_aligned16(in) || !lfd_aligned16(out)) return LFD_ERR_INVALID_ARGUMENT;
  if ((d->ks != 1 && d->ks != 3) || (d->stride != 1 && d->stride != 2)) return LFD_ERR_UNSUPPORTED;
  if ((running_mean == nullptr) != (running_var == nullptr)) return LFD_ERR_INVALID_ARGUMENT;
  // rows_only: `workspace` is the caller's own row buffer (512 rows x 2 x cout floats suffice)
  if (workspace_bytes < (rows_only ? (size_t)512 * 2 * d->cout * sizeof(float) : lfd_train_workspace_bytes())) return LFD_ERR_WORKSPACE_TOO_SMALL;
  ConvArgs a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out; a.w = (const half8*)w_packed; a.bias = bias;
  a.res_px = d->cout; a.zeros = (const _Float16*)zeros;
  a.N = d->n; a.H = d->h; a.W = d->w;
  const int pad = d->ks / 2;
  a.OH = (d->h + 2 * pad - d->ks) / d->stride + 1;
  a.OW = (d->w + 2 * pad - d->ks) / d->stride + 1;
  a.cout = d->cout;
  a.stat_partials = reinterpret_cast<float*>(workspace);
  a.pro_stats = in_stats; a.pro_gamma = in_gamma; a.pro_beta = in_beta;
  const int64_t pixels = (int64_t)a.N * a.OH * a.OW;
  int blocks = 0, rc = LFD_ERR_UNSUPPORTED;
  switch (d->cin * 10000 + d->ks * 1000 + d->stride * 100 + (d->cout / 32) * 10 + (d->cout % 32 ? 1 : 0)) {
    case 64 * 10000 + 3100 + 20: rc = launch_stats<64, 3, 1, 2, true>(a, st, &blocks); break;
    case 64 * 10000 + 3200 + 20: rc = launch_stats<64, 3, 2, 2, true>(a, st, &blocks); break;
    case 64 * 10000 + 3200 + 40: rc = launch_stats<64, 3, 2, 4, true>(a, st, &blocks); break;
    case 64 * 10000 + 1100 + 20: rc = launch_stats<64, 1, 1, 2, true>(a, st, &blocks); break;
    case 64 * 10000 + 1100 + 40: rc = launch_stats<64, 1, 1, 4, true>(a, st, &blocks); break;      // the neck on 64-channel taps
    case 64 * 10000 + 1200 + 20: rc = launch_stats<64, 1, 2, 2, true>(a, st, &blocks); break;
    case 64 * 10000 + 1200 + 40: rc = launch_stats<64, 1, 2, 4, true>(a, st, &blocks); break;
    case 128 * 10000 + 1100 + 40: rc = launch_stats<128, 1, 1, 4, true>(a, st, &blocks); break;
    default: {
      if (in_stats || rows_only) return LFD_ERR_UNSUPPORTED;
      rc = lfd_conv2d_nhwc_f16(d, in, out, w_packed, bias, nullptr, nullptr, nullptr, zeros, stream);
      if (rc != LFD_OK) return rc;
      return lfd_bn_train_stats_f16(out, pixels, d->cout, eps, momentum, running_mean, running_var, workspace, workspace_bytes,
                                    stats, stream);
    }
  }
  if (rc != LFD_OK) return rc;
  if (rows_only) { *rows_only = blocks; return LFD_OK; }
  return lfd_bn_stats_final_launch(a.stat_partials, blocks, d->cout, (double)pixels, eps, momentum, running_mean, running_var,
                                   stats, st);
}

int lfd_conv2d_bn_stats_nhwc_f16(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed,
                                 const float* bias, const void* zeros, float eps, float momentum, float* running_mean,
                                 float* running_var, void* workspace, size_t workspace_bytes, float* stats,
                                 lfd_stream_t stream) {
  return conv_bn_stats(d, in, nullptr, nullptr, nullptr, out, w_packed, bias, zeros, eps, momentum, running_mean, running_var,
                       workspace, workspace_bytes, stats, stream);
}

// the conv alone, its per-workgroup statistics rows left in the CALLER's buffer (the final pass comes later, for several convs
// at once: lfd_bn_train_finish_into_levels_f16); LFD_ERR_UNSUPPORTED for shapes without a STATS kernel
int lfd_conv2d_bn_partials_nhwc_f16(const lfd_conv_desc_t* d, const void* in, void* out, const void* w_packed, const float* bias,
                                    const void* zeros, float* rows, size_t rows_bytes, int32_t* nrows, lfd_stream_t stream) {
  if (!nrows || !rows) return LFD_ERR_INVALID_ARGUMENT;
  return conv_bn_stats(d, in, nullptr, nullptr, nullptr, out, w_packed, bias, zeros, 0.f, 0.f, nullptr, nullptr, rows, rows_bytes,
                       nullptr, stream, nrows);
}

int lfd_conv1x1_of_bn_relu_bn_stats_nhwc_f16(const lfd_conv_desc_t* d, const void* y_in, const float* in_stats,
                                             const float* in_gamma, const float* in_beta, void* out, const void* w_packed,
                                             const float* bias, const void* zeros, float eps, float momentum,
                                             float* running_mean, float* running_var, void* workspace,
                                             size_t workspace_bytes, float* stats, lfd_stream_t stream) {
  if (!d || !in_stats || !in_gamma || !in_beta || d->ks != 1 || d->stride != 1) return LFD_ERR_INVALID_ARGUMENT;
  return conv_bn_stats(d, y_in, in_stats, in_gamma, in_beta, out, w_packed, bias, zeros, eps, momentum, running_mean,
                       running_var, workspace, workspace_bytes, stats, stream);
}

// The 1x1 stride-1 data-gradient conv of a stem pair with the BatchNorm backward sums of the unit it feeds taken in its
// epilogue (conv_impl.h BSUM): rows of [2][cout] partial sums in `workspace`, *sum_rows of them -- the input of
// lfd_bn_train_bwd_rows_f16 / lfd_stem_conv0_bn_bwd_wgrad_rows, which skip their own sums pass.
int lfd_conv1x1_dgrad_bn_bwd_sums_nhwc_f16(const lfd_conv_desc_t* d, const void* dy, void* dz, const void* w_packed,
                                           const float* bias, const void* zeros, const void* y_unit, const float* unit_stats,
                                           const float* unit_gamma, const float* unit_beta, void* workspace,
                                           size_t workspace_bytes, int32_t* sum_rows, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !dy || !dz || !w_packed || !bias || !zeros || !y_unit || !unit_stats || !unit_gamma || !unit_beta || !workspace || !sum_rows)
    return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->h < 1 || d->w < 1 || d->tail_cout || d->relu || d->ks != 1 || d->stride != 1) return LFD_ERR_INVALID_ARGUMENT;
  if (d->cin != 64 || d->cout != 64) return LFD_ERR_UNSUPPORTED;
  if (!lfd_aligned16(dy) || !lfd_aligned16(dz) || !lfd_aligned16(y_unit)) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_train_workspace_bytes()) return LFD_ERR_WORKSPACE_TOO_SMALL;
  ConvArgs a{};
  a.in = (const _Float16*)dy; a.out = (_Float16*)dz; a.w = (const half8*)w_packed; a.bias = bias;
  a.res_px = d->cout; a.zeros = (const _Float16*)zeros;
  a.N = d->n; a.H = d->h; a.W = d->w; a.OH = d->h; a.OW = d->w;
  a.cout = d->cout;
  a.stat_partials = reinterpret_cast<float*>(workspace);
  a.bsum_y = (const _Float16*)y_unit; a.bsum_stats = unit_stats; a.bsum_gamma = unit_gamma; a.bsum_beta = unit_beta;
  int blocks = 0;
  const int rc = launch_conv_<64, 1, 1, 2, true, false, false, false, false, false, false, true>(a, st, &blocks);
  if (rc != LFD_OK) return rc;
  *sum_rows = blocks;
  return LFD_OK;
}

}  // extern "C"
