// csrc/planes_ml.hip -- lfd_pl_conv2d_levels: one 1x1 conv launch of the neck / head over ALL pyramid levels
// (simple_neck.py:67-74, lfd_head.py:164-185: the same three convs per level, each level with its own filters).  Per level
// they were 3 launches x 5 levels; the four small levels' 12 launches took 159 us for a third of the first level's work --
// launch latency and tail, not arithmetic.  The persistent workgroups of planes_impl.h walk the concatenated tile list of
// the levels and switch filters / GroupNorm sums / outputs when they cross into another level.
#include "planes_impl.h"

using namespace pl;

extern "C" int lfd_pl_conv2d_levels(const lfd_pl_conv_desc_t* d, const lfd_pl_level_t* levels, int32_t num_levels, const void* zeros,
                                    lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !levels || !zeros || num_levels < 1 || num_levels > LFD_MAX_LEVELS) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->cout < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (d->ks != 1 || d->stride != 1) return LFD_ERR_UNSUPPORTED;
  const int outm = d->out_mode;
  if (outm < 0 || outm > 2) return LFD_ERR_INVALID_ARGUMENT;
  if (outm == 2 && (d->f_c0 < 0 || d->f_c1 < 0 || d->f_c0 + d->f_c1 < 1 || d->f_c0 + d->f_c1 > d->cout)) return LFD_ERR_INVALID_ARGUMENT;
  if (outm != 2 && (d->cout % 32)) return LFD_ERR_UNSUPPORTED;
  const bool tail = d->tail_cout > 0;
  if (tail && d->tail_cout != d->cout) return LFD_ERR_UNSUPPORTED;
  const bool gnin = levels[0].gn_in_sums != nullptr;
  if (gnin && d->cin != 128) return LFD_ERR_INVALID_ARGUMENT;
  const int nslab = (d->cout + 31) / 32, nk = d->cin / 16;
  PlArgs a{};
  PlLevels L{};
  L.n = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    const lfd_pl_level_t& s = levels[i];
    if (!s.in || !s.w_packed || !s.bias || s.h < 1 || s.w < 1) return LFD_ERR_INVALID_ARGUMENT;
    if (outm == 2 ? ((d->f_c0 > 0 && !s.f_out0) || (d->f_c1 > 0 && !s.f_out1)) : !s.out) return LFD_ERR_INVALID_ARGUMENT;
    if (outm == 1 && !s.gn_sums) return LFD_ERR_INVALID_ARGUMENT;
    if (tail && (!s.tail_w_packed || !s.tail_bias)) return LFD_ERR_INVALID_ARGUMENT;
    if ((s.gn_in_sums != nullptr) != gnin || (gnin && (!s.gn_in_gamma || !s.gn_in_beta))) return LFD_ERR_INVALID_ARGUMENT;
    if (!lfd_aligned16(s.in) || !lfd_aligned16(s.out) || (s.in_plane_halfs & 7) || (s.out_plane_halfs & 7)) return LFD_ERR_INVALID_ARGUMENT;
    // every level's filter is reloaded when the walk enters it; the kernel tells "another level" by the filter pointer
    for (int j = 0; j < i; ++j)
      if (levels[j].w_packed == s.w_packed && (levels[j].bias != s.bias || levels[j].tail_w_packed != s.tail_w_packed ||
                                                levels[j].tail_bias != s.tail_bias))
        return LFD_ERR_INVALID_ARGUMENT;
    PlLevel& l = L.lv[i];
    l.in = (const _Float16*)s.in; l.out = (_Float16*)s.out;
    l.w = (const half8*)s.w_packed; l.bias = s.bias;
    l.w2 = tail ? (const half8*)s.tail_w_packed : nullptr; l.bias2 = s.tail_bias;
    l.gnin_gamma = s.gn_in_gamma; l.gnin_beta = s.gn_in_beta;
    l.gn_acc = (unsigned long long*)s.gn_sums; l.gnin_acc = (const unsigned long long*)s.gn_in_sums;
    l.f_out0 = s.f_out0; l.f_out1 = s.f_out1; l.scale1 = s.scale1;
    l.in_plane = s.in_plane_halfs; l.out_plane = s.out_plane_halfs;
    l.H = s.h; l.W = s.w;
  }
  a.w_plane = (long)nslab * nk * 64;
  a.w2_plane = tail ? (long)(d->tail_cout / 32) * (d->cout / 16) * 64 : 0;
  a.zeros = (const _Float16*)zeros;
  a.N = d->n;
  a.cout = d->cout; a.cout2 = d->tail_cout; a.relu = d->relu; a.relu2 = d->tail_relu;
  a.f_c0 = d->f_c0; a.f_c1 = d->f_c1; a.f_img0 = d->f_image_stride0; a.f_img1 = d->f_image_stride1;
  a.gnin_eps = d->gn_in_eps;
  const int key = d->cin * 10 + nslab;
  if (outm == 2) {
    // cls / reg outputs (<= 32 | <= 64 channels), GroupNorm of the producer applied to the landed tile
    if (!gnin || tail) return LFD_ERR_UNSUPPORTED;
    if (key == 1281) return launch_pl_ml_<128, 1, false, 2, 1, true>(a, L, st);
    if (key == 1282) return launch_pl_ml_<128, 2, false, 2, 0, true>(a, L, st);
    return LFD_ERR_UNSUPPORTED;
  }
  if (nslab != 4) return LFD_ERR_UNSUPPORTED;
  if (gnin) {
    if (tail || d->cin != 128) return LFD_ERR_UNSUPPORTED;
    if (outm == 1) return launch_pl_ml_<128, 4, false, 1, 0, true>(a, L, st);      // later tower convs
    return LFD_ERR_UNSUPPORTED;
  }
  if (tail) {
    if (outm != 1) return LFD_ERR_UNSUPPORTED;
    // neck conv + first tower conv
    if (d->cin == 64) return launch_pl_ml_<64, 4, true, 1, 0, false>(a, L, st);
    if (d->cin == 128) return launch_pl_ml_<128, 4, true, 1, 0, false>(a, L, st);
    return LFD_ERR_UNSUPPORTED;
  }
  if (outm == 0) {
    // separate towers (no merge path): the neck conv alone
    if (d->cin == 64) return launch_pl_ml_<64, 4, false, 0, 0, false>(a, L, st);
    if (d->cin == 128) return launch_pl_ml_<128, 4, false, 0, 0, false>(a, L, st);
    return LFD_ERR_UNSUPPORTED;
  }
  if (d->cin == 128) return launch_pl_ml_<128, 4, false, 1, 0, false>(a, L, st);   // ... and each tower's first conv
  return LFD_ERR_UNSUPPORTED;
}
