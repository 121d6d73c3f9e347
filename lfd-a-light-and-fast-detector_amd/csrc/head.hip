// csrc/head.hip -- neck + shared detection head for one pyramid level, on gfx950 MFMA.
//
// Replaces SimpleNeck.forward (reference lfd/model/neck/simple_neck.py:67-74: conv1x1+BN+ReLU)
// chained into LFDHead.forward (reference lfd/model/head/lfd_head.py:164-185: two
// conv1x1 + GroupNorm(16) + ReLU, then cls / reg conv1x1 (+ per-level Scale :179-180)) and the
// NCHW -> [N, P, C] re-layout + level concat of LFD.forward (reference lfd/model/lfd.py:526-542).
//
// GroupNorm needs per-(image, group) statistics over ALL pixels of the level, i.e. a grid-wide
// reduction between two pointwise convs.  Instead of materialising the 128-channel pre-norm
// tensors in HBM (write + re-read per conv), the chain is RECOMPUTED from the 64/128-channel
// backbone tap in three passes of the same kernel:
//   pass 1: neck -> conv1                         -> per-tile (sum, sumsq) of conv1 per group
//   pass 2: neck -> conv1 -> GN1+ReLU -> conv2    -> per-tile (sum, sumsq) of conv2 per group
//   pass 3: neck -> conv1 -> GN1+ReLU -> conv2 -> GN2+ReLU -> cls/reg conv -> fp32 outputs
// (+13 % MFMA work for the whole network, -65 % head HBM bytes; pre-norm values never leave fp32
// registers).  Partial sums are combined in fp64 in a fixed order (deterministic).
//
// Per workgroup: 4 waves, wave = one 32-channel output tile of the 128 head channels, all
// waves share the same 64 pixels; stage outputs are exchanged through swizzled LDS tiles;
// weights of every stage stay in VGPRs across the persistent tile loop.
#include <type_traits>
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HC = 128;        // head / neck channels
constexpr int TPX = 64;        // pixels per tile (2 MFMA pixel tiles)
constexpr int NGROUP_MAX = 32;

struct HeadArgs {
  const _Float16* x;     // [N, HW, CIN] backbone tap (NHWC, flattened pixels)
  const half8* wn;       // neck   [4][CIN/16][64]
  const float* bn;       // neck bias (BN folded) [128]
  const half8* w1;       // tower conv 1 [4][8][64]
  const half8* w2;       // tower conv 2 [4][8][64]
  const half8* wf;       // final conv   [FT][8][64]  (cout padded to FT*32)
  const float* bf;       // final bias   [FT*32]
  const float* ab1;      // [N][128][2]  GN1 (scale, shift) per image/channel   (pass >= 2)
  const float* ab2;      // [N][128][2]  GN2                                    (pass == 3)
  float* part;           // [N*tiles][16 groups... up to 128/gsize][2] partial stats (pass 1, 2)
  float* out_cls;        // [N, P, CC]
  float* out_reg;        // [N, P, 4]
  const float* scale;    // per-level Scale parameter (device scalar) or nullptr
  int N, HW;             // images, pixels per image at this level
  int P, p_off;          // total points per image, offset of this level
  int CC;                // classification channels
  int split;             // final couts [0,split) -> cls, [split, split+4) -> reg (split = fcout if no reg)
  int fcout;             // valid final couts
  int gshift;            // log2(channels per group)  (GroupNorm(16,128) -> 3)
  int tiles_per_img, ntiles;
  const _Float16* zeros;
};

__device__ __forceinline__ void dma16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// swizzled LDS tile of [TPX pixels][C channels] fp16: byte address of (pixel, 16-byte chunk c)
template <int CPP>
__device__ __forceinline__ int lds_addr(int px, int c) {
  constexpr int PPR = (CPP >= 16) ? 1 : 16 / CPP;
  return px * (CPP * 16) + ((c ^ ((px / PPR) % CPP)) * 16);
}

template <int CIN, int PASS, int FT>
__global__ __launch_bounds__(256, 2) void k_head(HeadArgs a) {
  constexpr int CPPX = CIN / 8;           // chunks per pixel of the input tile
  constexpr int XBYTES = TPX * CIN * 2;
  constexpr int ABYTES = TPX * HC * 2;
  constexpr int NKN = CIN / 16, NKH = HC / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xbuf0 = smem;                        // 2 x XBYTES (double-buffered input tile)
  char* bufA = smem + 2 * XBYTES;            // stage ping
  char* bufB = bufA + ABYTES;                // stage pong

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pix = lane & 31;
  const int ct = wave;                       // 32-channel tile of the 128 head channels

  half8 wn[NKN], w1[NKH], w2[PASS >= 2 ? NKH : 1], wf[PASS == 3 ? NKH : 1];
#pragma unroll
  for (int k = 0; k < NKN; ++k) wn[k] = a.wn[(ct * NKN + k) * 64 + lane];
#pragma unroll
  for (int k = 0; k < NKH; ++k) w1[k] = a.w1[(ct * NKH + k) * 64 + lane];
  if constexpr (PASS >= 2) {
#pragma unroll
    for (int k = 0; k < NKH; ++k) w2[k] = a.w2[(ct * NKH + k) * 64 + lane];
  }
  // final stage: wave -> (cout tile fct, pixel tile fpt); FT=1: waves 0,1 ; FT=2: all four
  const int fct = wave % FT, fpt = wave / FT;
  const bool f_active = fpt < 2;
  if constexpr (PASS == 3) if (f_active) {
#pragma unroll
    for (int k = 0; k < NKH; ++k) wf[k] = a.wf[(fct * NKH + k) * 64 + lane];
  }
  float bnv[16];
  {
    const float* bp = a.bn + ct * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
      bnv[4 * g] = b4.x; bnv[4 * g + 1] = b4.y; bnv[4 * g + 2] = b4.z; bnv[4 * g + 3] = b4.w;
    }
  }
  const float scale = (PASS == 3 && a.scale) ? a.scale[0] : 1.f;

  auto issue_dma = [&](int t, int buf) {
    const int n = t / a.tiles_per_img;
    const int p0 = (t - n * a.tiles_per_img) * TPX;
    constexpr int SPW = 64 / CPPX;
    char* lbase = xbuf0 + buf * XBYTES;
    for (int slot0 = wave * SPW; slot0 < TPX; slot0 += 4 * SPW) {
      const int px = slot0 + lane / CPPX;
      const int cs = lane % CPPX;
      constexpr int PPR = (CPPX >= 16) ? 1 : 16 / CPPX;
      const int c = cs ^ ((px / PPR) % CPPX);
      const bool valid = (p0 + px) < a.HW;
      const _Float16* src = valid ? a.x + ((size_t)n * a.HW + p0 + px) * CIN + c * 8 : a.zeros + c * 8;
      dma16(src, lbase + slot0 * (CIN * 2));
    }
  };

  // one MFMA stage: acc[pt] (+)= W(ct) x tile(src)   (K = NK*16 channels)
  auto stage = [&](f32x16 (&acc)[2], const half8* w, const char* src, auto cpp_tag, int nk) {
    constexpr int CPP = decltype(cpp_tag)::value;
#pragma unroll
    for (int q = 0; q < (CPP / 2); ++q) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const half8 xf = *reinterpret_cast<const half8*>(src + lds_addr<CPP>(pt * 32 + pix, 2 * q + h));
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[q], xf, acc[pt], 0, 0, 0);
      }
    }
    (void)nk;
  };
  // write relu(acc*sc + sh) as fp16 into a [TPX][128] LDS tile (this wave's 32 channels)
  auto store_tile = [&](const f32x16 (&acc)[2], char* dst, const float* sc, const float* sh) {
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int px = pt * 32 + pix;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = acc[pt][4 * g + j];
          if (sc) x = x * sc[4 * g + j] + sh[4 * g + j];
          v[j] = (_Float16)fmaxf(x, 0.f);
        }
        *reinterpret_cast<half4*>(dst + lds_addr<HC / 8>(px, ct * 4 + g) + 8 * h) = v;
      }
    }
  };
  // per-tile GroupNorm partial statistics of this wave's 32 channels
  auto write_stats = [&](const f32x16 (&acc)[2], int t, int p0) {
    const float v0 = (p0 + pix) < a.HW ? 1.f : 0.f, v1 = (p0 + 32 + pix) < a.HW ? 1.f : 0.f;
    float s[16], ss[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float y0 = acc[0][i] * v0, y1 = acc[1][i] * v1;
      s[i] = y0 + y1;
      ss[i] = y0 * y0 + y1 * y1;
    }
    // channel of acc index i = 8*(i>>2) + 4*h + (i&3).  Reduce to groups of 2^gshift channels.
    // gshift <= 2: (i&3)>>gshift-subgroups stay lane-local per h; gshift == 3: whole g, both h.
    const int ngl = 32 >> a.gshift;  // groups in this wave's tile
    float* dst = a.part + ((size_t)t * (HC >> a.gshift) + (size_t)ct * ngl) * 2;
    if (a.gshift >= 3) {
      const int gg = 1 << (a.gshift - 3);  // 8-channel blocks per group
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float x = s[4 * g] + s[4 * g + 1] + s[4 * g + 2] + s[4 * g + 3];
        float xx = ss[4 * g] + ss[4 * g + 1] + ss[4 * g + 2] + ss[4 * g + 3];
        x = wave_sum_f(x);
        xx = wave_sum_f(xx);
        if (lane == 0) { atomicAdd(dst + (g / gg) * 2, x); atomicAdd(dst + (g / gg) * 2 + 1, xx); }
      }
    } else {
      // groups of 4 (gshift 2), 2 or 1 channels: reduce over the 32 pixel lanes of each half
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float x = s[i], xx = ss[i];
#pragma unroll
        for (int sft = 16; sft > 0; sft >>= 1) { x += __shfl_xor(x, sft, 64); xx += __shfl_xor(xx, sft, 64); }
        if (pix == 0) {
          const int ch = 8 * (i >> 2) + 4 * h + (i & 3);
          atomicAdd(dst + (ch >> a.gshift) * 2, x);
          atomicAdd(dst + (ch >> a.gshift) * 2 + 1, xx);
        }
      }
    }
  };

  // ---- persistent tile loop
  const int nblk = gridDim.x;
  int t = blockIdx.x, buf = 0;
  if (t < a.ntiles) issue_dma(t, 0);
  for (; t < a.ntiles; t += nblk, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + nblk < a.ntiles) issue_dma(t + nblk, buf ^ 1);
    const int n = t / a.tiles_per_img;
    const int p0 = (t - n * a.tiles_per_img) * TPX;

    f32x16 acc[2];
    // neck: relu(Wn x + bn)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = bnv[i]; acc[1][i] = bnv[i]; }
    stage(acc, wn, xbuf0 + buf * XBYTES, std::integral_constant<int, CPPX>{}, NKN);
    store_tile(acc, bufA, nullptr, nullptr);
    __syncthreads();
    // conv1 (no bias: norm follows, lfd_head.py:97)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    stage(acc, w1, bufA, std::integral_constant<int, HC / 8>{}, NKH);
    if constexpr (PASS == 1) {
      write_stats(acc, t, p0);
    } else {
    float sc[16], sh[16];
    {
      const float* ab = a.ab1 + ((size_t)n * HC + ct * 32 + 4 * h) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[4 * g + j] = ab[(8 * g + j) * 2]; sh[4 * g + j] = ab[(8 * g + j) * 2 + 1]; }
      }
    }
    store_tile(acc, bufB, sc, sh);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    stage(acc, w2, bufB, std::integral_constant<int, HC / 8>{}, NKH);
    if constexpr (PASS == 2) {
      write_stats(acc, t, p0);
    } else {
    {
      const float* ab = a.ab2 + ((size_t)n * HC + ct * 32 + 4 * h) * 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[4 * g + j] = ab[(8 * g + j) * 2]; sh[4 * g + j] = ab[(8 * g + j) * 2 + 1]; }
      }
    }
    store_tile(acc, bufA, sc, sh);   // bufA's neck tile was fully consumed before the last barrier
    __syncthreads();
    if (f_active) {
      f32x16 fa;
      {
        const float* bp = a.bf + fct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          fa[4 * g] = b4.x; fa[4 * g + 1] = b4.y; fa[4 * g + 2] = b4.z; fa[4 * g + 3] = b4.w;
        }
      }
#pragma unroll
      for (int q = 0; q < NKH; ++q) {
        const half8 xf = *reinterpret_cast<const half8*>(bufA + lds_addr<HC / 8>(fpt * 32 + pix, 2 * q + h));
        fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[q], xf, fa, 0, 0, 0);
      }
      const int p = p0 + fpt * 32 + pix;
      if (p < a.HW) {
        const size_t row = (size_t)n * a.P + a.p_off + p;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ch = fct * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
          if (ch < a.split) a.out_cls[row * a.CC + ch] = fa[i];
          else if (ch < a.fcout) a.out_reg[row * 4 + (ch - a.split)] = fa[i] * scale;
        }
      }
    }
    }  // PASS == 3
    }  // PASS >= 2
  }
}

// per (image, group): combine tile partials in fp64, emit per-channel (scale, shift)
__global__ void k_gn_finalize(const float* part, int tiles_per_img, int ngroups, int gsize, int hw,
                              const float* gamma, const float* beta, float eps, float* ab /*[N][128][2]*/) {
  const int n = blockIdx.x;
  const int g = threadIdx.x;
  if (g >= ngroups) return;
  double s = 0.0, ss = 0.0;
  for (int t = 0; t < tiles_per_img; ++t) {
    const float* p = part + ((size_t)(n * tiles_per_img + t) * ngroups + g) * 2;
    s += (double)p[0];
    ss += (double)p[1];
  }
  const double cnt = (double)hw * gsize;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int j = 0; j < gsize; ++j) {
    const int c = g * gsize + j;
    const double sc = (double)gamma[c] * rstd;
    ab[((size_t)n * HC + c) * 2] = (float)sc;
    ab[((size_t)n * HC + c) * 2 + 1] = (float)((double)beta[c] - mean * sc);
  }
}

template <int CIN, int PASS, int FT>
int launch_head(HeadArgs a, hipStream_t st) {
  constexpr int LDS = 2 * TPX * CIN * 2 + 2 * TPX * HC * 2;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head<CIN, PASS, FT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    done = true;
  }
  int blocks = a.ntiles < 512 ? a.ntiles : 512;
  if (blocks < 1) return LFD_OK;
  hipLaunchKernelGGL((k_head<CIN, PASS, FT>), dim3(blocks), dim3(256), LDS, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

template <int CIN>
int dispatch_head(int pass, int ft, const HeadArgs& a, hipStream_t st) {
  if (pass == 1) return launch_head<CIN, 1, 1>(a, st);
  if (pass == 2) return launch_head<CIN, 2, 1>(a, st);
  if (pass == 3) return ft == 2 ? launch_head<CIN, 3, 2>(a, st) : launch_head<CIN, 3, 1>(a, st);
  return LFD_ERR_INVALID_ARGUMENT;
}

}  // namespace

extern "C" {

size_t lfd_head_partial_floats(int32_t n, int32_t hw, int32_t num_groups) {
  const int tiles = (hw + TPX - 1) / TPX;
  return (size_t)n * tiles * num_groups * 2;
}

int lfd_head_level_f16(const lfd_head_desc_t* d, int32_t pass, const void* x, const void* wn_packed,
                       const float* bn, const void* w1_packed, const void* w2_packed, const void* wf_packed,
                       const float* bf, const float* ab1, const float* ab2, float* partial, float* out_cls,
                       float* out_reg, const float* scale, const void* zeros, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !x || !wn_packed || !bn || !w1_packed || !zeros) return LFD_ERR_INVALID_ARGUMENT;
  if (d->head_channels != HC) return LFD_ERR_UNSUPPORTED;
  if (d->cin != 64 && d->cin != 128) return LFD_ERR_UNSUPPORTED;
  const int gsize = HC / d->num_groups;
  if (d->num_groups < 1 || d->num_groups > HC || gsize * d->num_groups != HC || (gsize & (gsize - 1)) || gsize > 32)
    return LFD_ERR_UNSUPPORTED;
  int gshift = 0;
  while ((1 << gshift) < gsize) ++gshift;
  HeadArgs a{};
  a.x = (const _Float16*)x; a.wn = (const half8*)wn_packed; a.bn = bn; a.w1 = (const half8*)w1_packed;
  a.w2 = (const half8*)w2_packed; a.wf = (const half8*)wf_packed; a.bf = bf; a.ab1 = ab1; a.ab2 = ab2;
  a.part = partial; a.out_cls = out_cls; a.out_reg = out_reg; a.scale = scale;
  a.N = d->n; a.HW = d->hw; a.P = d->total_points; a.p_off = d->point_offset; a.CC = d->cls_channels;
  a.split = d->final_split; a.fcout = d->final_cout; a.gshift = gshift;
  a.tiles_per_img = (d->hw + TPX - 1) / TPX;
  a.ntiles = d->n * a.tiles_per_img;
  a.zeros = (const _Float16*)zeros;
  if (pass < 3) {
    if (!partial) return LFD_ERR_INVALID_ARGUMENT;
    if (hipMemsetAsync(partial, 0, sizeof(float) * lfd_head_partial_floats(d->n, d->hw, d->num_groups), st) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
  }
  if (pass >= 2 && (!ab1 || !w2_packed)) return LFD_ERR_INVALID_ARGUMENT;
  if (pass == 3 && (!ab2 || !wf_packed || !bf || !out_cls || !out_reg)) return LFD_ERR_INVALID_ARGUMENT;
  const int ft = (d->final_cout + 31) / 32;
  if (pass == 3 && (ft < 1 || ft > 2)) return LFD_ERR_UNSUPPORTED;
  return d->cin == 64 ? dispatch_head<64>(pass, ft, a, st) : dispatch_head<128>(pass, ft, a, st);
}

int lfd_groupnorm_finalize(const float* partial, int32_t n, int32_t hw, int32_t num_groups, const float* gamma,
                           const float* beta, float eps, float* ab, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!partial || !gamma || !beta || !ab || n < 1 || num_groups < 1 || num_groups > HC) return LFD_ERR_INVALID_ARGUMENT;
  const int tiles = (hw + TPX - 1) / TPX;
  hipLaunchKernelGGL(k_gn_finalize, dim3(n), dim3(128), 0, st, partial, tiles, num_groups, HC / num_groups, hw, gamma,
                     beta, eps, ab);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
