// csrc/precise.hip -- the 'fp32_storage' precision mode of the LFD eval forward (lfd/model/lfd.py:511-542 and everything
// it calls: lfd_resnet.py:354-501, simple_neck.py:67-74, lfd_head.py:164-185), a SHIPPED mode of the product
// (LFD.precision = 'fp32_storage', lfd_amd/engine_p32.py), not a test instrument.
//
// Why it exists: BASELINE.json's north_star asks for "cls/bbox tensors within 1e-3" of the reference's fp32 PyTorch path.
// The fp16-operand pipeline (engine.py) sits at 1.3e-3 .. 2.3e-3 on sigma(cls) / sigma(reg) -- that is the rounding of
// fp16 weights and fp16 stored activations, not kernel error.  This mode removes the rounding and keeps the
// matrix cores:
//   * every activation tensor between launches is fp32 NHWC in HBM;
//   * a convolution is ONE launch: the tile loader splits each fp32 operand EXACTLY into two fp16 parts on its way into
//     LDS, x = x_hi + 2^-11 x_lo (x_hi = fp16(x), x_lo = fp16(2^11 (x - x_hi)); the 2^11 keeps the low part out of the
//     fp16 subnormals), the BatchNorm-folded weights are packed the same way once per plan, and the contraction issues
//     three v_mfma_f32_32x32x16_f16 per k-step into two fp32 accumulator sets
//         main += w_hi x_hi,      corr += w_hi x_lo + w_lo x_hi,      result = main + 2^-11 corr     (x_lo w_lo ~ 2^-22: dropped)
//     -- 3x the MFMA work of the fp16 path, ~22 significant bits per product, fp32 accumulation;
//   * bias, residual add, Scale and ReLU are applied to the fp32 accumulators in the epilogue; fp32 leaves the kernel;
//   * the 3-channel first conv gathers its 3x3 stride-2 patches (27 taps -> one 32-wide k chunk) straight from the frame
//     (NCHW fp32 / NHWC fp16 / NHWC uint8 with the reference's simple_normalize, augmentation_pipeline.py:31-36);
//   * GroupNorm (+ ReLU) of the head towers: statistics in fp64 from fp32 values (fixed-order two-stage sum), normalise in
//     place (lfd_head.py:97-117 builds conv -> GroupNorm -> ReLU).
// Roofs: the convs are MFMA-bound at 3x the algorithmic FLOPs until the layer's AI drops below ~130 FLOP/B (fp32
// activations double the bytes): the stem and the 1x1s are HBM-bound.  Measured numbers: DESIGN 4 / bench.py `precise`.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kLo = 2048.f;           // 2^11
constexpr float kInvLo = 1.f / 2048.f;
constexpr int kPitch = 80;              // LDS bytes per pixel and plane: 32 channels x 2 B + 16 B of padding (bank spread)

struct P32Args {
  const void* in;        // fp32 NHWC [N,H,W,cin]; PATCH: the frames in `fmt`
  float* out;
  const half8* w;        // [slab][chunk][tap][kk][hi|lo][64 lanes] x 8 halfs
  const float* bias;     // [32 * nslab]
  const float* res;      // fp32 NHWC [N,OH,OW,cout] or null
  const float* scale;    // one float (lfd_head.py Scale) or null
  const half8* w2;       // TAIL: chained 1x1 64 -> 64, [2 slabs][2 chunks][1 tap][2 k-steps][hi | lo][64 lanes]
  const float* bias2;    // TAIL: [64]
  int relu2;
  int N, H, W, OH, OW, cin, cout, nslab, relu, fmt;
  long out_img_stride;   // floats between images of `out`
  int out_pix_stride;    // floats between pixels of `out`
  int tiles_x, tiles_y;
};

__device__ __forceinline__ void split8(const float4 a, const float4 b, half8& hi, half8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)((v[j] - (float)h) * kLo);     // v - h is exact in fp32, so is the power-of-two scaling
  }
}

__device__ __forceinline__ float frame_value(const void* in, int fmt, int n, int c, int y, int x, int H, int W) {
  if (fmt == 0) return reinterpret_cast<const float*>(in)[(((size_t)n * 3 + c) * H + y) * W + x];
  const size_t i = (((size_t)n * H + y) * W + x) * 3 + c;
  if (fmt == 1) return (float)reinterpret_cast<const _Float16*>(in)[i];
  const float v = (float)reinterpret_cast<const uint8_t*>(in)[i];
  return (v / 255.f - 0.5f) / 0.5f;
}

// One workgroup = 4 waves = TH x 16 output pixels x 64 output channels; wave w: channel slab (w & 1), pixel rows
// (w >> 1) * 2 NG + [0, 2 NG) as NG MFMA pixel tiles of 2 rows x 16 columns.  Input channels in chunks of 32.
// TAIL: the conv is followed by a 1x1 conv 64 -> 64 (+ bias + ReLU) in the SAME launch (the stem pairs, lfd_resnet.py:356-413:
// conv3x3 + BN + ReLU -> conv1x1 + BN + ReLU): the first conv's fp32 results (bias, ReLU applied) are split into hi / lo planes
// straight into LDS -- wave (slab s) fills channel chunk s of its 64 pixels -- and contracted again; the 64-channel fp32
// intermediate (1.06 GB per 8 x 1080p for the first pair) never reaches HBM.  Needs cout == 64 (one workgroup = both slabs).
template <int KS, int S, int NG, bool PATCH, bool TAIL = false>
__global__ __launch_bounds__(256) void k_p32_conv(const P32Args a) {
  constexpr int TW = 16, TH = 4 * NG;
  constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
  constexpr int KK = KS * KS, PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_hi = smem;
  char* const s_lo = smem + IH * IW * kPitch;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, p = lane & 31, kh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, n = t / a.tiles_y;
  const int slab = blockIdx.y * 2 + (wv & 1);
  const bool active = slab < a.nslab;
  const int prow0 = (wv >> 1) * 2 * NG;
  const int nchunk = PATCH ? 1 : a.cin / 32;

  f32x16 accm[NG], accc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[g][r] = accc[g][r] = 0.f;

  const half8* wslab = a.w + (size_t)(active ? slab : 0) * nchunk * KK * 4 * 64 + lane;

  for (int c = 0; c < nchunk; ++c) {
    if (c) __syncthreads();          // every wave is done with the previous chunk's tile
    // ---- tile loader: fp32 -> (hi, lo) fp16 planes in LDS
    if constexpr (PATCH) {
      for (int i = tid; i < TH * TW * 4; i += 256) {
        const int pix = i >> 2, cg = i & 3;
        const int oy = ty * TH + pix / TW, ox = tx * TW + pix % TW;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = cg * 8 + j;                    // k = (dy * 3 + dx) * 3 + channel; 27..31: zero weights
          const int tap = k / 3, ch = k - tap * 3, dy = tap / 3, dx = tap - dy * 3;
          const int gy = 2 * oy + dy - 1, gx = 2 * ox + dx - 1;
          const bool ok = k < 27 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          const float f = frame_value(a.in, a.fmt, n, ok ? ch : 0, ok ? gy : 0, ok ? gx : 0, a.H, a.W);
          v[j] = ok ? f : 0.f;
        }
        half8 hi, lo;
        split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        *reinterpret_cast<half8*>(s_hi + pix * kPitch + cg * 16) = hi;
        *reinterpret_cast<half8*>(s_lo + pix * kPitch + cg * 16) = lo;
      }
    } else {
      const float* inp = reinterpret_cast<const float*>(a.in);
      for (int i = tid; i < IH * IW * 4; i += 256) {
        const int pix = i >> 2, cg = i & 3;
        const int iy = pix / IW, ix = pix - iy * IW;
        const int gy = ty * TH * S - PAD + iy, gx = tx * TW * S - PAD + ix;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        // unconditional loads from a clamped address + select (a conditional load is a branch, DESIGN lesson 22)
        const float* q = inp + (((size_t)n * a.H + (ok ? gy : 0)) * a.W + (ok ? gx : 0)) * a.cin + c * 32 + cg * 8;
        float4 v0 = *reinterpret_cast<const float4*>(q), v1 = *reinterpret_cast<const float4*>(q + 4);
        if (!ok) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
        half8 hi, lo;
        split8(v0, v1, hi, lo);
        *reinterpret_cast<half8*>(s_hi + pix * kPitch + cg * 16) = hi;
        *reinterpret_cast<half8*>(s_lo + pix * kPitch + cg * 16) = lo;
      }
    }
    __syncthreads();
    if (active) {
      const half8* wc = wslab + (size_t)c * KK * 4 * 64;
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        const int dy = tap / KS, dx = tap % KS;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const half8 wh = wc[((tap * 2 + kk) * 2 + 0) * 64];
          const half8 wl = wc[((tap * 2 + kk) * 2 + 1) * 64];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int orow = prow0 + 2 * g + (p >> 4), ocol = p & 15;
            const int off = ((orow * S + dy) * IW + ocol * S + dx) * kPitch + (kk * 16 + kh * 8) * 2;
            const half8 xh = *reinterpret_cast<const half8*>(s_hi + off);
            const half8 xl = *reinterpret_cast<const half8*>(s_lo + off);
            accm[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accm[g], 0, 0, 0);
            accc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accc[g], 0, 0, 0);
            accc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accc[g], 0, 0, 0);
          }
        }
      }
    }
  }
  if constexpr (TAIL) {
    // ---- first conv's epilogue into LDS planes: channel chunk `slab`, pixel (prow0 + 2g + (p >> 4)) * 16 + (p & 15)
    constexpr int CHB = TH * TW * kPitch;                 // bytes of one plane of one 32-channel chunk
    __syncthreads();                                      // every wave is done reading the input tile
    char* const t_hi = smem;                              // [2 chunks][TH*TW px][kPitch]
    char* const t_lo = smem + 2 * CHB;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int pix = (prow0 + 2 * g + (p >> 4)) * TW + (p & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        union { _Float16 h[4]; uint2 u; } hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = accm[g][4 * q + e] + accc[g][4 * q + e] * kInvLo + a.bias[slab * 32 + 8 * q + 4 * kh + e];
          if (a.relu) v = fmaxf(v, 0.f);
          const _Float16 hh = (_Float16)v;
          hi.h[e] = hh;
          lo.h[e] = (_Float16)((v - (float)hh) * kLo);
        }
        const int off = slab * CHB + pix * kPitch + (8 * q + 4 * kh) * 2;
        *reinterpret_cast<uint2*>(t_hi + off) = hi.u;
        *reinterpret_cast<uint2*>(t_lo + off) = lo.u;
      }
    }
    __syncthreads();
    // ---- chained 1x1: 2 chunks x 2 k-steps, A = w2 fragments of this wave's output slab
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) accm[g][r] = accc[g][r] = 0.f;
    const half8* w2s = a.w2 + (size_t)slab * 2 * 4 * 64 + lane;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const half8 wh = w2s[((c2 * 2 + kk) * 2 + 0) * 64];
        const half8 wl = w2s[((c2 * 2 + kk) * 2 + 1) * 64];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int pix = (prow0 + 2 * g + (p >> 4)) * TW + (p & 15);
          const int off = c2 * CHB + pix * kPitch + (kk * 16 + kh * 8) * 2;
          const half8 xh = *reinterpret_cast<const half8*>(t_hi + off);
          const half8 xl = *reinterpret_cast<const half8*>(t_lo + off);
          accm[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accm[g], 0, 0, 0);
          accc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accc[g], 0, 0, 0);
          accc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accc[g], 0, 0, 0);
        }
      }
  }
  if (!active) return;
  // ---- epilogue on the fp32 accumulators: register 4q + e of lane (kh, p) = channel 32 slab + 8q + 4kh + e of pixel p
  const float* const ebias = TAIL ? a.bias2 : a.bias;
  const int erelu = TAIL ? a.relu2 : a.relu;
  const float sc = a.scale ? *a.scale : 1.f;
  const bool vec = (a.cout & 3) == 0 && (a.out_pix_stride & 3) == 0 && (a.out_img_stride & 3) == 0;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int oy = ty * TH + prow0 + 2 * g + (p >> 4), ox = tx * TW + (p & 15);
    if (oy >= a.OH || ox >= a.OW) continue;
    const size_t pixi = (size_t)oy * a.OW + ox;
    float* o = a.out + (size_t)n * a.out_img_stride + pixi * a.out_pix_stride;
    const float* r = a.res ? a.res + ((size_t)n * a.OH * a.OW + pixi) * a.cout : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = slab * 32 + 8 * q + 4 * kh;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = accm[g][4 * q + e] + accc[g][4 * q + e] * kInvLo + ebias[c0 + e];
      if (vec) {
        if (c0 >= a.cout) continue;
        if (r) {
          const float4 rv = *reinterpret_cast<const float4*>(r + c0);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] *= sc;
          if (erelu) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<float4*>(o + c0) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c0 + e >= a.cout) continue;
          float x = v[e];
          if (r) x += r[c0 + e];
          x *= sc;
          if (erelu) x = fmaxf(x, 0.f);
          o[c0 + e] = x;
        }
      }
    }
  }
}

template <int KS, int S, int NG, bool PATCH, bool TAIL = false>
int launch_p32(P32Args a, hipStream_t st) {
  constexpr int TH = 4 * NG, TW = 16;
  constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS;
  constexpr int lds_in = 2 * IH * IW * kPitch, lds_tail = TAIL ? 4 * TH * TW * kPitch : 0;
  constexpr int lds = lds_in > lds_tail ? lds_in : lds_tail;
  a.tiles_x = (a.OW + TW - 1) / TW;
  a.tiles_y = (a.OH + TH - 1) / TH;
  const long tiles = (long)a.tiles_x * a.tiles_y * a.N;
  if (tiles > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  auto kern = k_p32_conv<KS, S, NG, PATCH, TAIL>;
  if (lds > 64 * 1024) {
    static unsigned long long set_mask = 0;   // (idempotent; races only repeat the same call)
    const int set_dev = lfd_device_ordinal();
    if (LFD_ONCE_PER_DEVICE(set_mask, set_dev)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return LFD_ERR_LAUNCH_FAILED;
      LFD_DONE_ON_DEVICE(set_mask, set_dev);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)((a.nslab + 1) / 2)), dim3(256), lds, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// ---------------------------------------------------------------------------------------------- GroupNorm (+ ReLU)
constexpr int kGnChunks = 64;

// partial[(n * kGnChunks + chunk) * groups + g] = {sum, sum of squares} (fp64) over the chunk's pixels
__global__ __launch_bounds__(256) void k_p32_gn_stats(const float* x, long hw, int c, int groups, double* partial) {
  __shared__ double red[256][2];
  const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int c4 = c >> 2, rows = 256 / c4;                  // c4 divides 256 (checked by the launcher)
  const int cq = tid % c4, pr = tid / c4;
  const long per = (hw + kGnChunks - 1) / kGnChunks;
  const long p0 = chunk * per, p1 = p0 + per < hw ? p0 + per : hw;
  double s = 0., ss = 0.;
  const float* xi = x + (size_t)n * hw * c;
  for (long q = p0 + pr; q < p1; q += rows) {
    const float4 v = *reinterpret_cast<const float4*>(xi + q * c + cq * 4);
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  red[tid][0] = s;
  red[tid][1] = ss;
  __syncthreads();
  if (tid < groups) {
    const int gs4 = c4 / groups;                           // channel quads per group
    double a = 0., b = 0.;
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < gs4; ++k) {
        a += red[r * c4 + tid * gs4 + k][0];
        b += red[r * c4 + tid * gs4 + k][1];
      }
    double* o = partial + (((size_t)n * kGnChunks + chunk) * groups + tid) * 2;
    o[0] = a;
    o[1] = b;
  }
}

__global__ __launch_bounds__(256) void k_p32_gn_apply(float* x, long hw, int c, int groups, const double* partial,
                                                      const float* gamma, const float* beta, float eps, int relu) {
  __shared__ float s_mean[64], s_rstd[64];
  const int n = blockIdx.y, tid = threadIdx.x;
  if (tid < groups) {
    double a = 0., b = 0.;
    for (int k = 0; k < kGnChunks; ++k) {                  // fixed order: deterministic
      const double* q = partial + (((size_t)n * kGnChunks + k) * groups + tid) * 2;
      a += q[0];
      b += q[1];
    }
    const double cnt = (double)hw * (c / groups);
    const double m = a / cnt;
    double var = b / cnt - m * m;
    var = var > 0. ? var : 0.;
    s_mean[tid] = (float)m;
    s_rstd[tid] = (float)(1. / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int c4 = c >> 2, gs = c / groups;
  const long total = hw * c4;
  float* xi = x + (size_t)n * hw * c;
  for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) {
    const int cq = (int)(i % c4), ch = cq * 4;
    float4 v = *reinterpret_cast<float4*>(xi + i * 4);
    float* e = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = (ch + j) / gs;
      float y = (e[j] - s_mean[g]) * s_rstd[g] * gamma[ch + j] + beta[ch + j];
      e[j] = relu ? fmaxf(y, 0.f) : y;
    }
    *reinterpret_cast<float4*>(xi + i * 4) = v;
  }
}

}  // namespace

extern "C" {

size_t lfd_p32_conv_packed_weight_halfs(int32_t cin, int32_t cout, int32_t ks) {
  const size_t nslab = (size_t)(cout + 31) / 32, nchunk = (size_t)(cin + 31) / 32;
  return nslab * nchunk * ks * ks * 2 * 2 * 64 * 8;
}

int lfd_p32_conv2d_nhwc_f32(const lfd_p32_conv_desc_t* d, const void* in, float* out, const void* w_packed,
                            const float* bias, const float* residual, const float* scale, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !in || !out || !w_packed || !bias) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->h < 1 || d->w < 1 || d->cout < 1) return LFD_ERR_INVALID_ARGUMENT;
  const bool patch = d->in_format >= 0;
  if (patch) {
    if (d->in_format > 2 || d->cin != 3 || d->ks != 3 || d->stride != 2) return LFD_ERR_UNSUPPORTED;
  } else {
    if (d->cin < 32 || d->cin % 32) return LFD_ERR_UNSUPPORTED;
    if ((d->ks != 1 && d->ks != 3) || (d->stride != 1 && d->stride != 2)) return LFD_ERR_UNSUPPORTED;
  }
  P32Args a{};
  a.in = in; a.out = out; a.w = (const half8*)w_packed; a.bias = bias; a.res = residual; a.scale = scale;
  a.N = d->n; a.H = d->h; a.W = d->w; a.cin = d->cin; a.cout = d->cout; a.nslab = (d->cout + 31) / 32;
  a.relu = d->relu; a.fmt = d->in_format;
  const int pad = d->ks / 2;
  a.OH = (d->h + 2 * pad - d->ks) / d->stride + 1;
  a.OW = (d->w + 2 * pad - d->ks) / d->stride + 1;
  a.out_pix_stride = d->out_pixel_stride > 0 ? d->out_pixel_stride : d->cout;
  a.out_img_stride = d->out_image_stride > 0 ? d->out_image_stride : (long)a.OH * a.OW * a.out_pix_stride;
  if (patch) return launch_p32<1, 1, 2, true>(a, st);
  switch (d->ks * 10 + d->stride) {
    case 31: return launch_p32<3, 1, 2, false>(a, st);
    case 32: return launch_p32<3, 2, 1, false>(a, st);
    case 11: return launch_p32<1, 1, 2, false>(a, st);
    case 12: return launch_p32<1, 2, 1, false>(a, st);
    default: return LFD_ERR_UNSUPPORTED;
  }
}

int lfd_p32_conv2d_tail_nhwc_f32(const lfd_p32_conv_desc_t* d, const void* in, float* out, const void* w_packed,
                                 const float* bias, const void* tail_w_packed, const float* tail_bias, int32_t tail_relu,
                                 lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !in || !out || !w_packed || !bias || !tail_w_packed || !tail_bias) return LFD_ERR_INVALID_ARGUMENT;
  if (d->n < 1 || d->h < 1 || d->w < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (d->cout != 64) return LFD_ERR_UNSUPPORTED;          // both 32-channel slabs in one workgroup; the chained 1x1 is 64 -> 64
  const bool patch = d->in_format >= 0;
  if (patch) {
    if (d->in_format > 2 || d->cin != 3 || d->ks != 3 || d->stride != 2) return LFD_ERR_UNSUPPORTED;
  } else if (d->cin < 32 || d->cin % 32 || d->ks != 3 || d->stride != 2) {
    return LFD_ERR_UNSUPPORTED;                           // the stem pairs: 3x3 stride 2 -> 1x1
  }
  P32Args a{};
  a.in = in; a.out = out; a.w = (const half8*)w_packed; a.bias = bias; a.w2 = (const half8*)tail_w_packed; a.bias2 = tail_bias;
  a.relu2 = tail_relu;
  a.N = d->n; a.H = d->h; a.W = d->w; a.cin = d->cin; a.cout = 64; a.nslab = 2; a.relu = d->relu; a.fmt = d->in_format;
  a.OH = (d->h + 2 - 3) / 2 + 1;
  a.OW = (d->w + 2 - 3) / 2 + 1;
  a.out_pix_stride = d->out_pixel_stride > 0 ? d->out_pixel_stride : 64;
  a.out_img_stride = d->out_image_stride > 0 ? d->out_image_stride : (long)a.OH * a.OW * a.out_pix_stride;
  if (patch) return launch_p32<1, 1, 2, true, true>(a, st);
  return launch_p32<3, 2, 1, false, true>(a, st);
}

size_t lfd_p32_groupnorm_workspace_bytes(int32_t n, int32_t groups) {
  return (size_t)(n > 0 ? n : 0) * kGnChunks * (groups > 0 ? groups : 0) * 2 * sizeof(double);
}

int lfd_p32_groupnorm_relu_f32(float* x, int32_t n, int64_t hw, int32_t c, int32_t groups, const float* gamma,
                               const float* beta, float eps, int32_t relu, void* workspace, size_t workspace_bytes,
                               lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!x || !gamma || !beta || !workspace || n < 1 || hw < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (c < 4 || c % 4 || 256 % (c / 4) || groups < 1 || groups > 64 || c % groups || (c / groups) % 4)
    return LFD_ERR_UNSUPPORTED;
  if (workspace_bytes < lfd_p32_groupnorm_workspace_bytes(n, groups)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(k_p32_gn_stats, dim3(kGnChunks, n), dim3(256), 0, st, x, (long)hw, c, groups, partial);
  LFD_CHECK_LAUNCH();
  long blocks = (hw * (c / 4) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_p32_gn_apply, dim3((unsigned)blocks, n), dim3(256), 0, st, x, (long)hw, c, groups, partial, gamma,
                     beta, eps, relu);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
