// csrc/stem_fused.hip -- the whole 'faster' stem in ONE kernel:
//   conv3x3 s2 (3->C) + BN + ReLU -> conv1x1 (C->C) + BN + ReLU -> conv3x3 s2 (C->C) + BN + ReLU
//   -> conv1x1 (C->C) + BN + ReLU                 (reference lfd/model/backbone/lfd_resnet.py:376-413)
//
// Why: the output of the first pair is the largest activation of the network (540x960xC per 1080p
// image, 66 MB fp16 at C = 64).  As two kernels it is written once and read once -- 1.06 GB per
// batch of 8 frames, ~40 % of the bytes the whole forward moves.  Here it never leaves the CU: per
// 4x16 output tile the 9x33-pixel stride-2 halo region of that intermediate is produced straight into
// LDS (in exactly the swizzled, column-de-interleaved layout the 3x3 s2 contraction reads) from a
// 19x67-pixel copy of the raw frame, then consumed by the weights-stationary 3x3 (144 VGPRs of
// filter per wave) and its chained 1x1.  Cost: the halo is recomputed (x1.16 of 6 GFLOP/img).
//
// Phase A (per 32-pixel group, one wave computes ALL C channels so no cross-wave exchange is needed):
//   im2col(K=27->32) -> 2 MFMA/cout-tile -> ReLU -> fp16 -> wave-private LDS -> 1x1 MFMAs -> ReLU
//   -> fp16 -> mid1[slot(iy,ix)] (zero outside the intermediate image = the 3x3's zero padding)
// Phase B: identical to k_conv<C,3,2,NCT,true,true> (conv.hip) with its input tile = mid1.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

enum { IN_NCHW_F32 = 0, IN_NHWC_F16 = 1, IN_NHWC_U8 = 2 };

struct FusedArgs {
  const void* in;
  _Float16* out;         // [N, H2, W2, C]
  const half8* w1;       // [NCT][2][64]        stem conv 1 (3x3 s2, K=27 padded)
  const float* b1;
  const half8* w2;       // [NCT][C/16][64]     stem conv 2 (1x1)
  const float* b2;
  const half8* w3;       // [NCT][9*C/16][64]   stem conv 3 (3x3 s2)
  const float* b3;
  const half8* w4;       // [NCT][C/16][64]     stem conv 4 (1x1)
  const float* b4;
  int N, H, W;           // raw frame
  int H1, W1;            // after pair 1
  int H2, W2;            // after pair 2 (output)
  int tiles_x, tiles_y, ntiles;
};

template <int FMT>
__device__ __forceinline__ _Float16 load_px(const void* in, int n, int H, int W, int gy, int gx, int c) {
  if (FMT == IN_NCHW_F32) {
    return (_Float16) reinterpret_cast<const float*>(in)[(((size_t)n * 3 + c) * H + gy) * W + gx];
  } else if (FMT == IN_NHWC_F16) {
    return reinterpret_cast<const _Float16*>(in)[(((size_t)n * H + gy) * W + gx) * 3 + c];
  } else {
    const float v = (float)reinterpret_cast<const uint8_t*>(in)[(((size_t)n * H + gy) * W + gx) * 3 + c];
    return (_Float16)((v / 255.f - 0.5f) / 0.5f);
  }
}

template <int NCT>
struct FCfg {
  static constexpr int C = NCT * 32;
  // ---- phase B geometry == Cfg<C,3,2,NCT> of conv.hip
  static constexpr int TW = 16, RPT = 2, PG = 4 / NCT, TH = PG * RPT;
  static constexpr int IH = 2 * TH + 1, IW = 2 * TW + 1, IWh = (IW + 1) / 2, IWs = 2 * IWh;
  static constexpr int CPP = C / 8, PIXB = C * 2, PPR = 16 / CPP;
  static constexpr int NSLOT = IH * IWs;
  static constexpr int NQ = C / 16, NK = 9 * NQ, NK2 = C / 16;
  // ---- phase A geometry
  static constexpr int R = IH * IW;                 // intermediate pixels needed by one tile
  static constexpr int NG = (R + 31) / 32;          // 32-pixel MFMA groups
  static constexpr int RH = 2 * IH + 1, RW = 2 * IW + 1;
  static constexpr int RS = ((RW * 3 + 1) / 2) * 2; // halfs per raw LDS row
  // ---- LDS map (bytes)
  static constexpr int RAW_BYTES = ((RH * RS * 2 + 16 + 255) / 256) * 256;
  static constexpr int MID0_BYTES = 4 * 32 * PIXB;  // wave-private [32 px][C] tiles; reused as the tail's mid tile
  static constexpr int MID1_BYTES = ((NSLOT * PIXB + 255) / 256) * 256;
  static constexpr int WA_BYTES = (NCT * 2 + NCT * NK2) * 1024 + 4 * C * 4;   // phase-A filters + the 4 bias vectors
  static constexpr int OFF_MID0 = RAW_BYTES, OFF_MID1 = OFF_MID0 + MID0_BYTES, OFF_WA = OFF_MID1 + MID1_BYTES;
  static constexpr int LDS_BYTES = OFF_WA + WA_BYTES;
  static_assert(PG * 32 * PIXB <= MID0_BYTES, "tail mid tile aliases mid0");
  static_assert(PG * 32 * PIXB <= MID1_BYTES, "output staging aliases mid1");
};

template <int NCT, int FMT>
__global__ __launch_bounds__(256, 2) void k_stem_fused(FusedArgs a) {
  using F = FCfg<NCT>;
  constexpr int C = F::C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  _Float16* s_raw = reinterpret_cast<_Float16*>(smem);
  char* s_mid0 = smem + F::OFF_MID0;
  char* s_mid1 = smem + F::OFF_MID1;
  half8* s_w1 = reinterpret_cast<half8*>(smem + F::OFF_WA);           // [NCT][2][64]
  half8* s_w2 = s_w1 + NCT * 2 * 64;                                   // [NCT][NK2][64]
  float* s_b = reinterpret_cast<float*>(s_w2 + NCT * F::NK2 * 64);      // [4][C] biases b1..b4

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ct = wave % NCT, pg = wave / NCT;
  const int h = lane >> 5, pix = lane & 31;
  const int oyl = pix / F::TW, oxl = pix % F::TW;

  // ---- stationary weights: 3x3 s2 (C->C) slab + chained 1x1 in VGPRs, phase-A filters in LDS
  half8 w3r[F::NK], w4r[F::NK2];
#pragma unroll
  for (int k = 0; k < F::NK; ++k) w3r[k] = a.w3[((size_t)ct * F::NK + k) * 64 + lane];
#pragma unroll
  for (int k = 0; k < F::NK2; ++k) w4r[k] = a.w4[((size_t)ct * F::NK2 + k) * 64 + lane];
  for (int i = threadIdx.x; i < NCT * 2 * 64; i += 256) s_w1[i] = a.w1[i];
  for (int i = threadIdx.x; i < NCT * F::NK2 * 64; i += 256) s_w2[i] = a.w2[i];
  if (threadIdx.x < C) {
    s_b[threadIdx.x] = a.b1[threadIdx.x]; s_b[C + threadIdx.x] = a.b2[threadIdx.x];
    s_b[2 * C + threadIdx.x] = a.b3[threadIdx.x]; s_b[3 * C + threadIdx.x] = a.b4[threadIdx.x];
  }

  // ---- phase-B LDS read offsets (see conv.hip)
  int xoff[3][F::NQ];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = oxl * 2 + s;
    const int rem = (ix & 1) * F::IWh + (ix >> 1);
    const int f = (rem / F::PPR) % F::CPP;
    const int rowbase = ((pg * F::RPT + oyl) * 2) * F::IWs + rem;
#pragma unroll
    for (int q = 0; q < F::NQ; ++q) xoff[s][q] = rowbase * F::PIXB + (((2 * q + h) ^ f) * 16);
  }

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int tiles_per_img = a.tiles_x * a.tiles_y;

  for (int t = t_begin + bix; t < t_end; t += t_step) {
    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const int gy1_0 = ty0 * F::TH * 2 - 1, gx1_0 = tx0 * F::TW * 2 - 1;   // intermediate-image origin of the region
    const int gyr0 = 2 * gy1_0 - 1, gxr0 = 2 * gx1_0 - 1;                 // raw-frame origin

    __syncthreads();   // previous tile fully done with every LDS region (also covers s_w1/s_w2 fill)
    // ---- stage the raw frame tile (zero padded) into LDS.  All global loads of a thread are issued
    //      back to back (fixed trip count, predicated) and only then written: one memory round trip
    //      per tile instead of one per element.
    {
      constexpr int NE = F::RH * F::RW * 3;
      constexpr int NIT = (NE + 255) / 256;
      _Float16 rv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + threadIdx.x;
        const int iy = i / (F::RW * 3), e = i - iy * (F::RW * 3);
        const int ix = e / 3, c = e - ix * 3;
        const int gy = gyr0 + iy, gx = gxr0 + ix;
        // unconditional load from a clamped address + select: a branch around the load would make the
        // compiler wait for each element separately (15 serialized memory round trips per tile)
        const bool ok = i < NE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int cy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
        const _Float16 v = load_px<FMT>(a.in, n, a.H, a.W, cy, cx, c);
        rv[it] = ok ? v : (_Float16)0.f;
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + threadIdx.x;
        const int iy = i / (F::RW * 3), e = i - iy * (F::RW * 3);
        if (i < NE) s_raw[iy * F::RS + e] = rv[it];
      }
    }
    __syncthreads();

    // ---- phase A: intermediate region -> mid1
    char* m0 = s_mid0 + wave * (32 * F::PIXB);
    for (int grp = wave; grp < F::NG; grp += 4) {
      const int j = grp * 32 + pix;
      const bool inreg = j < F::R;
      const int iy = inreg ? j / F::IW : 0, ix = inreg ? j - (j / F::IW) * F::IW : 0;
      const _Float16* base = s_raw + (2 * iy) * F::RS + 6 * ix;
      union { half8 v; uint32_t u[4]; } f0, f1;
      {
        const uint32_t* p0 = reinterpret_cast<const uint32_t*>(base + h * F::RS);
        f0.u[0] = p0[0]; f0.u[1] = p0[1]; f0.u[2] = p0[2]; f0.u[3] = p0[3];
      }
      if (h == 0) {
        const uint32_t* p2 = reinterpret_cast<const uint32_t*>(base + 2 * F::RS);
        f1.u[0] = p2[0]; f1.u[1] = p2[1]; f1.u[2] = p2[2]; f1.u[3] = p2[3];
      } else {
        const uint32_t e0 = reinterpret_cast<const uint32_t*>(base + 8)[0] & 0xffffu;
        const uint32_t e1 = reinterpret_cast<const uint32_t*>(base + F::RS + 8)[0] & 0xffffu;
        const uint32_t e2 = reinterpret_cast<const uint32_t*>(base + 2 * F::RS + 8)[0] & 0xffffu;
        f1.u[0] = e0 | (e1 << 16); f1.u[1] = e2; f1.u[2] = 0u; f1.u[3] = 0u;
      }
      const int fm = (pix / F::PPR) % F::CPP;   // swizzle key of the wave-private [32][C] tile
#pragma unroll
      for (int c1 = 0; c1 < NCT; ++c1) {
        f32x16 acc;
        const float* bp = s_b + c1 * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w1[(c1 * 2 + 0) * 64 + lane], f0.v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w1[(c1 * 2 + 1) * 64 + lane], f1.v, acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4 v;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) v[jj] = (_Float16)fmaxf(acc[4 * g + jj], 0.f);
          *reinterpret_cast<half4*>(m0 + pix * F::PIXB + (((c1 * 4 + g) ^ fm) * 16) + 8 * h) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();   // wave-private tile: LDS ops of one wave complete in issue order
      // intermediate pixel inside its image?  (outside == zero padding of the following 3x3)
      const int gy1 = gy1_0 + iy, gx1 = gx1_0 + ix;
      const bool vis = inreg && gy1 >= 0 && gy1 < a.H1 && gx1 >= 0 && gx1 < a.W1;
      const int rem = (ix & 1) * F::IWh + (ix >> 1);
      const int f1k = (rem / F::PPR) % F::CPP;
      char* dstpix = s_mid1 + (iy * F::IWs + rem) * F::PIXB + 8 * h;
#pragma unroll
      for (int c2 = 0; c2 < NCT; ++c2) {
        f32x16 acc;
        const float* bp = s_b + C + c2 * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        }
#pragma unroll
        for (int q = 0; q < F::NK2; ++q) {
          const half8 xf = *reinterpret_cast<const half8*>(m0 + pix * F::PIXB + (((2 * q + h) ^ fm) * 16));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w2[(c2 * F::NK2 + q) * 64 + lane], xf, acc, 0, 0, 0);
        }
        if (inreg) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4 v;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = vis ? (_Float16)fmaxf(acc[4 * g + jj], 0.f) : (_Float16)0.f;
            *reinterpret_cast<half4*>(dstpix + (((c2 * 4 + g) ^ f1k) * 16)) = v;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();   // next group's writes to m0 stay behind this group's reads
    }
    __syncthreads();

    // ---- phase B: 3x3 s2 over mid1 (weights stationary), explicit 3-deep LDS prefetch ring
    f32x16 acc;
    {
      const float* bp = s_b + 2 * C + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
        acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
      }
    }
    auto xfrag = [&](int k) {
      const int r = k / (3 * F::NQ), s = (k / F::NQ) % 3, q = k % F::NQ;
      return *reinterpret_cast<const half8*>(s_mid1 + xoff[s][q] + r * F::IWs * F::PIXB);
    };
    {
      constexpr int PD = 3;
      half8 xq[PD + 1];
#pragma unroll
      for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
#pragma unroll
      for (int k = 0; k < F::NK; ++k) {
        if (k + PD < F::NK) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3r[k], xq[k % (PD + 1)], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // chained 1x1: ReLU -> fp16 -> mid tile (aliases mid0) -> MFMA
    {
      const int pb = pg * 32 + pix;
      const int fm = (pb / F::PPR) % F::CPP;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 v;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[jj] = (_Float16)fmaxf(acc[4 * g + jj], 0.f);
        *reinterpret_cast<half4*>(s_mid0 + pb * F::PIXB + (((ct * 4 + g) ^ fm) * 16) + 8 * h) = v;
      }
      __syncthreads();
      const float* bp = s_b + 3 * C + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
        acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
      }
#pragma unroll
      for (int q = 0; q < F::NK2; ++q) {
        const half8 xf = *reinterpret_cast<const half8*>(s_mid0 + pb * F::PIXB + (((2 * q + h) ^ fm) * 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4r[q], xf, acc, 0, 0, 0);
      }
      // ---- epilogue: ReLU -> fp16 -> staging (mid1 is dead: every wave passed the barrier above
      //      only after finishing its phase-B reads) -> full-line stores
      char* sout = s_mid1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 v;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[jj] = (_Float16)fmaxf(acc[4 * g + jj], 0.f);
        *reinterpret_cast<half4*>(sout + pb * F::PIXB + (((ct * 4 + g) ^ fm) * 16) + 8 * h) = v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F::PG * 32 * F::CPP; i += 256) {
      const int pb = i / F::CPP, c = i - pb * F::CPP;
      const int oy = ty0 * F::TH + (pb >> 5) * F::RPT + (pb & 31) / F::TW;
      const int ox = tx0 * F::TW + (pb & 31) % F::TW;
      if (oy < a.H2 && ox < a.W2) {
        const int fm = (pb / F::PPR) % F::CPP;
        const uint4 v = *reinterpret_cast<const uint4*>(s_mid1 + pb * F::PIXB + ((c ^ fm) * 16));
        *reinterpret_cast<uint4*>(a.out + (((size_t)n * a.H2 + oy) * a.W2 + ox) * C + c * 8) = v;
      }
    }
  }
}

template <int NCT, int FMT>
int launch_fused(FusedArgs a, hipStream_t st) {
  using F = FCfg<NCT>;
  a.tiles_x = (a.W2 + F::TW - 1) / F::TW;
  a.tiles_y = (a.H2 + F::TH - 1) / F::TH;
  const long long nt = (long long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffLL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fused<NCT, FMT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    done = true;
  }
  int blocks = a.ntiles < 512 ? a.ntiles : 512;
  if (blocks < 1) return LFD_OK;
  hipLaunchKernelGGL((k_stem_fused<NCT, FMT>), dim3(blocks), dim3(256), F::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

template <int NCT>
int dispatch_fmt(int fmt, const FusedArgs& a, hipStream_t st) {
  switch (fmt) {
    case IN_NCHW_F32: return launch_fused<NCT, IN_NCHW_F32>(a, st);
    case IN_NHWC_F16: return launch_fused<NCT, IN_NHWC_F16>(a, st);
    case IN_NHWC_U8: return launch_fused<NCT, IN_NHWC_U8>(a, st);
    default: return LFD_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace

extern "C" {

int lfd_stem_faster_fused_f16(const void* in, int32_t in_format, int32_t n, int32_t h, int32_t w, int32_t channels,
                              const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                              const void* w3_packed, const float* b3, const void* w4_packed, const float* b4,
                              void* out, lfd_stream_t stream) {

  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || !w1_packed || !b1 || !w2_packed || !b2 || !w3_packed || !b3 || !w4_packed || !b4)
    return LFD_ERR_INVALID_ARGUMENT;
  if (n < 1 || h < 1 || w < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (channels != 32 && channels != 64) return LFD_ERR_UNSUPPORTED;
  FusedArgs a{};
  a.in = in; a.out = (_Float16*)out;
  a.w1 = (const half8*)w1_packed; a.b1 = b1; a.w2 = (const half8*)w2_packed; a.b2 = b2;
  a.w3 = (const half8*)w3_packed; a.b3 = b3; a.w4 = (const half8*)w4_packed; a.b4 = b4;
  a.N = n; a.H = h; a.W = w;
  a.H1 = (h - 1) / 2 + 1; a.W1 = (w - 1) / 2 + 1;
  a.H2 = (a.H1 - 1) / 2 + 1; a.W2 = (a.W1 - 1) / 2 + 1;
  return channels == 64 ? dispatch_fmt<2>(in_format, a, st) : dispatch_fmt<1>(in_format, a, st);
}

}  // extern "C"
