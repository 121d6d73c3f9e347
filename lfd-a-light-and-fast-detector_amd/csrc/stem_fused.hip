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
#include <type_traits>

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
  int stagger;           // 1: de-phase the workgroups at start (k_stem2x)
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

// =====================================================================================================
// k_stem2x: second-generation fused stem for C = 64, NHWC fp16 frames ("one wave per SIMD" design).
//
//   * ONE 256-thread workgroup per CU, 512 registers per wave: every wave keeps ALL four filters resident
//     (3x3 s2 64->64: 72 fragments = 288 registers; the three small ones: 20 fragments) and produces whole
//     128-byte pixel lines, so nothing but the intermediate tile itself is exchanged between waves.
//   * 4 x 32 output tile; its 9 x 65-pixel stride-2 halo of the 540x960x64 intermediate is computed into LDS
//     (76 KB, column-de-interleaved + XOR-swizzled: exactly what the 3x3 s2 contraction reads) from a 19 x 131
//     pixel copy of the raw frame.  Halo recompute: x1.14 on the (small) first pair.
//   * chained 1x1s without an LDS round trip: the accumulator layout of one MFMA (lane = pixel, register =
//     output row) IS a valid B operand of the next one if the next filter's K order is permuted to match --
//     registers 8(q&1)..8(q&1)+7 of channel tile q>>1 are k-step q.  The permutation (and a row permutation
//     that makes every lane own 8 consecutive output channels = one 16-byte chunk) is applied when the
//     filters are loaded, from the standard packed layout.
//   * raw frame rows arrive as ALIGNED 16-byte chunks (never cross a page, so the over-read at a row's ends
//     is always safe), prefetched one tile ahead into registers; the row's sub-chunk misalignment stays in the
//     LDS copy and is folded into the im2col read address (unaligned LDS reads are legal on gfx950).
// =====================================================================================================
struct X2 {
  static constexpr int TH = 4, TW = 32;
  static constexpr int IH = 9, IW = 65, IWh = 33, IWs = 66;   // intermediate region (+ de-interleaved row pitch)
  // (585 pixels = 18 row-half groups of 32 + the 65th column)
  static constexpr int RH = 19, RW = 131;                      // raw frame region
  static constexpr int NCH = 51, RSB = NCH * 16 + 16;          // aligned 16-B chunks per raw row / bytes per raw LDS row (+ shift slack)
  static constexpr int NCHUNK = RH * NCH;                      // 969
  static constexpr int RAW_BYTES = ((RH * RSB + 255) / 256) * 256;
  static constexpr int MID_BYTES = IH * IWs * 128;
  static constexpr int STG_BYTES = 4 * 32 * 128;
  static constexpr int OFF_MID = RAW_BYTES, OFF_STG = OFF_MID + MID_BYTES, OFF_B = OFF_STG + STG_BYTES;
  static constexpr int OFF_W4 = OFF_B + 4 * 64 * 4;            // conv4 filter (permuted fragments), 8 KB
  static constexpr int OFF_DUMMY = OFF_W4 + 8 * 1024;          // write target of lanes / groups past the region's end: [32 px][128 B]
  // per-thread constant tables (with one wave per SIMD every dependent VALU instruction costs ~8 cycles and nothing
  // else is there to hide it: a 16-byte LDS read replaces the 20-30 instruction chains that rebuild these values)
  static constexpr int OFF_TCH = OFF_DUMMY + 32 * 128;         // [4 chunks][256 threads] int4 {row, first half, byte offset, drift}
  static constexpr int OFF_TXO = OFF_TCH + 4 * 256 * 16;       // [3 taps][256 threads] int4: phase-B read offsets
  static constexpr int OFF_WB4 = OFF_TXO + 3 * 256 * 16;       // conv3 / conv4 bias k-step fragments [4][64]
  static constexpr int OFF_W3T = OFF_WB4 + 4 * 64 * 16;        // conv3 filter, k-steps 34, 35 of both cout tiles
  static constexpr int LDS_BYTES = OFF_W3T + 4 * 64 * 16;
};

#ifdef LFD_X2_TIMING
__device__ unsigned long long g_x2_dbg[8 * 16];
#define X2_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && itn < 8) g_x2_dbg[itn * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define X2_T(i)
#endif
typedef uint32_t __attribute__((aligned(2))) u32_a2;
typedef uint16_t __attribute__((aligned(2))) u16_a2;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(2))) u32x4_a2;

// two fp32 -> packed fp16 (round to nearest even) with ReLU: v_cvt_pk_f16_f32 + v_pk_max_f16.
// (Compiler-visible operations, not inline asm: the hazard recogniser has to see the reads of the MFMA
// result registers to insert the wait states they need.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float x, float y) {
  f32x2 f; f[0] = x; f[1] = y;
  union { half2v v; uint32_t u; } r; r.v = __builtin_convertvector(f, half2v);
  return r.u;
}
__device__ __forceinline__ uint32_t relu_pk(uint32_t h) {
  union { half2v v; uint32_t u; } r; r.u = h;
  half2v z; z[0] = (_Float16)0.f; z[1] = (_Float16)0.f;
  r.v = __builtin_elementwise_max(r.v, z);
  return r.u;
}
// registers [8*half .. 8*half+7] of an accumulator -> 8 fp16 (ReLU) = one B operand / one 16-byte chunk.
// All conversions first, then all maxes: with a single wave per SIMD a VALU instruction that depends on the one
// right before it stalls ~4 extra cycles (tools/ub/valu.hip: 4.4 vs 8.0 cycles), cvt/max/cvt/max would pay that
// on every pair.
__device__ __forceinline__ half8 relu8(const f32x16& acc, int half) {
  union { half8 v; uint32_t u[4]; } r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r.u[e] = cvt_pk(acc[8 * half + 2 * e], acc[8 * half + 2 * e + 1]);
#pragma unroll
  for (int e = 0; e < 4; ++e) r.u[e] = relu_pk(r.u[e]);
  return r.v;
}

// U8: the frame is NHWC uint8 and simple_normalize ((x/255 - 0.5)/0.5, augmentation_pipeline.py:31-36) is applied while the
// raw tile is staged: the LDS image is the one the fp16 path builds for a frame at a 16-byte aligned (virtual) address, so
// everything after the staging is shared.
// ALN: the frame's rows are 16-byte aligned (W % 8 == 0, 16-byte aligned base: every video format): all rows of all
// tiles have the same sub-chunk misalignment (14 bytes: tiles start 3 pixels = 18 bytes left of a 128-pixel boundary),
// so the realignment shift, the LDS read offsets and the global addresses of a tile's chunks are compile-time / per-thread
// constants instead of ~90 instructions per chunk -- with one wave per SIMD nothing hides them (fetch + store of the raw
// tile were 2.4 k of the 14.3 k cycles of a tile).
template <bool U8, bool ALN>
__global__ __launch_bounds__(256, 1) void k_stem2x(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_raw = smem;
  char* s_mid = smem + X2::OFF_MID;
  float* s_b = reinterpret_cast<float*>(smem + X2::OFF_B);   // [4][64]: b1 b2 b3 b4

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int hh = lane >> 5, pix = lane & 31;

  // ---- filters.  conv1 / conv3 in the standard order; the two 1x1s with permuted K (to consume the previous
  //      accumulator registers directly) and permuted rows (8 consecutive channels per lane).
  // conv3's filter (72 fragments): the first W3A k-steps of both cout tiles are loaded straight into AccVGPRs by inline
  // asm ("=a" outputs) and read from there by the MFMAs.  Left to the register allocator, ~47 of the 72 fragments were
  // "spilled" to AccVGPRs and copied back with v_accvgpr_read in front of every use (190 VALU instructions per tile that
  // nothing overlaps with one wave per SIMD).  The next W3V k-steps are ordinary registers, the last W3L live in LDS
  // (W3L * 2 reads per tile): everything resident was a handful of registers over 512.
#ifndef X2_W3A
#define X2_W3A 30
#endif
  constexpr int W3A = X2_W3A, W3L = 2, W3R = 36 - W3L, W3V = W3R - W3A;
  static_assert(W3A % 2 == 0 && W3A >= 0 && W3V >= 0 && 8 * W3A <= 256, "AccVGPR budget");
  half8 w1r[2][2], w2r[2][4], w3r[2][W3V > 0 ? W3V : 1];
  u32x4 w3a[2][W3A > 0 ? W3A : 1];
  half8* s_w3t = reinterpret_cast<half8*>(smem + X2::OFF_W3T);   // [2][W3L][64]
  half8* s_w4 = reinterpret_cast<half8*>(smem + X2::OFF_W4);   // [2][4][64]; only needed for 8 MFMAs per tile
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int k = 0; k < 2; ++k) w1r[c][k] = a.w1[(c * 2 + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < W3A; k += 2) {
      const half8* p = a.w3 + (c * 36 + k) * 64 + lane;
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"
                   : "=&a"(w3a[c][k]), "=&a"(w3a[c][k + 1]) : "v"(p) : "memory");
    }
#pragma unroll
    for (int k = 0; k < W3V; ++k) w3r[c][k] = a.w3[(c * 36 + W3A + k) * 64 + lane];
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < W3L; ++k) s_w3t[(c * W3L + k) * 64 + lane] = a.w3[(c * 36 + W3R + k) * 64 + lane];
    }
  }
  {
    const int m = lane & 31, hk = lane >> 5;
    const int g = m >> 3, hm = (m >> 2) & 1, j = m & 3;
    const int row = 16 * (g >> 1) + 8 * hm + 4 * (g & 1) + j;   // output channel (within its tile of 32) of MFMA row m
    auto perm = [&](const half8* wstd, int c, int q) {
      const _Float16* base = reinterpret_cast<const _Float16*>(wstd + (c * 4 + q) * 64);
      const half4 lo = *reinterpret_cast<const half4*>(base + row * 8 + 4 * hk);          // k = 16q + 4hk + 0..3
      const half4 hi = *reinterpret_cast<const half4*>(base + (row + 32) * 8 + 4 * hk);   // k = 16q + 8 + 4hk + 0..3
      half8 r;
      r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
      return r;
    };
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w2r[c][q] = perm(a.w2, c, q);
        if (wave == 0) s_w4[(c * 4 + q) * 64 + lane] = perm(a.w4, c, q);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the AccVGPR loads above are invisible to the compiler's counters
  if (threadIdx.x < 64) {
    s_b[threadIdx.x] = a.b1[threadIdx.x]; s_b[64 + threadIdx.x] = a.b2[threadIdx.x];
    s_b[128 + threadIdx.x] = a.b3[threadIdx.x]; s_b[192 + threadIdx.x] = a.b4[threadIdx.x];
  }
  // ---- biases of conv1 / conv2 ride on the MFMAs instead of being read from LDS for every 32-pixel group
  //      (16 KB of LDS traffic per group and wave -- phase A was LDS-bandwidth bound on it): a K slot whose B
  //      element is 1.0 and whose A element is the bias, split hi + lo in fp16 (exact to ~2^-22 relative).
  //      conv1 has 5 unused K slots (27 -> 32): slots 27, 28 = elements 3, 4 of k-step 1, lane half 1.
  //      conv2 gets a fifth k-step whose B fragment is {1, 1, 0, ...} in lane half 0.
  //      conv3 / conv4: the same (one extra MFMA instead of 16 LDS-fed register moves per accumulator).
  half8 wb2[2];
  half8* s_wb4 = reinterpret_cast<half8*>(smem + X2::OFF_WB4);   // [4][64]: conv3 c0, c1, conv4 c0, c1
  {
    const int m = lane & 31, hk = lane >> 5;
    const int g = m >> 3, hm = (m >> 2) & 1, j = m & 3;
    const int row = 16 * (g >> 1) + 8 * hm + 4 * (g & 1) + j;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float b3v = a.b3[c * 32 + m], b4v = a.b4[c * 32 + row];
      const _Float16 b3h = (_Float16)b3v, b3l = (_Float16)(b3v - (float)b3h);
      const _Float16 b4h = (_Float16)b4v, b4l = (_Float16)(b4v - (float)b4h);
      half8 f3, f4;
#pragma unroll
      for (int e = 0; e < 8; ++e) { f3[e] = (_Float16)0.f; f4[e] = (_Float16)0.f; }
      if (!hk) { f3[0] = b3h; f3[1] = b3l; f4[0] = b4h; f4[1] = b4l; }
      if (wave == 0) { s_wb4[c * 64 + lane] = f3; s_wb4[(2 + c) * 64 + lane] = f4; }
      const float b1v = a.b1[c * 32 + m];
      const _Float16 b1h = (_Float16)b1v, b1l = (_Float16)(b1v - (float)b1h);
      if (hk) { w1r[c][1][3] = b1h; w1r[c][1][4] = b1l; }
      const float b2v = a.b2[c * 32 + row];
      const _Float16 b2h = (_Float16)b2v, b2l = (_Float16)(b2v - (float)b2h);
#pragma unroll
      for (int e = 0; e < 8; ++e) wb2[c][e] = (_Float16)0.f;
      if (!hk) { wb2[c][0] = b2h; wb2[c][1] = b2l; }
    }
  }
  union { half8 v; uint32_t u[4]; } ones_f;
  ones_f.u[0] = hh ? 0u : 0x3c003c00u; ones_f.u[1] = 0u; ones_f.u[2] = 0u; ones_f.u[3] = 0u;
  const half8 ones = ones_f.v;

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const long rowbytes = (long)a.W * 6;
  const int dsh = ALN ? 0 : (int)(rowbytes & 15);   // change of a row's 16-byte misalignment from one row to the next

  // ---- raw frame tile: 19 rows x 51 aligned chunks, chunk c = tid + 256u (row r = c / 51, i = c % 51)
  struct TileGeo { long rs0; int n, ty0, tx0, gyr0, gxr0, sh0; bool live; };
  // tile walk: (image, tile row, tile column) of tile t advance by the decomposed stride -- no divisions in the loop
  struct Walk { int n, ty, tx; };
  Walk w_first, w_cur;
  {
    const int t0 = t_begin + bix < a.ntiles ? t_begin + bix : 0;
    w_first.n = t0 / tiles_per_img;
    const int tr = t0 - w_first.n * tiles_per_img;
    w_first.ty = tr / a.tiles_x; w_first.tx = tr - w_first.ty * a.tiles_x;
    w_cur = w_first;
  }
  const int st_n = t_step / tiles_per_img, st_r = t_step - st_n * tiles_per_img;
  const int st_y = st_r / a.tiles_x, st_x = st_r - st_y * a.tiles_x;
  auto walk_advance = [&]() {
    w_cur.tx += st_x; w_cur.ty += st_y; w_cur.n += st_n;
    if (w_cur.tx >= a.tiles_x) { w_cur.tx -= a.tiles_x; ++w_cur.ty; }
    if (w_cur.ty >= a.tiles_y) { w_cur.ty -= a.tiles_y; ++w_cur.n; }
  };
  auto geo = [&](int t) {       // t must be the tile w_cur stands on
    TileGeo g;
    g.live = t < t_end;
    const Walk wk = g.live ? w_cur : w_first;      // past the end: any valid tile (its loads are discarded)
    g.n = wk.n; g.ty0 = wk.ty; g.tx0 = wk.tx;
    g.gyr0 = 4 * g.ty0 * X2::TH - 3; g.gxr0 = 4 * g.tx0 * X2::TW - 3;   // raw origin = 2 * (2 * out - 1) - 1
    // byte address of raw pixel (gyr0, gxr0): outside the frame for border tiles -- only its low bits and
    // rows / chunks that are inside the frame are ever used
    g.rs0 = (U8 ? 0L : (long)reinterpret_cast<uintptr_t>(a.in)) + (((long)g.n * a.H + g.gyr0) * a.W + g.gxr0) * 6;
    g.sh0 = ALN ? 14 : (int)(g.rs0 & 15);
    return g;
  };
  // this thread's four chunks: c = tid + 256u -> row r = c / 51, chunk i = c % 51.  Recomputed where needed
  // (a handful of VALU per tile) rather than held in 16 registers across the whole kernel.
  struct Chunk { int r, e, off, rd; };
  {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = threadIdx.x + 256 * u;
      const int r = c / X2::NCH, i = c - r * X2::NCH;
      int4 k;
      k.x = c < X2::NCHUNK ? r : -1; k.y = 8 * i; k.z = c < X2::NCHUNK ? r * (int)rowbytes + 16 * i : 0;
      k.w = ALN ? (c < X2::NCHUNK ? r * X2::RSB + 16 * i : X2::OFF_DUMMY + 16 * (int)(threadIdx.x & 63)) : (r * dsh) & 15;   // ALN: LDS address of the chunk
      reinterpret_cast<int4*>(smem + X2::OFF_TCH)[u * 256 + threadIdx.x] = k;
    }
    // phase-B LDS read offsets: lane pix = output column; tap column s reads intermediate column 2*pix + s, i.e.
    // de-interleaved slot (s & 1) * 33 + pix + (s >> 1); wave w = output row w
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int rem = (s3 & 1) * X2::IWh + pix + (s3 >> 1);
      const int f = (rem >> 1) & 7;
      const int rowbase = (2 * wave) * X2::IWs + rem;
      int4 k;
      k.x = rowbase * 128 + (((0 + hh) ^ f) * 16); k.y = rowbase * 128 + (((2 + hh) ^ f) * 16);
      k.z = rowbase * 128 + (((4 + hh) ^ f) * 16); k.w = rowbase * 128 + (((6 + hh) ^ f) * 16);
      reinterpret_cast<int4*>(smem + X2::OFF_TXO)[s3 * 256 + threadIdx.x] = k;
    }
  }
  auto chunk_of = [&](int u) {
    const int4 v = reinterpret_cast<const int4*>(smem + X2::OFF_TCH)[u * 256 + threadIdx.x];
    Chunk k; k.r = v.x; k.e = v.y; k.off = v.z; k.rd = v.w;
    return k;
  };
  const char* safe_src = reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(a.in) & ~(uintptr_t)15);
  // chunk (row r, i) holds halfs [8i - sh/2, 8i - sh/2 + 8) of the row, counted from column gxr0; it is needed
  // iff its row is inside the frame and it contains a half of an in-frame column
  // chunk (row r, i) holds halfs [e0, e0 + 8) of the row, counted from column gxr0 (e0 = 8i - sh/2), the dword in
  // front of it ends with half e0 - 1.  okv / okp: the chunk / that half has something inside the frame.
  auto chunk_ok = [&](const TileGeo& g, const Chunk& k, int& e0, int& elo, int& ehi, bool& okp) {
    const int shr = ALN ? 14 : (g.sh0 + k.rd) & 15;
    e0 = k.e - (shr >> 1);
    elo = (g.gxr0 < 0 ? -g.gxr0 : 0) * 3;
    ehi = ((a.W - g.gxr0) < X2::RW ? (a.W - g.gxr0) : X2::RW) * 3;
    const int gy = g.gyr0 + k.r;
    const bool rowok = g.live && k.r >= 0 && gy >= 0 && gy < a.H;
    okp = rowok && e0 - 1 >= elo && e0 - 1 < ehi;
    return rowok && e0 + 8 > elo && e0 < ehi;
  };
  uint4 rawv[4];
  uint32_t rawp[4];   // the dword in front of each chunk (its upper half moves into the chunk when the row is shifted)
  // tile whose 19 x 131 raw region lies completely inside the frame (84 % of the tiles of a 1080p frame): every
  // chunk is needed and nothing has to be masked
  auto interior = [&](const TileGeo& g) {
    return g.live && g.gyr0 >= 0 && g.gyr0 + X2::RH <= a.H && g.gxr0 >= 0 && g.gxr0 + X2::RW <= a.W;
  };
  // The loads are unconditional (clamped address) and their results are NOT touched here: any use -- even the
  // zero-select for skipped chunks -- would make the compiler wait for the load right away and serialise the
  // four round trips.  The select happens a tile later, in raw_store.
  // U8: a chunk = 8 consecutive bytes of the frame at byte offset (virtual fp16 byte address) / 2, plus the byte in front
  // of it.  Fetched as three ALIGNED dwords starting at the dword that holds the byte in front (clamped into the frame
  // buffer: a clamped dword only ever supplies bytes of out-of-frame columns, which raw_store masks), realigned in
  // raw_put with v_alignbyte.
  const long u8_last = U8 ? (((long)a.N * a.H * a.W * 3 - 1) & ~3L) : 0;
  auto raw_fetch1 = [&](const TileGeo& g, bool inner, int u) {
    const Chunk k = chunk_of(u);
    const int shr = ALN ? 14 : (g.sh0 + k.rd) & 15;
    if constexpr (U8) {
      long b = (g.rs0 + (k.off - shr)) / 2 - 1;             // byte offset of the byte in front of the chunk
      bool ok = true;
      if (!inner) {
        int e0, elo, ehi;
        bool okp;
        ok = chunk_ok(g, k, e0, elo, ehi, okp);
        ok = ok || okp;      // the byte in front can be the last in-frame element of a row whose chunk is all padding
      } else {
        ok = (u < 3 || k.r >= 0);
      }
      if (!ok) b = 0;
      long d0 = b & ~3L;
      const char* base = reinterpret_cast<const char*>(a.in);
      uint32_t w[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        long d = d0 + 4 * j;
        d = d < 0 ? 0 : (d > u8_last ? u8_last : d);
        w[j] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(reinterpret_cast<uintptr_t>(base + d));
      }
      rawv[u] = make_uint4(w[0], w[1], w[2], (uint32_t)(b & 3));
      rawp[u] = 0;
      return;
    }
    const char* real = reinterpret_cast<const char*>(g.rs0) + (k.off - shr);
    const char* src;
    const char* srp;
    if (inner) {
      src = (u < 3 || k.r >= 0) ? real : safe_src + 16;
      srp = src - 4;
    } else {
      int e0, elo, ehi;
      bool okp;
      const bool ok = chunk_ok(g, k, e0, elo, ehi, okp);
      src = ok ? real : safe_src;
      srp = okp ? real - 4 : safe_src;
    }
    const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(reinterpret_cast<uintptr_t>(src));
    rawv[u] = make_uint4(v[0], v[1], v[2], v[3]);
    rawp[u] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(reinterpret_cast<uintptr_t>(srp));
  };
  // U8: bytes [prev, c0..c7] realigned to the low end of (lo, hi, top): lo = prev c0 c1 c2, hi = c3..c6, top = c7;
  // converted to the 8 halfs of the chunk (-> v) and the half in front of it (-> upper half of prev)
  auto u8_expand = [&](const uint4& raw, uint4& v, uint32_t& prev) {
    const uint32_t sh = raw.w;   // 0..3: position of the byte in front inside the first dword
    uint32_t lo = raw.x, hi = raw.y, top = raw.z;
    if (sh == 1) { lo = __builtin_amdgcn_alignbyte(raw.y, raw.x, 1); hi = __builtin_amdgcn_alignbyte(raw.z, raw.y, 1); top = raw.z >> 8; }
    else if (sh == 2) { lo = __builtin_amdgcn_alignbyte(raw.y, raw.x, 2); hi = __builtin_amdgcn_alignbyte(raw.z, raw.y, 2); top = raw.z >> 16; }
    else if (sh == 3) { lo = __builtin_amdgcn_alignbyte(raw.y, raw.x, 3); hi = __builtin_amdgcn_alignbyte(raw.z, raw.y, 3); top = raw.z >> 24; }
    auto cv = [](uint32_t byte) {
      // fp16(f * (2/255) - 1) == fp16((f/255 - 0.5)/0.5) for all 256 inputs (tests/test_host_logic.py checks the table)
      const float f = (float)(byte & 255u);
      const _Float16 h = (_Float16)(f * (2.0f / 255.0f) - 1.0f);
      return (uint32_t)__builtin_bit_cast(unsigned short, h);
    };
    prev = cv(lo) << 16;
    v.x = cv(lo >> 8) | (cv(lo >> 16) << 16);
    v.y = cv(lo >> 24) | (cv(hi) << 16);
    v.z = cv(hi >> 8) | (cv(hi >> 16) << 16);
    v.w = cv(hi >> 24) | (cv(top) << 16);
  };
  auto raw_fetch = [&](const TileGeo& g) {
    const bool inner = interior(g);
    if constexpr (ALN && !U8) {
      if (inner) {
        // one basic block: the four table reads first (one LDS round trip instead of four), then the eight loads from
        // a scalar tile base + per-thread constant 32-bit offsets (offset 0 for the chunk slots past the region's end)
        const char* tb = reinterpret_cast<const char*>(g.rs0 - 14);
        uint32_t off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) off[u] = (uint32_t)reinterpret_cast<const int4*>(smem + X2::OFF_TCH)[u * 256 + threadIdx.x].z;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const char* src = tb + off[u];
          const u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(reinterpret_cast<uintptr_t>(src));
          rawv[u] = make_uint4(v[0], v[1], v[2], v[3]);
          rawp[u] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(reinterpret_cast<uintptr_t>(src - 4));
        }
        return;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) raw_fetch1(g, inner, u);
  };
  // Rows whose misalignment is 2 mod 4 are stored 2 bytes further right, so that every pixel of the LDS copy starts
  // on a dword boundary (12 bytes per column pair) and the im2col reads are plain aligned dword reads.  The shift
  // is done in registers (funnel shift with the dword in front of the chunk): a 2-byte-misaligned 16-byte LDS
  // write costs 100+ cycles, 400+ with four waves writing (tools/ub/vmem3.hip).
  auto raw_put = [&](const TileGeo& g, const Chunk& k, const uint4& v0, uint32_t prev0) {
    uint4 v = v0;
    uint32_t prev = prev0;
    if constexpr (U8) u8_expand(v0, v, prev);
    const int shr_s = ALN ? 14 : (g.sh0 + k.rd) & 15;
    u32x4 o;
    if (shr_s & 2) {
      o[0] = __builtin_amdgcn_alignbit(v.x, prev, 16); o[1] = __builtin_amdgcn_alignbit(v.y, v.x, 16);
      o[2] = __builtin_amdgcn_alignbit(v.z, v.y, 16);  o[3] = __builtin_amdgcn_alignbit(v.w, v.z, 16);
    } else {
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    if constexpr (ALN) *reinterpret_cast<u32x4*>(s_raw + k.rd) = o;
    else *reinterpret_cast<u32x4*>(s_raw + (k.r < 0 ? 0 : k.r) * X2::RSB + 2 * k.e) = o;
  };
  auto raw_store = [&](const TileGeo& g) {
    if (interior(g)) {
      if constexpr (ALN) {
        // unconditional: the table sends the chunk slots past the region's end to the dummy area
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[u] = reinterpret_cast<const int4*>(smem + X2::OFF_TCH)[u * 256 + threadIdx.x].w;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          Chunk k; k.r = 0; k.e = 0; k.off = 0; k.rd = dst[u];
          raw_put(g, k, rawv[u], rawp[u]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const Chunk k = chunk_of(u);
          if (u < 3 || k.r >= 0) raw_put(g, k, rawv[u], rawp[u]);   // only the 4th chunk of a thread can be past the end (969 = 3*256 + 201)
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const Chunk k = chunk_of(u);
        if (u < 3 || k.r >= 0) {
          int e0, elo, ehi;
          bool okp;
          const bool ok = chunk_ok(g, k, e0, elo, ehi, okp);
          uint4 vx = rawv[u];
          uint32_t px = rawp[u];
          if constexpr (U8) u8_expand(rawv[u], vx, px);
          const uint4 v = ok ? vx : make_uint4(0u, 0u, 0u, 0u);
          const uint32_t pv = okp ? px : 0u;
          // after the (optional) one-half shift position kk of the chunk holds half e0s + kk of the row: zero the
          // ones that belong to out-of-frame columns
          const int shr_s = ALN ? 14 : (g.sh0 + k.rd) & 15;
          const int e0s = e0 - ((shr_s >> 1) & 1);
          u32x4 o;
          if (shr_s & 2) {
            o[0] = __builtin_amdgcn_alignbit(v.x, pv, 16); o[1] = __builtin_amdgcn_alignbit(v.y, v.x, 16);
            o[2] = __builtin_amdgcn_alignbit(v.z, v.y, 16); o[3] = __builtin_amdgcn_alignbit(v.w, v.z, 16);
          } else {
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int ea = e0s + 2 * kk, eb = ea + 1;
            const uint32_t m = ((ea >= elo && ea < ehi) ? 0xffffu : 0u) | ((eb >= elo && eb < ehi) ? 0xffff0000u : 0u);
            o[kk] &= m;
          }
          *reinterpret_cast<u32x4*>(s_raw + k.r * X2::RSB + 2 * k.e) = o;
        }
      }
    }
  };

  int t = t_begin + bix;
  // De-phase the workgroups.  All 256 of them run the same tile loop at the same speed; started together, they
  // would all hit the memory system in the same few hundred cycles of every tile (raw-frame loads, output stores)
  // and leave it idle for the rest -- and a wave stalled on a full memory queue cannot issue anything else.
  // A one-off start delay of 0..15/16 of a tile time spreads those bursts evenly.
  {
    const int steps = (a.stagger ? (bix & 15) : 0);
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(14);   // 14 * 64 = 896 cycles
  }
  TileGeo g_cur = geo(t);
  raw_fetch(g_cur);
  raw_store(g_cur);

  char* wst = smem + X2::OFF_STG + wave * 4096;   // wave-private output staging: [32 px][128 B], XOR-swizzled
  const int fo = (pix >> 1) & 7;

  // ---- phase A works on one 32-pixel group of the intermediate region per call chain; the stages are
  //      written branch-free so that two groups can be interleaved in one basic block (with a single wave per
  //      SIMD the only latency hiding there is comes from independent instructions of the same wave)
  struct Grp {
    int A0, A1, A2;          // byte addresses of raw rows 2my, 2my+1, 2my+2 at column 2mx
    int dst;                 // byte address of the pixel's 128-byte slot in the intermediate tile (or the dummy)
    int fk;                  // its swizzle key
    uint32_t vmask;          // ~0 inside the intermediate image, 0 = zero padding of conv3
    half8 f0, f1;            // im2col fragments (K = 27 -> 32)
    f32x16 acc[2];
    half8 bq[4];
  };
  int itn = 0;
  for (; t < t_end; t += t_step, ++itn) {
    X2_T(0);
    const int gy1_0 = 2 * g_cur.ty0 * X2::TH - 1, gx1_0 = 2 * g_cur.tx0 * X2::TW - 1;   // intermediate-image origin
    const int sh0 = ALN ? 14 : g_cur.sh0;
    __syncthreads();   // raw tile of this tile is in LDS; every wave is done reading the previous intermediate tile
    X2_T(1);

    // ================= phase A: raw -> conv1 -> conv2 -> intermediate tile (all in this wave) =================
    // Group -> pixels: groups 0..17 are the two 32-column halves of the nine rows (row = grp >> 1 is wave-uniform,
    // so nearly all of the address arithmetic is scalar); group 18 is the left-over 65th column (9 pixels).
    // MASKED = false: the whole 9 x 65 region lies inside the intermediate image (all tiles but the frame's border
    // ones) -- no zero-padding mask to compute and apply (22 instructions per group)
    auto phase_a = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // Group -> pixels: groups 0..17 are the two 32-column halves of the nine rows (row = grp >> 1 is wave-uniform,
      // so nearly all of the address arithmetic is scalar); group 18 is the left-over 65th column (9 pixels).
      // Every wave runs FIVE groups: wave w has w, w+4, w+8, w+12, then 16 / 17 / 18 for waves 0 / 1 / 2 and a dummy
      // (valid reads, results into the dummy area) for wave 3 -- the pipeline below is straight-line code.
      auto s_addr_px = [&](Grp& G, int my, int mx, bool inreg) {
        const int r0 = 2 * my;
        const int shb = sh0 + r0 * dsh;
        const int a0 = r0 * X2::RSB + 12 * mx;
        const int s0 = shb & 15, s1 = (shb + dsh) & 15, s2 = (shb + 2 * dsh) & 15;
        G.A0 = a0 + s0 + (s0 & 2); G.A1 = a0 + X2::RSB + s1 + (s1 & 2); G.A2 = a0 + 2 * X2::RSB + s2 + (s2 & 2);   // dword aligned
        const int gy1 = gy1_0 + my, gx1 = gx1_0 + mx;
        G.vmask = (!MASKED || (gy1 >= 0 && gy1 < a.H1 && gx1 >= 0 && gx1 < a.W1)) ? 0xffffffffu : 0u;
        const int rem = (mx & 1) * X2::IWh + (mx >> 1);
        G.fk = inreg ? (rem >> 1) & 7 : 0;
        G.dst = inreg ? X2::OFF_MID + (my * X2::IWs + rem) * 128 : X2::OFF_DUMMY + pix * 128;
      };
      // Lane -> column within a 32-column half (round 3): lane p works on column 4 (p & 7) + {0, 2, 1, 3}[p >> 3], not on
      // column p.  A pixel's 16-byte chunks are XOR-swizzled with the key (slot >> 1) & 7 of its de-interleaved slot -- what
      // makes conv3's B-fragment reads conflict-free -- and ds_write_b128 is serviced 8 consecutive lanes at a time over 32
      // banks: with lane = column, 8 consecutive lanes are 4 even + 4 odd columns whose keys coincide in pairs and across
      // the two parities -> 4-way bank conflicts on every one of the 80 tile writes (32 instead of 8 LDS cycles each;
      // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.53 in profiles/r02b_pmc_sq_counters.txt).  With this order the 8 lanes of
      // a group hold columns of ONE parity, 4 apart = slots 2 apart = 8 distinct keys: conflict-free (tools/stem_lds_sim.py).
      // The raw-tile reads keep their bank pattern (the same 32 addresses per instruction, assigned to other lanes), and a
      // lane's pixel is computed by that lane alone from im2col to the store, so the values do not change.
#ifdef X2_LINEAR_LANES
      const int pcol = pix;
#else
      const int pcol = 4 * (pix & 7) + ((0x3120 >> (4 * (pix >> 3))) & 3);
#endif
      auto s_addr_k = [&](Grp& G, int k) {
        if (k < 4) {                               // wave-uniform row / half
          const int grp = wave + 4 * k;
          s_addr_px(G, grp >> 1, ((grp & 1) << 5) + pcol, true);
        } else {                                   // selects, no branches
          const int my = wave < 2 ? X2::IH - 1 : (wave == 2 ? (pix < X2::IH ? pix : X2::IH - 1) : 0);
          const int mx = wave < 2 ? (wave << 5) + pcol : (wave == 2 ? X2::IW - 1 : pix);
          s_addr_px(G, my, mx, wave < 2 || (wave == 2 && pix < X2::IH));
        }
      };
      // im2col: k-step 0 = {h0: row0 e0..7 | h1: row1 e0..7}, k-step 1 = {h0: row2 e0..7 | h1: row0 e8, row1 e8,
      // row2 e8, 1, 1(bias slots), 0 x3}, e = 3 * s + c (pack_stem_weight).  Every lane issues the same reads
      // (addresses differ by lane half); issue and use are separate steps of the pipeline.
      struct Raw { uint32_t f[4], x[4], y1, y2; };
      auto s_load_issue = [&](const Grp& G, Raw& R) {
        const uint32_t* p0 = reinterpret_cast<const uint32_t*>(s_raw + (hh ? G.A1 : G.A0));
        R.f[0] = p0[0]; R.f[1] = p0[1]; R.f[2] = p0[2]; R.f[3] = p0[3];
        const uint32_t* px = reinterpret_cast<const uint32_t*>(s_raw + (hh ? G.A0 + 16 : G.A2));   // h0: row2 e0..7 | h1: row0 e8 (+7 unused)
        R.x[0] = px[0]; R.x[1] = px[1]; R.x[2] = px[2]; R.x[3] = px[3];
        R.y1 = *reinterpret_cast<const uint32_t*>(s_raw + G.A1 + 16);                               // row1 e8 (low half)
        R.y2 = *reinterpret_cast<const uint32_t*>(s_raw + G.A2 + 16);                               // row2 e8 (low half)
      };
      auto s_load_finish = [&](Grp& G, const Raw& R) {
        union { half8 v; uint32_t u[4]; } f0, f1;
        f0.u[0] = R.f[0]; f0.u[1] = R.f[1]; f0.u[2] = R.f[2]; f0.u[3] = R.f[3];
        f1.u[0] = hh ? ((R.x[0] & 0xffffu) | (R.y1 << 16)) : R.x[0];
        f1.u[1] = hh ? ((R.y2 & 0xffffu) | 0x3c000000u) : R.x[1];   // element 3 = 1.0: bias slot (hi)
        f1.u[2] = hh ? 0x00003c00u : R.x[2];                        // element 4 = 1.0: bias slot (lo)
        f1.u[3] = hh ? 0u : R.x[3];
        G.f0 = f0.v; G.f1 = f1.v;
      };
      // the four stages, one MFMA / one 8-channel octet at a time
      auto conv1_m = [&](Grp& G, int i) {          // i = 0..3: cout tile i >> 1, k-step i & 1
        const int c = i >> 1;
        if ((i & 1) == 0) G.acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1r[c][0], G.f0, zero, 0, 0, 0);
        else G.acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1r[c][1], G.f1, G.acc[c], 0, 0, 0);
      };
      auto relu1_o = [&](Grp& G, int o) { G.bq[o] = relu8(G.acc[o >> 1], o & 1); };   // o = 0..3
      auto conv2_m = [&](Grp& G, int i) {          // i = 0..9: cout tile i / 5; step 0 = bias k-step
        const int c = i / 5, q = i % 5;
        if (q == 0) G.acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb2[c], ones, zero, 0, 0, 0);
        else G.acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2r[c][q - 1], G.bq[q - 1], G.acc[c], 0, 0, 0);
      };
      auto write_o = [&](Grp& G, int o) {          // o = 0..3: 16-byte chunk 2o + hh of the pixel's line
        union { half8 v; uint32_t u[4]; } r;
        r.v = relu8(G.acc[o >> 1], o & 1);
        if constexpr (MASKED) {
#pragma unroll
          for (int e = 0; e < 4; ++e) r.u[e] &= G.vmask;   // pixels outside the intermediate image = conv3's zero padding
        }
        *reinterpret_cast<half8*>(smem + G.dst + (((2 * o + hh) ^ G.fk) * 16)) = r.v;
      };
      // ---- software pipeline over the five groups.  A 32x32x16 MFMA occupies the matrix pipe for 32 cycles and the
      //      wave can issue ~7 independent VALU instructions in its shadow -- but only if they are THERE: the stages
      //      of one group depend on each other (conv1 -> ReLU/pack -> conv2 -> ReLU/pack/store), so each step pairs
      //      the MFMAs of one group with the VALU work of another (left to itself the compiler emits all of conv2's
      //      MFMAs back to back and then 70 VALU instructions with an idle matrix pipe: 1300 cycles per group).
      //        step A_k: conv2(G_k)   [10 MFMA]  ||  ReLU1(G_k+1), im2col loads of G_k+2
      //        step B_k: conv1(G_k+2) [ 4 MFMA]  ||  ReLU2 + store of G_k
      //      sched_barrier(0) pins one {MFMA, VALU octet} slot after the other.
#define X2_SB() __builtin_amdgcn_sched_barrier(0)
      Grp G[5];
      Raw R[2];
      s_addr_k(G[0], 0); s_load_issue(G[0], R[0]);
      s_addr_k(G[1], 1); s_load_issue(G[1], R[1]);
      s_load_finish(G[0], R[0]);
      X2_SB();
      conv1_m(G[0], 0); conv1_m(G[0], 1); X2_SB();
      s_load_finish(G[1], R[1]); X2_SB();
      conv1_m(G[0], 2); conv1_m(G[0], 3); X2_SB();
#pragma unroll
      for (int i = 0; i < 4; ++i) { conv1_m(G[1], i); X2_SB(); relu1_o(G[0], i); X2_SB(); }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        // step A_k
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          conv2_m(G[k], i); X2_SB();
          if (i < 4 && k + 1 < 5) { relu1_o(G[k + 1], i); X2_SB(); }
          if (i == 4 && k + 2 < 5) { s_addr_k(G[k + 2], k + 2); s_load_issue(G[k + 2], R[k & 1]); X2_SB(); }
          if (i == 8 && k + 2 < 5) { s_load_finish(G[k + 2], R[k & 1]); X2_SB(); }
        }
        // step B_k
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (k + 2 < 5) { conv1_m(G[k + 2], i); X2_SB(); }
          write_o(G[k], i); X2_SB();
        }
      }
#undef X2_SB
    };   // phase_a
    if (gy1_0 >= 0 && gy1_0 + X2::IH <= a.H1 && gx1_0 >= 0 && gx1_0 + X2::IW <= a.W1) phase_a(std::false_type{});
    else phase_a(std::true_type{});
    X2_T(2);
    __syncthreads();   // intermediate tile complete; raw tile consumed
    X2_T(3);

    // next tile's raw frame region: fetched now (registers), written to LDS at the end of this tile -- behind
    // the barrier above nobody reads the raw tile any more, and the round trip hides under phase B
    const TileGeo g_out = g_cur;
    // phase-B read offsets: requested before the raw fetch so that their LDS round trip hides under its address work
    int xoff[3][4];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int4 k = reinterpret_cast<const int4*>(smem + X2::OFF_TXO)[s3 * 256 + threadIdx.x];
      xoff[s3][0] = k.x; xoff[s3][1] = k.y; xoff[s3][2] = k.z; xoff[s3][3] = k.w;
    }
    walk_advance();
    g_cur = geo(t + t_step);
    raw_fetch(g_cur);

    // ================= phase B: conv3 (3x3 s2, 64 -> 64) over the intermediate tile, this wave = output row =====
    f32x16 acc3[2];
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) acc3[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_wb4[c * 64 + lane], ones, zero, 0, 0, 0);   // bias k-step
    }
    X2_T(4);
    auto xfrag = [&](int k) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return *reinterpret_cast<const half8*>(s_mid + xoff[s][q] + r * X2::IWs * 128);
    };
    {
      constexpr int PD = 3;
      half8 xq[PD + 1], w3t[2][W3L];
#pragma unroll
      for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
#pragma unroll
      for (int k = 0; k < 36; ++k) {
        if (k + PD < 36) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (k == W3R - 8) {                     // the LDS-resident tail of the filter: requested eight k-steps ahead
#pragma unroll
            for (int kk = 0; kk < W3L; ++kk) w3t[c][kk] = s_w3t[(c * W3L + kk) * 64 + lane];
          }
          const half8 wk = k < W3A ? __builtin_bit_cast(half8, w3a[c][k < W3A ? k : 0])
                                   : (k < W3R ? w3r[c][(k >= W3A && k < W3R) ? k - W3A : 0] : w3t[c][k < W3R ? 0 : k - W3R]);
          acc3[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wk, xq[k % (PD + 1)], acc3[c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    X2_T(5);
    // conv4 (1x1) straight from conv3's accumulators.  Its ten filter fragments (bias step + four k-steps per cout tile) come
    // from LDS through a 4-deep register ring requested BEFORE the ReLU / pack of conv3's result: read right in front of
    // their MFMA, each of them waited a full LDS round trip.
    auto w4frag = [&](int i) {                     // i = 0..9: cout tile i / 5; step 0 = the bias fragment
      const int c = i / 5, q = i % 5;
      return q == 0 ? s_wb4[(2 + c) * 64 + lane] : s_w4[(c * 4 + q - 1) * 64 + lane];
    };
    constexpr int W4PD = 3;
    half8 w4ring[W4PD + 1];
#pragma unroll
    for (int i = 0; i < W4PD; ++i) w4ring[i] = w4frag(i);
    half8 bq[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) { bq[2 * c] = relu8(acc3[c], 0); bq[2 * c + 1] = relu8(acc3[c], 1); }
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        if (i + W4PD < 10) w4ring[(i + W4PD) % (W4PD + 1)] = w4frag(i + W4PD);
        const int c = i / 5, q = i % 5;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4ring[i % (W4PD + 1)], q == 0 ? ones : bq[q == 0 ? 0 : q - 1], q == 0 ? zero : acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q == 4) {
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2)
            *reinterpret_cast<half8*>(wst + pix * 128 + (((4 * c + 2 * g2 + hh) ^ fo) * 16)) = relu8(acc, g2);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    X2_T(5 + 3);   // slot 8 unused by the reader; keeps numbering simple
#ifdef LFD_X2_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X2_T(9);
#endif
    // ---- this wave's output row leaves as whole 128-byte lines
    {
      // instruction jc: pixels 8jc..8jc+7 x 8 chunks = 1 KB contiguous in the output row -> scalar row base
      // + lane * 16 + jc * 1024; LDS side: pixel p's chunk c8 sits at ((c8 ^ (p >> 1)) & 7) -- (p >> 1) & 7 =
      // 4 * (jc & 1) + (lane >> 4), two lane constants
      const int oy = g_out.ty0 * X2::TH + wave;
      char* ob = reinterpret_cast<char*>(a.out) + (((long)g_out.n * a.H2 + oy) * a.W2 + (long)g_out.tx0 * X2::TW) * 128 + lane * 16;
      const int ox0 = g_out.tx0 * X2::TW + (lane >> 3);
      const int lo0 = (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 4)) ^ 0) * 16), lo1 = (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 4)) ^ 4) * 16);
#pragma unroll
      for (int jc = 0; jc < 4; ++jc) {
        const uint4 v = *reinterpret_cast<const uint4*>(wst + jc * 1024 + ((jc & 1) ? lo1 : lo0));
        if (oy < a.H2 && ox0 + jc * 8 < a.W2) *reinterpret_cast<uint4*>(ob + jc * 1024) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    X2_T(6);
    raw_store(g_cur);   // (a tile index past the end stores zeros; nobody reads them)
    X2_T(7);
  }
}
#ifdef LFD_X2_TIMING
}  // namespace
extern "C" __attribute__((visibility("default"))) int lfd_debug_x2_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_x2_dbg), sizeof(unsigned long long) * 128);
}
namespace {
#endif

template <bool U8, bool ALN>
int launch_stem2x(FusedArgs a, hipStream_t st) {
  a.tiles_x = (a.W2 + X2::TW - 1) / X2::TW;
  a.tiles_y = (a.H2 + X2::TH - 1) / X2::TH;
  const long long nt = (long long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffLL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  static unsigned long long done_mask = 0;
  const int done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(done_mask, done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem2x<U8, ALN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            X2::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(done_mask, done_dev);
  }
  const int blocks = 8 * ((a.ntiles + 7) / 8) < 256 ? 8 * ((a.ntiles + 7) / 8) : 256;   // (XCD-contiguous tile ranges)
  if (blocks < 1) return LFD_OK;
  // (round 1's kernel gained 3 % from de-phasing the workgroups' memory bursts with a start delay of up to one tile time; with
  //  the round-2 staging -- one basic block of loads per tile -- the delay only costs: 178 vs 172 us at 8 x 1080p.  Opt-in.)
  a.stagger = lfd_tune(LFD_TUNE_X2_STAGGER) && a.ntiles >= 16 * blocks;
  hipLaunchKernelGGL((k_stem2x<U8, ALN>), dim3(blocks), dim3(256), X2::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

template <int NCT, int FMT>
int launch_fused(FusedArgs a, hipStream_t st) {
  using F = FCfg<NCT>;
  a.tiles_x = (a.W2 + F::TW - 1) / F::TW;
  a.tiles_y = (a.H2 + F::TH - 1) / F::TH;
  const long long nt = (long long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffLL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  static unsigned long long done_mask = 0;
  const int done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(done_mask, done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fused<NCT, FMT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(done_mask, done_dev);
  }
  int blocks = 8 * ((a.ntiles + 7) / 8) < 512 ? 8 * ((a.ntiles + 7) / 8) : 512;
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
  const int use_x2 = lfd_tune(LFD_TUNE_STEM2X);
  // aligned rows: W % 8 == 0 pixels (16-byte row pitch in fp16 / in the virtual fp16 image of a uint8 frame) and, for
  // fp16 frames, a 16-byte aligned base (LFD_X2_ALN=0 forces the general kernel: tests compare the two bit for bit)
  const int use_aln = lfd_tune(LFD_TUNE_X2_ALN);
  const bool aln = use_aln && (w % 8) == 0;
  // (the row-streaming form of this kernel, k_stem_rows, measured equal at 8 x 1080p in round 3: tools/negative_results/stem_rows.hip)
  if (use_x2 && channels == 64 && in_format == IN_NHWC_F16) {
    if (aln && (reinterpret_cast<uintptr_t>(in) & 15) == 0) return launch_stem2x<false, true>(a, st);
    return launch_stem2x<false, false>(a, st);
  }
  if (use_x2 && channels == 64 && in_format == IN_NHWC_U8) return aln ? launch_stem2x<true, true>(a, st) : launch_stem2x<true, false>(a, st);
  return channels == 64 ? dispatch_fmt<2>(in_format, a, st) : dispatch_fmt<1>(in_format, a, st);
}

}  // extern "C"
