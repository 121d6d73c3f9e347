// csrc/planes_stem2xs.hip -- k_pl_stem2xs: the whole 'faster' stem (lfd_resnet.py:376-413: conv3x3 s2 3 -> 64, conv1x1,
// conv3x3 s2 64 -> 64, conv1x1, each + BN + ReLU) on hi/lo planes in one launch, as a ROW STREAM with producer waves and
// consumer waves (round 6; fp16 / uint8 NHWC and fp32 NCHW frames with 16-byte aligned rows; everything else runs k_pl_stem2x).
//
// k_pl_stem2x (planes_stem2x.hip, planes_impl.h PROD) runs one wave per SIMD through a serial chain per producer round; the
// matrix pipe is busy 0.35 of the time at 1.3 kW and the full clock (LESSONS 50-53).  Its eight-wave form with every wave in
// the same phase between two workgroup barriers was slower (LESSONS 60, tools/negative_results/planes_stem2x8.hip.txt): a
// second wave per SIMD pays when it does something ELSE.  Here a 512-thread workgroup walks down a strip of 16 output columns,
// two output rows ("chunk") per slot, one barrier per slot:
//   * waves 0-3, PRODUCERS: the pixels of the 33-column mid-tensor strip as ONE continuous stream in groups of 32 -- no
//     rounding up per tile, and no vertical halo: a chunk adds 4 new mid rows (132 pixels = 4.125 groups) to a ring of 10 rows
//     in LDS (the 4 x 16 tile of k_pl_stem2x recomputes 297 pixels per 64 outputs, this 264).  A wave takes a whole group
//     through conv0 (both 32-channel slabs, 8 MFMAs) -> ReLU -> split -> conv1x1 (24 MFMAs; its K fragments ARE the split
//     accumulators, all in the wave's own registers: no fragment exchange, no barrier) -> ReLU -> split -> ring.  They also
//     fetch the frame patch of the next chunk by LDS-DMA.
//   * waves 4-7, CONSUMERS of the chunk produced one slot earlier: (slab s, K half kh).  kh = 1 contracts k-steps [KA, 36)
//     and hands main + 2^-11 corr to its partner through LDS; kh = 0 contracts [0, KA), and under those MFMAs finishes the
//     chunk of the slot before (partial sums -> ReLU -> split -> operand planes of the chained 1x1) and the chunk before
//     that (chained 1x1 -> ReLU -> split -> wave-private staging -> stores).
// Every hand-over is one slot old at its reader, so ONE workgroup barrier per slot orders all of them (double buffers).
// Same packed filters as lfd_pl_stem2x; sums in a different order (K halves): equal to fp32 rounding, not bit for bit.
#include "planes_impl.h"

namespace pl {

struct SS {
  static constexpr int TW = 16, IW = 33, IWh = 17, IWs = 34, PIXB = 128;
  static constexpr int RR = 10;                                   // ring rows: 5 being read + 4 new + 1 overhang of the last group
  static constexpr int ROWB = IWs * PIXB;
  static constexpr int RING_PLANE = RR * ROWB;                    // (< 64 KB: the lo plane is an immediate offset of the LDS read)
  static constexpr int RING_OFF = 0;
  static constexpr int PROLOGUE_PX = 5 * IW;                      // a segment's first chunk: 5 rows
  static constexpr int PR = 12, FPITCH = 256, JUNK = 7, FJ = (2 * IW + 2) * 3;   // frame patch: 5 mid rows = 11 frame rows
  static constexpr int PATCH_BYTES = PR * FPITCH * 2;
  static constexpr int PATCH_OFF = RING_OFF + 2 * RING_PLANE;
  static constexpr int MID_OFF = PATCH_OFF + 2 * PATCH_BYTES;     // operand planes of the chained 1x1, two chunks
  static constexpr int MID_PLANE = 32 * 128, MID_BUF = 2 * MID_PLANE;
  static constexpr int XCH_OFF = MID_OFF + 2 * MID_BUF;           // partial sums kh = 1 -> kh = 0: [2 chunks][2 slabs][4][64 lanes] x 16 B
  static constexpr int XCH_SLAB = 64 * 16 * 4, XCH_BUF = 2 * XCH_SLAB;
  static constexpr int STG_OFF = XCH_OFF + 2 * XCH_BUF;           // wave-private output staging of the two kh = 0 waves
  static constexpr int STG_PITCH = 80, STG_PLANE = 32 * STG_PITCH, STG_WAVE = 2 * STG_PLANE;
  static constexpr int TW_OFF = STG_OFF + 2 * STG_WAVE;           // chained 1x1 filters [2 planes][2 slabs][4][64 lanes] x 16 B
  static constexpr int BIAS_OFF = TW_OFF + 2 * 2 * 4 * 64 * 16;   // producer 1x1 bias [64] | consumer bias [64] | chained 1x1 bias [64]
  static constexpr int LUT_OFF = BIAS_OFF + 3 * 256;              // uint8 frames: byte -> (hi | lo << 16) of simple_normalize(byte); entry 256 = zero
  static constexpr int LDS_BYTES = LUT_OFF + 264 * 4;
  static constexpr int U8_PITCH = 256, U8_JB = 4;                 // uint8 frames: bytes per patch row; bytes left of the patch's first tap pixel
  // fp32 NCHW frames: a patch row = the three channels' 72 floats side by side (frame columns 64 tx - 4 ..: 18 lanes x 16 B per
  // channel, one LDS-DMA instruction per patch row); 12 rows = 10368 B -- buffer 0 in the patch area, buffer 1 where the other formats
  // keep the chained 1x1's filters (this format keeps them in registers)
  static constexpr int F32_CH = 288, F32_ROW = 3 * F32_CH, F32_PATCH = PR * F32_ROW;
  static_assert(F32_PATCH <= 2 * PATCH_BYTES && F32_PATCH <= 2 * 2 * 4 * 64 * 16, "fp32 patch buffers");
  static_assert(RING_PLANE < 65536, "lo plane as an immediate offset");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS capacity");
};

#ifndef PL_SS_KA
#define PL_SS_KA 19
#endif
#ifndef PL_SS_PD
#define PL_SS_PD 2
#endif

#ifdef LFD_PL_TIMING
#define SS_T(w, i) do { if (blockIdx.x == PL_DBG_BLOCK && threadIdx.x == (w) * 64 && slot >= 8 && slot < 16) g_pl_dbg[(slot - 8) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SS_T(w, i)
#endif

// the walk of a workgroup over its run of chunks: image, strip, chunk of the strip, chunk of the SEGMENT (a segment starts where
// the run starts and at every strip change: its first chunk needs 5 rows, the others 4), ring row of the chunk's first row
struct Walk {
  int n, tx, cy, j, rc;
};
__device__ __forceinline__ void walk_next(Walk& w, int CY, int tiles_x) {
  ++w.cy;
  if (w.cy == CY) {
    w.cy = 0;
    if (++w.tx == tiles_x) { w.tx = 0; ++w.n; }
    w.j = 0;
    w.rc += 5;
  } else {
    ++w.j;
    w.rc += 4;
  }
  if (w.rc >= SS::RR) w.rc -= SS::RR;
}

// FMT: IN_NHWC_F16 | IN_NHWC_U8 | IN_NCHW_F32 (planes_impl.h)
template <int KA, int FMT>
__global__ __launch_bounds__(512, 1) void k_pl_stem2xs(PlArgs a, PlProd P, int CY, int total) {
  constexpr bool U8 = FMT == IN_NHWC_U8, F32 = FMT == IN_NCHW_F32;
  using C = SS;
  constexpr int KB = 36 - KA;
  constexpr int PD = PL_SS_PD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pix = lane & 31;

#ifdef LFD_PL_TIMING
  if (blockIdx.x == PL_DBG_BLOCK && threadIdx.x == 0) { g_pl_dbg[128] = __builtin_readcyclecounter(); g_pl_dbg[129] = __builtin_amdgcn_s_memrealtime(); }
#endif
  // ---- shared constants
  {
    if constexpr (!F32) {
      half8* tw = reinterpret_cast<half8*>(smem + C::TW_OFF);
      for (int i = threadIdx.x; i < 2 * 2 * 4 * 64; i += 512) {
        const int pl = i >> 9, r = i & 511;
        tw[i] = a.w2[(size_t)pl * a.w2_plane + r];
      }
    }
    float* sb = reinterpret_cast<float*>(smem + C::BIAS_OFF);
    if (threadIdx.x < 64) {
      sb[threadIdx.x] = P.b2[threadIdx.x];
      sb[64 + threadIdx.x] = a.bias[threadIdx.x];
      sb[128 + threadIdx.x] = a.bias2[threadIdx.x];
    }
    if constexpr (U8) {
      // simple_normalize (augmentation_pipeline.py:31-36) in fp32 like the reference, split like every plane value -- per byte VALUE,
      // once per workgroup (the tile kernel evaluates the same expressions per patch element: planes_impl.h frame_fill)
      uint32_t* lut = reinterpret_cast<uint32_t*>(smem + C::LUT_OFF);
      if (threadIdx.x < 264) {
        const float v = threadIdx.x < 256 ? px_value<IN_NHWC_U8>(threadIdx.x) : 0.f;
        const _Float16 hh = (_Float16)v;
        const _Float16 ll = (_Float16)((v - (float)hh) * kLo);
        lut[threadIdx.x] = (uint32_t)__builtin_bit_cast(unsigned short, hh) | ((uint32_t)__builtin_bit_cast(unsigned short, ll) << 16);
      }
    }
  }

  // ---- this workgroup's run of chunks
  const long u0 = (long)blockIdx.x * total / gridDim.x, u1 = (long)(blockIdx.x + 1) * total / gridDim.x;
  const int nchunks = (int)(u1 - u0);
  Walk first;
  {
    const int per_img = a.tiles_x * CY;
    first.n = (int)(u0 / per_img);
    const int r = (int)(u0 - (long)first.n * per_img);
    first.tx = r / CY;
    first.cy = r - first.tx * CY;
    first.j = 0;
    first.rc = 0;
  }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nslots = nchunks + 3;

  if (wave < 4) {
    // ========================================================= producers =====================================================
    half8 w1h[2][2], w1l[2][2];               // conv0 (+ bias slot) [slab][k-step]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        w1h[ct][ks] = P.w1[((0 * 2 + ct) * 2 + ks) * 64 + lane];
        w1l[ct][ks] = P.w1[((1 * 2 + ct) * 2 + ks) * 64 + lane];
      }
    half8 p2h[2][4], p2l[2][4];               // 1x1 [output slab][k-step q: consumes the fragment of conv0 slab q >> 1, half q & 1]
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        p2h[co][q] = P.w2[(co * 4 + q) * 64 + lane];
        p2l[co][q] = P.w2[2 * 4 * 64 + (co * 4 + q) * 64 + lane];
      }
    const float* pb = reinterpret_cast<const float*>(smem + C::BIAS_OFF);
    constexpr int RD = C::FPITCH / 2;          // dwords per patch row
    const int go0 = h ? 4 : RD, go1 = h ? RD + 4 : RD + 1, go2 = h ? 2 * RD + 4 : RD + 2;

    // first mid row of the stream part a chunk's producer slot covers (segment rows): 0 | 4 j + 1
    auto patch_dma = [&](const Walk& w, int pbuf) {
      const int mlo = w.j ? 4 * w.j + 1 : 0;
      const int gy_lo = 4 * (w.cy - w.j) - 1 + mlo;
      const int fy0 = 2 * gy_lo - 1;
      const int gx0 = 2 * C::TW * w.tx - 1, fxm = 2 * gx0 - 2;
      if constexpr (F32) {
        // floats: patch column j of (row, channel) = frame column 64 tx - 4 + j; lane = channel * 18 + 16-byte piece (17 used); whole
        // pieces are inside or outside the frame (W % 4 == 0), outside reads the zero line
        const float* fr32 = reinterpret_cast<const float*>(P.frame);
        const int sub = lane / 18, l18 = lane - sub * 18;
        const int fcol = 4 * C::TW * w.tx - 4 + 4 * l18;
        const bool act = lane < 54 && l18 < 17;
        const bool cok = fcol >= 0 && fcol < P.FW;
        char* lb32 = smem + (pbuf ? C::TW_OFF : C::PATCH_OFF);
        for (int ii = wave; ii < C::PR; ii += 4) {
          const int fy = fy0 + ii;
          const bool ok = cok && fy >= 0 && fy < P.FH;
          const void* src = ok ? (const void*)(fr32 + (((size_t)w.n * 3 + (act ? sub : 0)) * P.FH + fy) * P.FW + fcol) : (const void*)a.zeros;
          if (act) dma16(src, lb32 + ii * C::F32_ROW);
        }
        return;
      }
      if constexpr (U8) {
        // bytes: patch byte j of a row = frame byte fxm * 3 - U8_JB + j (a multiple of 16: 192 tx - 16); 16 lanes x 16 B per row,
        // four rows per instruction
        const unsigned char* fr8 = reinterpret_cast<const unsigned char*>(P.frame);
        const int bcol = fxm * 3 - C::U8_JB + 16 * (lane & 15);
        const bool cok = bcol >= 0 && bcol < P.FW * 3;
        char* lb8 = smem + C::PATCH_OFF + pbuf * C::PATCH_BYTES;
        for (int ii = wave; ii < C::PR / 4; ii += 4) {
          const int fy = fy0 + 4 * ii + (lane >> 4);
          const bool ok = cok && fy >= 0 && fy < P.FH;
          const void* src = ok ? (const void*)(fr8 + ((size_t)w.n * P.FH + fy) * P.FW * 3 + bcol) : (const void*)a.zeros;
          dma16(src, lb8 + ii * 1024);
        }
        return;
      }
      const _Float16* fr = reinterpret_cast<const _Float16*>(P.frame);
      constexpr int NL = (C::FJ + C::JUNK - 3 + 7) / 8;
      const int hcol = fxm * 3 - C::JUNK + 3 + 8 * (lane & 31);
      const bool colok = (lane & 31) < NL && hcol >= 0 && hcol < P.FW * 3;
      char* lbase = smem + C::PATCH_OFF + pbuf * C::PATCH_BYTES;
      for (int ii = wave; ii < C::PR / 2; ii += 4) {
        const int fy = fy0 + 2 * ii + (lane >> 5);
        const bool ok = colok && fy >= 0 && fy < P.FH;
        const _Float16* src = ok ? fr + ((size_t)w.n * P.FH + fy) * P.FW * 3 + hcol : a.zeros;
        dma16(src, lbase + ii * 1024);
      }
    };

    // one group of 32 stream pixels: q0 = stream index of its first pixel, `limit` = pixels behind it are not written
    auto produce_group = [&](const Walk& w, int q0, int limit, int mlo, const uint32_t* fh) {
      // lanes 0-15 take the group's even stream pixels, lanes 16-31 the odd ones: a 16-lane group of a ring store then covers
      // consecutive slots of one de-interleaved half-row (8 chunk positions twice) instead of both halves (up to 4 lanes per
      // bank): 381 -> 377 us
      const int q = q0 + 2 * (pix & 15) + (pix >> 4);
      const bool valid = q < limit;
      const int m = valid ? q / C::IW : mlo;
      const int mx = valid ? q - m * C::IW : 0;
      const int my = m - mlo;
      // conv0 (3x3 s2 on the frame, K = 27 + bias slot): two k-steps gathered as aligned dwords.  k-slots: step 0 {h = 0: frame row 0
      // [junk, e0..e6], h = 1: row 2 [junk, e0..e6]}, step 1 {h = 0: row 1 [junk, e0..e6], h = 1: (row 0 e7 e8, row 1 e7 e8, row 2 e7 e8,
      // ONE, pad)}, e = 3 dx + c
      union { half8 v; uint32_t u[4]; } f0, f1, f0l, f1l;
      if constexpr (F32) {
        // element e = 3 dx + c of frame row r at patch (row 2 my + r, channel c, column 1 + 2 mx + dx); split like every plane value
        const char* pb32 = reinterpret_cast<const char*>(fh) + (2 * my) * C::F32_ROW + (1 + 2 * mx) * 4;
        auto elem = [&](int r, int e) { return *reinterpret_cast<const float*>(pb32 + r * C::F32_ROW + (e % 3) * C::F32_CH + (e / 3) * 4); };
        auto row8 = [&](int r, uint32_t (&uh)[4], uint32_t (&ul)[4]) {        // [junk, e0 .. e6]
          split2(0.f, elem(r, 0), uh[0], ul[0]);
          split2(elem(r, 1), elem(r, 2), uh[1], ul[1]);
          split2(elem(r, 3), elem(r, 4), uh[2], ul[2]);
          split2(elem(r, 5), elem(r, 6), uh[3], ul[3]);
        };
        if (h == 0) {
          row8(0, f0.u, f0l.u);
          row8(1, f1.u, f1l.u);
        } else {
          row8(2, f0.u, f0l.u);
#pragma unroll
          for (int r = 0; r < 3; ++r) split2(elem(r, 7), elem(r, 8), f1.u[r], f1l.u[r]);
          f1.u[3] = 0x3c00u; f1l.u[3] = 0u;
        }
      } else if constexpr (!U8) {
        const int base0 = (2 * my) * RD + 3 * mx + (C::JUNK - 1) / 2;
        const uint32_t* a0p = fh + base0 + (h ? 2 * RD : 0);
        f0.u[0] = a0p[0]; f0.u[1] = a0p[1]; f0.u[2] = a0p[2]; f0.u[3] = a0p[3];
        const uint32_t* b0 = fh + base0;
        f1.u[0] = b0[go0]; f1.u[1] = b0[go1]; f1.u[2] = b0[go2];
        const uint32_t last = b0[RD + 3];
        f1.u[3] = h ? 0x3c00u : last;
      } else {
        // bytes [j0, j0 + 10) of the three frame rows, j0 = 6 mx + U8_JB + 2 (the byte left of the first tap pixel's blue... junk), as
        // three aligned dwords per row + a byte shift of 0 or 2; each byte through the table -> (hi, lo)
        const int j0 = 6 * mx + C::U8_JB + 2;
        const unsigned sh = (unsigned)(j0 & 3);
        const char* rb = reinterpret_cast<const char*>(fh) + (2 * my) * C::U8_PITCH + (j0 & ~3);
        uint32_t d[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const uint32_t* pr = reinterpret_cast<const uint32_t*>(rb + r * C::U8_PITCH);
          d[r][0] = pr[0]; d[r][1] = pr[1]; d[r][2] = pr[2];
        }
        uint32_t w0[3], w1[3], t2[3];               // bytes 0-3, 4-7, 8-9 of [j0, j0 + 10) per row
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          w0[r] = __builtin_amdgcn_alignbyte(d[r][1], d[r][0], sh);
          w1[r] = __builtin_amdgcn_alignbyte(d[r][2], d[r][1], sh);
          t2[r] = (d[r][2] >> (8 * sh)) & 0xffffu;
        }
        // which taps lie inside the frame (the conv's zero padding is zero AFTER normalisation: table entry 256)
        const int gyf = 2 * (4 * (w.cy - w.j) - 1 + m) - 1, gxf = 2 * (2 * C::TW * w.tx - 1 + mx) - 1;   // frame row / column of tap (0, 0)
        bool rok[3], cok[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          rok[r] = gyf + r >= 0 && gyf + r < P.FH;
          cok[r] = gxf + r >= 0 && gxf + r < P.FW;
        }
        const uint32_t* lut = reinterpret_cast<const uint32_t*>(smem + C::LUT_OFF);
        auto look = [&](uint32_t word, int k, bool ok) {
          const uint32_t b = (word >> (8 * k)) & 0xffu;
          return lut[ok ? b : 256u];
        };
        auto row8 = [&](uint32_t lo4, uint32_t hi4, bool rowok, uint32_t (&uh)[4], uint32_t (&ul)[4]) {
          // [junk, e0 .. e6]: pixel of element e = e / 3
          const uint32_t v0 = look(lo4, 0, true), v1 = look(lo4, 1, rowok && cok[0]), v2 = look(lo4, 2, rowok && cok[0]), v3 = look(lo4, 3, rowok && cok[0]);
          const uint32_t v4 = look(hi4, 0, rowok && cok[1]), v5 = look(hi4, 1, rowok && cok[1]), v6 = look(hi4, 2, rowok && cok[1]), v7 = look(hi4, 3, rowok && cok[2]);
          uh[0] = __builtin_amdgcn_perm(v1, v0, 0x05040100u); ul[0] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
          uh[1] = __builtin_amdgcn_perm(v3, v2, 0x05040100u); ul[1] = __builtin_amdgcn_perm(v3, v2, 0x07060302u);
          uh[2] = __builtin_amdgcn_perm(v5, v4, 0x05040100u); ul[2] = __builtin_amdgcn_perm(v5, v4, 0x07060302u);
          uh[3] = __builtin_amdgcn_perm(v7, v6, 0x05040100u); ul[3] = __builtin_amdgcn_perm(v7, v6, 0x07060302u);
        };
        // step 0: row 0 (h = 0) | row 2 (h = 1)
        row8(h ? w0[2] : w0[0], h ? w1[2] : w1[0], h ? rok[2] : rok[0], f0.u, f0l.u);
        // step 1: row 1 (h = 0) | the e7 e8 pairs of the three rows + the constant one (byte 255 -> exactly 1.0, 0) + pad (h = 1)
        if (h == 0) {
          row8(w0[1], w1[1], rok[1], f1.u, f1l.u);
        } else {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const uint32_t va = look(t2[r], 0, rok[r] && cok[2]), vb = look(t2[r], 1, rok[r] && cok[2]);
            f1.u[r] = __builtin_amdgcn_perm(vb, va, 0x05040100u); f1l.u[r] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
          }
          f1.u[3] = 0x3c00u; f1l.u[3] = 0u;
        }
      }
      half8 xh[4], xl[4];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x16 am = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[ct][0], f0.v, zero16, 0, 0, 0);
        f32x16 ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[ct][0], f0.v, zero16, 0, 0, 0);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[ct][1], f1.v, am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l[ct][1], f1.v, ac, 0, 0, 0);
        if constexpr (U8 || F32) {   // (normalised bytes / fp32 pixels are not fp16 values: their low parts enter like any activation's)
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[ct][0], f0l.v, ac, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h[ct][1], f1l.v, ac, 0, 0, 0);
        }
        // (main, corr) -> ReLU -> the two k-step fragments of the 1x1 they form (its K runs in the accumulator layout's order)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          union { half8 v; uint32_t wd[4]; } vh, vl;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y0 = fmaxf(comb(am[8 * u + 2 * k], ac[8 * u + 2 * k]), 0.f);
            const float y1 = fmaxf(comb(am[8 * u + 2 * k + 1], ac[8 * u + 2 * k + 1]), 0.f);
            split2(y0, y1, vh.wd[k], vl.wd[k]);
          }
          xh[2 * ct + u] = vh.v; xl[2 * ct + u] = vl.v;
        }
      }
      // ring slot of the pixel (column-de-interleaved stride-2 row: even columns, then odd; chunk XOR swizzle)
      const int gy = 4 * (w.cy - w.j) - 1 + m, gx = 2 * C::TW * w.tx - 1 + mx;
      const float vmax = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? __builtin_inff() : 0.f;   // (zero outside the mid tensor: the 3x3's padding)
      const int rem = valid ? (mx & 1) * C::IWh + (mx >> 1) : C::IWs - 1;      // (not written: the slot no tap reads)
      const int fk = (rem >> 1) & 7;
      int rr = w.rc + (w.j ? 1 : 0) + my;
      if (rr >= C::RR) rr -= C::RR;
      char* dst = smem + C::RING_OFF + (rr * C::IWs + rem) * C::PIXB + 8 * h;
#pragma unroll
      for (int co = 0; co < 2; ++co) {
        f32x16 tm, tc;
        {
          const float* bp = pb + co * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
            tm[4 * g + 0] = b4.x; tm[4 * g + 1] = b4.y; tm[4 * g + 2] = b4.z; tm[4 * g + 3] = b4.w;
          }
        }
        tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2h[co][0], xh[0], tm, 0, 0, 0);
        tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2h[co][0], xl[0], zero16, 0, 0, 0);
        tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2l[co][0], xh[0], tc, 0, 0, 0);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2h[co][q], xh[q], tm, 0, 0, 0);
          tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2h[co][q], xl[q], tc, 0, 0, 0);
          tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2l[co][q], xh[q], tc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = __builtin_amdgcn_fmed3f(comb(tm[4 * g + e], tc[4 * g + e]), 0.f, vmax);
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = ((co * 4 + g) ^ fk) * 16;
          *reinterpret_cast<uint2*>(dst + o) = vh;
          *reinterpret_cast<uint2*>(dst + C::RING_PLANE + o) = vl;
        }
      }
    };

    Walk cur = first, nxt = first;
    walk_next(nxt, CY, a.tiles_x);
    patch_dma(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    block_barrier();
    for (int slot = 0; slot < nslots; ++slot) {
      SS_T(0, 0);
      if (slot < nchunks) {
        if (slot + 1 < nchunks) patch_dma(nxt, (slot + 1) & 1);
        const uint32_t* fh = reinterpret_cast<const uint32_t*>(F32 ? smem + ((slot & 1) ? C::TW_OFF : C::PATCH_OFF)
                                                                   : smem + C::PATCH_OFF + (slot & 1) * C::PATCH_BYTES);
        // groups of this chunk: the segment's first chunk = stream pixels [0, 165) in 6 groups (the last one's overhang is left
        // to the next slot); chunk j >= 1 = groups [G(j - 1), G(j)) behind pixel 165, G(j) = ceil(132 j / 32)
        const int j = cur.j;
        int q_first, ng, limit, mlo;
        if (j == 0) {
          q_first = 0; ng = 6; limit = C::PROLOGUE_PX; mlo = 0;
        } else {
          const int ga = (33 * (j - 1) + 7) >> 3, gb = (33 * j + 7) >> 3;
          q_first = C::PROLOGUE_PX + 32 * ga; ng = gb - ga; limit = 0x7fffffff; mlo = 4 * j + 1;
        }
        const int mine = (wave - j) & 3;        // (the wave that takes a fifth group rotates)
        produce_group(cur, q_first + 32 * mine, limit, mlo, fh);
        SS_T(0, 1);
        if (mine + 4 < ng) produce_group(cur, q_first + 32 * (mine + 4), limit, mlo, fh);
        cur = nxt;
        walk_next(nxt, CY, a.tiles_x);
      }
      SS_T(0, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      block_barrier();
      SS_T(0, 3);
    }
#ifdef LFD_PL_TIMING
    if (blockIdx.x == PL_DBG_BLOCK && threadIdx.x == 0) { g_pl_dbg[130] = __builtin_readcyclecounter(); g_pl_dbg[131] = __builtin_amdgcn_s_memrealtime(); g_pl_dbg[132] = nslots; }
#endif
  } else {
    // ========================================================= consumers =====================================================
    const int s = wave & 1, kh = (wave >> 1) & 1;
    const int oyl = pix >> 4, oxl = pix & 15;
    int xoff[3][4];
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int ix = oxl * 2 + st;
      const int rem = (ix & 1) * C::IWh + (ix >> 1);
      const int f = (rem >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) xoff[st][q] = rem * C::PIXB + (((2 * q + h) ^ f) * 16);
    }
    const half8* wsrc = a.w + ((size_t)s * 36) * 64 + lane;
    const char* ring = smem + C::RING_OFF;
    Walk cw = first;                            // the chunk contracted in this slot (slot - 1)

    if (kh == 1) {
      // ---------------- k-steps [KA, 36) of chunk slot - 1, partial sums to the partner; chained 1x1 + stores of chunk slot - 3
      half8 wh[KB], wl[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        wh[k] = wsrc[(size_t)(KA + k) * 64];
        wl[k] = wsrc[a.w_plane + (size_t)(KA + k) * 64];
      }
      const float* sb = reinterpret_cast<const float*>(smem + C::BIAS_OFF);
      const half8* tws = reinterpret_cast<const half8*>(smem + C::TW_OFF) + (s * 4) * 64 + lane;
      half8 twrh[F32 ? 4 : 1], twrl[F32 ? 4 : 1];      // fp32 frames: the chained 1x1's filters in registers (their LDS area is a patch buffer)
      if constexpr (F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          twrh[q] = a.w2[((size_t)s * 4 + q) * 64 + lane];
          twrl[q] = a.w2[a.w2_plane + ((size_t)s * 4 + q) * 64 + lane];
        }
      }
      char* const stg = smem + C::STG_OFF + s * C::STG_WAVE;
      const int fm = (pix >> 1) & 7;
      int p1_n = 0, p1_tx = 0, p1_cy = 0, p2_n = 0, p2_tx = 0, p2_cy = 0;     // chunks slot - 2, slot - 3
      block_barrier();
      for (int slot = 0; slot < nslots; ++slot) {
        if (slot >= 2) walk_next(cw, CY, a.tiles_x);
        const bool do_tail = slot >= 3;
        int rowoff[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          int rr = cw.rc + 2 * oyl + r;
          if (rr >= C::RR) rr -= C::RR;
          rowoff[r] = rr * C::ROWB;
        }
        auto xaddr = [&](int k) { return ring + xoff[(k / 4) % 3][k % 4] + rowoff[k / 12]; };
        const char* const midr = smem + C::MID_OFF + ((slot + 1) & 1) * C::MID_BUF;
        auto tail_stage = [&](const f32x16& tm, const f32x16& tc, int g) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = comb(tm[4 * g + e], tc[4 * g + e]);
            if (a.relu2) y[e] = fmaxf(y[e], 0.f);
          }
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = pix * C::STG_PITCH + 16 * g + 8 * h;
          *reinterpret_cast<uint2*>(stg + o) = vh;
          *reinterpret_cast<uint2*>(stg + C::STG_PLANE + o) = vl;
        };
        auto epi_read = [&](int r, uint4& cvh, uint4& cvl) {
          const int i = lane + 64 * r;
          const int o = (i >> 2) * C::STG_PITCH + (i & 3) * 16;
          cvh = *reinterpret_cast<const uint4*>(stg + o);
          cvl = *reinterpret_cast<const uint4*>(stg + C::STG_PLANE + o);
        };
        auto epi_store = [&](int r, const uint4& cvh, const uint4& cvl) {
          _Float16* trash = const_cast<_Float16*>(a.zeros) + 1024 + (threadIdx.x & 127) * 8;
          const int i = lane + 64 * r;
          const int px = i >> 2, c4 = i & 3;
          const int oy = 2 * p2_cy + (px >> 4), ox = C::TW * p2_tx + (px & 15);
          const bool ok = oy < a.OH && ox < a.OW;
          _Float16* dst = a.out + (((size_t)p2_n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * 64 + s * 32 + c4 * 8;
          *reinterpret_cast<uint4*>(ok ? dst : trash) = cvh;
          *reinterpret_cast<uint4*>(ok ? dst + a.out_plane : trash) = cvl;
        };
        SS_T(6, 8);
        // ---- the chained 1x1 of chunk slot - 3 (its operand planes were written one slot ago), staged before the contraction
        //      starts (its 32 accumulator registers are free again; the producer wave of this SIMD runs under it)
        if (do_tail) {
          f32x16 tm, tc;
          const float* bp = sb + 128 + s * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
            tm[4 * g + 0] = b4.x; tm[4 * g + 1] = b4.y; tm[4 * g + 2] = b4.z; tm[4 * g + 3] = b4.w;
          }
          tc = zero16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = pix * 128 + (((2 * q + h) ^ fm) * 16);
            const half8 xh = *reinterpret_cast<const half8*>(midr + o);
            const half8 xl = *reinterpret_cast<const half8*>(midr + C::MID_PLANE + o);
            half8 twh, twl;
            if constexpr (F32) { twh = twrh[q]; twl = twrl[q]; } else { twh = tws[q * 64]; twl = tws[2 * 4 * 64 + q * 64]; }
            tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(twh, xh, tm, 0, 0, 0);
            tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(twh, xl, tc, 0, 0, 0);
            tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(twl, xh, tc, 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) tail_stage(tm, tc, g);
        }
        SS_T(6, 9);
        f32x16 bm = zero16, bc = zero16;
        half8 xqh[PD + 1], xql[PD + 1];
        uint4 cvh, cvl;
#pragma unroll
        for (int k = 0; k < PD; ++k) {
          const char* p = xaddr(KA + k);
          xqh[k] = *reinterpret_cast<const half8*>(p);
          xql[k] = *reinterpret_cast<const half8*>(p + C::RING_PLANE);
        }
        static_assert(KB >= 9, "the store pieces sit in front of k-steps 2 .. 8 of the kh = 1 role");
        static_for([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (k == 2) { if (do_tail) epi_read(0, cvh, cvl); }
          if constexpr (k == 4) { if (do_tail) epi_store(0, cvh, cvl); }
          if constexpr (k == 6) { if (do_tail) epi_read(1, cvh, cvl); }
          if constexpr (k == 8) { if (do_tail) epi_store(1, cvh, cvl); }
          if constexpr (k + PD < KB) {
            const char* p = xaddr(KA + k + PD);
            xqh[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p);
            xql[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p + C::RING_PLANE);
          }
          PL_SB();
          bm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)], bm, 0, 0, 0);
          bc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)], bc, 0, 0, 0);
          bc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)], bc, 0, 0, 0);
          PL_SB();
        }, std::make_integer_sequence<int, KB>{});
        SS_T(6, 10);
        {
          float* xw = reinterpret_cast<float*>(smem + C::XCH_OFF + (slot & 1) * C::XCH_BUF + s * C::XCH_SLAB) + lane * 4;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = comb(bm[4 * g + 0], bc[4 * g + 0]);
            v.y = comb(bm[4 * g + 1], bc[4 * g + 1]);
            v.z = comb(bm[4 * g + 2], bc[4 * g + 2]);
            v.w = comb(bm[4 * g + 3], bc[4 * g + 3]);
            *reinterpret_cast<float4*>(xw + g * 256) = v;
          }
        }
        p2_n = p1_n; p2_tx = p1_tx; p2_cy = p1_cy;
        p1_n = cw.n; p1_tx = cw.tx; p1_cy = cw.cy;
        block_barrier();
        SS_T(6, 11);
      }
    } else {
      // ---------------- k-steps [0, KA) of chunk slot - 1; both halves' partial sums -> operand planes of chunk slot - 2
      half8 wh[KA], wl[KA];
#pragma unroll
      for (int k = 0; k < KA; ++k) {
        wh[k] = wsrc[(size_t)k * 64];
        wl[k] = wsrc[a.w_plane + (size_t)k * 64];
      }
      const float* sb = reinterpret_cast<const float*>(smem + C::BIAS_OFF);
      const int fm = (pix >> 1) & 7;
      float yp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) yp[r] = 0.f;

      block_barrier();
      for (int slot = 0; slot < nslots; ++slot) {
        if (slot >= 2) walk_next(cw, CY, a.tiles_x);
        const bool do_mid = slot >= 2 && slot <= nchunks + 1;
        int rowoff[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          int rr = cw.rc + 2 * oyl + r;
          if (rr >= C::RR) rr -= C::RR;
          rowoff[r] = rr * C::ROWB;
        }
        auto xaddr = [&](int k) { return ring + xoff[(k / 4) % 3][k % 4] + rowoff[k / 12]; };
        const float* xr = reinterpret_cast<const float*>(smem + C::XCH_OFF + ((slot + 1) & 1) * C::XCH_BUF + s * C::XCH_SLAB) + lane * 4;
        char* const midw = smem + C::MID_OFF + (slot & 1) * C::MID_BUF;
        auto mid_piece = [&](int g) {
          // conv + bias (this wave's half) + the partner's half -> ReLU -> planes: 32 channels of the chunk's 32 pixels
          const float4 pa = *reinterpret_cast<const float4*>(xr + g * 256);
          float y[4];
          y[0] = yp[4 * g + 0] + pa.x; y[1] = yp[4 * g + 1] + pa.y; y[2] = yp[4 * g + 2] + pa.z; y[3] = yp[4 * g + 3] + pa.w;
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
          }
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = pix * 128 + (((s * 4 + g) ^ fm) * 16) + 8 * h;
          *reinterpret_cast<uint2*>(midw + o) = vh;
          *reinterpret_cast<uint2*>(midw + C::MID_PLANE + o) = vl;
        };
        SS_T(4, 4);
        f32x16 am, ac;
        {
          const float* bp = sb + 64 + s * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
            am[4 * g + 0] = b4.x; am[4 * g + 1] = b4.y; am[4 * g + 2] = b4.z; am[4 * g + 3] = b4.w;
          }
          ac = zero16;
        }
        half8 xqh[PD + 1], xql[PD + 1];
#pragma unroll
        for (int k = 0; k < PD; ++k) {
          const char* p = xaddr(k);
          xqh[k] = *reinterpret_cast<const half8*>(p);
          xql[k] = *reinterpret_cast<const half8*>(p + C::RING_PLANE);
        }
        static_assert(KA >= 9, "the pieces sit in front of k-steps 2 .. 8 of the kh = 0 role");
        static_for([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (k >= 2 && k <= 8 && k % 2 == 0) { if (do_mid) mid_piece(k / 2 - 1); }
          if constexpr (k + PD < KA) {
            const char* p = xaddr(k + PD);
            xqh[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p);
            xql[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p + C::RING_PLANE);
          }
          PL_SB();
          am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)], am, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)], ac, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)], ac, 0, 0, 0);
          PL_SB();
        }, std::make_integer_sequence<int, KA>{});
        SS_T(4, 6);
#pragma unroll
        for (int r = 0; r < 16; ++r) yp[r] = comb(am[r], ac[r]);
        block_barrier();
        SS_T(4, 7);
      }
    }
  }
}

}  // namespace pl

#ifdef LFD_PL_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_pl_stem2xs_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pl::g_pl_dbg), sizeof(unsigned long long) * 136);
}
#endif

template <int FMT>
static int launch_stem2xs(pl::PlArgs a, const pl::PlProd& p, hipStream_t st) {
  using C = pl::SS;
  a.tiles_x = (a.OW + C::TW - 1) / C::TW;
  a.tiles_y = (a.OH + 1) / 2;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  auto kern = pl::k_pl_stem2xs<PL_SS_KA, FMT>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  int blocks = 256;
  if (blocks > a.ntiles) blocks = a.ntiles;
  hipLaunchKernelGGL(kern, dim3(blocks, 1), dim3(512), C::LDS_BYTES, st, a, p, a.tiles_y, a.ntiles);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// in_format (lfd_stem_conv_f16's codes): fp16 NHWC frames with w % 8 == 0, uint8 NHWC frames with w % 16 == 0, fp32 NCHW frames with
// w % 4 == 0 -- 16-byte aligned bases; the caller (lfd_pl_stem2x) checks
int lfd_pl_stem2xs_launch(pl::PlArgs a, const pl::PlProd& p, int in_format, hipStream_t st) {
  switch (in_format) {
    case pl::IN_NHWC_F16: return launch_stem2xs<pl::IN_NHWC_F16>(a, p, st);
    case pl::IN_NHWC_U8: return launch_stem2xs<pl::IN_NHWC_U8>(a, p, st);
    case pl::IN_NCHW_F32: return launch_stem2xs<pl::IN_NCHW_F32>(a, p, st);
    default: return LFD_ERR_INVALID_ARGUMENT;
  }
}
