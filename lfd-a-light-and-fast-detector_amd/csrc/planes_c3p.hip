// csrc/planes_c3p.hip -- k_pl_c3p, the 3x3 stride-1 64-channel conv (+ residual) of the planes mode (lfd_resnet.py:96-154)
// with TWO waves per SIMD (round 6).
//
// k_pl_c3 (planes_c3.hip) keeps the hi AND lo filter slab of a wave's 32 output channels in registers -- 288 of 512 -- so one
// wave per SIMD issues everything: its MFMAs, the LDS-DMA of the next tile, the split / staging / stores of the previous one.
// The matrix pipe was busy 41-43 % of the time at 2.45 GHz (profiles/r05_precise_pmc_sq_counters.txt): not power, not HBM --
// issue slots of a lone wave (LESSONS 46: a VMEM instruction stalls its wave 60-100 cycles wherever it stands).
// Here the contraction index is split over a wave PAIR that shares a SIMD (waves w and w + 4 of a 512-thread workgroup):
//   role A (waves 0-3, dispatched first): k-steps [0, KA) + the residual k-steps; issues every LDS-DMA of the NEXT tile
//          between its MFMAs; hands main + 2^-11 corr of its part (16 floats per lane) to its partner through LDS;
//   role B (waves 4-7): k-steps [KA, 36); adds A's part of the PREVIOUS tile (one barrier per tile orders the exchange,
//          two exchange buffers), ReLU, splits into planes, stages in a wave-private LDS slab and stores -- as pieces between
//          its MFMAs.
// 144 weight registers per wave instead of 288: both fit the 256 a wave may hold at two waves per SIMD, and whatever one wave
// issues besides MFMAs, the other one's MFMAs cover.  Same tile (4 x 16 output pixels x 64 channels per workgroup), same LDS
// image of the input tile, same packed filters as k_pl_c3.  The sum order differs from k_pl_c3's (two partial sums per
// element instead of one chain): equal to fp32 rounding, not bit for bit.
#include "planes_impl.h"

namespace pl {

template <bool RES>
struct C3P {
  static constexpr int TW = 16, TH = 4, IH = 6, IW = 18, IWs = 18, PIXB = 128;
  static constexpr int NSLOT = IH * IWs;
  static constexpr int IN_BYTES = ((NSLOT * PIXB + 1023) / 1024) * 1024;      // one plane of one buffer
  static constexpr int NK = 36;
  static constexpr int RES_PLANE = 64 * 128;
  static constexpr int RES_OFF = 4 * IN_BYTES;
  static constexpr int STG_OFF = RES_OFF + (RES ? 4 * RES_PLANE : 0);
  static constexpr int STG_PITCH = 80, STG_PLANE = 32 * STG_PITCH, STG_WAVE = 2 * STG_PLANE;
  static constexpr int XCH_OFF = STG_OFF + 4 * STG_WAVE;
  static constexpr int XCH_PAIR = 64 * 16 * 4, XCH_BUF = 4 * XCH_PAIR;      // [pair][4 chunks][64 lanes] x 16 B
  static constexpr int BIAS_OFF = XCH_OFF + 2 * XCH_BUF;
  static constexpr int LDS_BYTES = BIAS_OFF + 64 * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS capacity");
};

// depth of the operand ring (k-steps requested ahead of their MFMAs).  The A role of the residual variant sits at the
// 256-register limit: one step less there (a second wave on the SIMD covers the LDS latency the ring was sized for)
#ifndef PL_C3P_PDA
#define PL_C3P_PDA (RES ? 2 : 3)
#endif
#ifndef PL_C3P_PDB
#define PL_C3P_PDB 3
#endif

#ifdef LFD_PL_TIMING
#define C3P_T(i) do { if (blockIdx.x == PL_DBG_BLOCK && blockIdx.y == 0 && (threadIdx.x & 255) == 0 && it < 8) g_pl_dbg[it * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define C3P_T(i)
#endif

template <bool RES, int KA>
__global__ __launch_bounds__(512, 1) void k_pl_c3p(PlArgs a) {
  using C = C3P<RES>;
  constexpr int KB = C::NK - KA;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wave >> 2, w4 = wave & 3;
  const int ct = w4 & 1, pg = w4 >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const int oyl = pix >> 4, oxl = pix & 15;
  const int cog = blockIdx.y;                       // 64 output channels per workgroup

  float* sbias = reinterpret_cast<float*>(smem + C::BIAS_OFF);
  if (threadIdx.x < 64) sbias[threadIdx.x] = a.bias[cog * 64 + threadIdx.x];
#ifdef LFD_PL_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_pl_dbg[128] = __builtin_readcyclecounter(); g_pl_dbg[129] = __builtin_amdgcn_s_memrealtime(); }
#endif

  int xoff[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = oxl + s;
    const int f = (ix >> 1) & 7;
    const int rowbase = (pg * 2 + oyl) * C::IWs + ix;
#pragma unroll
    for (int q = 0; q < 4; ++q) xoff[s][q] = rowbase * C::PIXB + (((2 * q + h) ^ f) * 16);
  }
  auto xaddr = [&](const char* xb, int k) {
    const int r = k / 12, s = (k / 4) % 3, q = k % 4;
    return xb + xoff[s][q] + r * C::IWs * C::PIXB;
  };

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const half8* wsrc = a.w + ((size_t)(cog * 2 + ct) * C::NK) * 64 + lane;
  int t = t_begin + bix;
  int buf = 0, it = 0;

  if (role == 0) {
    // =================================================== role A ===================================================
#ifdef PL_C3P_PRIO_A
    __builtin_amdgcn_s_setprio(PL_C3P_PRIO_A);
#endif
    constexpr int PD = PL_C3P_PDA;
    half8 wh[KA], wl[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      wh[k] = wsrc[(size_t)k * 64];
      wl[k] = wsrc[a.w_plane + (size_t)k * 64];
    }
    // identity fragments of the residual "tap": slab ct reads channels 32 ct .. 32 ct + 31 of the identity tile = 16-channel
    // groups 2 ct + qq; lane (h, co = pix) of fragment qq holds a one at j = co - 16 qq - 8 h.  Formed where they are used
    // (8 registers that would otherwise live across the whole walk: the A role of the residual variant is the one at the
    // 256-register limit)
    // (per-lane constants that differ by a constant are kept ONCE and re-derived at the use, behind an empty asm the
    //  optimiser cannot hoist across: roff[1] = roff[0] ^ 32, vo_main1 = (vo_main0 ^ 64) + 1024, vo_res[1] = vo_res[0] + two rows)
    int roff0 = 0;
    if constexpr (RES) {
      const int slot = 32 * pg + pix;
      roff0 = slot * 128 + (((2 * (2 * ct) + h) ^ ((slot >> 1) & 7)) * 16);
    }
    auto identity_fragment = [&](int qq) {
      int j0 = pix - 16 * qq - 8 * h;
      asm volatile("" : "+v"(j0));              // (not hoisted out of the tile loop)
      union { half8 v; uint32_t u[4]; } f;
#pragma unroll
      for (int r = 0; r < 4; ++r) f.u[r] = (j0 == 2 * r) ? 0x3c00u : ((j0 == 2 * r + 1) ? 0x3c000000u : 0u);
      return f.v;
    };
    const long in_plane_b = a.in_plane * 2;
    const long rowpitch = (long)a.W * 128;
    auto dma2 = [&](const char* src, bool valid, const char* zsrc, char* ldst, long plane_b, int lds_plane) {
      dma16(valid ? src : zsrc, ldst);
      dma16(valid ? src + plane_b : zsrc, ldst + lds_plane);
    };
    // ---- DMA pieces of the tile being fetched (7 per A wave): 3 x 8-pixel halves of halo rows, 2 x the two right-most
    //      columns, 2 x the identity tile (RES) -- k_pl_c3's, the four A waves in the place of its four waves
    int d_n = 0, d_ty0 = 0, d_tx0 = 0, d_buf = 0;
    bool d_interior = false;
    auto dma_setup = [&](int tt, int b) {
      d_n = tt / tiles_per_img;
      const int tr = tt - d_n * tiles_per_img;
      d_ty0 = tr / a.tiles_x;
      d_tx0 = tr - d_ty0 * a.tiles_x;
      d_buf = b;
      d_interior = d_ty0 > 0 && d_tx0 > 0 && (d_ty0 + 1) * C::TH + 1 <= a.H && (d_tx0 + 1) * C::TW + 1 <= a.W;
    };
    const unsigned vo_main0 = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 4)) * 16);
    unsigned vo_res0 = 0;
    if constexpr (RES) {
      const int slot = 8 * w4 + (lane >> 3);
      vo_res0 = (unsigned)(((slot >> 4) * a.OW + (slot & 15)) * a.cout * 2 + ((lane & 7) ^ ((slot >> 1) & 7)) * 16);
    }
    const unsigned res_two_rows = (unsigned)(2 * a.OW * a.cout * 2);
    auto launder = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
    auto dma_piece = [&](int p) {
      const int gy0 = d_ty0 * C::TH - 1, gx0 = d_tx0 * C::TW - 1;
      char* lbase = smem + d_buf * 2 * C::IN_BYTES;
      const char* p00 = reinterpret_cast<const char*>(a.in) + ((long)d_n * a.H + gy0) * rowpitch + (long)gx0 * 128;
      if (d_interior) {
        if (p < 3) {
          const int m = w4 + 4 * p;
          const int iy = m >> 1, hf = m & 1;
          const char* rb = p00 + iy * rowpitch;
          char* ld = lbase + (iy * C::IWs + 8 * hf) * C::PIXB;
          const unsigned vo = hf ? (launder(vo_main0) ^ 64u) + 1024u : vo_main0;
          dma16s(rb, vo, ld);
          dma16s(rb + in_plane_b, vo, ld + C::IN_BYTES);
        } else if (p < 5) {
          const int iy = w4 + 4 * (p - 3);
          if (iy < C::IH && lane < 16) {
            const char* rb = p00 + iy * rowpitch;
            char* ld = lbase + (iy * C::IWs + 16) * C::PIXB;
            const unsigned vo_small = 2048u + launder((unsigned)lane) * 16u;
            dma16s(rb, vo_small, ld);
            dma16s(rb + in_plane_b, vo_small, ld + C::IN_BYTES);
          }
        } else if constexpr (RES) {
          const int r = w4 + 4 * (p - 5);
          const char* rb = reinterpret_cast<const char*>(a.res + (((size_t)d_n * a.OH + d_ty0 * C::TH) * a.OW + d_tx0 * C::TW) * a.cout + cog * 64);
          char* ld = smem + C::RES_OFF + d_buf * 2 * C::RES_PLANE + r * 1024;
          const unsigned vo = p == 5 ? vo_res0 : launder(vo_res0) + res_two_rows;
          dma16s(rb, vo, ld);
          dma16s(rb + a.res_plane * 2, vo, ld + C::RES_PLANE);
        }
        return;
      }
      if (p < 3) {
        const int lpx = lane >> 3;
        const int m = w4 + 4 * p;
        const int iy = m >> 1, hf = m & 1;
        const int cc = (((lane & 7) ^ (lpx >> 1)) * 16) ^ (hf ? 64 : 0);
        const int gy = gy0 + iy, gx = gx0 + 8 * hf + lpx;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        dma2(p00 + iy * rowpitch + hf * 1024 + lpx * 128 + cc, ok, reinterpret_cast<const char*>(a.zeros) + cc,
             lbase + (iy * C::IWs + 8 * hf) * C::PIXB, in_plane_b, C::IN_BYTES);
      } else if (p < 5) {
        const int iy = w4 + 4 * (p - 3);
        if (iy < C::IH && lane < 16) {
          const int gy = gy0 + iy, gx = gx0 + 16 + (lane >> 3);
          const bool ok = gy >= 0 && gy < a.H && gx < a.W;
          dma2(p00 + iy * rowpitch + 2048 + lane * 16, ok, reinterpret_cast<const char*>(a.zeros) + (lane & 7) * 16,
               lbase + (iy * C::IWs + 16) * C::PIXB, in_plane_b, C::IN_BYTES);
        }
      } else if constexpr (RES) {
        const int r = w4 + 4 * (p - 5);               // 8 instructions of 8 pixels x 8 chunks per plane
        const int slot = 8 * r + (lane >> 3);
        const int c = (lane & 7) ^ ((slot >> 1) & 7);
        const int oy = d_ty0 * C::TH + (slot >> 4), ox = d_tx0 * C::TW + (slot & 15);
        const bool ok = oy < a.OH && ox < a.OW;
        const _Float16* src = a.res + (((size_t)d_n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.cout + cog * 64 + c * 8;
        dma2(reinterpret_cast<const char*>(src), ok, reinterpret_cast<const char*>(a.zeros) + c * 16,
             smem + C::RES_OFF + d_buf * 2 * C::RES_PLANE + r * 1024, a.res_plane * 2, C::RES_PLANE);
      }
    };
    constexpr int NPIECE = RES ? 7 : 5;
    static_assert(2 * NPIECE <= KA, "the DMA pieces sit in front of the odd k-steps of role A");

    if (t < t_end) {
      dma_setup(t, 0);
#pragma unroll
      for (int p = 0; p < NPIECE; ++p) dma_piece(p);
    }
    while (t < t_end) {
      // every VMEM operation of an A wave is a DMA of the tile it is about to contract
      C3P_T(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      C3P_T(1);
      block_barrier();
      C3P_T(2);
      const bool has_next = t + t_step < t_end;
      if (has_next) dma_setup(t + t_step, buf ^ 1);
      const char* xb = smem + buf * 2 * C::IN_BYTES;
      f32x16 am, ac;
      {
        const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          am[4 * g + 0] = b4.x; am[4 * g + 1] = b4.y; am[4 * g + 2] = b4.z; am[4 * g + 3] = b4.w;
          ac[4 * g + 0] = 0.f; ac[4 * g + 1] = 0.f; ac[4 * g + 2] = 0.f; ac[4 * g + 3] = 0.f;
        }
      }
      half8 xqh[PD + 1], xql[PD + 1];
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        const char* p = xaddr(xb, k);
        xqh[k] = *reinterpret_cast<const half8*>(p);
        xql[k] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
      }
      static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k % 2 == 1 && (k - 1) / 2 < NPIECE) {
          if (has_next) dma_piece((k - 1) / 2);
        }
        if constexpr (k + PD < KA) {
          const char* p = xaddr(xb, k + PD);
          xqh[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p);
          xql[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
        }
        PL_SB();
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)], am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)], ac, 0, 0, 0);
        PL_SB();
      }, std::make_integer_sequence<int, KA>{});
      if constexpr (RES) {
        // the residual is a k-step: 1.0 x hi into the main set, 1.0 x lo into the correction set (exact products)
        const char* rb = smem + C::RES_OFF + buf * 2 * C::RES_PLANE;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int ro = qq ? (int)(launder((unsigned)roff0) ^ 32u) : roff0;
          const half8 rh = *reinterpret_cast<const half8*>(rb + ro);
          const half8 rl = *reinterpret_cast<const half8*>(rb + C::RES_PLANE + ro);
          const half8 idf = identity_fragment(qq);
          am = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf, rh, am, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf, rl, ac, 0, 0, 0);
        }
      }
      C3P_T(3);
      // hand-off: this wave's part of the tile (bias + k-steps [0, KA) + identity), lane-linear 16-byte chunks
      {
        float* xw = reinterpret_cast<float*>(smem + C::XCH_OFF + (it & 1) * C::XCH_BUF + w4 * C::XCH_PAIR) + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v;
          v.x = comb(am[4 * g + 0], ac[4 * g + 0]);
          v.y = comb(am[4 * g + 1], ac[4 * g + 1]);
          v.z = comb(am[4 * g + 2], ac[4 * g + 2]);
          v.w = comb(am[4 * g + 3], ac[4 * g + 3]);
          *reinterpret_cast<float4*>(xw + g * 256) = v;
        }
      }
      C3P_T(4);
      t += t_step; buf ^= 1; ++it;
    }
    block_barrier();      // (the partner's last epilogue reads the last hand-off behind this one)
  } else {
    // =================================================== role B ===================================================
#ifdef PL_C3P_PRIO_B
    __builtin_amdgcn_s_setprio(PL_C3P_PRIO_B);
#endif
    constexpr int PD = PL_C3P_PDB;
    half8 wh[KB], wl[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      wh[k] = wsrc[(size_t)(KA + k) * 64];
      wl[k] = wsrc[a.w_plane + (size_t)(KA + k) * 64];
    }
    char* const stg = smem + C::STG_OFF + w4 * C::STG_WAVE;
    float yp[16];                                 // this wave's part of the previous tile: main + 2^-11 corr
#pragma unroll
    for (int r = 0; r < 16; ++r) yp[r] = 0.f;
    int e_n = 0, e_ty0 = 1 << 24, e_tx0 = 0;      // (no previous tile yet: rows far outside the image -> the trash line)
    const float* xr = reinterpret_cast<const float*>(smem + C::XCH_OFF + w4 * C::XCH_PAIR) + lane * 4;
    // ---- epilogue pieces of the PREVIOUS tile
    auto epi_stage = [&](int g, int xbuf) {
      const float4 pa = *reinterpret_cast<const float4*>(xr + xbuf * (C::XCH_BUF / 4) + g * 256);
      float y[4];
      y[0] = yp[4 * g + 0] + pa.x; y[1] = yp[4 * g + 1] + pa.y; y[2] = yp[4 * g + 2] + pa.z; y[3] = yp[4 * g + 3] + pa.w;
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
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
      const int oy = e_ty0 * C::TH + 2 * pg + (px >> 4), ox = e_tx0 * C::TW + (px & 15);
      const bool ok = oy < a.OH && ox < a.OW;
      _Float16* dst = a.out + (((size_t)e_n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.cout + cog * 64 + ct * 32 + c4 * 8;
      *reinterpret_cast<uint4*>(ok ? dst : trash) = cvh;
      *reinterpret_cast<uint4*>(ok ? dst + a.out_plane : trash) = cvl;
    };
    static_assert(KB >= 15, "the epilogue pieces sit in front of k-steps 1 .. 14 of role B");

    while (t < t_end) {
      C3P_T(8);
      block_barrier();
      C3P_T(9);
      const int n = t / tiles_per_img;
      const int tr = t - n * tiles_per_img;
      const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
      const char* xb = smem + buf * 2 * C::IN_BYTES;
      const int xprev = (it + 1) & 1;
      f32x16 bm, bc;
#pragma unroll
      for (int r = 0; r < 16; ++r) bm[r] = bc[r] = 0.f;
      half8 xqh[PD + 1], xql[PD + 1];
      uint4 cvh, cvl;
#pragma unroll
      for (int k = 0; k < PD; ++k) {
        const char* p = xaddr(xb, KA + k);
        xqh[k] = *reinterpret_cast<const half8*>(p);
        xql[k] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
      }
      static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#ifndef PL_C3_NOPIECES
        if constexpr (k >= 1 && k <= 4) epi_stage(k - 1, xprev);
        if constexpr (k == 8) epi_read(0, cvh, cvl);
        if constexpr (k == 10) epi_store(0, cvh, cvl);
        if constexpr (k == 12) epi_read(1, cvh, cvl);
        if constexpr (k == 14) epi_store(1, cvh, cvl);
#endif
        if constexpr (k + PD < KB) {
          const char* p = xaddr(xb, KA + k + PD);
          xqh[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p);
          xql[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
        }
        PL_SB();
        bm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)], bm, 0, 0, 0);
        bc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)], bc, 0, 0, 0);
        bc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)], bc, 0, 0, 0);
        PL_SB();
      }, std::make_integer_sequence<int, KB>{});
      C3P_T(10);
#pragma unroll
      for (int r = 0; r < 16; ++r) yp[r] = comb(bm[r], bc[r]);
      e_n = n; e_ty0 = ty0; e_tx0 = tx0;
      t += t_step; buf ^= 1; ++it;
    }
    block_barrier();
    if (it > 0) {
      // the last tile's epilogue has no contraction to hide under
      const int xprev = (it + 1) & 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) epi_stage(g, xprev);
      uint4 cvh, cvl;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        epi_read(r, cvh, cvl);
        epi_store(r, cvh, cvl);
      }
    }
  }
#ifdef LFD_PL_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_pl_dbg[130] = __builtin_readcyclecounter(); g_pl_dbg[131] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

#ifndef PL_C3P_KA
#define PL_C3P_KA 18
#endif

template <bool RES>
int launch_pl_c3p(const PlArgs& a0, hipStream_t st) {
  using C = C3P<RES>;
  PlArgs a = a0;
  a.tiles_x = (a.OW + C::TW - 1) / C::TW;
  a.tiles_y = (a.OH + C::TH - 1) / C::TH;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  const int cgroups = a.cout / 64;
  auto kern = k_pl_c3p<RES, PL_C3P_KA>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  int blocks = 256 / cgroups;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(blocks, cgroups), dim3(512), C::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // namespace pl

#ifdef LFD_PL_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_pl_c3p_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pl::g_pl_dbg), sizeof(unsigned long long) * 136);
}
#endif

int lfd_pl_c3p_launch(const pl::PlArgs& a, bool residual, hipStream_t st) {
  return residual ? pl::launch_pl_c3p<true>(a, st) : pl::launch_pl_c3p<false>(a, st);
}
