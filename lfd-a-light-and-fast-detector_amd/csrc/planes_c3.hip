// csrc/planes_c3.hip -- k_pl_c3, the pipelined-epilogue 3x3 stride-1 64-channel conv (+ residual) of the planes mode
// (lfd_resnet.py:96-154), in its own translation unit: planes.hip is compiled with the MFMA results in VGPRs (its serial
// epilogues read them with VALU right away); here the epilogue pieces run in the shadow of the next tile's MFMAs, the two
// accumulator sets belong in AccVGPRs and the 256 architectural VGPRs to the operand ring and the pieces.
#include "planes_impl.h"

namespace pl {

// ------------------------------------------------------------------------------------------------------------------------
// k_pl_c3: the 3x3 stride-1 64-channel conv (+ residual) of the residual blocks (lfd_resnet.py:96-154) -- 13 of the 18 launches
// of a WIDERFACE_LFD_S backbone -- with the WHOLE epilogue of a tile hidden under the next tile's contraction.
// pl_block runs one wave per SIMD (288 weight registers), so whatever a wave issues between two contractions is exposed:
// measured per 8 x 16 tile (tools/timing/pl_phases.py) 8.8 k cycles of contraction + 3.5 k (plain) / 6.3 k (residual) of
// accumulator -> planes -> LDS -> HBM.  Here a wave owns ONE 32-pixel MFMA tile per 4 x 16 workgroup tile and TWO accumulator
// sets: while the matrix pipe works on tile t into one set, the same wave splits tile t-1's set into planes, stages it in a
// wave-private LDS slab (no barrier: nothing crosses waves) and stores it, as "pieces" placed between the MFMAs of the unrolled
// k loop, next to the DMA pieces of tile t+1.  The residual is not an epilogue term at all: the identity tile arrives by DMA in
// operand layout and is ADDED BY THE MATRIX PIPE as two extra k-steps against an identity fragment (1.0 x hi into the main set,
// 1.0 x lo into the correction set: exact products, 4 MFMAs per wave and tile instead of ~220 VALU instructions).
template <bool RES>
struct C3 {
  static constexpr int TW = 16, TH = 4, IH = 6, IW = 18, IWs = 18, PIXB = 128;
  static constexpr int NSLOT = IH * IWs;
  static constexpr int IN_BYTES = ((NSLOT * PIXB + 1023) / 1024) * 1024;      // one plane of one buffer
  static constexpr int NK = 36;
  static constexpr int RES_PLANE = 64 * 128;
  static constexpr int RES_OFF = 4 * IN_BYTES;
  static constexpr int STG_OFF = RES_OFF + (RES ? 4 * RES_PLANE : 0);
  static constexpr int STG_PITCH = 80, STG_PLANE = 32 * STG_PITCH, STG_WAVE = 2 * STG_PLANE;
  static constexpr int BIAS_OFF = STG_OFF + 4 * STG_WAVE;
  static constexpr int LDS_BYTES = BIAS_OFF + 64 * 4;
};

template <bool RES>
__global__ __launch_bounds__(256, 1) void k_pl_c3(PlArgs a) {
  using C = C3<RES>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ct = wave & 1, pg = wave >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const int oyl = pix >> 4, oxl = pix & 15;
  const int cog = blockIdx.y;                       // 64 output channels per workgroup

  float* sbias = reinterpret_cast<float*>(smem + C::BIAS_OFF);
  if (threadIdx.x < 64) sbias[threadIdx.x] = a.bias[cog * 64 + threadIdx.x];
#ifdef LFD_PL_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_pl_dbg[128] = __builtin_readcyclecounter(); g_pl_dbg[129] = __builtin_amdgcn_s_memrealtime(); }
#endif

  half8 wh[C::NK], wl[C::NK];
  {
    const half8* wsrc = a.w + ((size_t)(cog * 2 + ct) * C::NK) * 64 + lane;
#pragma unroll
    for (int k = 0; k < C::NK; ++k) {
      wh[k] = wsrc[(size_t)k * 64];
      wl[k] = wsrc[a.w_plane + (size_t)k * 64];
    }
  }
  // identity fragments of the residual "tap": slab ct reads channels 32 ct .. 32 ct + 31 of the identity tile = 16-channel
  // groups 2 ct + qq; lane (h, co = pix) of fragment qq holds a one at j = co - 16 qq - 8 h
  half8 idf[RES ? 2 : 1];
  if constexpr (RES) {
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int j = 0; j < 8; ++j) idf[qq][j] = (pix - 16 * qq - 8 * h == j) ? (_Float16)1.f : (_Float16)0.f;
  }

  int xoff[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = oxl + s;
    const int f = (ix >> 1) & 7;
    const int rowbase = (pg * 2 + oyl) * C::IWs + ix;
#pragma unroll
    for (int q = 0; q < 4; ++q) xoff[s][q] = rowbase * C::PIXB + (((2 * q + h) ^ f) * 16);
  }
  int roff[RES ? 2 : 1];
  if constexpr (RES) {
    const int slot = 32 * pg + pix;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) roff[qq] = slot * 128 + (((2 * (2 * ct + qq) + h) ^ ((slot >> 1) & 7)) * 16);
  }
  char* const stg = smem + C::STG_OFF + wave * C::STG_WAVE;

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const long in_plane_b = a.in_plane * 2;
  const long rowpitch = (long)a.W * 128;

  auto dma2 = [&](const char* src, bool valid, const char* zsrc, char* ldst, long plane_b, int lds_plane) {
    dma16(valid ? src : zsrc, ldst);
    dma16(valid ? src + plane_b : zsrc, ldst + lds_plane);
  };
  // ---- DMA pieces of the tile being fetched (7 per wave): 3 x 8-pixel halves of halo rows, 2 x the two right-most columns,
  //      2 x the identity tile (RES)
  int d_n = 0, d_ty0 = 0, d_tx0 = 0, d_buf = 0;
  bool d_interior = false;
  auto dma_setup = [&](int t, int buf) {
    d_n = t / tiles_per_img;
    const int tr = t - d_n * tiles_per_img;
    d_ty0 = tr / a.tiles_x;
    d_tx0 = tr - d_ty0 * a.tiles_x;
    d_buf = buf;
    // every halo pixel (and every pixel of the identity tile) inside the image: ~80 % of the tiles of a 135 x 240 map
    d_interior = d_ty0 > 0 && d_tx0 > 0 && (d_ty0 + 1) * C::TH + 1 <= a.H && (d_tx0 + 1) * C::TW + 1 <= a.W;
  };
  // per-lane byte offsets of the interior fast path (per-kernel constants)
  const unsigned vo_main0 = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 4)) * 16);
  const unsigned vo_main1 = 1024 + (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 4)) * 16) ^ 64);
  const unsigned vo_small = 2048 + lane * 16;
  unsigned vo_res[2] = {0, 0};
  if constexpr (RES) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int slot = 8 * (wave + 4 * j) + (lane >> 3);
      vo_res[j] = (unsigned)(((slot >> 4) * a.OW + (slot & 15)) * a.cout * 2 + ((lane & 7) ^ ((slot >> 1) & 7)) * 16);
    }
  }
  auto dma_piece = [&](int p) {
    const int gy0 = d_ty0 * C::TH - 1, gx0 = d_tx0 * C::TW - 1;
    char* lbase = smem + d_buf * 2 * C::IN_BYTES;
    const char* p00 = reinterpret_cast<const char*>(a.in) + ((long)d_n * a.H + gy0) * rowpitch + (long)gx0 * 128;
    if (d_interior) {
      // wave-uniform row base in SGPRs + the lane's constant offset: scalar arithmetic only at the issue site
      if (p < 3) {
        const int m = wave + 4 * p;
        const int iy = m >> 1, hf = m & 1;
        const char* rb = p00 + iy * rowpitch;
        char* ld = lbase + (iy * C::IWs + 8 * hf) * C::PIXB;
        const unsigned vo = hf ? vo_main1 : vo_main0;
        dma16s(rb, vo, ld);
        dma16s(rb + in_plane_b, vo, ld + C::IN_BYTES);
      } else if (p < 5) {
        const int iy = wave + 4 * (p - 3);
        if (iy < C::IH && lane < 16) {
          const char* rb = p00 + iy * rowpitch;
          char* ld = lbase + (iy * C::IWs + 16) * C::PIXB;
          dma16s(rb, vo_small, ld);
          dma16s(rb + in_plane_b, vo_small, ld + C::IN_BYTES);
        }
      } else if constexpr (RES) {
        const int r = wave + 4 * (p - 5);
        const char* rb = reinterpret_cast<const char*>(a.res + (((size_t)d_n * a.OH + d_ty0 * C::TH) * a.OW + d_tx0 * C::TW) * a.cout + cog * 64);
        char* ld = smem + C::RES_OFF + d_buf * 2 * C::RES_PLANE + r * 1024;
        dma16s(rb, vo_res[p - 5], ld);
        dma16s(rb + a.res_plane * 2, vo_res[p - 5], ld + C::RES_PLANE);
      }
      return;
    }
    if (p < 3) {
      const int lpx = lane >> 3;
      const int m = wave + 4 * p;
      const int iy = m >> 1, hf = m & 1;
      const int cc = (((lane & 7) ^ (lpx >> 1)) * 16) ^ (hf ? 64 : 0);
      const int gy = gy0 + iy, gx = gx0 + 8 * hf + lpx;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      dma2(p00 + iy * rowpitch + hf * 1024 + lpx * 128 + cc, ok, reinterpret_cast<const char*>(a.zeros) + cc,
           lbase + (iy * C::IWs + 8 * hf) * C::PIXB, in_plane_b, C::IN_BYTES);
    } else if (p < 5) {
      const int iy = wave + 4 * (p - 3);
      if (iy < C::IH && lane < 16) {
        const int gy = gy0 + iy, gx = gx0 + 16 + (lane >> 3);
        const bool ok = gy >= 0 && gy < a.H && gx < a.W;
        dma2(p00 + iy * rowpitch + 2048 + lane * 16, ok, reinterpret_cast<const char*>(a.zeros) + (lane & 7) * 16,
             lbase + (iy * C::IWs + 16) * C::PIXB, in_plane_b, C::IN_BYTES);
      }
    } else if constexpr (RES) {
      const int r = wave + 4 * (p - 5);               // 8 instructions of 8 pixels x 8 chunks per plane
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

  // ---- epilogue pieces of the PREVIOUS tile (its accumulator set pm / pc, its coordinates e_n / e_ty0 / e_tx0)
  int e_n = 0, e_ty0 = 1 << 24, e_tx0 = 0;      // (no previous tile yet: rows far outside the image)
  auto epi_stage = [&](const f32x16& pm, const f32x16& pc, int g) {
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = comb(pm[4 * g + e], pc[4 * g + e]);
      if (a.relu) y[e] = fmaxf(y[e], 0.f);
    }
    uint2 vh, vl;
    split2(y[0], y[1], vh.x, vl.x);
    split2(y[2], y[3], vh.y, vl.y);
    const int o = pix * C::STG_PITCH + 16 * g + 8 * h;
    *reinterpret_cast<uint2*>(stg + o) = vh;
    *reinterpret_cast<uint2*>(stg + C::STG_PLANE + o) = vl;
  };
  // round r (0 | 1): 16 pixels x 4 chunks of the wave's slab per plane; read and store are separate pieces (the LDS latency
  // passes under MFMAs), 8 registers live in between
  auto epi_read = [&](int r, uint4& cvh, uint4& cvl) {
    const int i = lane + 64 * r;
    const int o = (i >> 2) * C::STG_PITCH + (i & 3) * 16;
    cvh = *reinterpret_cast<const uint4*>(stg + o);
    cvl = *reinterpret_cast<const uint4*>(stg + C::STG_PLANE + o);
  };
  auto epi_store = [&](int r, const uint4& cvh, const uint4& cvl) {
    // exactly two stores per lane and round = four per tile (the counted wait at the tile top): out-of-image pixels go to
    // the trash line
    _Float16* trash = const_cast<_Float16*>(a.zeros) + 1024 + (threadIdx.x & 127) * 8;
    const int i = lane + 64 * r;
    const int px = i >> 2, c4 = i & 3;
    const int oy = e_ty0 * C::TH + 2 * pg + (px >> 4), ox = e_tx0 * C::TW + (px & 15);
    const bool ok = oy < a.OH && ox < a.OW;
    _Float16* dst = a.out + (((size_t)e_n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.cout + cog * 64 + ct * 32 + c4 * 8;
    *reinterpret_cast<uint4*>(ok ? dst : trash) = cvh;
    *reinterpret_cast<uint4*>(ok ? dst + a.out_plane : trash) = cvl;
  };

  int t = t_begin + bix;
  int buf = 0;
  bool first = true;
  if (t < t_end) {
    dma_setup(t, 0);
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) dma_piece(p);
  }
  int dbg_it = 0; (void)dbg_it;

  // one tile: contraction into (am, ac); the previous tile's set (pm, pc) leaves through the pieces
  // (two accumulator sets per tile: a third one for the second correction product -- so that no MFMA follows another on the
  //  same registers -- measured the same k loop, 5.6 k cycles per 108 MFMAs, and spilled in the residual variant)
  auto tile = [&](f32x16& am, f32x16& ac, const f32x16& pm, const f32x16& pc) {
    PL_T(0);
    // the VMEM operations younger than this tile's DMA pieces are the four output stores issued later in the previous
    // contraction: vmcnt retires in order, four may stay in flight
    if (!first) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first = false;
    PL_T(1);
    block_barrier();
    PL_T(2);
    const bool has_next = t + t_step < t_end;
    if (has_next) dma_setup(t + t_step, buf ^ 1);
    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const char* xb = smem + buf * 2 * C::IN_BYTES;
    {
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
        am[4 * g + 0] = b4.x; am[4 * g + 1] = b4.y; am[4 * g + 2] = b4.z; am[4 * g + 3] = b4.w;
        ac[4 * g + 0] = 0.f; ac[4 * g + 1] = 0.f; ac[4 * g + 2] = 0.f; ac[4 * g + 3] = 0.f;
      }
    }
    PL_T(3);
    auto xaddr = [&](int k) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return xb + xoff[s][q] + r * C::IWs * C::PIXB;
    };
    constexpr int PD = 3;
    half8 xqh[PD + 1], xql[PD + 1];
    uint4 cvh, cvl;
#pragma unroll
    for (int k = 0; k < PD; ++k) {
      const char* p = xaddr(k);
      xqh[k] = *reinterpret_cast<const half8*>(p);
      xql[k] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
    }
    static_for([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      // pieces: DMA of the next tile at k = 1, 3, .. ; staging of the previous tile at k = 2, 4, 6, 8; its LDS -> register
      // reads at k = 16 / 24, its stores at k = 20 / 28 (behind every DMA piece: the counted wait above)
#ifndef PL_C3_NOPIECES
      if constexpr (k % 2 == 1 && (k - 1) / 2 < NPIECE) {
        if (has_next) dma_piece((k - 1) / 2);
      }
      // (unconditional: in front of the first tile the "previous tile" is a zero set at coordinates outside the image, its
      //  four stores go to the trash line -- no branch around the pieces, one store count for the wait at the tile top)
      if constexpr (k == 2 || k == 4 || k == 6 || k == 8) epi_stage(pm, pc, k / 2 - 1);
      if constexpr (k == 16) epi_read(0, cvh, cvl);
      if constexpr (k == 20) epi_store(0, cvh, cvl);
      if constexpr (k == 24) epi_read(1, cvh, cvl);
      if constexpr (k == 28) epi_store(1, cvh, cvl);
#endif
      if constexpr (k + PD < C::NK) {
        const char* p = xaddr(k + PD);
        xqh[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p);
        xql[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
      }
      PL_SB();
      am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)], am, 0, 0, 0);
      ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)], ac, 0, 0, 0);
      ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)], ac, 0, 0, 0);
      PL_SB();
    }, std::make_integer_sequence<int, C::NK>{});
    if constexpr (RES) {
      const char* rb = smem + C::RES_OFF + buf * 2 * C::RES_PLANE;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const half8 rh = *reinterpret_cast<const half8*>(rb + roff[qq]);
        const half8 rl = *reinterpret_cast<const half8*>(rb + C::RES_PLANE + roff[qq]);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[qq], rh, am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[qq], rl, ac, 0, 0, 0);
      }
    }
    PL_T(4);
    e_n = n; e_ty0 = ty0; e_tx0 = tx0;
  };

  f32x16 am0, ac0, am1, ac1;
#pragma unroll
  for (int r = 0; r < 16; ++r) am0[r] = ac0[r] = am1[r] = ac1[r] = 0.f;
  int last = -1;
  while (t < t_end) {
    tile(am0, ac0, am1, ac1);
    last = 0;
    t += t_step; buf ^= 1; ++dbg_it;
    if (t >= t_end) break;
    tile(am1, ac1, am0, ac0);
    last = 1;
    t += t_step; buf ^= 1; ++dbg_it;
  }
  if (last >= 0) {
    // the last tile's epilogue has no contraction to hide under
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (last == 0) epi_stage(am0, ac0, g);
      else epi_stage(am1, ac1, g);
    }
    uint4 cvh, cvl;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      epi_read(r, cvh, cvl);
      epi_store(r, cvh, cvl);
    }
  }
#ifdef LFD_PL_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_pl_dbg[130] = __builtin_readcyclecounter(); g_pl_dbg[131] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <bool RES>
int launch_pl_c3(const PlArgs& a0, hipStream_t st) {
  using C = C3<RES>;
  PlArgs a = a0;
  a.tiles_x = (a.OW + C::TW - 1) / C::TW;
  a.tiles_y = (a.OH + C::TH - 1) / C::TH;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  const int cgroups = a.cout / 64;
  auto kern = k_pl_c3<RES>;
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
  hipLaunchKernelGGL(kern, dim3(blocks, cgroups), dim3(256), C::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}


}  // namespace pl

#ifdef LFD_PL_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_pl_c3_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pl::g_pl_dbg), sizeof(unsigned long long) * 136);
}
#endif

int lfd_pl_c3_launch(const pl::PlArgs& a, bool residual, hipStream_t st) {
  return residual ? pl::launch_pl_c3<true>(a, st) : pl::launch_pl_c3<false>(a, st);
}
