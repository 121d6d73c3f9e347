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
// All pyramid levels run in ONE launch per pass (persistent workgroups walk a level-major tile
// list and reload the per-level neck weights on a level switch): 5 launches per forward.
// (+13 % MFMA work for the whole network, -65 % head HBM bytes; pre-norm values never leave fp32
// registers).  Partial sums are combined in fp64 in a fixed order (deterministic).
//
// Per workgroup: 4 waves, wave = one 32-channel output tile of the 128 head channels, all
// waves share the same 64 pixels; stage outputs are exchanged through swizzled LDS tiles;
// weights of every stage stay in VGPRs across the persistent tile loop.
#include <type_traits>
#include "common.h"
#include "decode_impl.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int HC = 128;        // head / neck channels
constexpr int TPX = 64;        // pixels per tile (2 MFMA pixel tiles)

struct HeadLevel {
  const _Float16* x;     // [N, HW, cin] backbone tap (NHWC, flattened pixels)
  const half8* wn;       // neck   [4][cin/16][64]
  const float* bn;       // neck bias (BN folded) [128]
  const half8* w1;       // tower conv 1 [4][8][64]
  const half8* w2;       // tower conv 2 [4][8][64]
  const half8* wf;       // final conv   [FT][8][64]  (cout padded to FT*32)
  const float* bf;       // final bias   [FT*32]
  const float* scale;    // per-level Scale parameter (device scalar) or nullptr
  const half8* w1f;      // [N][4][9][64] tower conv 1 / 2 with GroupNorm folded in (k_gn_finalize), K-permuted fragments,
  const half8* w2f;      //               k-step 8 = the shift as a bias fragment; or nullptr (fold per work chunk)
  const half8* w1p;      // [4][8][64] tower conv 1 / 2 unscaled, K-permuted (host-made copy), or nullptr (permute on the fly)
  const half8* w2p;
  half8* a1;             // [N][groups of 32 pixels][8][64]: ReLU(GN1(conv1)) of every pixel as conv2's B fragments, written by
                         // pass 2 and read by the output pass instead of recomputing neck + conv1; or nullptr
  int cin, hw, p_off;    // tap channels, pixels per image, first point of the level
  int tile_start;        // first global tile of this level
  int tiles_per_img;
};

struct HeadArgs {
  HeadLevel lv[LFD_MAX_LEVELS];
  int nlev;
  const float* ab1;      // [L][N][128][2]  GN1 (scale, shift) per level/image/channel  (pass >= 2)
  const float* ab2;      // [L][N][128][2]  GN2                                         (pass == 3)
  float* part;           // [ntiles][128 >> gshift][2] per-tile partial statistics       (pass 1, 2)
  float* out_cls;        // [N, P, CC]
  float* out_reg;        // [N, P, 4]
  int N, P, CC;
  int reg_rows;          // final conv rows [0,reg_rows) -> reg (0 or 4), then cls_rows rows -> cls
  int cls_rows;
  int grp_levels[LFD_MAX_LEVELS];      // levels handled by this launch (same tap channel count)
  int grp_tile_start[LFD_MAX_LEVELS];  // first launch-local tile of each of them
  int grp_n, grp_ntiles;
  int grp_gpi[LFD_MAX_LEVELS];         // k_head2: 32-pixel groups per image of each of them
  int h2_chg[LFD_MAX_LEVELS];          //          groups per work chunk (even: whole statistics tiles)
  int h2_item_start[LFD_MAX_LEVELS];   //          first work item (= 4 consecutive chunks) of each of them
  int h2_nitems;
  int gshift;            // log2(channels per group)  (GroupNorm(16,128) -> 3)
  int ntiles;
  const _Float16* zeros;
  LfdAppendTarget dec;   // k_head2<3, 1, true>: where the candidates go (lfd_head_forward_decode_f16)
};

// global -> LDS DMA through inline asm and a fence-free barrier: see conv.hip.  (With the builtin, the compiler
// put s_waitcnt vmcnt(0) in front of the first LDS read / every __syncthreads(), i.e. each tile waited for the
// whole ring -- including the tiles it had just requested -- and the 4-deep prefetch bought nothing.)
__device__ __forceinline__ void dma16(const void* g, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory");
}
__device__ __forceinline__ void block_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
// sum over the 64 lanes with DPP adds (6 VALU instructions; the result is valid in lane 63 only).  __shfl_xor goes
// through the LDS crossbar (ds_bpermute): 32 reductions of 6 dependent steps cost ~9.7 k cycles per chunk in k_head2.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  int x = __float_as_int(v);
#define LFD_DPP_ADD(ctrl, rmask)                                                                       \
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false)))
  LFD_DPP_ADD(0x111, 0xf);   // row_shr:1
  LFD_DPP_ADD(0x112, 0xf);   // row_shr:2
  LFD_DPP_ADD(0x114, 0xf);   // row_shr:4
  LFD_DPP_ADD(0x118, 0xf);   // row_shr:8   -> lane 15 of every row of 16 holds its row's sum
  LFD_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  LFD_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef LFD_DPP_ADD
  return __int_as_float(x);
}

// Sums of 32 per-lane values over the 64 lanes in ~100 VALU instead of 32 x 6: a butterfly in which every step
// halves the number of values a lane carries (lane bit k decides which of a pair it keeps; the other one goes to
// the partner lane, whose bit k differs) -- quad permutes for bits 0/1, row rotations for bits 2/3 (the source
// lane i -+ 4 / 8 has that bit flipped and the lower bits equal), the LDS crossbar for bits 4/5.  Returns, in
// lane l, the total of v[l & 31].  Fixed order -> deterministic.
__device__ __forceinline__ float wave_sum_transpose32(const float (&v)[32], int lane) {
  float a[16], b[8], c[4], d[2];
#define LFD_TR_STEP(dst, src, n, bit, ctrl)                                                                       \
  _Pragma("unroll") for (int i = 0; i < n; ++i) {                                                                 \
    const bool up = (lane >> bit) & 1;                                                                            \
    const float keep = up ? src[2 * i + 1] : src[2 * i], give = up ? src[2 * i] : src[2 * i + 1];                  \
    dst[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), ctrl, 0xf, 0xf, false));   \
  }
  LFD_TR_STEP(a, v, 16, 0, 0xB1);    // quad_perm [1,0,3,2]
  LFD_TR_STEP(b, a, 8, 1, 0x4E);     // quad_perm [2,3,0,1]
  LFD_TR_STEP(c, b, 4, 2, 0x124);    // row_ror:4
  LFD_TR_STEP(d, c, 2, 3, 0x128);    // row_ror:8
#undef LFD_TR_STEP
  const bool up = (lane >> 4) & 1;
  const float keep = up ? d[1] : d[0], give = up ? d[0] : d[1];
  float r = keep + __shfl_xor(give, 16, 64);
  r += __shfl_xor(r, 32, 64);
  return r;
}

// swizzled LDS tile of [TPX pixels][C channels] fp16: byte address of (pixel, 16-byte chunk c)
template <int CPP>
__device__ __forceinline__ int lds_addr(int px, int c) {
  constexpr int PPR = (CPP >= 16) ? 1 : 16 / CPP;
  return px * (CPP * 16) + ((c ^ ((px / PPR) % CPP)) * 16);
}

template <int CIN, int PASS, int FT>
__global__ __launch_bounds__(256, 2) void k_head(HeadArgs a) {
  constexpr int CPPX = CIN / 8;              // 16-byte chunks per input pixel
  constexpr int XBYTES = TPX * CIN * 2;      // one input tile
  constexpr int NB = 32768 / XBYTES;         // input ring depth: 4 (64 ch) / 2 (128 ch) tiles in flight
  constexpr int KD = (TPX * CPPX) / 256;     // DMA instructions per wave per tile
  constexpr int ABYTES = TPX * HC * 2;
  constexpr int NKN = CIN / 16, NKH = HC / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xring = smem;                        // NB x XBYTES = 32 KB
  char* bufA = smem + 32768;                 // stage ping
  char* bufB = bufA + ABYTES;                // stage pong
  float* s_ab = reinterpret_cast<float*>(bufB + ABYTES);   // [2 stages][128][2] GN (scale, shift) of the current (level, image)
  float* s_bf = s_ab + 512;                                // [64] final bias of the current level
  float* s_bn = s_bf + 64;                                 // [128] neck bias of the current level
  half8* s_wf = reinterpret_cast<half8*>(s_bn + 128);      // [FT][8][64] final-conv fragments (LDS, not VGPRs)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pix = lane & 31;
  const int ct = wave;                       // 32-channel tile of the 128 head channels
  const int fct = wave % FT, fpt = wave / FT;   // final stage: (cout tile, pixel tile) of this wave
  const bool f_active = fpt < 2;

  half8 wn[NKN], w1[NKH], w2[PASS >= 2 ? NKH : 1];
  float scale = 1.f;
  int cur_level = -1, cur_n = -1;

  struct TileId { int l, n, p0, tg; };
  auto tile_of = [&](int u) {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < LFD_MAX_LEVELS; ++i)
      if (i < a.grp_n && u >= a.grp_tile_start[i]) gi = i;
    TileId t;
    t.l = a.grp_levels[gi];
    const int tl = u - a.grp_tile_start[gi];
    t.n = tl / a.lv[t.l].tiles_per_img;
    t.p0 = (tl - t.n * a.lv[t.l].tiles_per_img) * TPX;
    t.tg = a.lv[t.l].tile_start + tl;      // global tile index (partial statistics row)
    return t;
  };

  auto load_level = [&](int l) {
    const HeadLevel& L = a.lv[l];
#pragma unroll
    for (int k = 0; k < NKN; ++k) wn[k] = L.wn[(ct * NKN + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < NKH; ++k) w1[k] = L.w1[(ct * NKH + k) * 64 + lane];
    if constexpr (PASS >= 2) {
#pragma unroll
      for (int k = 0; k < NKH; ++k) w2[k] = L.w2[(ct * NKH + k) * 64 + lane];
    }
    if constexpr (PASS == 3) {
      for (int i = threadIdx.x; i < FT * NKH * 64; i += 256) s_wf[i] = L.wf[i];
      scale = L.scale ? L.scale[0] : 1.f;
      if (threadIdx.x < FT * 32) s_bf[threadIdx.x] = L.bf[threadIdx.x];
    }
    if (threadIdx.x < HC) s_bn[threadIdx.x] = L.bn[threadIdx.x];
  };

  auto issue_dma = [&](int u, int slot) {
    const TileId t = tile_of(u);
    const HeadLevel& L = a.lv[t.l];
    constexpr int SPW = 64 / CPPX, PPR = (CPPX >= 16) ? 1 : 16 / CPPX;
    char* lbase = xring + slot * XBYTES;
#pragma unroll
    for (int i = 0; i < KD; ++i) {
      const int slot0 = (wave + 4 * i) * SPW;
      const int px = slot0 + lane / CPPX;
      const int c = (lane % CPPX) ^ ((px / PPR) % CPPX);
      const bool valid = (t.p0 + px) < L.hw;
      const _Float16* src = valid ? L.x + ((size_t)t.n * L.hw + t.p0 + px) * CIN + c * 8 : a.zeros + c * 8;
      dma16(src, lbase + slot0 * (CIN * 2));
    }
  };

  // one MFMA stage over a [TPX][CPP*8] swizzled LDS tile
  auto stage = [&](f32x16 (&acc)[2], const half8* w, const char* src, auto cpp_tag) {
    constexpr int CPP = decltype(cpp_tag)::value;
#pragma unroll
    for (int q = 0; q < (CPP / 2); ++q) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const half8 xf = *reinterpret_cast<const half8*>(src + lds_addr<CPP>(pt * 32 + pix, 2 * q + h));
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[q], xf, acc[pt], 0, 0, 0);
      }
    }
  };
  // write relu(acc*sc + sh) as fp16 into a [TPX][128] LDS tile (this wave's 32 channels);
  // ab == nullptr: plain ReLU (neck).  ab: LDS [128][2] (scale, shift) of the current image.
  auto store_tile = [&](const f32x16 (&acc)[2], char* dst, const float* ab) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
      if (ab) {
        const float4* p = reinterpret_cast<const float4*>(ab + (ct * 32 + 8 * g + 4 * h) * 2);
        const float4 a0 = p[0], a1 = p[1];
        sc[0] = a0.x; sh[0] = a0.y; sc[1] = a0.z; sh[1] = a0.w; sc[2] = a1.x; sh[2] = a1.y; sc[3] = a1.z; sh[3] = a1.w;
      }
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = acc[pt][4 * g + j];
          if (ab) x[j] = x[j] * sc[j] + sh[j];
        }
        uint2 v;
        v.x = lfd_cvt_pk_max(x[0], x[1], LFD_PK_RELU);
        v.y = lfd_cvt_pk_max(x[2], x[3], LFD_PK_RELU);
        *reinterpret_cast<uint2*>(dst + lds_addr<HC / 8>(pt * 32 + pix, ct * 4 + g) + 8 * h) = v;
      }
    }
  };
  // per-tile GroupNorm partial statistics of this wave's 32 channels.  Each (tile, group) slot has
  // exactly one writer for group sizes >= 8 (one 8-byte store per group, no zero-init needed);
  // smaller groups accumulate with atomics into a buffer the host zeroes.
  auto write_stats = [&](const f32x16 (&acc)[2], int tg, int p0, int hw) {
    const float v0 = (p0 + pix) < hw ? 1.f : 0.f, v1 = (p0 + 32 + pix) < hw ? 1.f : 0.f;
    float s[16], ss[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float y0 = acc[0][i] * v0, y1 = acc[1][i] * v1;
      s[i] = y0 + y1;
      ss[i] = y0 * y0 + y1 * y1;
    }
    const int ngl = 32 >> a.gshift;  // groups in this wave's 32-channel tile
    float* dst = a.part + ((size_t)tg * (HC >> a.gshift) + (size_t)ct * ngl) * 2;
    if (a.gshift >= 3) {
      const int gg = 1 << (a.gshift - 3);  // 8-channel blocks per group
      float b8[4], b8q[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        b8[g] = wave_sum_f(s[4 * g] + s[4 * g + 1] + s[4 * g + 2] + s[4 * g + 3]);
        b8q[g] = wave_sum_f(ss[4 * g] + ss[4 * g + 1] + ss[4 * g + 2] + ss[4 * g + 3]);
      }
      if (lane < ngl) {            // lane k owns group k: sum its gg 8-channel blocks
        float x = 0.f, xx = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (g / gg == lane) { x += b8[g]; xx += b8q[g]; }
        *reinterpret_cast<float2*>(dst + lane * 2) = make_float2(x, xx);
      }
    } else {
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

  // ---- persistent loop over a CONTIGUOUS range of this launch's tiles (same level / image for
  //      long stretches: weights and GN tables reload rarely), input tiles NB-1 ahead in flight
  const int per = (a.grp_ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int u0 = blockIdx.x * per;
  const int u1 = (u0 + per) < a.grp_ntiles ? (u0 + per) : a.grp_ntiles;
#pragma unroll
  for (int i = 0; i < NB - 1; ++i)
    if (u0 + i < u1) issue_dma(u0 + i, i);
  for (int u = u0; u < u1; ++u) {
    const int slot = (u - u0) % NB;
    const TileId t = tile_of(u);
    const HeadLevel& L = a.lv[t.l];
    if (t.l != cur_level || t.n != cur_n) {
      // (level, image) switch: the ring stays in flight; the compiler waits for the filter loads it can see
      block_barrier();                      // previous tile's readers of s_ab / s_bf are done
      if (t.l != cur_level) {
        load_level(t.l);
        // Touch the freshly loaded filters HERE: the compiler then waits for them inside this (rare) branch.
        // Otherwise it places conservative vmcnt(k..0) waits at their uses in every iteration, and since vmcnt
        // counts the DMA ring too, every tile would drain the whole prefetch ring.
#pragma unroll
        for (int k = 0; k < NKN; ++k) asm volatile("" : "+v"(wn[k]));
#pragma unroll
        for (int k = 0; k < NKH; ++k) asm volatile("" : "+v"(w1[k]));
        if constexpr (PASS >= 2) {
#pragma unroll
          for (int k = 0; k < NKH; ++k) asm volatile("" : "+v"(w2[k]));
        }
      }
      if constexpr (PASS >= 2) {
        for (int i = threadIdx.x; i < 256 * (PASS - 1); i += 256) {
          const float* src = (i < 256 ? a.ab1 : a.ab2) + ((size_t)t.l * a.N + t.n) * HC * 2;
          s_ab[i] = src[i & 255];
        }
      }
      cur_level = t.l; cur_n = t.n;
    }
    if (u + NB - 1 < u1) {
      issue_dma(u + NB - 1, (u - u0 + NB - 1) % NB);
      // loads retire in order: at most (NB-1)*KD younger DMA instructions may still be in flight
      if constexpr (NB == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    block_barrier();
    static_assert((NB == 4 && KD == 2) || (NB == 2 && KD == 4), "vmcnt immediates above assume these");

    f32x16 acc[2];
    // neck: relu(Wn x + bn)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(s_bn + ct * 32 + 8 * g + 4 * h);
      acc[0][4 * g] = b4.x; acc[0][4 * g + 1] = b4.y; acc[0][4 * g + 2] = b4.z; acc[0][4 * g + 3] = b4.w;
      acc[1][4 * g] = b4.x; acc[1][4 * g + 1] = b4.y; acc[1][4 * g + 2] = b4.z; acc[1][4 * g + 3] = b4.w;
    }
    stage(acc, wn, xring + slot * XBYTES, std::integral_constant<int, CPPX>{});
    store_tile(acc, bufA, nullptr);
    block_barrier();
    // conv1 (no bias: norm follows, lfd_head.py:97)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    stage(acc, w1, bufA, std::integral_constant<int, HC / 8>{});
    if constexpr (PASS == 1) {
      write_stats(acc, t.tg, t.p0, L.hw);
    } else {
      store_tile(acc, bufB, s_ab);
      block_barrier();
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
      stage(acc, w2, bufB, std::integral_constant<int, HC / 8>{});
      if constexpr (PASS == 2) {
        write_stats(acc, t.tg, t.p0, L.hw);
      } else {
        store_tile(acc, bufA, s_ab + 256);   // bufA's neck tile was fully consumed before the last barrier
        block_barrier();
        if (f_active) {
          f32x16 fa;
          {
            const float* bp = s_bf + fct * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
              fa[4 * g] = b4.x; fa[4 * g + 1] = b4.y; fa[4 * g + 2] = b4.z; fa[4 * g + 3] = b4.w;
            }
          }
#pragma unroll
          for (int q = 0; q < NKH; ++q) {
            const half8 xf = *reinterpret_cast<const half8*>(bufA + lds_addr<HC / 8>(fpt * 32 + pix, 2 * q + h));
            fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_wf[(fct * NKH + q) * 64 + lane], xf, fa, 0, 0, 0);
          }
          const int p = t.p0 + fpt * 32 + pix;
          if (p < L.hw) {
            const size_t row = (size_t)t.n * a.P + L.p_off + p;
            // final rows: [reg x reg_rows][cls x cls_rows]; lane (pixel, h) holds rows 8g + 4h + j
            if (a.reg_rows == 4 && fct == 0 && h == 0)
              *reinterpret_cast<float4*>(a.out_reg + row * 4) =
                  make_float4(fa[0] * scale, fa[1] * scale, fa[2] * scale, fa[3] * scale);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int ch = fct * 32 + 8 * (i >> 2) + 4 * h + (i & 3) - a.reg_rows;
              if (ch >= 0 && ch < a.cls_rows) a.out_cls[row * a.CC + ch] = fa[i];
            }
          }
        }
      }
    }
  }
}

// =====================================================================================================
// k_head2 (64-channel taps): one WAVE carries 32 pixels through the whole chain, registers only.
//
// k_head above splits the 128 head channels over the four waves of a workgroup, so every stage ends in an LDS
// exchange + workgroup barrier (5 per 64-pixel tile) -- the kernel is barrier / latency bound.  Here each wave owns
// all 128 channels of its 32 pixels:
//   * the accumulators of one stage (lane = pixel, register = channel) are the B operand of the next stage
//     directly -- registers 8(q&1)..8(q&1)+7 of channel tile q>>1 are k-step q; the next filter's K order is permuted
//     to match when it is loaded.  No LDS traffic for activations, no barrier, no tile staging;
//   * GroupNorm's per-(image, channel) scale is folded into the rows of the conv that produced the tensor (the same
//     kind of fold the host does for BatchNorm, done on the device once per image with one fp16 rounding), its
//     shift and the neck / final biases ride on an extra MFMA k-step (B = 1, A = value split hi + lo in fp16);
//     between stages only ReLU + the fp16 pack remain (2 instructions per pair of values);
//   * one 256-thread workgroup per CU, 512 registers per wave: conv1 and conv2 filters resident (256 registers);
//     the level's neck / final filters sit in LDS, shared by the workgroup (a workgroup only ever works on one level
//     at a time);
//   * work unit = chunk of consecutive 32-pixel groups of one image.  GroupNorm partial sums are accumulated per
//     lane over ONE 64-pixel statistics tile (2 groups), reduced across the lanes and written to that tile's slot:
//     the statistics -- and therefore the outputs -- of an image depend neither on the batch it is in nor on the
//     chunk length, which the host is free to choose for load balance (h2_plan_chunks).
// =====================================================================================================
constexpr int H2_CH = 12;      // longest chunk (384 pixels): 904 chunks for 8 x 1080p, one per wave of 256 CUs
// Chunk length per level.  Every chunk pays ~9 k cycles of per-image filter setup, a 32-pixel group costs ~7 k:
// long chunks amortise the setup, short ones fill the chip when the batch is small (1 x 1080p has 1353 groups:
// 12-group chunks would keep 31 of 256 CUs busy).  Target: about one chunk per wave of the chip (1024), at most
// H2_CH groups; the small levels get shorter chunks (their few chunks would otherwise be the longest-running
// work items of the launch).  8 x 1080p: (85 + 22 + 8 + 4 + 2) chunks per image = 968 chunks = 242 work items.
inline void h2_plan_chunks(int n, int nlev, const int* gpi, int* chg, int waves_per_simd = 1) {
  const int forced = lfd_tune(LFD_TUNE_H2_CHUNK);   // tests: any even value
  long total = 0;
  for (int j = 0; j < nlev; ++j) total += (long)gpi[j] * n;
  const long slots = 2048L * waves_per_simd;       // two groups per chunk at least; about one chunk per resident wave
  int c = 2 * (int)((total + slots - 1) / slots);
  c = c < 2 ? 2 : (c > H2_CH ? H2_CH : c);
  if (forced >= 2) c = forced & ~1;
  for (int j = 0; j < nlev; ++j) {
    const int cap = gpi[j] >= 128 ? H2_CH : (gpi[j] >= 32 ? 8 : 4);
    chg[j] = (forced >= 2 || c < cap) ? c : cap;
  }
}

__device__ __forceinline__ uint32_t h2_cvt_pk(float x, float y) {
  lfd_f32x2 f; f[0] = x; f[1] = y;
  union { lfd_f16x2 v; uint32_t u; } r; r.v = __builtin_convertvector(f, lfd_f16x2);
  return r.u;
}
__device__ __forceinline__ uint32_t h2_relu_pk(uint32_t h) {
  union { lfd_f16x2 v; uint32_t u; } r, z; r.u = h; z.u = 0u;
  r.v = __builtin_elementwise_max(r.v, z.v);
  return r.u;
}

#ifdef LFD_H2_TIMING
__device__ unsigned long long g_h2_dbg[3 * 32];
#define H2_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_h2_dbg[(PASS - 1) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define H2_T(i)
#endif

// FOLD (output pass only, every level has w1_folded / w2_folded): both tower filters are loaded straight into AccVGPRs by
// inline asm and read from there by the MFMAs.  As ordinary values, 256 registers of filter + the working set exceed the
// 256 architectural VGPRs and the allocator parks fragments in AccVGPRs, copying them back in front of every use: 144
// v_accvgpr_read per pixel group in this pass, with nothing to overlap them (one wave per SIMD).
// A1 (output pass, with FOLD): conv2's B fragments -- ReLU(GN1(conv1(neck(x)))) in fp16, exactly the registers pass 2 fed its own
// conv2 with -- are loaded from the buffer pass 2 wrote (lfd_head_level_ptrs_t.tower1_out) instead of recomputed: 45 instead of
// 101 MFMAs per pixel group, no neck / conv1 filters, bit-identical outputs; costs 256 B per pixel of HBM write + read.
// Waves per SIMD: one for every pass.  -DH2_P1_TWO gives the statistics pass of conv1 (PASS 1, which keeps only ONE tower
// filter resident) two workgroups per CU so that the VALU half of a group could overlap the other workgroup's MFMAs --
// measured round 3: the 256-register budget spills 116 dwords per lane into scratch and the serial step got 31 us SLOWER
// (0.661 vs 0.629 ms under rocprofv3).  A negative result, kept as a switch.
#ifdef H2_P1_TWO
#define H2_WPS(PASS) ((PASS) == 1 ? 2 : 1)
#else
#define H2_WPS(PASS) 1
#endif
template <int PASS, int FT, bool DEC = false, bool FOLD = false, bool A1 = false>
__global__ __launch_bounds__(256, H2_WPS(PASS)) void k_head2(HeadArgs a) {
  static_assert(!DEC || (PASS == 3 && FT == 1), "decode rides on the output pass of a merged single-class tower");
  static_assert(!FOLD || PASS == 3, "AccVGPR-resident filters: output pass");
  static_assert(!A1 || FOLD, "stored tower-1 activations: output pass with folded filters");
  constexpr int NKN = 4, NKNX = 8, NKH = HC / 16;                      // neck k-steps per 64 input channels / maximum (128 channels)
  constexpr int WN_FRAGS = 4 * (NKNX + 1);                             // neck: 4 cout tiles x (up to 8 k-steps + bias step)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half8* s_wn = reinterpret_cast<half8*>(smem);                         // [4][NKNX + 1][64 lanes]; bias step at index NKNX
  half8* s_wf = s_wn + WN_FRAGS * 64;                                   // [FT][NKH + 1][64 lanes]
  half8* s_wb = s_wf + (PASS == 3 ? FT * (NKH + 1) : 0) * 64 + (threadIdx.x >> 6) * 8 * 64;   // wave-private: GN shift k-steps [2][4][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int hh = lane >> 5, pix = lane & 31;
  // DEC: wave-private candidate staging, 2 x [64] x {box 16 B | score 4 B | point 4 B}: candidates are decoded where they
  // are found, collected here and appended with ONE atomic per chunk (per-group atomics on the per-image counter serialise
  // in L2: 1850 of them on 8 addresses cost the 8 x 1080p launch 19 us).  The staging traffic goes through inline-asm
  // ds_write / ds_read WITHOUT a memory clobber, on purpose: a compiler-visible LDS store in the pixel loop makes every
  // loop-invariant LDS fragment (final conv, bias steps: 17 of them) be re-read per group behind serialised waits
  // (+8 us per launch); nothing else touches this array and LDS operations of one wave execute in order.
  constexpr int DEC_ST = 64;
  __shared__ __attribute__((aligned(16))) char s_stage[DEC ? 4 * 2 * DEC_ST * 24 : 16];
  const uint32_t st_base = (uint32_t)(uintptr_t)s_stage + (DEC ? wave * (2 * DEC_ST * 24) : 0);
  int npend = 0;       // wave-uniform
  uint32_t st_tok = 0; // orders the staging asm statements (data dependence instead of `volatile`)

  half8 w1[FOLD ? 1 : 4][NKH], w2[(PASS >= 2 && !FOLD) ? 4 : 1][NKH];
  u32x4v w1a[(FOLD && !A1) ? 4 : 1][NKH], w2a[FOLD ? 4 : 1][NKH];
  float scale = 1.f;
  union { half8 v; uint32_t u[4]; } ones_f;
  ones_f.u[0] = hh ? 0u : 0x3c003c00u; ones_f.u[1] = 0u; ones_f.u[2] = 0u; ones_f.u[3] = 0u;
  const half8 ones = ones_f.v;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // K-permuted fragment of a [128 -> 32*T] 1x1 filter in the standard packing: element e of lane (m, hk) is
  // W[m][16q + 8(e>>2) + 4hk + (e&3)] = standard lane (m, e>>2), element 4hk + (e&3); rows optionally scaled
  auto perm = [&](const half8* wstd, int c, int q, float rs) {
    const int m = lane & 31, hk = lane >> 5;
    const _Float16* base = reinterpret_cast<const _Float16*>(wstd + (c * NKH + q) * 64);
    const half4 lo = *reinterpret_cast<const half4*>(base + m * 8 + 4 * hk);
    const half4 hi = *reinterpret_cast<const half4*>(base + (m + 32) * 8 + 4 * hk);
    half8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (_Float16)((float)lo[e] * rs); r[4 + e] = (_Float16)((float)hi[e] * rs); }
    return r;
  };
  // bias k-step fragment: lane (m, hk = 0) = {hi, lo, 0 ...}
  auto bias_frag = [&](float v) {
    const _Float16 bh = (_Float16)v, bl = (_Float16)(v - (float)bh);
    half8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (_Float16)0.f;
    if (!(lane >> 5)) { r[0] = bh; r[1] = bl; }
    return r;
  };
  // accumulator registers [8*half, 8*half+8) -> 8 fp16 after ReLU = one B operand of the next stage
  auto to_b = [&](const f32x16& acc, int half) {
    union { half8 v; uint32_t u[4]; } r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r.u[e] = h2_cvt_pk(acc[8 * half + 2 * e], acc[8 * half + 2 * e + 1]);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.u[e] = h2_relu_pk(r.u[e]);
    return r.v;
  };

  // Append the staged candidates: ONE atomic per chunk and wave reserves the slots; its result is not awaited -- the
  // staging buffer is double-buffered and the reserved range is filled when the NEXT chunk ends (or at kernel end), by
  // which time the counter value has long arrived (a wave waiting ~2-5 us per chunk for a contended L2 atomic cost the
  // 8 x 1080p launch 9 us).  No max-coordinate atomic: with one class the reference's class offsets are exactly zero
  // (label 0 * (max + 1), nms.py:148-150) and lfd_detect_from_candidates skips them.
  int pend_base = 0, pend_cnt = 0, pend_n = 0, sbuf = 0;       // reservation in flight (wave-uniform but for pend_base)
  auto dec_copy_out = [&]() {
    if constexpr (DEC) {
      if (pend_cnt) {
        const int base = __builtin_amdgcn_readfirstlane(pend_base);
        const uint32_t sa = st_base + (sbuf ^ 1) * (DEC_ST * 24);
        f32x4v bv;
        float scv;
        int ptv;
        asm("ds_read_b128 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(bv), "=&v"(scv), "=&v"(ptv), "+v"(st_tok)
            : "v"(sa + lane * 16), "v"(sa + DEC_ST * 16 + lane * 4), "v"(sa + DEC_ST * 20 + lane * 4));
        if (lane < pend_cnt && base + lane < a.dec.cap) {
          // (explicit global address space: a flat store / atomic may alias LDS as far as the compiler knows, and it would
          //  re-read the loop-invariant LDS fragments it keeps in registers after each of them)
          const size_t o = (size_t)pend_n * a.dec.cap + base + lane;
          typedef __attribute__((address_space(1))) f32x4v gfloat4;
          typedef __attribute__((address_space(1))) float gfloat;
          typedef __attribute__((address_space(1))) int gint;
          ((gfloat4*)(uintptr_t)a.dec.cand_box)[o] = bv;
          ((gfloat*)(uintptr_t)a.dec.cand_score)[o] = scv;
          ((gint*)(uintptr_t)a.dec.cand_label)[o] = 0;
          ((gint*)(uintptr_t)a.dec.cand_point)[o] = ptv;
        }
        pend_cnt = 0;
      }
    }
  };
  auto dec_flush = [&](int n) {
    if constexpr (DEC) {
      dec_copy_out();                                  // the previous reservation (its staging buffer becomes free)
      int base = 0;
      if (lane == 0)
        base = __hip_atomic_fetch_add((__attribute__((address_space(1))) int*)(uintptr_t)(a.dec.total + n), npend, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
      pend_base = base; pend_cnt = npend; pend_n = n;
      sbuf ^= 1;
    }
  };
  H2_T(0);
  int cur_j = -1;
  for (int item = blockIdx.x; item < a.h2_nitems; item += gridDim.x) {
    // work item = 4 consecutive chunks of one level (wave w takes chunk 4 * local + w)
    int j = 0;
#pragma unroll
    for (int i = 1; i < LFD_MAX_LEVELS; ++i)
      if (i < a.grp_n && item >= a.h2_item_start[i]) j = i;
    const int l = a.grp_levels[j];
    const HeadLevel& L = a.lv[l];
    // Level switch: the shared LDS filters (neck, final conv) are requested FIRST, into registers, and written to LDS
    // only after this wave's own per-image filter setup below -- ~9 k cycles of loads and scaling that do not touch
    // the shared area -- so their (cold) fetch latency and the two workgroup barriers overlap with useful work.
    const bool new_level = j != cur_j;
    const int nkl = L.cin / 16;              // 4 or 8 neck k-steps for this level
    half8 t_wn[NKNX], t_wf[(PASS == 3) ? (FT * NKH + 3) / 4 : 1];
    float t_bn = 0.f, t_bf = 0.f;
    if (new_level) {
#pragma unroll
      for (int k = 0; k < NKNX; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (!A1 && i < 4 * nkl * 64) t_wn[k] = L.wn[i];
      }
      t_bn = L.bn[wave * 32 + (lane & 31)];
      if constexpr (PASS == 3) {
#pragma unroll
        for (int k = 0; k < (FT * NKH + 3) / 4; ++k) {
          const int fq = wave + 4 * k;
          if (fq < FT * NKH) t_wf[k] = perm(L.wf, fq / NKH, fq % NKH, 1.f);
        }
        if (wave < FT) t_bf = L.bf[wave * 32 + (lane & 31)];
        scale = L.scale ? L.scale[0] : 1.f;
      }
    }
    const int gpi = a.grp_gpi[j], chg = a.h2_chg[j], cpi = (gpi + chg - 1) / chg;
    const int c = (item - a.h2_item_start[j]) * 4 + wave;
    const bool have_chunk = c < cpi * a.N;
    const int n = have_chunk ? c / cpi : 0, ci = c - n * cpi;
    const int g0 = ci * chg, g1 = (g0 + chg) < gpi ? (g0 + chg) : gpi;

    H2_T(1);
    // ---- this image's filters: GroupNorm scale folded into the rows, shift as a bias k-step
    {
      const int m = lane & 31;
      const float* t1 = a.ab1 + ((size_t)l * a.N + n) * HC * 2;
      const float* t2 = a.ab2 + ((size_t)l * a.N + n) * HC * 2;
      // (folded copies written once per (level, image) by k_gn_finalize: 36 plain 16-byte loads per filter instead of
      //  ~600 instructions of permute + scale + round -- 9.8 k cycles per chunk, 40 % of a two-group chunk at batch 1)
      const half8* f1 = (PASS >= 2 && L.w1f) ? L.w1f + (size_t)n * (4 * 9 * 64) : nullptr;
      const half8* f2 = (PASS == 3 && L.w2f) ? L.w2f + (size_t)n * (4 * 9 * 64) : nullptr;
      if constexpr (FOLD && A1) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
          for (int k = 0; k < NKH; k += 8) {
            const half8* p2 = f2 + (ct * 9 + k) * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"
                         "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"
                         "global_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:1024\n\t"
                         "global_load_dwordx4 %6, %9, off offset:2048\n\tglobal_load_dwordx4 %7, %9, off offset:3072"
                         : "=&a"(w2a[ct][k]), "=&a"(w2a[ct][k + 1]), "=&a"(w2a[ct][k + 2]), "=&a"(w2a[ct][k + 3]),
                           "=&a"(w2a[ct][k + 4]), "=&a"(w2a[ct][k + 5]), "=&a"(w2a[ct][k + 6]), "=&a"(w2a[ct][k + 7])
                         : "v"(p2), "v"(p2 + 4 * 64) : "memory");
          }
          s_wb[(4 + ct) * 64 + lane] = f2[(ct * 9 + 8) * 64 + lane];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if constexpr (FOLD) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
          for (int k = 0; k < NKH; k += 4) {
            const half8* p1 = f1 + (ct * 9 + k) * 64 + lane;
            const half8* p2 = f2 + (ct * 9 + k) * 64 + lane;
            asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"
                         "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"
                         "global_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:1024\n\t"
                         "global_load_dwordx4 %6, %9, off offset:2048\n\tglobal_load_dwordx4 %7, %9, off offset:3072"
                         : "=&a"(w1a[ct][k]), "=&a"(w1a[ct][k + 1]), "=&a"(w1a[ct][k + 2]), "=&a"(w1a[ct][k + 3]),
                           "=&a"(w2a[ct][k]), "=&a"(w2a[ct][k + 1]), "=&a"(w2a[ct][k + 2]), "=&a"(w2a[ct][k + 3])
                         : "v"(p1), "v"(p2) : "memory");
          }
          s_wb[ct * 64 + lane] = f1[(ct * 9 + 8) * 64 + lane];
          s_wb[(4 + ct) * 64 + lane] = f2[(ct * 9 + 8) * 64 + lane];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the AccVGPR loads are invisible to the compiler's counters
      } else
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (f1) {
#pragma unroll
          for (int k = 0; k < NKH; ++k) w1[ct][k] = f1[(ct * 9 + k) * 64 + lane];
          s_wb[ct * 64 + lane] = f1[(ct * 9 + 8) * 64 + lane];
        } else if (PASS == 1 && L.w1p) {
#pragma unroll
          for (int k = 0; k < NKH; ++k) w1[ct][k] = L.w1p[(ct * NKH + k) * 64 + lane];
        } else {
          const float sc1 = PASS >= 2 ? t1[(ct * 32 + m) * 2] : 1.f;
#pragma unroll
          for (int k = 0; k < NKH; ++k) w1[ct][k] = perm(L.w1, ct, k, sc1);
          if constexpr (PASS >= 2) s_wb[ct * 64 + lane] = bias_frag(t1[(ct * 32 + m) * 2 + 1]);
        }
        if constexpr (PASS >= 2) {
          if (f2) {
#pragma unroll
            for (int k = 0; k < NKH; ++k) w2[ct][k] = f2[(ct * 9 + k) * 64 + lane];
            s_wb[(4 + ct) * 64 + lane] = f2[(ct * 9 + 8) * 64 + lane];
          } else if (PASS == 2 && L.w2p) {
#pragma unroll
            for (int k = 0; k < NKH; ++k) w2[ct][k] = L.w2p[(ct * NKH + k) * 64 + lane];
          } else {
            const float sc2 = PASS == 3 ? t2[(ct * 32 + m) * 2] : 1.f;
#pragma unroll
            for (int k = 0; k < NKH; ++k) w2[ct][k] = perm(L.w2, ct, k, sc2);
            if constexpr (PASS == 3) s_wb[(4 + ct) * 64 + lane] = bias_frag(t2[(ct * 32 + m) * 2 + 1]);
          }
        }
      }
    }

    if (new_level) {
      __syncthreads();                       // everybody is done with the previous level's LDS filters
#pragma unroll
      for (int k = 0; k < NKNX; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (!A1 && i < 4 * nkl * 64) {
          const int cc = i / (nkl * 64), r = i - cc * (nkl * 64);
          s_wn[cc * (NKNX + 1) * 64 + r] = t_wn[k];
        }
      }
      s_wn[(wave * (NKNX + 1) + NKNX) * 64 + lane] = bias_frag(t_bn);     // neck bias step of cout tile `wave`
      if constexpr (PASS == 3) {
#pragma unroll
        for (int k = 0; k < (FT * NKH + 3) / 4; ++k) {
          const int fq = wave + 4 * k;
          if (fq < FT * NKH) s_wf[((fq / NKH) * (NKH + 1) + (fq % NKH)) * 64 + lane] = t_wf[k];
        }
        if (wave < FT) s_wf[(wave * (NKH + 1) + NKH) * 64 + lane] = bias_frag(t_bf);
      }
      __syncthreads();
      cur_j = j;
    }
    if (!have_chunk) continue;               // (no workgroup barrier below this point in the iteration)

    // GroupNorm partial sums of this lane: block b = 4ct + g covers channels 8b .. 8b+7, of which this lane holds 4
    // (registers 4g .. 4g+3 of tile ct).  Pass 1 has registers to spare and keeps two fp32 lanes per accumulator
    // (v_pk_add_f32 / v_pk_fma_f32: half the instructions); pass 2 (conv2's filter resident as well) keeps one.
    constexpr int SW = (PASS == 1) ? 2 : 1;
    float s[PASS < 3 ? 16 * SW : 1], ss[PASS < 3 ? 16 * SW : 1];
#pragma unroll
    for (int i = 0; i < (PASS < 3 ? 16 * SW : 1); ++i) { s[i] = 0.f; ss[i] = 0.f; }
    auto add_stats = [&](const f32x16& acc, int ct, float vm, bool masked) {
      if constexpr (PASS < 3) {
        if (masked) {        // wave-uniform: only the last group of an image has pixels past the end
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = acc[r] * vm;
            const int i = (ct * 4 + (r >> 2)) * SW + (SW == 2 ? (r & 1) : 0);
            s[i] += y;
            ss[i] = __builtin_fmaf(y, y, ss[i]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = acc[r];
            const int i = (ct * 4 + (r >> 2)) * SW + (SW == 2 ? (r & 1) : 0);
            s[i] += y;
            ss[i] = __builtin_fmaf(y, y, ss[i]);
          }
        }
      }
    };

    // x fragments of a group: lane (pixel, hh) reads channels 16q + 8hh .. +7 of its pixel (clamped inside the level)
    // (128-channel levels -- a few hundred pixels per image -- fetch their second 64 channels synchronously in the
    //  neck stage: no extra registers for the big 64-channel levels, the stall lands on otherwise idle CUs)
    half8 xq[NKN];
    const int cin = L.cin;
    const _Float16* ximg = L.x + (size_t)n * L.hw * cin + 8 * hh;
    auto load_x = [&](int g, int half) {
      int p = g * 32 + pix;
      p = p < L.hw ? p : L.hw - 1;
#pragma unroll
      for (int q = 0; q < NKN; ++q) xq[q] = *reinterpret_cast<const half8*>(ximg + (size_t)p * cin + 64 * half + 16 * q);
    };
    // tower-1 activations of this image: fragments q = 0..7 of group g at ((n * gpi + g) * 8 + q) * 64 + lane
    half8* a1img = L.a1 ? L.a1 + ((size_t)n * gpi) * (NKH * 64) + lane : nullptr;
    half8 a1n[A1 ? NKH : 1];
    auto load_a1 = [&](int g) {
      if constexpr (A1) {
#pragma unroll
        for (int q = 0; q < NKH; ++q) a1n[q] = a1img[((size_t)g * NKH + q) * 64];
      }
    };
    H2_T(2);
    if constexpr (A1) load_a1(g0); else load_x(g0, 0);
    for (int g = g0; g < g1; ++g) {
      if (g - g0 < 12) H2_T(3 + (g - g0));
      const int p0 = g * 32;
      const bool tail = p0 + 32 > L.hw;                 // wave-uniform
      const bool lane_ok = p0 + pix < L.hw;
      // Each stage goes channel tile by channel tile: 9 MFMAs into one 16-register accumulator, which is consumed
      // (statistics / ReLU + fp16 pack into the next stage's B operands) before the next tile starts -- at most
      // bq (inputs) + bqn (outputs) + one accumulator are live, instead of four accumulators per stage.
      half8 bq[NKH], bqn[NKH];
      if constexpr (A1) {
#pragma unroll
        for (int q = 0; q < NKH; ++q) bqn[q] = a1n[q];
        if (g + 1 < g1) load_a1(g + 1);
      } else {
      // ---- neck: relu(Wn x + bn).  The filter fragments come from LDS through a 4-deep register ring: read right in front
      //      of their MFMA (what the compiler emits on its own) every one of the 20 / 36 k-steps waited a full LDS round
      //      trip -- ~1300 cycles per group, a quarter of pass 1.
#ifndef LFD_H2_WPD
#define LFD_H2_WPD 3        // (5 measured the same: 15.7-15.85 k images/s either way, same session)
#endif
      constexpr int WPD = LFD_H2_WPD;
      half8 wring[WPD + 1];
      if (cin == 64) {
        auto fidx = [](int i) { return (i / 5) * (NKNX + 1) + ((i % 5) == 0 ? NKNX : (i % 5) - 1); };   // bias step, then q = 0..3
#pragma unroll
        for (int i = 0; i < WPD; ++i) wring[i] = s_wn[fidx(i) * 64 + lane];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 20; ++i) {
          if (i + WPD < 20) wring[(i + WPD) % (WPD + 1)] = s_wn[fidx(i + WPD) * 64 + lane];
          const int ct = i / 5, j = i % 5;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wring[i % (WPD + 1)], j == 0 ? ones : xq[j == 0 ? 0 : j - 1], j == 0 ? zero : acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (j == 4) { bq[2 * ct] = to_b(acc, 0); bq[2 * ct + 1] = to_b(acc, 1); }
        }
      } else {
        f32x16 acc4[4];
        auto fidx0 = [](int i) { return (i / 5) * (NKNX + 1) + ((i % 5) == 0 ? NKNX : (i % 5) - 1); };   // first 64 channels
        auto fidx1 = [](int i) { return (i / 4) * (NKNX + 1) + NKN + (i % 4); };                       // second 64 channels
#pragma unroll
        for (int i = 0; i < WPD; ++i) wring[i] = s_wn[fidx0(i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 20; ++i) {
          if (i + WPD < 20) wring[(i + WPD) % (WPD + 1)] = s_wn[fidx0(i + WPD) * 64 + lane];
          const int ct = i / 5, j = i % 5;
          acc4[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wring[i % (WPD + 1)], j == 0 ? ones : xq[j == 0 ? 0 : j - 1], j == 0 ? zero : acc4[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        load_x(g, 1);
#pragma unroll
        for (int i = 0; i < WPD; ++i) wring[i] = s_wn[fidx1(i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i + WPD < 16) wring[(i + WPD) % (WPD + 1)] = s_wn[fidx1(i + WPD) * 64 + lane];
          const int ct = i / 4, q = i % 4;
          acc4[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wring[i % (WPD + 1)], xq[q], acc4[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (q == 3) { bq[2 * ct] = to_b(acc4[ct], 0); bq[2 * ct + 1] = to_b(acc4[ct], 1); }
        }
      }
      if (g + 1 < g1) load_x(g + 1, 0);    // next group's pixels: requested now, consumed at the top of the next iteration
      // ---- conv1 (no bias: norm follows, lfd_head.py:97; passes 2, 3: GN1 folded in, its shift on the bias step)
      half8 wbv[PASS >= 2 ? 4 : 1];       // the stage's four bias-step fragments: one LDS round trip, not one per cout tile
      if constexpr (PASS >= 2) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wbv[ct] = s_wb[ct * 64 + lane];
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x16 acc;
        if constexpr (PASS >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbv[ct], ones, zero, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NKH; ++q)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(FOLD ? __builtin_bit_cast(half8, w1a[FOLD ? ct : 0][q]) : w1[FOLD ? 0 : ct][q], bq[q], (PASS >= 2 || q > 0) ? acc : zero, 0, 0, 0);
        if constexpr (PASS == 1) add_stats(acc, ct, lane_ok ? 1.f : 0.f, tail);
        else { bqn[2 * ct] = to_b(acc, 0); bqn[2 * ct + 1] = to_b(acc, 1); }
      }
      }   // !A1
      if constexpr (PASS == 2) {
        if (a1img) {           // wave-uniform: hand conv2's operands to the output pass
#pragma unroll
          for (int q = 0; q < NKH; ++q) a1img[((size_t)g * NKH + q) * 64] = bqn[q];
        }
      }
      if constexpr (PASS >= 2) {
        // ---- conv2 (pass 3: GN2 folded in)
        half8 wbv2[PASS == 3 ? 4 : 1];
        if constexpr (PASS == 3) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) wbv2[ct] = s_wb[(4 + ct) * 64 + lane];
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          f32x16 acc;
          if constexpr (PASS == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbv2[ct], ones, zero, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < NKH; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(FOLD ? __builtin_bit_cast(half8, w2a[FOLD ? ct : 0][q]) : w2[FOLD ? 0 : ct][q], bqn[q], (PASS == 3 || q > 0) ? acc : zero, 0, 0, 0);
          if constexpr (PASS == 2) add_stats(acc, ct, lane_ok ? 1.f : 0.f, tail);
          else { bq[2 * ct] = to_b(acc, 0); bq[2 * ct + 1] = to_b(acc, 1); }
        }
        if constexpr (PASS == 3) {
          const size_t row = (size_t)n * a.P + L.p_off + p0 + pix;
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            f32x16 fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_wf[(f * (NKH + 1) + NKH) * 64 + lane], ones, zero, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NKH; ++q)
              fa = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_wf[(f * (NKH + 1) + q) * 64 + lane], bq[q], fa, 0, 0, 0);
            if constexpr (DEC) {
              // ---- threshold + decode (front half of lfd_detect_batched: lfd.py:449-499, nms.py:202-207).
              //      Rows 0..3 (regression) sit in the pixel's hh = 0 lane, row 4 (the class logit) in its hh = 1 lane.
              //      Common case (no candidate in the group): one compare against a conservative logit bound + a ballot;
              //      the exact test sigma(x) > thr, the partner's regression rows and the decode only behind it.
              const bool pre = hh == 1 && lane_ok && fa[0] > a.dec.logit_lo;
              if (__ballot(pre)) {
                float sc = 0.f;
                bool cand = false;
                if (pre) { sc = lfd_sigmoidf_ref(fa[0]); cand = sc > a.dec.score_thr; }
                const unsigned long long cm = __ballot(cand);
                if (cm) {
                  const float r0 = __shfl(fa[0] * scale, pix, 64), r1 = __shfl(fa[1] * scale, pix, 64);
                  const float r2 = __shfl(fa[2] * scale, pix, 64), r3 = __shfl(fa[3] * scale, pix, 64);
                  if (cand) {
                    const int slot = npend + (int)__popcll(cm & ((1ull << lane) - 1ull));
                    const int q = p0 + pix, lw = a.dec.w[l];
                    const int iy = q / lw, ix = q - iy * lw;
                    const float4 bx = lfd_decode_core(a.dec.decode_mode, r0, r1, r2, r3, (float)(ix * a.dec.stride[l]),
                                                      (float)(iy * a.dec.stride[l]), a.dec.m[l], a.dec.meta[n * 3 + 0],
                                                      a.dec.meta[n * 3 + 1], a.dec.meta[n * 3 + 2]);
                    f32x4v bv; bv[0] = bx.x; bv[1] = bx.y; bv[2] = bx.z; bv[3] = bx.w;
                    const uint32_t sa = st_base + sbuf * (DEC_ST * 24);
                    // (not volatile, no memory clobber: ordered against the read-back only through the token operand)
                    asm("ds_write_b128 %1, %2\n\tds_write_b32 %3, %4\n\tds_write_b32 %5, %6"
                        : "+v"(st_tok)
                        : "v"(sa + slot * 16), "v"(bv), "v"(sa + DEC_ST * 16 + slot * 4), "v"(sc),
                          "v"(sa + DEC_ST * 20 + slot * 4), "v"(L.p_off + q));
                  }
                  npend += (int)__popcll(cm);
                  if (npend > DEC_ST - 32) { dec_flush(n); npend = 0; }
                }
              }
            }
            if (!DEC && lane_ok) {
              // final rows: [reg x reg_rows][cls x cls_rows]; lane (pixel, hh) holds rows 8g + 4hh + j
              if (a.reg_rows == 4 && f == 0 && hh == 0)
                *reinterpret_cast<float4*>(a.out_reg + row * 4) = make_float4(fa[0] * scale, fa[1] * scale, fa[2] * scale, fa[3] * scale);
              if (a.reg_rows == 4 && a.cls_rows == 1) {
                if (f == 0 && hh == 1) a.out_cls[row * a.CC] = fa[0];
              } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                  const int ch = f * 32 + 8 * (i >> 2) + 4 * hh + (i & 3) - a.reg_rows;
                  if (ch >= 0 && ch < a.cls_rows) a.out_cls[row * a.CC + ch] = fa[i];
                }
              }
            }
          }
        }
      }
      // ---- statistics tile complete (second group of the tile, or the image's last group): transpose-reduce the
      //      32 per-lane sums over the 64 lanes and write the tile's slot (chunks start on tile boundaries: g0 even)
      if constexpr (PASS < 3) {
        if (((g - g0) & 1) || g + 1 == g1) {
          float v[32];
#pragma unroll
          for (int b = 0; b < 16; ++b) {
            v[2 * b] = SW == 2 ? s[2 * b] + s[2 * b + 1] : s[b];
            v[2 * b + 1] = SW == 2 ? ss[2 * b] + ss[2 * b + 1] : ss[b];
          }
          const float tot = wave_sum_transpose32(v, lane);      // lane l (and l + 32): total of v[l & 31]
          const int gg = 1 << (a.gshift - 3);                   // 8-channel blocks per GroupNorm group
          float r = tot;
          for (int m = 1; m < gg; m <<= 1) r += __shfl_xor(r, 2 * m, 64);
          float* dst = a.part + ((size_t)L.tile_start + (size_t)n * L.tiles_per_img + (g >> 1)) * (HC >> a.gshift) * 2;
          if (lane < 32 && (((lane >> 1) & (gg - 1)) == 0)) dst[((lane >> 1) >> (a.gshift - 3)) * 2 + (lane & 1)] = r;
#pragma unroll
          for (int i = 0; i < 16 * SW; ++i) { s[i] = 0.f; ss[i] = 0.f; }
        }
      }
    }
    if constexpr (DEC) { if (npend) { dec_flush(n); npend = 0; } }
    H2_T(16);
    H2_T(17);
  }
  dec_copy_out();
}

#ifdef LFD_H2_TIMING
}  // namespace
extern "C" __attribute__((visibility("default"))) int lfd_debug_h2_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_h2_dbg), sizeof(unsigned long long) * 96);
}
namespace {
#endif

template <int PASS, int FT, bool DEC = false, bool FOLD = false, bool A1 = false>
int launch_head2(const HeadArgs& a, hipStream_t st) {
  constexpr int LDS = (4 * 9 + ((PASS == 3) ? FT * 9 : 0) + 4 * 8) * 1024;
  static unsigned long long done_mask = 0;
  const int done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(done_mask, done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head2<PASS, FT, DEC, FOLD, A1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(done_mask, done_dev);
  }
  constexpr int cap = 256 * H2_WPS(PASS);
  int blocks = a.h2_nitems < cap ? a.h2_nitems : cap;
  if (blocks < 1) return LFD_OK;
  hipLaunchKernelGGL((k_head2<PASS, FT, DEC, FOLD, A1>), dim3(blocks), dim3(256), LDS, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// per (level, image, group): combine tile partials in fp64 (fixed order), emit per-channel (scale, shift)
struct FinalizeArgs {
  int tile_start[LFD_MAX_LEVELS], tiles_per_img[LFD_MAX_LEVELS], hw[LFD_MAX_LEVELS];
  int tile_stride[LFD_MAX_LEVELS];   // 1 (every 64-pixel tile has its own slot)
  const float* gamma[LFD_MAX_LEVELS];
  const float* beta[LFD_MAX_LEVELS];
  const float* part;
  float* ab;   // [L][N][128][2]
  int N, ngroups, gsize;
  float eps;
  const _Float16* wsrc[LFD_MAX_LEVELS];   // fold: packed tower conv (standard layout) of each level, or nullptr
  half8* wdst[LFD_MAX_LEVELS];            //       [N][4][9][64] folded, K-permuted fragments (see k_head2)
};

__global__ __launch_bounds__(256) void k_gn_finalize(FinalizeArgs f) {
  // block = (image n, level l, cout tile ct): the statistics of the 32 channels [32 ct, 32 ct + 32) -- 32 / gsize groups, two
  // values (sum, sum of squares) each -- combined over the level's tile partials in fp64 in a fixed order (deterministic), their
  // per-channel (scale, shift), and this cout tile's share of the folded filter.  thread = (tile lane tl, value v): coalesced
  // row reads, four independent accumulators per thread.  (One block per (image, level) took 12.6 us once it also folded.)
  __shared__ double sm[256];
  __shared__ double s_tot[64];
  __shared__ float s_ab[32][2];
  const int n = blockIdx.x, l = blockIdx.y, ct = blockIdx.z;
  const int nv = f.ngroups * 2;                  // values per tile row
  const int nvc = 64 / f.gsize;                  // values of this cout tile: 2 * (32 / gsize), 2 .. 64
  int nvp = 2;
  while (nvp < nvc) nvp <<= 1;                   // (gsize is a power of two: nvp == nvc)
  const int TLN = 256 / nvp;                     // tile lanes
  const int tl = threadIdx.x / nvp, v0 = threadIdx.x - tl * nvp;
  // fold: this thread's source fragments (k-steps of cout tile ct) are requested up front -- their L2 round trip hides under
  // the statistics reduction below.  Items i = tid + 256 j < 9 * 64: k = i / 64 (8 = the bias fragment), lane = i % 64.
  const bool fold = f.wsrc[l] && f.wdst[l];
  half4 f_lo[3], f_hi[3];
  if (fold) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = threadIdx.x + 256 * j, k = i >> 6, lane = i & 63, m = lane & 31, hk = lane >> 5;
      if (k < 8) {
        const _Float16* base = f.wsrc[l] + (size_t)(ct * 8 + k) * 64 * 8;
        f_lo[j] = *reinterpret_cast<const half4*>(base + m * 8 + 4 * hk);
        f_hi[j] = *reinterpret_cast<const half4*>(base + (m + 32) * 8 + 4 * hk);
      }
    }
  }
  // affine parameters of this thread's group (threads < nvc / 2 own one group each): requested now, so that their round trip
  // runs beside the statistics reduction instead of behind it (groups of up to 8 channels; larger ones load in place)
  float ga_pre[8], be_pre[8];
  if ((int)threadIdx.x < nvc / 2 && f.gsize <= 8) {
    const int g = ct * (nvc / 2) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = g * f.gsize + (j < f.gsize ? j : 0);
      ga_pre[j] = f.gamma[l][c];
      be_pre[j] = f.beta[l][c];
    }
  }
  const int t0 = f.tile_start[l] + n * f.tiles_per_img[l];
  const int st = f.tile_stride[l];
  const int T = (f.tiles_per_img[l] + st - 1) / st;   // slots that carry data
  double acc = 0.0;
  if (v0 < nvc) {
    const float* col = f.part + (size_t)t0 * nv + ct * nvc + v0;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int t = tl;
    // 16 loads requested per thread before the first add (round 3; 4 before): the 1080p level has 507 tile rows per image =
    // 16 per thread, so the reduction is ONE L2 round trip instead of four dependent ones.  The summation order (four
    // accumulators striding the rows, combined pairwise) is unchanged: identical bits.
    for (; t + 15 * TLN < T; t += 16 * TLN) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = col[(size_t)((t + j * TLN) * st) * nv];
#pragma unroll
      for (int j = 0; j < 16; j += 4) { a0 += (double)v[j]; a1 += (double)v[j + 1]; a2 += (double)v[j + 2]; a3 += (double)v[j + 3]; }
    }
    for (; t + 3 * TLN < T; t += 4 * TLN) {     // 4 independent loads in flight per thread
      a0 += (double)col[(size_t)(t * st) * nv];
      a1 += (double)col[(size_t)((t + TLN) * st) * nv];
      a2 += (double)col[(size_t)((t + 2 * TLN) * st) * nv];
      a3 += (double)col[(size_t)((t + 3 * TLN) * st) * nv];
    }
    for (; t < T; t += TLN) a0 += (double)col[(size_t)(t * st) * nv];
    acc = (a0 + a1) + (a2 + a3);
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < nvc) {                       // fixed-order combine over the tile lanes
    double sum = 0.0;
    for (int i = 0; i < TLN; ++i) sum += sm[i * nvp + threadIdx.x];
    s_tot[threadIdx.x] = sum;
  }
  __syncthreads();
  if (threadIdx.x < nvc / 2) {                   // one group each
    const int g = ct * (nvc / 2) + threadIdx.x;
    const double sum = s_tot[2 * threadIdx.x], ss = s_tot[2 * threadIdx.x + 1];
    const double cnt = (double)f.hw[l] * f.gsize;
    const double mean = sum / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)f.eps);
    for (int j = 0; j < f.gsize; ++j) {
      const int c = g * f.gsize + j;
      const float gam = f.gsize <= 8 ? ga_pre[j & 7] : f.gamma[l][c], bet = f.gsize <= 8 ? be_pre[j & 7] : f.beta[l][c];
      const double sc = (double)gam * rstd;
      float* o = f.ab + (((size_t)l * f.N + n) * HC + c) * 2;
      o[0] = (float)sc;
      o[1] = (float)((double)bet - mean * sc);
      s_ab[c - ct * 32][0] = o[0]; s_ab[c - ct * 32][1] = o[1];
    }
  }
  __syncthreads();
  // ---- fold: this (level, image)'s copy of the tower conv with the scale in its rows (one fp16 rounding, the same
  //      arithmetic as k_head2's own fold) and the shift as a bias fragment, in k_head2's K-permuted register layout:
  //      element e of lane (m, hk) of fragment (ct, k) = W[32ct + m][16k + 8(e >> 2) + 4hk + (e & 3)]
  if (fold) {
    half8* __restrict__ dst = f.wdst[l] + (size_t)n * (4 * 9 * 64) + (size_t)ct * 9 * 64;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = threadIdx.x + 256 * j, k = i >> 6, lane = i & 63, m = lane & 31, hk = lane >> 5;
      if (i >= 9 * 64) break;
      half8 r;
      if (k < 8) {
        const float rs = s_ab[m][0];
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = (_Float16)((float)f_lo[j][e] * rs); r[4 + e] = (_Float16)((float)f_hi[j][e] * rs); }
      } else {
        const float v = s_ab[m][1];
        const _Float16 bh = (_Float16)v, bl = (_Float16)(v - (float)bh);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (_Float16)0.f;
        if (!hk) { r[0] = bh; r[1] = bl; }
      }
      dst[i] = r;
    }
  }
}

template <int CIN, int PASS, int FT>
int launch_head(const HeadArgs& a, hipStream_t st) {
  constexpr int LDS = 32768 + 2 * TPX * HC * 2 + 512 * 4 + 64 * 4 + 128 * 4 + (PASS == 3 ? FT * 8 * 1024 : 0);
  static unsigned long long done_mask = 0;
  const int done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(done_mask, done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head<CIN, PASS, FT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(done_mask, done_dev);
  }
  int blocks = a.grp_ntiles < 512 ? a.grp_ntiles : 512;
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

// levels whose passes run in k_head2 (and whose statistics therefore occupy one slot per chunk)
static bool head2_enabled() {
  return lfd_tune(LFD_TUNE_HEAD2) != 0;
}
int fill_levels(const lfd_head_desc_t* d, int* tile_start, int* tiles_per_img, int* ntiles) {
  if (!d || d->num_levels < 1 || d->num_levels > LFD_MAX_LEVELS || d->n < 1) return LFD_ERR_INVALID_ARGUMENT;
  int t = 0;
  for (int i = 0; i < d->num_levels; ++i) {
    if (d->level_hw[i] < 0) return LFD_ERR_INVALID_ARGUMENT;
    tile_start[i] = t;
    tiles_per_img[i] = (d->level_hw[i] + TPX - 1) / TPX;
    t += tiles_per_img[i] * d->n;
  }
  *ntiles = t;
  return LFD_OK;
}

}  // namespace

extern "C" {

size_t lfd_head_partial_floats(const lfd_head_desc_t* d) {
  int ts[LFD_MAX_LEVELS], tp[LFD_MAX_LEVELS], nt = 0;
  if (fill_levels(d, ts, tp, &nt) != LFD_OK || d->num_groups < 1) return 0;
  return (size_t)nt * d->num_groups * 2;
}

static int head_forward_impl(const lfd_head_desc_t* d, int32_t pass, const lfd_head_level_ptrs_t* lv,
                             const float* ab1, const float* ab2, float* partial, float* out_cls, float* out_reg,
                             const void* zeros, const LfdAppendTarget* dec, hipStream_t st) {
  if (!d || !lv || !zeros) return LFD_ERR_INVALID_ARGUMENT;
  if (d->head_channels != HC) return LFD_ERR_UNSUPPORTED;
  const int gsize = d->num_groups > 0 ? HC / d->num_groups : 0;
  if (d->num_groups < 1 || d->num_groups > HC || gsize * d->num_groups != HC || (gsize & (gsize - 1)) || gsize > 32)
    return LFD_ERR_UNSUPPORTED;
  int gshift = 0;
  while ((1 << gshift) < gsize) ++gshift;
  HeadArgs a{};
  int ts[LFD_MAX_LEVELS], tp[LFD_MAX_LEVELS];
  int rc = fill_levels(d, ts, tp, &a.ntiles);
  if (rc != LFD_OK) return rc;
  a.nlev = d->num_levels;
  for (int i = 0; i < d->num_levels; ++i) {
    if (d->level_cin[i] != 64 && d->level_cin[i] != 128) return LFD_ERR_UNSUPPORTED;
    if (!lv[i].x || !lv[i].wn_packed || !lv[i].bn || !lv[i].w1_packed) return LFD_ERR_INVALID_ARGUMENT;
    if (pass >= 2 && !lv[i].w2_packed) return LFD_ERR_INVALID_ARGUMENT;
    if (pass == 3 && (!lv[i].wf_packed || !lv[i].bf)) return LFD_ERR_INVALID_ARGUMENT;
    HeadLevel& L = a.lv[i];
    L.x = (const _Float16*)lv[i].x; L.wn = (const half8*)lv[i].wn_packed; L.bn = lv[i].bn;
    L.w1 = (const half8*)lv[i].w1_packed; L.w2 = (const half8*)lv[i].w2_packed;
    L.wf = (const half8*)lv[i].wf_packed; L.bf = lv[i].bf; L.scale = lv[i].scale;
    L.w1f = (const half8*)lv[i].w1_folded; L.w2f = (const half8*)lv[i].w2_folded;
    L.a1 = (half8*)lv[i].tower1_out;
    L.w1p = (const half8*)lv[i].w1_perm; L.w2p = (const half8*)lv[i].w2_perm;
    L.cin = d->level_cin[i]; L.hw = d->level_hw[i]; L.p_off = d->level_point_offset[i];
    L.tile_start = ts[i]; L.tiles_per_img = tp[i];
  }
  a.ab1 = ab1; a.ab2 = ab2; a.part = partial; a.out_cls = out_cls; a.out_reg = out_reg;
  a.N = d->n; a.P = d->total_points; a.CC = d->cls_channels; a.reg_rows = d->final_reg_rows; a.cls_rows = d->final_cls_rows;
  a.gshift = gshift; a.zeros = (const _Float16*)zeros;
  if (pass < 3) {
    if (!partial) return LFD_ERR_INVALID_ARGUMENT;
    if (gshift < 3 &&
        hipMemsetAsync(partial, 0, sizeof(float) * lfd_head_partial_floats(d), st) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
  }
  if (pass >= 2 && !ab1) return LFD_ERR_INVALID_ARGUMENT;
  if (pass == 3 && (!ab2 || (!dec && (!out_cls || !out_reg)))) return LFD_ERR_INVALID_ARGUMENT;
  if ((d->final_reg_rows != 0 && d->final_reg_rows != 4) || d->final_cls_rows < 0) return LFD_ERR_INVALID_ARGUMENT;
  const int ft = (d->final_reg_rows + d->final_cls_rows + 31) / 32;
  if (pass == 3 && (ft < 1 || ft > 2)) return LFD_ERR_UNSUPPORTED;
  if (pass < 1 || pass > 3) return LFD_ERR_INVALID_ARGUMENT;
  if (head2_enabled() && gshift >= 3) {
    // wave-per-32-pixels kernel, all levels in one launch: work items = 4 consecutive chunks of one level
    a.grp_n = 0;
    a.h2_nitems = 0;
    for (int i = 0; i < d->num_levels; ++i) {
      a.grp_levels[a.grp_n] = i;
      a.grp_gpi[a.grp_n++] = (d->level_hw[i] + 31) / 32;
    }
    h2_plan_chunks(d->n, a.grp_n, a.grp_gpi, a.h2_chg, H2_WPS(pass));
    for (int j = 0; j < a.grp_n; ++j) {
      const int cpi = (a.grp_gpi[j] + a.h2_chg[j] - 1) / a.h2_chg[j];
      a.h2_item_start[j] = a.h2_nitems;
      a.h2_nitems += (cpi * d->n + 3) / 4;
    }
    if (pass == 1) return launch_head2<1, 1>(a, st);
    if (pass == 2) return launch_head2<2, 1>(a, st);
    bool fold = true;      // every level brings both folded filters -> AccVGPR-resident variant
    for (int i = 0; i < d->num_levels; ++i) fold = fold && lv[i].w1_folded && lv[i].w2_folded;
    fold = fold && lfd_tune(LFD_TUNE_H2_AGPR);
    bool a1 = fold;        // ... and pass 2 left conv2's operands behind -> no neck / conv1 recompute
    for (int i = 0; i < d->num_levels; ++i) a1 = a1 && lv[i].tower1_out;
    a1 = a1 && lfd_tune(LFD_TUNE_H2_A1);
    if (dec) {
      if (ft != 1 || d->final_reg_rows != 4 || d->final_cls_rows != 1) return LFD_ERR_UNSUPPORTED;
      a.dec = *dec;
      if (a1) return launch_head2<3, 1, true, true, true>(a, st);
      return fold ? launch_head2<3, 1, true, true>(a, st) : launch_head2<3, 1, true>(a, st);
    }
    if (ft == 2) return a1 ? launch_head2<3, 2, false, true, true>(a, st) : (fold ? launch_head2<3, 2, false, true>(a, st) : launch_head2<3, 2>(a, st));
    return a1 ? launch_head2<3, 1, false, true, true>(a, st) : (fold ? launch_head2<3, 1, false, true>(a, st) : launch_head2<3, 1>(a, st));
  }
  if (dec) return LFD_ERR_UNSUPPORTED;
  // one launch per tap-channel class (64 / 128): homogeneous tiles, compile-time ring geometry
  for (int cin = 64; cin <= 128; cin += 64) {
    a.grp_n = 0;
    a.grp_ntiles = 0;
    for (int i = 0; i < d->num_levels; ++i)
      if (d->level_cin[i] == cin) {
        a.grp_levels[a.grp_n] = i;
        a.grp_tile_start[a.grp_n] = a.grp_ntiles;
        a.grp_ntiles += tp[i] * d->n;
        ++a.grp_n;
      }
    if (a.grp_ntiles == 0) continue;
    rc = cin == 64 ? dispatch_head<64>(pass, ft, a, st) : dispatch_head<128>(pass, ft, a, st);
    if (rc != LFD_OK) return rc;
  }
  return LFD_OK;
}

int lfd_head_forward_f16(const lfd_head_desc_t* d, int32_t pass, const lfd_head_level_ptrs_t* lv,
                         const float* ab1, const float* ab2, float* partial, float* out_cls, float* out_reg,
                         const void* zeros, lfd_stream_t stream) {
  return head_forward_impl(d, pass, lv, ab1, ab2, partial, out_cls, out_reg, zeros, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int lfd_head_forward_decode_f16(const lfd_head_desc_t* d, const lfd_head_level_ptrs_t* lv, const float* ab1,
                                const float* ab2, float* out_cls, float* out_reg, const void* zeros,
                                const lfd_detect_desc_t* det, const float* img_meta, void* det_workspace,
                                size_t det_workspace_bytes, lfd_stream_t stream) {
  if (!d || !det) return LFD_ERR_INVALID_ARGUMENT;
  // one foreground class scored with a sigmoid; the detector's levels are the head's levels
  if (det->num_classes != 1 || det->num_cls_channels != 1 || det->score_mode != 0 || d->cls_channels != 1)
    return LFD_ERR_UNSUPPORTED;
  if (det->num_levels != d->num_levels) return LFD_ERR_INVALID_ARGUMENT;
  int p = 0;
  for (int i = 0; i < d->num_levels; ++i) {
    if (det->level_h[i] * det->level_w[i] != d->level_hw[i] || d->level_point_offset[i] != p) return LFD_ERR_INVALID_ARGUMENT;
    p += d->level_hw[i];
  }
  if (p != d->total_points) return LFD_ERR_INVALID_ARGUMENT;
  LfdAppendTarget dec;
  const int rc = lfd_detect_bind_append(det, d->n, img_meta, det_workspace, det_workspace_bytes, &dec);
  if (rc != LFD_OK) return rc;
  return head_forward_impl(d, 3, lv, ab1, ab2, nullptr, out_cls, out_reg, zeros, &dec, reinterpret_cast<hipStream_t>(stream));
}

int lfd_groupnorm_finalize(const lfd_head_desc_t* d, const float* partial, const float* const* gamma,
                           const float* const* beta, float eps, float* ab, lfd_stream_t stream) {
  return lfd_groupnorm_finalize_fold(d, partial, gamma, beta, eps, ab, nullptr, 0, stream);
}

int lfd_groupnorm_finalize_fold(const lfd_head_desc_t* d, const float* partial, const float* const* gamma,
                                const float* const* beta, float eps, float* ab, const lfd_head_level_ptrs_t* lv,
                                int32_t which, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !partial || !gamma || !beta || !ab) return LFD_ERR_INVALID_ARGUMENT;
  if (lv && which != 1 && which != 2) return LFD_ERR_INVALID_ARGUMENT;
  FinalizeArgs f{};
  int nt = 0;
  int rc = fill_levels(d, f.tile_start, f.tiles_per_img, &nt);
  if (rc != LFD_OK) return rc;
  if (d->num_groups < 1 || d->num_groups > HC || HC % d->num_groups) return LFD_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < d->num_levels; ++i) {
    if (!gamma[i] || !beta[i]) return LFD_ERR_INVALID_ARGUMENT;
    f.hw[i] = d->level_hw[i]; f.gamma[i] = gamma[i]; f.beta[i] = beta[i];
    f.tile_stride[i] = 1;
    if (lv) {
      f.wsrc[i] = (const _Float16*)(which == 1 ? lv[i].w1_packed : lv[i].w2_packed);
      f.wdst[i] = (half8*)(which == 1 ? lv[i].w1_folded : lv[i].w2_folded);
    }
  }
  f.part = partial; f.ab = ab; f.N = d->n; f.ngroups = d->num_groups; f.gsize = HC / d->num_groups; f.eps = eps;
  if (f.gsize > 32) return LFD_ERR_UNSUPPORTED;      // a cout tile of 32 channels holds whole groups
  hipLaunchKernelGGL(k_gn_finalize, dim3(d->n, d->num_levels, 4), dim3(256), 0, st, f);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
