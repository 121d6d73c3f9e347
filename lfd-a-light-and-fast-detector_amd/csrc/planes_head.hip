// csrc/planes_head.hip -- k_pl_head<MODE>: the neck + head 1x1 convs of the planes mode (simple_neck.py:67-74,
// lfd_head.py:88-139,164-185; the [N,P,C] re-layout of lfd.py:526-542) as LEAN kernels over FLAT pixel tiles (round 6).
//
// The generic conv of planes_impl.h ran these layers at ~3 TB/s and 0.05-0.28 MFMA utilisation: two workgroups per CU, and the
// VALU the bottleneck (profiles/r05_precise_pmc_sq_counters.txt; ~770 VALU instructions per wave and 64-pixel tile) -- 2-D
// tiles with per-lane 64-bit addresses and validity selects for every DMA and store, a staging round trip through LDS for
// every output, a hi/lo split of every intermediate on the way out and a join on the way back in.  A 1x1 conv has no
// geometry: here a pyramid level of an image is a FLAT list of pixels,
//   * tile = 64 consecutive pixels: every DMA is a scalar base + a per-wave constant lane offset, every store likewise;
//   * the intermediates of the tower -- pre-GroupNorm conv outputs, private to these kernels -- are plain fp32 [N][P][128]
//     (the bytes of a plane pair): the producer stores its accumulators straight from registers, 16 bytes per lane and
//     32-byte runs per pixel that the four stores of a wave complete to full 128-byte lines (no split, no staging, no
//     barrier); the consumer's GroupNorm + ReLU pass reads fp32 (no join) and forms the hi / lo operand planes in LDS;
//   * GroupNorm sums come from the fp32 accumulators in the producer's epilogue (a wave owns its 4 groups of 8 channels: no
//     cross-wave reduction), tiles of one image accumulated in fp64, flushed as order-independent 64-bit fixed-point atomics
//     (the consumer's side of planes_impl.h, unchanged: bit-reproducible statistics).
// MODE 0 (A): tap planes -> neck 1x1 + bias + ReLU -> (LDS) -> first tower 1x1 + bias -> fp32 t + sums
// MODE 1 (B): fp32 t -> GroupNorm + ReLU (LDS) -> tower 1x1 + bias -> fp32 t' + sums
// MODE 2 (C): fp32 t -> GroupNorm + ReLU (LDS) -> cls | reg 1x1 + bias (+ Scale) -> fp32 [N,P,C'] / [N,P,4] at the level's offset
// MODE 3 (D): planes [N][P][128] (a neck output) -> tower 1x1 + bias -> fp32 t + sums  (separate cls / reg towers: TT100K)
// All pyramid levels of a launch in one persistent grid (each level its own filters, lfd_head.py:88-139).
#include "planes_impl.h"

namespace pl {

struct PhLevel {
  const void* in;                 // A: tap planes hi [N][P][cin] fp16, lo `in_plane` halfs behind; B / C: fp32 [N][P][128]
  long in_plane;
  float* out;                     // A / B: fp32 [N][P][128]
  const half8* w0;                // A: neck [2][4][cin/16][64]; B / C: the conv [2][nslab][8][64]
  const float* b0;
  const half8* w1;                // A: first tower conv [2][4][8][64]
  const float* b1;
  unsigned long long* gn_out;     // A / B: [kGnRep][N][16][2] fixed-point sums of the values written to `out`
  const unsigned long long* gn_in;  // B / C: sums of the producer of `in`
  const float* gamma;
  const float* beta;
  float* f_out0;                  // C: channels [0, f_c0) -> f_out0[n * f_img0 + pixel * f_c0 + c]
  float* f_out1;                  //    channels [f_c0, f_c0 + f_c1) -> f_out1[n * f_img1 + pixel * f_c1 + c - f_c0] * scale1
  const float* scale1;
  int P, tiles_per_img, tile_start, pad_;
};
struct PhArgs {
  PhLevel lv[LFD_MAX_LEVELS];
  const _Float16* zeros;
  long w_plane0, w_plane1;        // half8 units between the hi and lo plane of w0 / w1
  long f_img0, f_img1;
  int n_levels, N, ntiles, relu0;
  int f_c0, f_c1;
  float eps;
};

template <int MODE, int CIN, int NSLAB>
struct PhCfg {
  static constexpr int TPX = 64;                                   // pixels per tile
  static constexpr bool PLANES_IN = MODE == 0 || MODE == 3;        // the input tile is a plane pair (A: the tap; D: a neck output)
  static constexpr int NK0 = (PLANES_IN ? CIN : 128) / 16;
  static constexpr int IN_PIXB = CIN * 2;                          // A: bytes per pixel and plane of the tap tile
  static constexpr int IN_PLANE = TPX * IN_PIXB;
  static constexpr int NBUF = ((MODE == 0 && CIN == 64) || MODE == 3) ? 2 : 1;    // A on the large levels, D: double-buffered tap tiles
  static constexpr int RAW_BYTES = TPX * 512;                      // B / C: the fp32 tile as it lands
  static constexpr int OP_PLANE = TPX * 256;                       // one operand plane [64 px][128 ch] fp16
  static constexpr int IN_OFF = 0;
  static constexpr int IN_BYTES = PLANES_IN ? NBUF * 2 * IN_PLANE : RAW_BYTES;
  static constexpr int OP_OFF = IN_OFF + IN_BYTES;                 // A: the neck's output (operand of the tower conv); B / C: GN output
  static constexpr int BIAS_OFF = OP_OFF + (MODE == 3 ? 0 : 2 * OP_PLANE);      // (D contracts the landed tile itself)
  static constexpr int LDS_BYTES = BIAS_OFF + 2 * 128 * 4;
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

// 64-lane reduction of a double (every lane returns the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

#ifdef LFD_PL_TIMING
#define PH_T(i) do { if (blockIdx.x == PL_DBG_BLOCK && threadIdx.x == 0 && dbg_it < 8) g_pl_dbg[dbg_it * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PH_T(i)
#endif

template <int MODE, int CIN, int NSLAB>
__global__ __launch_bounds__(256, 2) void k_pl_head(PhArgs a) {
  using C = PhCfg<MODE, CIN, NSLAB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pix = lane & 31;
  // C: a wave owns ONE 32-pixel MFMA tile of one cout slab; A / B: both pixel tiles of its slab
  constexpr int PT = MODE == 2 ? 1 : 2;
  const int slab = MODE == 2 ? (wave % NSLAB) : wave;
  const int pt0 = MODE == 2 ? (wave / NSLAB) : 0;
  const bool mfma_wave = MODE != 2 || wave < 2 * NSLAB;

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  // CONTIGUOUS tile ranges per workgroup inside the XCD's range (PH_STRIDED: the strided walk of planes_impl.h): a workgroup then
  // stays inside one image for many tiles -- every image change costs the consumer a dependent chain of global loads (the
  // producer's GroupNorm sums -> mean / rstd -> scale / shift, ~2-3 us) and the producer a flush of its sums; with the strided
  // walk EVERY tile of the small pyramid levels (fewer tiles per image than the stride) paid that
  const int wgs_xcd = (nblk + 7 - xcd) / 8;
#ifdef PH_STRIDED
  const int t_step = wgs_xcd;
  const int t_first = t_begin + bix, t_last = t_end;
#else
  const int t_step = 1;
  const int chunk = (t_end - t_begin + wgs_xcd - 1) / (wgs_xcd > 0 ? wgs_xcd : 1);
  const int t_first = t_begin + bix * chunk;
  const int t_last = (t_first + chunk) < t_end ? (t_first + chunk) : t_end;
#endif
  int lstart[LFD_MAX_LEVELS];
#pragma unroll
  for (int i = 0; i < LFD_MAX_LEVELS; ++i) lstart[i] = i < a.n_levels ? a.lv[i].tile_start : 0x7fffffff;
  auto level_of = [&](int t) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < LFD_MAX_LEVELS; ++i) l += (t >= lstart[i]) ? 1 : 0;
    return l;
  };

  // a.lv[l] with a run-time l would move the whole argument block to scratch (and every pointer into VGPRs: the scalar-base
  // DMA needs SGPR bases): the level is selected by an unrolled compare chain over compile-time indices instead
  auto with_level = [&](int l, auto&& f) {
    static_for([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (l == i) f(a.lv[i]);
    }, std::make_integer_sequence<int, LFD_MAX_LEVELS>{});
  };
  // level of the tile being FETCHED (it may lie in the next level)
  int d_l = -1, d_P = 0, d_tpi = 1, d_t0 = 0;
  const char* d_in = nullptr;
  long d_plane_b = 0;
  // level of the tile being COMPUTED
  PhLevel lv{};

  float* sbias = reinterpret_cast<float*>(smem + C::BIAS_OFF);
  half8 w0h[C::NK0], w0l[C::NK0];
  half8 w1h[MODE == 0 ? 8 : 1], w1l[MODE == 0 ? 8 : 1];

  // ---- per-lane constants of the tile loaders (a DMA instruction = 64 lanes x 16 B = 1 KB, lane-linear in LDS)
  // A: tap planes, [px][cin] fp16 -> the operand layout (pixel-major, 16-byte chunk c of pixel p at chunk c ^ key(p)): the
  //    swizzle is applied on the SOURCE side.  Instruction i of a plane covers pixels (64 / CPP) i ..; wave w takes
  //    i = w, w + 4, ..: key(p) only depends on (4 w + lane / 16) -- a per-wave constant lane offset
  // B / C: fp32 rows of 512 B, two pixels per instruction, copied as they are (a permuted lane -> address map halves what the
  //    16-lane groups of the address coalescer merge: measured 2.5 TB/s of reads); the GroupNorm pass is organised around
  //    16-byte pieces (4 channels) so that its LDS reads are lane-linear too
  constexpr bool PLANES_IN = C::PLANES_IN;
  constexpr int CPP = (PLANES_IN ? CIN : 128) / 8;
  unsigned vo_in;
  if constexpr (PLANES_IN) {
    const int lp = lane / CPP, cs = lane % CPP;
    const int key = (CIN == 64) ? ((4 * wave + (lane >> 4)) & 7) : ((4 * wave + (lane >> 4)) & 15);
    vo_in = (unsigned)(lp * C::IN_PIXB + ((cs ^ key) * 16));
  } else {
    vo_in = (unsigned)(lane * 16);      // a plain copy: 1 KB = two pixel rows per instruction, lanes in address order (coalesced)
  }
  constexpr int IN_INSTR = PLANES_IN ? (C::IN_PLANE / 1024) : (C::RAW_BYTES / 1024);     // per plane (A) / per tile (B, C)
  static_assert(IN_INSTR % 4 == 0, "whole rounds of the four waves");

  // tile being fetched
  auto issue_in = [&](int t, int buf) {
    const int l = level_of(t);
    if (l != d_l) {
      d_l = l;
      with_level(l, [&](const PhLevel& v) {
        d_in = reinterpret_cast<const char*>(v.in); d_plane_b = v.in_plane * 2; d_P = v.P; d_tpi = v.tiles_per_img; d_t0 = v.tile_start;
      });
    }
    const int rel = t - d_t0;
    const int n = rel / d_tpi;
    const int j = rel - n * d_tpi;
    const int p0 = j * C::TPX;
    const bool full = p0 + C::TPX <= d_P;
    if constexpr (PLANES_IN) {
      const char* base = d_in + ((long)n * d_P + p0) * C::IN_PIXB;
      const long plane_b = d_plane_b;
      char* ld = smem + C::IN_OFF + buf * 2 * C::IN_PLANE;
#pragma unroll
      for (int jj = 0; jj < IN_INSTR / 4; ++jj) {
        const int i = wave + 4 * jj;
        if (full) {
          dma16s(base + i * 1024, vo_in, ld + i * 1024);
          dma16s(base + plane_b + i * 1024, vo_in, ld + C::IN_PLANE + i * 1024);
        } else {
          const int px = i * (64 / CPP) + lane / CPP;
          const bool ok = p0 + px < d_P;
          const char* src = ok ? base + i * 1024 + vo_in : reinterpret_cast<const char*>(a.zeros) + (lane % CPP) * 16;
          dma16(src, ld + i * 1024);
          dma16(ok ? src + plane_b : src, ld + C::IN_PLANE + i * 1024);
        }
      }
    } else {
      const char* base = d_in + ((long)n * d_P + p0) * 512;
      char* ld = smem + C::IN_OFF;
#pragma unroll
      for (int jj = 0; jj < IN_INSTR / 4; ++jj) {
        const int i = wave + 4 * jj;
        if (full) {
          dma16s(base + i * 1024, vo_in, ld + i * 1024);
        } else {
          const bool ok = p0 + 2 * i + (lane >> 5) < d_P;
          dma16(ok ? base + i * 1024 + vo_in : reinterpret_cast<const char*>(a.zeros) + (lane & 31) * 16, ld + i * 1024);
        }
      }
    }
  };

  // ---- B-fragment read offsets of the 128-channel operand planes (chunk 2 q + h of pixel p at chunk ^ (p % 16))
  int xo[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) xo[pt] = ((pt0 + pt) * 32 + pix) * 256;
  const int xkey = (pix & 15) ^ h;              // (2 q + h) ^ (p % 16) = (2 q) ^ (h ^ (p % 16)): p % 16 = pix % 16 for both tiles
  auto op_addr = [&](const char* base, int q, int pt) { return base + xo[pt] + (((2 * q) ^ xkey) << 4); };
  // A: the tap tile's own layout (CIN channels per pixel)
  int xi[PLANES_IN ? 2 : 1];
  int xikey = 0;
  if constexpr (PLANES_IN) {
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) xi[pt] = (pt * 32 + pix) * C::IN_PIXB;
    xikey = ((CIN == 64) ? ((pix >> 1) & 7) : (pix & 15)) ^ h;
  }

  // ---- GroupNorm of the input (B / C): thread = one 16-byte piece (channels 4 j .. 4 j + 3, group j / 2) of the pixels tid / 32 + 8 r
  float gn_a[PLANES_IN ? 1 : 4], gn_b[PLANES_IN ? 1 : 4];
  int gnin_key = -1;
  // ---- GroupNorm sums of the output (A / B): lane (h, pix) of slab s holds channels 32 s + 8 g + 4 h + e of its pixels
  double gs[MODE == 2 ? 1 : 4], gq[MODE == 2 ? 1 : 4];
  int gn_n = -1, gn_l = -1;
  unsigned long long* gn_dst = nullptr;     // gn_out of the level the running sums belong to
  if constexpr (MODE != 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) gs[g] = gq[g] = 0.;
  }
  auto gn_flush = [&]() {
    if constexpr (MODE != 2) {
      if (gn_n >= 0) {
        unsigned long long* dst = gn_dst + ((((size_t)(blockIdx.x % kGnRep)) * a.N + gn_n) * 16 + slab * 4) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double s = wave_sum(gs[g]), q = wave_sum(gq[g]);
          if (lane == 0) {
            atomicAdd(dst + 2 * g, (unsigned long long)__double2ll_rn(s * kGnFix));
            atomicAdd(dst + 2 * g + 1, (unsigned long long)__double2ll_rn(q * kGnFix));
          }
          gs[g] = gq[g] = 0.;
        }
      }
    }
  };

  int cur_l = -1;
  int dbg_it = 0; (void)dbg_it;
  int n_st = 0;                  // C: fp32 output store instructions of this wave per tile (the counted wait)
  int t = t_first;
  int buf = 0;
  bool first = true;
  if (t < t_last) issue_in(t, 0);

  for (; t < t_last; t += t_step, ++dbg_it) {
    PH_T(0);
    const int l = level_of(t);
    const bool new_level = l != cur_l;
    if (new_level) with_level(l, [&](const PhLevel& v) { lv = v; });
    const int rel = t - lv.tile_start;
    const int n = rel / lv.tiles_per_img;
    const int j = rel - n * lv.tiles_per_img;
    const int p0 = j * C::TPX;
    const bool has_next = t + t_step < t_last;

    // the VMEM operations younger than this tile's DMA are the previous tile's output stores (vmcnt retires in order)
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (MODE != 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else {
      switch (n_st) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
    }
    first = false;
    PH_T(1);
    block_barrier();             // tile t landed for every wave; every wave is done with the previous tile's operand planes
    PH_T(2);
    if constexpr (PLANES_IN && C::NBUF == 2) {
      if (has_next) issue_in(t + t_step, buf ^ 1);
    }

    if (new_level) {
      // the walk enters another level: its filters, biases, GroupNorm parameters
      cur_l = l;
      const half8* ws = lv.w0 + ((size_t)slab * C::NK0) * 64 + lane;
#pragma unroll
      for (int k = 0; k < C::NK0; ++k) {
        w0h[k] = ws[(size_t)k * 64];
        w0l[k] = ws[a.w_plane0 + (size_t)k * 64];
      }
      if constexpr (MODE == 0) {
        const half8* w1s = lv.w1 + ((size_t)slab * 8) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          w1h[k] = w1s[(size_t)k * 64];
          w1l[k] = w1s[a.w_plane1 + (size_t)k * 64];
        }
      }
      if (threadIdx.x < 128) {
        sbias[threadIdx.x] = (MODE == 2 && (int)threadIdx.x >= NSLAB * 32) ? 0.f : lv.b0[threadIdx.x];
        if constexpr (MODE == 0) sbias[128 + threadIdx.x] = lv.b1[threadIdx.x];
      }
      if constexpr (MODE == 2) {
        n_st = 0;
        if (mfma_wave) {
#pragma unroll
          for (int i = 0; i < 16; ++i) n_st += (slab * 32 + 8 * (i >> 2) + (i & 3) < a.f_c0 + a.f_c1) ? 1 : 0;
        }
      }
      gnin_key = -1;
      block_barrier();
    }

    const char* op = smem + C::OP_OFF;
    if constexpr (!PLANES_IN) {
      // ---- GroupNorm + ReLU of the landed fp32 tile -> hi / lo operand planes (lfd_head.py:97-117 conv -> GroupNorm -> ReLU)
      const int pj = (int)threadIdx.x & 31, gi = pj >> 1;
      if (gnin_key != n) {
        gnin_key = n;
        long long s = 0, q = 0;
#pragma unroll
        for (int r = 0; r < kGnRep; ++r) {       // (integer adds: order-independent, the statistics stay bit-reproducible)
          s += (long long)lv.gn_in[(((size_t)r * a.N + n) * 16 + gi) * 2];
          q += (long long)lv.gn_in[(((size_t)r * a.N + n) * 16 + gi) * 2 + 1];
        }
        const double cnt = (double)lv.P * 8.0;
        const double m = (double)s / kGnFix / cnt;
        double var = (double)q / kGnFix / cnt - m * m;
        var = var > 0. ? var : 0.;
        const float rstd = (float)(1. / sqrt(var + (double)a.eps)), mean = (float)m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gn_a[e] = rstd * lv.gamma[pj * 4 + e];
          gn_b[e] = lv.beta[pj * 4 + e] - mean * gn_a[e];
        }
      }
      const char* raw = smem + C::IN_OFF;
      char* opw = smem + C::OP_OFF;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int p = ((int)threadIdx.x >> 5) + 8 * r;
        const float4 v = *reinterpret_cast<const float4*>(raw + p * 512 + pj * 16);
        const float y0 = fmaxf(fmaf(v.x, gn_a[0], gn_b[0]), 0.f), y1 = fmaxf(fmaf(v.y, gn_a[1], gn_b[1]), 0.f);
        const float y2 = fmaxf(fmaf(v.z, gn_a[2], gn_b[2]), 0.f), y3 = fmaxf(fmaf(v.w, gn_a[3], gn_b[3]), 0.f);
        uint2 oh, ol;
        split2(y0, y1, oh.x, ol.x);
        split2(y2, y3, oh.y, ol.y);
        const int o = p * 256 + ((gi ^ (p & 15)) * 16) + (pj & 1) * 8;
        *reinterpret_cast<uint2*>(opw + o) = oh;
        *reinterpret_cast<uint2*>(opw + C::OP_PLANE + o) = ol;
      }
      PH_T(3);
      block_barrier();           // operand planes complete; the raw tile is free
      PH_T(4);
      if (has_next) issue_in(t + t_step, 0);
      PH_T(5);
    }

    // ---- first contraction: A the neck conv on the tap tile, B / C the conv on the normalised planes
    f32x16 am[PT], ac[PT];
    auto init_acc = [&](const float* bp) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          am[pt][4 * g + 0] = b4.x; am[pt][4 * g + 1] = b4.y; am[pt][4 * g + 2] = b4.z; am[pt][4 * g + 3] = b4.w;
          ac[pt][4 * g + 0] = 0.f; ac[pt][4 * g + 1] = 0.f; ac[pt][4 * g + 2] = 0.f; ac[pt][4 * g + 3] = 0.f;
        }
      }
    };
    // K-loop over `nk` k-steps: fragments two k-steps ahead in a register ring
    auto contract = [&](auto addr, int lo_off, const half8* wh, const half8* wl, auto nk_c) {
      constexpr int NK = decltype(nk_c)::value;
      constexpr int PD = 1;      // (the register budget of two workgroups per CU: 256)
      half8 xqh[PD + 1][PT], xql[PD + 1][PT];
#pragma unroll
      for (int k = 0; k < PD && k < NK; ++k)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const char* p = addr(k, pt);
          xqh[k][pt] = *reinterpret_cast<const half8*>(p);
          xql[k][pt] = *reinterpret_cast<const half8*>(p + lo_off);
        }
      static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + PD < NK) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const char* p = addr(k + PD, pt);
            xqh[(k + PD) % (PD + 1)][pt] = *reinterpret_cast<const half8*>(p);
            xql[(k + PD) % (PD + 1)][pt] = *reinterpret_cast<const half8*>(p + lo_off);
          }
        }
        PL_SB();
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) am[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)][pt], am[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) ac[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)][pt], ac[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) ac[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)][pt], ac[pt], 0, 0, 0);
        PL_SB();
      }, std::make_integer_sequence<int, NK>{});
    };

    if constexpr (MODE == 0) {
      const char* xb = smem + C::IN_OFF + buf * 2 * C::IN_PLANE;
      init_acc(sbias + slab * 32 + 4 * h);
      contract([&](int q, int pt) { return xb + xi[pt] + (((2 * q) ^ xikey) << 4); }, C::IN_PLANE, w0h, w0l, std::integral_constant<int, C::NK0>{});
      // neck output (+ ReLU) -> the tower conv's operand planes: this wave's 32 channels = chunks 4 slab + g of every pixel
      char* mid = smem + C::OP_OFF;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int pb = pt * 32 + pix;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = comb(am[pt][4 * g + e], ac[pt][4 * g + e]);
            if (a.relu0) y[e] = fmaxf(y[e], 0.f);
          }
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = pb * 256 + (((slab * 4 + g) ^ (pb & 15)) * 16) + 8 * h;
          *reinterpret_cast<uint2*>(mid + o) = vh;
          *reinterpret_cast<uint2*>(mid + C::OP_PLANE + o) = vl;
        }
      }
      block_barrier();
      if constexpr (C::NBUF == 1) {
        if (has_next) issue_in(t + t_step, 0);       // (single tap buffer: free once every wave has contracted it)
      }
      init_acc(sbias + 128 + slab * 32 + 4 * h);
      contract([&](int q, int pt) { return op_addr(op, q, pt); }, C::OP_PLANE, w1h, w1l, std::integral_constant<int, 8>{});
    } else if constexpr (MODE == 3) {
      // D: the landed plane tile IS the operand (a tower's first conv on the neck's stored output, separate-tower heads)
      const char* xb = smem + C::IN_OFF + buf * 2 * C::IN_PLANE;
      init_acc(sbias + slab * 32 + 4 * h);
      contract([&](int q, int pt) { return xb + xi[pt] + (((2 * q) ^ xikey) << 4); }, C::IN_PLANE, w0h, w0l, std::integral_constant<int, C::NK0>{});
    } else {
      if (mfma_wave) {
        init_acc(sbias + slab * 32 + 4 * h);
        contract([&](int q, int pt) { return op_addr(op, q, pt); }, C::OP_PLANE, w0h, w0l, std::integral_constant<int, 8>{});
      }
    }

    PH_T(6);
    // ---- epilogue
    if constexpr (MODE != 2) {
      if (n != gn_n || l != gn_l) {
        gn_flush();
        gn_n = n; gn_l = l; gn_dst = lv.gn_out;
      }
      float* trash = reinterpret_cast<float*>(const_cast<_Float16*>(a.zeros) + 1024) + (threadIdx.x & 127) * 4;
      float* obase = lv.out + ((size_t)n * lv.P + p0) * 128 + slab * 32 + 4 * h;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int px = pt * 32 + pix;
        const bool ok = p0 + px < lv.P;
        float* orow = obase + (size_t)px * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v;
          v.x = comb(am[pt][4 * g + 0], ac[pt][4 * g + 0]);
          v.y = comb(am[pt][4 * g + 1], ac[pt][4 * g + 1]);
          v.z = comb(am[pt][4 * g + 2], ac[pt][4 * g + 2]);
          v.w = comb(am[pt][4 * g + 3], ac[pt][4 * g + 3]);
          // exactly eight stores per lane and tile (the counted wait at the loop top): pixels past the image go to the trash line
          *reinterpret_cast<float4*>(ok ? orow + 8 * g : trash) = v;
          if (ok) {
            const float s = (v.x + v.y) + (v.z + v.w);
            const float q = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
            gs[g] += (double)s;
            gq[g] += (double)q;
          }
        }
      }
    } else {
      if (mfma_wave) {
        // fp32 outputs straight from the accumulator layout (lane = pixel, registers = channels 8 g + 4 h + e of the slab):
        // exactly n_st store instructions per tile and wave; lanes without an output write the trash line
        const float sc1 = lv.scale1 ? *lv.scale1 : 1.f;
        const int ctot = a.f_c0 + a.f_c1;
        float* trash = reinterpret_cast<float*>(const_cast<_Float16*>(a.zeros) + 1024) + (threadIdx.x & 127);
        const int px = p0 + pt0 * 32 + pix;
        const bool in_img = px < lv.P;
        const size_t pixi = in_img ? (size_t)px : 0;
        float* q0 = lv.f_out0 + (size_t)n * a.f_img0 + pixi * a.f_c0;
        float* q1 = lv.f_out1 + (size_t)n * a.f_img1 + pixi * a.f_c1 - a.f_c0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (slab * 32 + 8 * g + e < ctot) {          // wave-uniform: the lower lane half's channel of this register
              const int co = slab * 32 + 8 * g + 4 * h + e;
              const float y = comb(am[0][4 * g + e], ac[0][4 * g + e]);
              const bool first_out = co < a.f_c0;
              float* dst = (in_img && co < ctot) ? (first_out ? q0 : q1) + co : trash;
              *dst = first_out ? y : y * sc1;
            }
          }
      }
    }
    if constexpr (PLANES_IN && C::NBUF == 2) buf ^= 1;
    PH_T(7);
  }
  gn_flush();
}

// ---- MODE 2 with the fp32 tile landing in REGISTERS (k_pl_head_out).  The output conv reads 178 MB to write 7 MB: with the
// tile arriving by LDS-DMA, the bytes a CU keeps in flight are the raw-tile buffers of its two workgroups times the fraction of
// the time a fetch is outstanding -- ~32 KB, 2.9 TB/s, the waves waiting 66 % of the time (profiles/r06_pmc_sq_counters.txt).
// This kernel has registers to spare (no 128-row filter): every thread loads ITS eight 16-byte pieces of tile t + 1 into
// registers before it transforms tile t (32 KB per workgroup in flight all the time, no raw buffer in LDS), and the operand
// planes are double-buffered, so one barrier per tile is enough.
template <int NSLAB>
__global__ __launch_bounds__(256, 2) void k_pl_head_out(PhArgs a) {
  constexpr int TPX = 64, OP_PLANE = TPX * 256, BIAS_OFF = 4 * OP_PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pix = lane & 31;
  const int slab = wave % NSLAB, pt0 = wave / NSLAB;
  const bool mfma_wave = wave < 2 * NSLAB;
  const int pj = (int)threadIdx.x & 31, gi = pj >> 1, prow = (int)threadIdx.x >> 5;
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int x_begin = xcd * per_xcd;
  const int x_end = (x_begin + per_xcd) < a.ntiles ? (x_begin + per_xcd) : a.ntiles;
  const int wgs_xcd = (nblk + 7 - xcd) / 8;
  const int chunk = (x_end - x_begin + wgs_xcd - 1) / (wgs_xcd > 0 ? wgs_xcd : 1);
  const int t_first = x_begin + bix * chunk;
  const int t_last = (t_first + chunk) < x_end ? (t_first + chunk) : x_end;
  int lstart[LFD_MAX_LEVELS];
#pragma unroll
  for (int i = 0; i < LFD_MAX_LEVELS; ++i) lstart[i] = i < a.n_levels ? a.lv[i].tile_start : 0x7fffffff;
  auto level_of = [&](int t) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < LFD_MAX_LEVELS; ++i) l += (t >= lstart[i]) ? 1 : 0;
    return l;
  };
  auto with_level = [&](int l, auto&& f) {
    static_for([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (l == i) f(a.lv[i]);
    }, std::make_integer_sequence<int, LFD_MAX_LEVELS>{});
  };
  int d_l = -1, d_P = 1, d_tpi = 1, d_t0 = 0;
  const char* d_in = nullptr;
  // the thread's eight pieces of a tile: pixel rows prow + 8 r.  Rows past the image are clamped to its last pixel: what the
  // lanes of such a row compute is never stored, and pixels do not mix
  auto load_tile = [&](int t, float4* dst) {
    const int l = level_of(t);
    if (l != d_l) {
      d_l = l;
      with_level(l, [&](const PhLevel& v) { d_in = reinterpret_cast<const char*>(v.in); d_P = v.P; d_tpi = v.tiles_per_img; d_t0 = v.tile_start; });
    }
    const int rel = t - d_t0;
    const int n = rel / d_tpi;
    const int p0 = (rel - n * d_tpi) * TPX;
    const char* base = d_in + (long)n * d_P * 512 + pj * 16;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int p = p0 + prow + 8 * r;
      p = p < d_P ? p : d_P - 1;
      dst[r] = *reinterpret_cast<const float4*>(base + (long)p * 512);
    }
  };
  float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
  half8 wh[8], wl[8];
  float gn_a[4], gn_b[4];
  PhLevel lv{};
  int cur_l = -1, gnin_key = -1;
  const int xo = (pt0 * 32 + pix) * 256;
  const int xkey = (pix & 15) ^ h;
  float4 cur[8], nxt[8];
  int buf = 0;
  int t = t_first;
  if (t < t_last) load_tile(t, cur);
  for (; t < t_last; ++t, buf ^= 1) {
    const int l = level_of(t);
    if (l != cur_l) {
      cur_l = l;
      with_level(l, [&](const PhLevel& v) { lv = v; });
      const half8* ws = lv.w0 + ((size_t)slab * 8) * 64 + lane;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        wh[k] = ws[(size_t)k * 64];
        wl[k] = ws[a.w_plane0 + (size_t)k * 64];
      }
      block_barrier();                 // (every wave has read the previous level's bias)
      if (threadIdx.x < 128) sbias[threadIdx.x] = ((int)threadIdx.x >= NSLAB * 32) ? 0.f : lv.b0[threadIdx.x];
      gnin_key = -1;
    }
    const int rel = t - lv.tile_start;
    const int n = rel / lv.tiles_per_img;
    const int p0 = (rel - n * lv.tiles_per_img) * TPX;
    if (t + 1 < t_last) load_tile(t + 1, nxt);
    if (gnin_key != n) {
      gnin_key = n;
      long long s = 0, q = 0;
#pragma unroll
      for (int r = 0; r < kGnRep; ++r) {       // (integer adds: order-independent, the statistics stay bit-reproducible)
        s += (long long)lv.gn_in[(((size_t)r * a.N + n) * 16 + gi) * 2];
        q += (long long)lv.gn_in[(((size_t)r * a.N + n) * 16 + gi) * 2 + 1];
      }
      const double cnt = (double)lv.P * 8.0;
      const double m = (double)s / kGnFix / cnt;
      double var = (double)q / kGnFix / cnt - m * m;
      var = var > 0. ? var : 0.;
      const float rstd = (float)(1. / sqrt(var + (double)a.eps)), mean = (float)m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gn_a[e] = rstd * lv.gamma[pj * 4 + e];
        gn_b[e] = lv.beta[pj * 4 + e] - mean * gn_a[e];
      }
    }
    // GroupNorm + ReLU of the tile in registers -> hi / lo operand planes (lfd_head.py:97-117)
    char* opw = smem + buf * 2 * OP_PLANE;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int p = prow + 8 * r;
      const float4 v = cur[r];
      const float y0 = fmaxf(fmaf(v.x, gn_a[0], gn_b[0]), 0.f), y1 = fmaxf(fmaf(v.y, gn_a[1], gn_b[1]), 0.f);
      const float y2 = fmaxf(fmaf(v.z, gn_a[2], gn_b[2]), 0.f), y3 = fmaxf(fmaf(v.w, gn_a[3], gn_b[3]), 0.f);
      uint2 oh, ol;
      split2(y0, y1, oh.x, ol.x);
      split2(y2, y3, oh.y, ol.y);
      const int o = p * 256 + ((gi ^ (p & 15)) * 16) + (pj & 1) * 8;
      *reinterpret_cast<uint2*>(opw + o) = oh;
      *reinterpret_cast<uint2*>(opw + OP_PLANE + o) = ol;
    }
    block_barrier();                   // operand planes of tile t complete (the other pair belongs to the previous tile's readers)
    if (mfma_wave) {
      f32x16 am, ac;
      {
        const float* bp = sbias + slab * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          am[4 * g + 0] = b4.x; am[4 * g + 1] = b4.y; am[4 * g + 2] = b4.z; am[4 * g + 3] = b4.w;
          ac[4 * g + 0] = 0.f; ac[4 * g + 1] = 0.f; ac[4 * g + 2] = 0.f; ac[4 * g + 3] = 0.f;
        }
      }
      const char* op = smem + buf * 2 * OP_PLANE + xo;
      half8 xh[2], xl[2];
      xh[0] = *reinterpret_cast<const half8*>(op + ((0 ^ xkey) << 4));
      xl[0] = *reinterpret_cast<const half8*>(op + OP_PLANE + ((0 ^ xkey) << 4));
      static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + 1 < 8) {
          xh[(k + 1) & 1] = *reinterpret_cast<const half8*>(op + (((2 * (k + 1)) ^ xkey) << 4));
          xl[(k + 1) & 1] = *reinterpret_cast<const half8*>(op + OP_PLANE + (((2 * (k + 1)) ^ xkey) << 4));
        }
        PL_SB();
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xh[k & 1], am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xl[k & 1], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xh[k & 1], ac, 0, 0, 0);
        PL_SB();
      }, std::make_integer_sequence<int, 8>{});
      // fp32 outputs straight from the accumulator layout (lane = pixel, registers = channels 8 g + 4 h + e of the slab)
      const float sc1 = lv.scale1 ? *lv.scale1 : 1.f;
      const int ctot = a.f_c0 + a.f_c1;
      const int px = p0 + pt0 * 32 + pix;
      const bool in_img = px < lv.P;
      float* q0 = lv.f_out0 + (size_t)n * a.f_img0 + (size_t)px * a.f_c0;
      float* q1 = lv.f_out1 + (size_t)n * a.f_img1 + (size_t)px * a.f_c1 - a.f_c0;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (slab * 32 + 8 * g + e < ctot) {          // wave-uniform: the lower lane half's channel of this register
            const int co = slab * 32 + 8 * g + 4 * h + e;
            const float y = comb(am[4 * g + e], ac[4 * g + e]);
            if (in_img && co < ctot) {
              if (co < a.f_c0) q0[co] = y;
              else q1[co] = y * sc1;
            }
          }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
  }
}

template <int NSLAB>
int launch_pl_head_out(PhArgs& a, hipStream_t st) {
  long nt = 0;
  for (int i = 0; i < a.n_levels; ++i) {
    a.lv[i].tiles_per_img = (a.lv[i].P + 63) / 64;
    a.lv[i].tile_start = (int)nt;
    nt += (long)a.N * a.lv[i].tiles_per_img;
    if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  }
  a.ntiles = (int)nt;
  constexpr int LDSB = 4 * 64 * 256 + 512;
  auto kern = k_pl_head_out<NSLAB>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  int blocks = 512;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), LDSB, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// ---- MODE 1 with PRODUCER and CONSUMER waves (k_pl_head_b2, round 6).  k_pl_head<1> is a chain per workgroup -- fetch,
// GroupNorm pass, contraction, stores -- and two workgroups per CU overlap only what happens to fall together (waves waiting
// 38 % of the time, 4.1 TB/s).  Here a 512-thread workgroup owns the CU: waves 0-3 (P) hold the fp32 tiles of the next TWO
// steps in registers (64 KB per CU in flight all the time, no raw tile in LDS), normalise tile i + 1 and write its hi / lo operand
// planes while waves 4-7 (Q, one 32-row slab each) contract tile i, store it and add up its GroupNorm sums; the operand planes
// are double-buffered, one barrier per tile.
__global__ __launch_bounds__(512, 1) void k_pl_head_b2(PhArgs a) {
  constexpr int TPX = 64, OP_PLANE = TPX * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int x_begin = xcd * per_xcd;
  const int x_end = (x_begin + per_xcd) < a.ntiles ? (x_begin + per_xcd) : a.ntiles;
  const int wgs_xcd = (nblk + 7 - xcd) / 8;
  const int chunk = (x_end - x_begin + wgs_xcd - 1) / (wgs_xcd > 0 ? wgs_xcd : 1);
  const int t_first = x_begin + bix * chunk;
  const int t_last = (t_first + chunk) < x_end ? (t_first + chunk) : x_end;
  const int nt = t_last > t_first ? t_last - t_first : 0;
  int lstart[LFD_MAX_LEVELS];
#pragma unroll
  for (int i = 0; i < LFD_MAX_LEVELS; ++i) lstart[i] = i < a.n_levels ? a.lv[i].tile_start : 0x7fffffff;
  auto level_of = [&](int t) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < LFD_MAX_LEVELS; ++i) l += (t >= lstart[i]) ? 1 : 0;
    return l;
  };
  auto with_level = [&](int l, auto&& f) {
    static_for([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (l == i) f(a.lv[i]);
    }, std::make_integer_sequence<int, LFD_MAX_LEVELS>{});
  };
  if (wave < 4) {
    // ================================================ P: fetch + GroupNorm + ReLU -> operand planes =================
    const int tid = (int)threadIdx.x;                  // 0 .. 255
    const int pj = tid & 31, gi = pj >> 1, prow = tid >> 5;
    int d_l = -1, d_P = 1, d_tpi = 1, d_t0 = 0;
    const char* d_in = nullptr;
    auto load_tile = [&](int t, float4* dst) {
      const int l = level_of(t);
      if (l != d_l) {
        d_l = l;
        with_level(l, [&](const PhLevel& v) { d_in = reinterpret_cast<const char*>(v.in); d_P = v.P; d_tpi = v.tiles_per_img; d_t0 = v.tile_start; });
      }
      const int rel = t - d_t0;
      const int n = rel / d_tpi;
      const int p0 = (rel - n * d_tpi) * TPX;
      const char* base = d_in + (long)n * d_P * 512 + pj * 16;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int p = p0 + prow + 8 * r;
        p = p < d_P ? p : d_P - 1;          // (rows past the image: clamped -- their results are never stored, pixels do not mix)
        dst[r] = *reinterpret_cast<const float4*>(base + (long)p * 512);
      }
    };
    // GroupNorm scale / shift of this thread's four channels for (level, image) of tile t
    int g_l = -1, g_P = 1, g_tpi = 1, g_t0 = 0;
    const unsigned long long* g_in = nullptr;
    const float* g_gamma = nullptr;
    const float* g_beta = nullptr;
    auto gn_params = [&](int t, float* ga, float* gb, int& key) {
      const int l = level_of(t);
      if (l != g_l) {
        g_l = l;
        with_level(l, [&](const PhLevel& v) { g_in = v.gn_in; g_gamma = v.gamma; g_beta = v.beta; g_P = v.P; g_tpi = v.tiles_per_img; g_t0 = v.tile_start; });
      }
      const int n = (t - g_t0) / g_tpi;
      const int k = l * 65536 + n;
      if (k == key) return;
      key = k;
      long long s = 0, q = 0;
#pragma unroll
      for (int r = 0; r < kGnRep; ++r) {       // (integer adds: order-independent, the statistics stay bit-reproducible)
        s += (long long)g_in[(((size_t)r * a.N + n) * 16 + gi) * 2];
        q += (long long)g_in[(((size_t)r * a.N + n) * 16 + gi) * 2 + 1];
      }
      const double cnt = (double)g_P * 8.0;
      const double m = (double)s / kGnFix / cnt;
      double var = (double)q / kGnFix / cnt - m * m;
      var = var > 0. ? var : 0.;
      const float rstd = (float)(1. / sqrt(var + (double)a.eps)), mean = (float)m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ga[e] = rstd * g_gamma[pj * 4 + e];
        gb[e] = g_beta[pj * 4 + e] - mean * ga[e];
      }
    };
    float4 r0[8], r1[8], r2[8];
    float ga[4], gb[4], na[4], nb[4];
    int key = -1, nkey = -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) ga[e] = gb[e] = na[e] = nb[e] = 0.f;
    if (nt > 0) load_tile(t_first, r0);
    if (nt > 1) load_tile(t_first + 1, r1);
    if (nt > 0) gn_params(t_first, na, nb, nkey);
    for (int i = 0; i <= nt; ++i) {
      if (i < nt) {
        if (i + 2 < nt) load_tile(t_first + i + 2, r2);
        if (nkey != key) {
          key = nkey;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ga[e] = na[e]; gb[e] = nb[e]; }
        }
        char* opw = smem + (i & 1) * 2 * OP_PLANE;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int p = prow + 8 * r;
          const float4 v = r0[r];
          const float y0 = fmaxf(fmaf(v.x, ga[0], gb[0]), 0.f), y1 = fmaxf(fmaf(v.y, ga[1], gb[1]), 0.f);
          const float y2 = fmaxf(fmaf(v.z, ga[2], gb[2]), 0.f), y3 = fmaxf(fmaf(v.w, ga[3], gb[3]), 0.f);
          uint2 oh, ol;
          split2(y0, y1, oh.x, ol.x);
          split2(y2, y3, oh.y, ol.y);
          const int o = p * 256 + ((gi ^ (p & 15)) * 16) + (pj & 1) * 8;
          *reinterpret_cast<uint2*>(opw + o) = oh;
          *reinterpret_cast<uint2*>(opw + OP_PLANE + o) = ol;
        }
        // the next tile's scale / shift (a dependent chain of global loads when the image changes) while the consumers work
        if (i + 1 < nt) gn_params(t_first + i + 1, na, nb, nkey);
#pragma unroll
        for (int r = 0; r < 8; ++r) { r0[r] = r1[r]; r1[r] = r2[r]; }
      }
      block_barrier();
    }
  } else {
    // ================================================ Q: contraction + fp32 stores + GroupNorm sums ==================
    const int slab = wave - 4;
    const int h = lane >> 5, pix = lane & 31;
    half8 wh[8], wl[8];
    float bias[16];
    PhLevel lv{};
    int cur_l = -1;
    double gs[4], gq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gs[g] = gq[g] = 0.;
    int gn_n = -1;
    unsigned long long* gn_dst = nullptr;
    auto gn_flush = [&]() {
      if (gn_n >= 0) {
        unsigned long long* dst = gn_dst + ((((size_t)(blockIdx.x % kGnRep)) * a.N + gn_n) * 16 + slab * 4) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double s = wave_sum(gs[g]), q = wave_sum(gq[g]);
          if (lane == 0) {
            atomicAdd(dst + 2 * g, (unsigned long long)__double2ll_rn(s * kGnFix));
            atomicAdd(dst + 2 * g + 1, (unsigned long long)__double2ll_rn(q * kGnFix));
          }
          gs[g] = gq[g] = 0.;
        }
      }
    };
    const int xkey = (pix & 15) ^ h;
    for (int i = 0; i <= nt; ++i) {
      if (i >= 1) {
        const int t = t_first + i - 1;
        const int l = level_of(t);
        if (l != cur_l) {
          gn_flush();
          gn_n = -1;
          cur_l = l;
          with_level(l, [&](const PhLevel& v) { lv = v; });
          const half8* ws = lv.w0 + ((size_t)slab * 8) * 64 + lane;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            wh[k] = ws[(size_t)k * 64];
            wl[k] = ws[a.w_plane0 + (size_t)k * 64];
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(lv.b0 + slab * 32 + 4 * h + 8 * g);
            bias[4 * g + 0] = b4.x; bias[4 * g + 1] = b4.y; bias[4 * g + 2] = b4.z; bias[4 * g + 3] = b4.w;
          }
        }
        const int rel = t - lv.tile_start;
        const int n = rel / lv.tiles_per_img;
        const int p0 = (rel - n * lv.tiles_per_img) * TPX;
        if (n != gn_n) {
          gn_flush();
          gn_n = n; gn_dst = lv.gn_out;
        }
        f32x16 am[2], ac[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { am[0][r] = am[1][r] = bias[r]; ac[0][r] = ac[1][r] = 0.f; }
        const char* op = smem + ((i - 1) & 1) * 2 * OP_PLANE;
        const char* o0 = op + pix * 256, * o1 = op + (32 + pix) * 256;
        half8 xh[2][2], xl[2][2];
        xh[0][0] = *reinterpret_cast<const half8*>(o0 + ((0 ^ xkey) << 4)); xl[0][0] = *reinterpret_cast<const half8*>(o0 + OP_PLANE + ((0 ^ xkey) << 4));
        xh[0][1] = *reinterpret_cast<const half8*>(o1 + ((0 ^ xkey) << 4)); xl[0][1] = *reinterpret_cast<const half8*>(o1 + OP_PLANE + ((0 ^ xkey) << 4));
        static_for([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (k + 1 < 8) {
            constexpr int q2 = 2 * (k + 1);
            xh[(k + 1) & 1][0] = *reinterpret_cast<const half8*>(o0 + ((q2 ^ xkey) << 4)); xl[(k + 1) & 1][0] = *reinterpret_cast<const half8*>(o0 + OP_PLANE + ((q2 ^ xkey) << 4));
            xh[(k + 1) & 1][1] = *reinterpret_cast<const half8*>(o1 + ((q2 ^ xkey) << 4)); xl[(k + 1) & 1][1] = *reinterpret_cast<const half8*>(o1 + OP_PLANE + ((q2 ^ xkey) << 4));
          }
          PL_SB();
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) am[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xh[k & 1][pt], am[pt], 0, 0, 0);
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) ac[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xl[k & 1][pt], ac[pt], 0, 0, 0);
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) ac[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xh[k & 1][pt], ac[pt], 0, 0, 0);
          PL_SB();
        }, std::make_integer_sequence<int, 8>{});
        // stores straight from the accumulator layout: 16 bytes per lane, 32-byte runs per pixel that the four stores of the wave
        // complete to full 128-byte lines (staged through LDS into whole-line stores the launch was 3-5 us SLOWER: the bytes, not
        // the store shape, bound it -- a plain copy of the same 357 MB takes 67 us on this part, tools/ub/hbm_rates.py)
        float* obase = lv.out + ((size_t)n * lv.P + p0) * 128 + slab * 32 + 4 * h;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const int px = pt * 32 + pix;
          const bool ok = p0 + px < lv.P;
          float* orow = obase + (size_t)px * 128;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = comb(am[pt][4 * g + 0], ac[pt][4 * g + 0]);
            v.y = comb(am[pt][4 * g + 1], ac[pt][4 * g + 1]);
            v.z = comb(am[pt][4 * g + 2], ac[pt][4 * g + 2]);
            v.w = comb(am[pt][4 * g + 3], ac[pt][4 * g + 3]);
            if (ok) {
              *reinterpret_cast<float4*>(orow + 8 * g) = v;
              const float s = (v.x + v.y) + (v.z + v.w);
              const float q = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)));
              gs[g] += (double)s;
              gq[g] += (double)q;
            }
          }
        }
      }
      block_barrier();
    }
    gn_flush();
  }
}

int launch_pl_head_b2(PhArgs& a, hipStream_t st) {
  long nt = 0;
  for (int i = 0; i < a.n_levels; ++i) {
    a.lv[i].tiles_per_img = (a.lv[i].P + 63) / 64;
    a.lv[i].tile_start = (int)nt;
    nt += (long)a.N * a.lv[i].tiles_per_img;
    if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  }
  a.ntiles = (int)nt;
  constexpr int LDSB = 4 * 64 * 256;
  auto kern = k_pl_head_b2;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  int blocks = 256;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), LDSB, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

template <int MODE, int CIN, int NSLAB>
int launch_pl_head(PhArgs& a, hipStream_t st) {
  using C = PhCfg<MODE, CIN, NSLAB>;
  long nt = 0;
  for (int i = 0; i < a.n_levels; ++i) {
    a.lv[i].tiles_per_img = (a.lv[i].P + C::TPX - 1) / C::TPX;
    a.lv[i].tile_start = (int)nt;
    nt += (long)a.N * a.lv[i].tiles_per_img;
    if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  }
  a.ntiles = (int)nt;
  auto kern = k_pl_head<MODE, CIN, NSLAB>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  int blocks = 512;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), C::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // namespace pl

#ifdef LFD_PL_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_pl_head_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pl::g_pl_dbg), sizeof(unsigned long long) * 136);
}
#endif

extern "C" int lfd_pl_head_levels(const lfd_pl_head_desc_t* d, const lfd_pl_head_level_t* levels, int32_t num_levels, const void* zeros,
                                  lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d || !levels || !zeros || num_levels < 1 || num_levels > LFD_MAX_LEVELS || d->n < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (d->mode < 0 || d->mode > 3) return LFD_ERR_INVALID_ARGUMENT;
  if (d->mode == 0 && d->cin != 64 && d->cin != 128) return LFD_ERR_UNSUPPORTED;
  if (d->mode == 2 && (d->f_c0 < 0 || d->f_c1 < 0 || d->f_c0 + d->f_c1 < 1 || d->f_c0 + d->f_c1 > 64)) return LFD_ERR_UNSUPPORTED;
  pl::PhArgs a{};
  a.zeros = (const _Float16*)zeros;
  a.n_levels = num_levels; a.N = d->n; a.relu0 = d->relu0; a.eps = d->gn_in_eps;
  a.f_c0 = d->f_c0; a.f_c1 = d->f_c1; a.f_img0 = d->f_image_stride0; a.f_img1 = d->f_image_stride1;
  const int nslab0 = d->mode == 2 ? (d->f_c0 + d->f_c1 + 31) / 32 : 4;
  const int nk0 = (d->mode == 0 ? d->cin : 128) / 16;
  a.w_plane0 = (long)nslab0 * nk0 * 64;
  a.w_plane1 = 4L * 8 * 64;
  for (int i = 0; i < num_levels; ++i) {
    const lfd_pl_head_level_t& s = levels[i];
    if (!s.in || !s.w0 || !s.b0 || s.pixels < 1) return LFD_ERR_INVALID_ARGUMENT;
    if (!lfd_aligned16(s.in) || !lfd_aligned16(s.out) || (s.in_plane_halfs & 7)) return LFD_ERR_INVALID_ARGUMENT;
    if (d->mode == 0 && (!s.w1 || !s.b1)) return LFD_ERR_INVALID_ARGUMENT;
    if (d->mode != 2 && (!s.out || !s.gn_sums)) return LFD_ERR_INVALID_ARGUMENT;
    if ((d->mode == 1 || d->mode == 2) && (!s.gn_in_sums || !s.gn_in_gamma || !s.gn_in_beta)) return LFD_ERR_INVALID_ARGUMENT;
    if (d->mode == 2 && ((d->f_c0 > 0 && !s.f_out0) || (d->f_c1 > 0 && !s.f_out1))) return LFD_ERR_INVALID_ARGUMENT;
    pl::PhLevel& l = a.lv[i];
    l.in = s.in; l.in_plane = s.in_plane_halfs; l.out = (float*)s.out;
    l.w0 = (const half8*)s.w0; l.b0 = s.b0; l.w1 = (const half8*)s.w1; l.b1 = s.b1;
    l.gn_out = (unsigned long long*)s.gn_sums; l.gn_in = (const unsigned long long*)s.gn_in_sums;
    l.gamma = s.gn_in_gamma; l.beta = s.gn_in_beta;
    l.f_out0 = s.f_out0; l.f_out1 = s.f_out1; l.scale1 = s.scale1;
    l.P = s.pixels;
  }
  if (d->mode == 0) return d->cin == 64 ? pl::launch_pl_head<0, 64, 4>(a, st) : pl::launch_pl_head<0, 128, 4>(a, st);
  if (d->mode == 3) return pl::launch_pl_head<3, 128, 4>(a, st);
  if (d->mode == 1) return lfd_tune(LFD_TUNE_PL_HEAD_ROLES) != 0 ? pl::launch_pl_head_b2(a, st) : pl::launch_pl_head<1, 128, 4>(a, st);
  if (lfd_tune(LFD_TUNE_PL_HEAD_OUT_REGS) != 0) return nslab0 == 1 ? pl::launch_pl_head_out<1>(a, st) : pl::launch_pl_head_out<2>(a, st);
  return nslab0 == 1 ? pl::launch_pl_head<2, 128, 1>(a, st) : pl::launch_pl_head<2, 128, 2>(a, st);
}
