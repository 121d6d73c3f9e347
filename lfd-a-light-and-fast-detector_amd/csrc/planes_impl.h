// csrc/planes_impl.h -- (device templates of planes.hip)
// Implicit-GEMM convolution (1x1 / 3x3, stride 1 / 2) on "hi/lo plane" tensors: the tolerance-compliant precision mode of the
// LFD eval forward (lfd/model/lfd.py:511-542; backbone lfd_resnet.py:96-154,354-439,458-468, neck simple_neck.py:67-74, head
// lfd_head.py:164-185) at the speed of the fused fp16 structures.
//
// Storage: an activation tensor x (fp32 in the reference) lives in HBM as TWO NHWC fp16 planes
//     hi = fp16(x),   lo = fp16(2^11 (x - hi))            x = hi + 2^-11 lo  to ~22 significant bits
// -- the bytes of an fp32 tensor, but already in the form the matrix cores consume: the tile loader is a plain global -> LDS
// DMA per plane (no conversion, no VALU), exactly the loader of the fp16 kernels (conv_impl.h) issued twice.  Weights
// (BatchNorm folded in fp64 on the host) are split the same way once per plan.  A k-step issues three
// v_mfma_f32_32x32x16_f16 into two fp32 accumulator sets
//     main += w_hi x_hi,      corr += w_hi x_lo + w_lo x_hi,      y = main + 2^-11 corr         (w_lo x_lo ~ 2^-22: dropped)
// and the epilogue (bias, residual, ReLU in fp32) splits y into planes again.  csrc/precise.hip computes the same three
// products from fp32 storage with the split in its loader (round 3); this file is what makes the mode fast:
//   * weights-stationary, one wave per SIMD: a wave keeps the hi AND lo [32 cout x K] slabs of its cout tile in registers
//     (3x3, 64 channels: 72 fragments = 288 of the 512 registers) for the lifetime of a persistent workgroup;
//   * both planes of the halo tile arrive by direct global -> LDS DMA, double-buffered behind the contraction;
//   * per pair of LDS fragment reads the matrix pipe gets 3 MFMAs (the fp16 kernels: 1 per read), so the LDS / DMA side that
//     bounds the fp16 kernels has 1.5x the slack here and the contraction runs at the MFMA issue rate;
//   * optional chained 1x1 (stem pair 3x3 s2 -> 1x1; neck 1x1 -> first tower 1x1), optional second output (a stage's 1x1
//     stride-2 identity branch from the centre tap), GroupNorm sums of the stored values as order-independent 64-bit
//     fixed-point atomics, fp32 outputs for the cls / reg convs (+ Scale, lfd_head.py:180-183).
#pragma once
#include "common.h"
#include <utility>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace pl {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) -- every piece condition and
// every register index of an unrolled contraction is a constant by construction (an `#pragma unroll` the optimiser declines
// leaves the weight registers dynamically indexed, i.e. in scratch)
template <class F, int... Ks>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Ks...>) {
  (f(std::integral_constant<int, Ks>{}), ...);
}

constexpr float kLo = 2048.f, kInvLo = 1.f / 2048.f;
constexpr double kGnFix = 16777216.0;   // 2^24: fixed-point scale of the GroupNorm sums
constexpr int kGnRep = LFD_PL_GN_REPLICAS;   // replicas of the sums a producer spreads its atomics over (consumers add them up)

struct PlArgs {
  const _Float16* in;      // hi plane [N,H,W,CIN]; the lo plane `in_plane` halfs behind it
  long in_plane;
  _Float16* out;           // [N,OH,OW,cout_out] hi plane, lo plane `out_plane` halfs behind it
  long out_plane;
  const half8* w;          // packed [2 = hi | lo][cout/32][NK][64 lanes] x 8 halfs (ops.pack_conv_weight order per plane)
  long w_plane;            // half8 units between the planes
  const float* bias;       // [cout]
  const _Float16* res;     // residual planes [N,OH,OW,cout] or null
  long res_plane;
  const half8* w2;         // TAIL: chained 1x1 CMID -> cout2, [2][cout2/32][CMID/16][64]
  long w2_plane;
  const float* bias2;
  const half8* wds;        // DS: 1x1 stride-2 identity branch [2][cout/32][CIN/16][64]
  long wds_plane;
  const float* bds;
  _Float16* out_ds;        // DS: [N,OH,OW,cout] planes (no ReLU)
  long ds_plane;
  const _Float16* zeros;   // 4 KB line: [0,2048) zero (source of out-of-image pixels), [2048,4096) write-only trash
  int N, H, W, OH, OW;
  int cout, cout2, relu, relu2;
  int tiles_x, tiles_y, ntiles;
  unsigned long long* gn_acc;   // OUTM 1: [N][cout_out / 8][2] fixed-point sums (sum | sum of squares) of the stored values
  float* f_out0;           // OUTM 2: channels [0, f_c0) -> f_out0[n * f_img0 + pixel * f_c0 + c]
  float* f_out1;           //         channels [f_c0, f_c0 + f_c1) -> f_out1[n * f_img1 + pixel * f_c1 + c - f_c0] * scale1
  int f_c0, f_c1;
  long f_img0, f_img1;
  const float* scale1;     // one float or null
  // GNIN: the input planes hold the PRE-normalisation output of a conv whose GroupNorm sums are gnin_acc (same layout as
  // gn_acc); the landed tile is normalised + ReLU'd in LDS before the contraction (lfd_head.py:97-117 conv -> GroupNorm -> ReLU)
  const unsigned long long* gnin_acc;
  const float* gnin_gamma;
  const float* gnin_beta;
  float gnin_eps;
};

// One pyramid level of a MULTI-LEVEL launch (ML: the 1x1 convs of the neck / head over all levels' tiles in one persistent grid --
// the small levels' launches are pure latency: 12 of them took 159 us for a third of the first level's work)
struct PlLevel {
  const _Float16* in;
  _Float16* out;
  const half8* w;          // every level has its own filters (lfd_head.py:88-139: head%d_* modules) and GroupNorm affine
  const float* bias;
  const half8* w2;
  const float* bias2;
  const float* gnin_gamma;
  const float* gnin_beta;
  unsigned long long* gn_acc;
  const unsigned long long* gnin_acc;
  float* f_out0;
  float* f_out1;
  const float* scale1;
  long in_plane, out_plane;
  int H, W, tiles_x, tiles_y, tile_start, pad_;
};
struct PlLevels {
  PlLevel lv[LFD_MAX_LEVELS];
  int n;
};

enum { IN_NCHW_F32 = 0, IN_NHWC_F16 = 1, IN_NHWC_U8 = 2 };

// the element as loaded (bit pattern) -- nothing is computed from it at the load site: any use would make the compiler wait for
// the load right there, i.e. (vmcnt retires in order) for the previous tile's output stores in front of it, a full HBM write
// round trip per tile (measured: 9-11 k of a tile's 16 k cycles)
template <int FMT>
__device__ __forceinline__ uint32_t load_px_raw(const void* in, int n, int H, int W, int gy, int gx, int c) {
  if (FMT == IN_NCHW_F32) {
    return reinterpret_cast<const uint32_t*>(in)[(((size_t)n * 3 + c) * H + gy) * W + gx];
  } else if (FMT == IN_NHWC_F16) {
    return reinterpret_cast<const uint16_t*>(in)[(((size_t)n * H + gy) * W + gx) * 3 + c];
  } else {
    return reinterpret_cast<const uint8_t*>(in)[(((size_t)n * H + gy) * W + gx) * 3 + c];
  }
}
template <int FMT>
__device__ __forceinline__ float px_value(uint32_t raw) {
  if (FMT == IN_NCHW_F32) return __uint_as_float(raw);
  if (FMT == IN_NHWC_F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)raw);
  return ((float)raw / 255.f - 0.5f) / 0.5f;   // simple_normalize (augmentation_pipeline.py:31-36) in fp32, like the reference
}

// PRODUCE (k_pl_stem2x, planes_stem2x.hip): the input tile of a 3x3 stride-2 conv on 64 channels is not fetched but COMPUTED
// in LDS from the frame -- the first stem pair conv3x3 s2 (3 -> 64) + BN + ReLU -> conv1x1 + BN + ReLU (lfd_resnet.py:376-395)
// feeding the second pair (:396-413) without its 1.06 GB round trip through HBM (bs 8 at 1080p: the largest tensor of the net).
struct PlProd {
  const void* frame;       // [N,FH,FW,3] fp16 | uint8, or [N,3,FH,FW] fp32 (in_format of lfd_stem_conv_f16)
  int FH, FW;
  int dma_ok;              // fp16 frames with 4-byte aligned rows: the raw tile arrives by LDS-DMA, double-buffered
  const half8* w1;         // conv0 + its bias, [2 = hi | lo][2 slabs][2 k-steps][64 lanes] in the producer's k-slot order (see gather)
  const half8* w2;         // 1x1, [2][2][4][64], K in the order the conv0 accumulators hold the channels (see produce)
  const float* b2;         // [64]
};
struct ProdCfg {
  static constexpr int MH = 9, MW = 33, NPX = MH * MW, NROUND = (NPX + 63) / 64;   // the consumer's 9 x 33 input tile
  static constexpr int FR = 2 * MH + 1;                  // frame rows of the tile
  static constexpr int FJ = (2 * MW + 2) * 3;            // halfs of a frame row the gather may touch: one junk pixel + 2 MW + 1 pixels
  static constexpr int JUNK = 7;                         // halfs left of the patch in an LDS row (3 = one pixel: dword-aligned gather;
                                                         // + 4: the row's first byte in the frame is 16-byte aligned for the DMA)
  static constexpr int FPITCH = 256;                     // halfs: 32 DMA lanes x 16 B
  static constexpr int FRAME_BYTES = (FR + 1) * FPITCH * 2;
  static constexpr int FRAME_OFF = 0;                    // 2 tiles: DMA double buffer (fp16 frames) | hi + lo of one tile
  static constexpr int W4_OFF = FRAME_OFF + 2 * FRAME_BYTES;       // the consumer's chained 1x1 [2][2][4][64]
  static constexpr int W1_OFF = W4_OFF + 2 * 2 * 4 * 64 * 16;      // conv0 [2][2][2][64]
  static constexpr int PB_OFF = W1_OFF + 2 * 2 * 2 * 64 * 16;      // the producer 1x1's bias [64]
  static constexpr int XCH_OFF = PB_OFF + 64 * 4;                  // fragment exchange between the two slab waves of a pixel group:
  static constexpr int XCH_BYTES = 4 * 4 * 1024;                   //   [wave][2 k-steps x (hi, lo)][64 lanes x 16 B], double-buffered
  static constexpr int BYTES = XCH_OFF + 2 * XCH_BYTES;
};

// 4-byte LDS-DMA (64 lanes x 4 B contiguous at the wave's LDS base)
__device__ __forceinline__ void dma4(const void* g, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(m0v) : "memory");
}

// PTO: 32-pixel MFMA tiles per wave (0: 2 for stride 1, 1 for stride 2)
template <int CIN, int KS, int S, int NCT, bool TAIL, bool RES, int PTO, bool PROD = false>
struct PCfg {
  static constexpr int PT = PTO ? PTO : ((S == 1) ? 2 : 1);
  static constexpr int TW = (S == 1 && !(CIN == 64 && KS == 3)) ? 32 : 16;
  static constexpr int RPT = 32 / TW;
  static constexpr int PG = 4 / NCT;
  static constexpr int TH = PG * PT * RPT;
  static constexpr int PAD = KS / 2;
  static constexpr int IH = (TH - 1) * S + KS;
  static constexpr int IW = (TW - 1) * S + KS;
  static constexpr int IWh = (IW + 1) / 2;
  static constexpr int IWs = (S == 2) ? 2 * IWh : ((IW + 1) & ~1);
  static constexpr int CPP = CIN / 8;
  static constexpr int PIXB = CIN * 2;
  static constexpr int PPR = (CPP >= 16) ? 1 : 16 / CPP;
  static constexpr int NSLOT = IH * IWs;
  static constexpr int IN_BYTES = ((NSLOT * PIXB + 1023) / 1024) * 1024;    // one plane of one buffer
  static constexpr int NK = KS * KS * CIN / 16;
  static constexpr int NQ = CIN / 16;
  static constexpr int CMID = NCT * 32;                 // channels of the main conv's output this workgroup holds
  static constexpr int MCPP = CMID / 8, MPIXB = CMID * 2;
  static constexpr int MPPR = (MCPP >= 16) ? 1 : 16 / MCPP;
  static constexpr int NK2 = CMID / 16;
  static constexpr int OPX = PG * PT * 32;              // pixels of the output tile
  static constexpr int MID_PLANE = TAIL ? OPX * MPIXB : 0;
  static constexpr int OUT_PLANE = OPX * NCT * 64;      // staging tile, one plane
  // scratch: the chained 1x1's operand planes, then (same bytes, one barrier later) the output staging planes; it lives in
  // the consumed input buffer when it fits there
  static constexpr int SCR_BYTES = 2 * (MID_PLANE > OUT_PLANE ? MID_PLANE : OUT_PLANE);
  static constexpr bool ALIAS = SCR_BYTES <= 2 * IN_BYTES;
  // (one 32-cout slab of a 128-channel 1x1 -- the cls / reg output conv: 4 x 32 pixels x 512 B = 64 KB per tile; single-buffered
  //  it fits a CU twice, and two workgroups hide each other's fetch AND each other's GroupNorm transform; double-buffered at
  //  one workgroup per CU the launch took 1.8 x as long)
  static constexpr int NBUF = (PROD || (CIN == 128 && KS == 1 && NCT == 1)) ? 1 : ((S == 1 || (CIN == 64 && KS == 3 && NCT == 2)) ? 2 : 1);
  static constexpr int SCR_OFF = NBUF * 2 * IN_BYTES;
  static constexpr int BIAS_OFF = SCR_OFF + (ALIAS ? 0 : SCR_BYTES);
  // RES: the residual tile (this workgroup's channel slice, both planes) arrives by DMA in copy-out order
  static constexpr int RES_OFF = BIAS_OFF + 3 * 128 * 4;
  static constexpr int RES_PLANE = RES ? OUT_PLANE : 0;
  static constexpr int GNR_OFF = RES_OFF + 2 * RES_PLANE;    // 1 KB: cross-wave reduction of the GroupNorm sums (OUTM 1)
  static constexpr int LDS_BYTES = GNR_OFF + 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS capacity");
};

// scheduling fence around the MFMAs of a k-step (pins the operand ring's prefetch distance)
#ifdef PL_NO_SB
#define PL_SB()
#else
#define PL_SB() __builtin_amdgcn_sched_barrier(0)
#endif

#ifdef LFD_PL_TIMING
#ifndef PL_DBG_BLOCK
#define PL_DBG_BLOCK 0
#endif
static __device__ unsigned long long g_pl_dbg[8 * 16 + 8 + 8];   // (one per translation unit)
#define PL_T(i) do { if (blockIdx.x == PL_DBG_BLOCK && blockIdx.y == 0 && threadIdx.x == 0 && dbg_it < 8) g_pl_dbg[dbg_it * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PL_T(i)
#endif

__device__ __forceinline__ void dma16(const void* g, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory");
}

// the same DMA with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit byte offset (global_load_lds saddr form): no
// per-lane 64-bit address arithmetic at the issue site -- the offsets are per-kernel constants of the lane
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory");
}

__device__ __forceinline__ void block_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// y (fp32) -> packed fp16 pair of the hi plane and of the lo plane.  hi = RNE(y) (v_cvt_pk_f16_f32), y - hi is exact in fp32,
// so is the scaling by 2^11; the second rounding (lo) leaves |y - hi - 2^-11 lo| <= 2^-23 |y|.
// (2048 y - 2048 hi is exact in fp32, so the fused form rounds once, to fp16, exactly like (y - hi) * 2048 -> fp16; it compiles
//  to v_mul + v_fma_mix{lo,hi}_f16 per element instead of cvt-back + sub + mul + cvt)
__device__ __forceinline__ void split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
  lfd_f32x2 f; f[0] = y0; f[1] = y1;
  union { lfd_f16x2 v; uint32_t u; } h;
  h.v = __builtin_convertvector(f, lfd_f16x2);
  const lfd_f32x2 sc = f * kLo;
  // lo halves straight out of the mixed-precision FMA (fp16 hi x fp32 -2048 + fp32 2048 y, rounded to fp16 into the low /
  // high half of one register): 4 instructions per pair -- v_cvt_pk, v_pk_mul, v_fma_mixlo, v_fma_mixhi.  Left to the
  // compiler the same expression became cvt-back x 2 + fmamk x 2 + cvt_pk (7 per pair); the splits are most of the VALU
  // work of every epilogue of this file.
  uint32_t l;
  const float m = -kLo;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h.u), "v"(m), "v"(sc[0]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h.u), "v"(m), "v"(sc[1]));
  hi = h.u; lo = l;
}

// hi + 2^-11 lo: one v_fma_mix_f32 on the fp16 operands (2^-11 lo is exact, the sum rounds once)
__device__ __forceinline__ float join1(_Float16 hi, _Float16 lo) { return fmaf((float)lo, kInvLo, (float)hi); }
// main + 2^-11 corr of the two accumulator sets
__device__ __forceinline__ float comb(float m, float c) { return fmaf(c, kInvLo, m); }

// OUTM: 0 = planes, 1 = planes + GroupNorm sums (groups of 8 channels), 2 = fp32 outputs (cls / reg)
template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, int OUTM, int PTO, bool GNIN, bool ML = false, int PFMT = -1>
__device__ __forceinline__ void pl_block(const PlArgs& a0, const PlLevels& L, char* smem, const PlProd& P) {
  constexpr bool PROD = PFMT >= 0;
  static_assert(!PROD || (CIN == 64 && KS == 3 && S == 2 && NCT == 2 && WREG && !RES && !DS && !ML && !GNIN), "PRODUCE: the second stem pair");
  static_assert(!ML || (KS == 1 && S == 1 && WREG && !RES && !DS), "multi-level launches: the 1x1 convs of the neck / head");
  PlArgs a = a0;            // (ML: the per-level fields are switched when the tile walk enters another level)
  static_assert(!GNIN || (KS == 1 && S == 1 && CIN == 128 && !TAIL && !RES && !DS), "GNIN: a 1x1 conv on a 128-channel GroupNorm(16) input");
  static_assert(!DS || (KS == 3 && S == 2 && !TAIL && !RES && OUTM == 0), "DS rides on a 3x3 stride-2 conv");
  static_assert(!TAIL || (!RES && !DS), "TAIL: conv -> 1x1 in one launch");
  static_assert(OUTM != 2 || (!TAIL && !RES && !DS), "fp32 outputs: the bare cls / reg conv");
  using C = PCfg<CIN, KS, S, NCT, TAIL, RES, PTO, PROD>;
  static_assert(!PROD || (C::IH == ProdCfg::MH && C::IW == ProdCfg::MW && C::LDS_BYTES + ProdCfg::BYTES <= 160 * 1024), "producer tile");
  constexpr int PT = C::PT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ct = wave % NCT;
  const int pg = wave / NCT;
  const int h = lane >> 5;
  const int pix = lane & 31;
  const int oyl = pix / C::TW, oxl = pix % C::TW;
  const int cog = blockIdx.y;
  const int co_base = (cog * NCT + ct) * 32;

  float* sbias = reinterpret_cast<float*>(smem + C::BIAS_OFF);
  if (threadIdx.x < NCT * 32) {
    sbias[threadIdx.x] = a.bias[cog * NCT * 32 + threadIdx.x];
    if constexpr (DS) sbias[128 + threadIdx.x] = a.bds[cog * NCT * 32 + threadIdx.x];
    if constexpr (TAIL) sbias[256 + threadIdx.x] = a.bias2[threadIdx.x];
  }

  // ---- stationary weights: hi and lo slabs of this wave's cout tile
  constexpr int NKR = WREG ? C::NK : 1;
  half8 wh[NKR], wl[NKR];
  const half8* wsrc = a.w + ((size_t)(cog * NCT + ct) * C::NK) * 64 + lane;
  if constexpr (WREG) {
#pragma unroll
    for (int k = 0; k < NKR; ++k) {
      wh[k] = wsrc[(size_t)k * 64];
      wl[k] = wsrc[a.w_plane + (size_t)k * 64];
    }
  }
  // (PRODUCE: the chained 1x1's fragments stay in LDS -- the producer's working set needs their 32 registers)
  constexpr bool TAILW_REG = TAIL && PFMT < 0;
  half8 w2h[TAILW_REG ? C::NK2 : 1], w2l[TAILW_REG ? C::NK2 : 1];
  if constexpr (TAILW_REG) {
#pragma unroll
    for (int k = 0; k < C::NK2; ++k) {
      w2h[k] = a.w2[((size_t)ct * C::NK2 + k) * 64 + lane];
      w2l[k] = a.w2[a.w2_plane + ((size_t)ct * C::NK2 + k) * 64 + lane];
    }
  }
  half8 wdh[DS ? C::NQ : 1], wdl[DS ? C::NQ : 1];
  if constexpr (DS) {
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) {
      wdh[q] = a.wds[((size_t)(cog * NCT + ct) * C::NQ + q) * 64 + lane];
      wdl[q] = a.wds[a.wds_plane + ((size_t)(cog * NCT + ct) * C::NQ + q) * 64 + lane];
    }
  }

  // ---- per-lane LDS read offsets (conv_impl.h: one per column tap and 16-channel group; XOR term formed at the read for 128 ch)
  constexpr bool XTAB = C::NQ <= 4;
  int xoff[KS][XTAB ? C::NQ : 1];
  int xbase[KS], xkey[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int ix = oxl * S + s;
    const int rem = (S == 2) ? ((ix & 1) * C::IWh + (ix >> 1)) : ix;
    const int f = (rem / C::PPR) % C::CPP;
    const int rowbase = ((pg * PT * C::RPT + oyl) * S) * C::IWs + rem;
    xbase[s] = rowbase * C::PIXB;
    xkey[s] = f ^ h;
#pragma unroll
    for (int q = 0; q < (XTAB ? C::NQ : 1); ++q) xoff[s][q] = rowbase * C::PIXB + (((2 * q + h) ^ f) * 16);
  }

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int x_begin = xcd * per_xcd;
  const int x_end = (x_begin + per_xcd) < a.ntiles ? (x_begin + per_xcd) : a.ntiles;
  const int wgs_xcd = (nblk + 7 - xcd) / 8;
  // the workgroup's tiles inside its XCD's range: strided by the workgroups of the XCD -- or, for the multi-level launches
  // (round 6), ONE contiguous run: a workgroup then stays inside an image for many tiles.  Every image change costs a GNIN
  // consumer a dependent chain of global loads (the producer's sums -> mean / rstd -> scale / shift, 2-3 us) and an OUTM 1
  // producer a flush of its sums (two barriers + atomics); with the strided walk every tile of the small pyramid levels
  // (fewer tiles per image than the stride) paid that: the 25 % of the pixels outside the first level took as long as the
  // first level (measured on k_pl_head, csrc/planes_head.hip: 121 -> 84 us)
  const int ml_chunk = (x_end - x_begin + wgs_xcd - 1) / (wgs_xcd > 0 ? wgs_xcd : 1);
  const int t_begin = ML ? x_begin + bix * ml_chunk : x_begin + bix;
  const int t_end = ML ? ((t_begin + ml_chunk) < x_end ? (t_begin + ml_chunk) : x_end) : x_end;
  const int t_step = ML ? 1 : wgs_xcd;
  int tiles_per_img = a.tiles_x * a.tiles_y, tile0 = 0, cur_l = ML ? -1 : 0;
  (void)cur_l;
  const long in_plane_b = a.in_plane * 2;    // bytes
  // ML: the levels' first tiles live in SGPRs (a lookup per tile from the kernel arguments costs a chain of scalar loads --
  // comparable to the 8 k-steps of a 1x1 tile)
  int lstart[ML ? LFD_MAX_LEVELS : 1];
  if constexpr (ML) {
#pragma unroll
    for (int i = 0; i < LFD_MAX_LEVELS; ++i) lstart[i] = i < L.n ? L.lv[i].tile_start : 0x7fffffff;
  }
  (void)lstart;
  auto level_of = [&](int t) {
    int l = 0;
    if constexpr (ML) {
#pragma unroll
      for (int i = 1; i < LFD_MAX_LEVELS; ++i) l += (t >= lstart[i]) ? 1 : 0;
    }
    return l;
  };
  (void)level_of;

  // ---- tile loaders: every DMA of the fp16 kernels (conv_impl.h) issued for both planes; the LDS image of a plane is the
  //      fp16 kernels' (lane-linear lines, XOR chunk swizzle on the source side), the lo plane IN_BYTES behind the hi plane
  // Issued in PIECES (one or two row-aligned DMA pairs each): with two input buffers the pieces of tile t+1 are spread over
  // the k-steps of tile t's contraction, where their address arithmetic and issue slots hide under the MFMAs (one wave per
  // SIMD: whatever is issued between two tiles is exposed; the ~16 DMA instructions per wave and tile of the 3x3 kernels were
  // ~1.6 k cycles of a ~12 k-cycle tile).
  constexpr bool FAST = (CIN == 64 && KS == 3 && S == 1 && NCT == 2 && !TAIL && !DS && PT == 2);
  constexpr bool FAST2 = (CIN == 64 && KS == 3 && S == 2);
  constexpr int SPW = 64 / C::CPP;                      // (generic walk) pixel slots per wave instruction
  constexpr int NP2 = (C::IWs + 7) / 8;                 // (FAST2) instructions per halo row
  constexpr int NPIECE = FAST ? 8 : (FAST2 ? NP2 * ((C::IH + 3) / 4) : (C::NSLOT + 4 * SPW - 1) / (4 * SPW));
  const long f_rowpitch = (long)a.W * (CIN * 2);
  // tile-level scalars of the tile being fetched (ML: of ITS level -- the prefetched tile may lie in the next level)
  int d_n = 0, d_gy0 = 0, d_gx0 = 0;
  bool d_interior = false;
  char* d_lbase = smem;
  const _Float16* d_in = a.in;
  long d_inpb = in_plane_b;
  int d_H = a.H, d_W = a.W;
  auto dma2 = [&](const char* src, bool valid, const char* zsrc, char* ldst) {
    dma16(valid ? src : zsrc, ldst);
    dma16(valid ? src + d_inpb : zsrc, ldst + C::IN_BYTES);
  };
  int d_l = -1, d_tpi = tiles_per_img, d_tx = a.tiles_x, d_t0 = 0;
  (void)d_l;
  auto dma_setup = [&](int t, int buf) {
    if constexpr (ML) {
      const int l = level_of(t);
      if (l != d_l) {
        const PlLevel& lv = L.lv[l];
        d_l = l;
        d_in = lv.in; d_inpb = lv.in_plane * 2; d_H = lv.H; d_W = lv.W;
        d_tpi = lv.tiles_x * lv.tiles_y; d_tx = lv.tiles_x; d_t0 = lv.tile_start;
      }
    }
    const int tpi = d_tpi, tx_ = d_tx, t0 = d_t0;
    d_n = (t - t0) / tpi;
    const int tr = (t - t0) - d_n * tpi;
    const int ty0 = tr / tx_, tx0 = tr - ty0 * tx_;
    d_gy0 = ty0 * C::TH * S - C::PAD;
    d_gx0 = tx0 * C::TW * S - C::PAD;
    d_lbase = smem + buf * 2 * C::IN_BYTES;
    d_interior = d_gy0 >= 0 && d_gx0 >= 0 && d_gy0 + C::IH <= a.H && d_gx0 + C::IW <= a.W;
  };
  auto dma_piece = [&](int p) {
    if constexpr (FAST) {
      // halo rows of 18 pixels: columns 0..15 are two 64-lane instructions (8 pixels x 8 chunks), columns 16..17 one 16-lane
      // instruction; source chunk swizzled (key = column >> 1), LDS lane-linear
      const char* p00 = reinterpret_cast<const char*>(a.in) + ((long)d_n * a.H + d_gy0) * f_rowpitch + (long)d_gx0 * (CIN * 2);
      if (p < 5) {
        const int f_lpx = lane >> 3;
        const int m = wave + 4 * p;
        const int iy = m >> 1, hf = m & 1;
        const int cc = (((lane & 7) ^ (f_lpx >> 1)) * 16) ^ (hf ? 64 : 0);
        const int gy = d_gy0 + iy, gx = d_gx0 + 8 * hf + f_lpx;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        dma2(p00 + iy * f_rowpitch + hf * 1024 + f_lpx * 128 + cc, ok, reinterpret_cast<const char*>(a.zeros) + cc,
             d_lbase + (iy * C::IWs + 8 * hf) * C::PIXB);
      } else {
        const int iy = wave + 4 * (p - 5);
        if (iy < C::IH && lane < 16) {
          const int gy = d_gy0 + iy, gx = d_gx0 + 16 + (lane >> 3);
          const bool ok = gy >= 0 && gy < a.H && gx < a.W;
          dma2(p00 + iy * f_rowpitch + 2048 + lane * 16, ok, reinterpret_cast<const char*>(a.zeros) + (lane & 7) * 16,
               d_lbase + (iy * C::IWs + 16) * C::PIXB);
        }
      }
    } else if constexpr (FAST2) {
      // column-de-interleaved stride-2 tile: a halo row is IWs slots (even columns, then odd columns), NP2 row-aligned
      // instructions of 8 slots; wave w owns rows w, w + 4, ...
      const int iy = wave + 4 * (p / NP2), j = p % NP2;
      if (iy < C::IH) {
        const int sl = lane >> 3, cs = lane & 7;
        const int rem = 8 * j + sl;
        const int ix = rem < C::IWh ? 2 * rem : 2 * rem - (2 * C::IWh - 1);
        const int c = cs ^ ((rem / C::PPR) % C::CPP);
        const int gy = d_gy0 + iy, gx = d_gx0 + ix;
        const bool needed = rem < C::IWs && ix < C::IW;
        if (d_interior) {
          // every halo pixel inside the image (all but the border tiles): wave-uniform row base in SGPRs + a 32-bit lane
          // offset -- no per-lane 64-bit address, no validity selects at the issue site
          if (8 * j + 8 <= C::IWs || needed) {
            const char* rb = reinterpret_cast<const char*>(a.in) + ((long)d_n * a.H + gy) * f_rowpitch + (long)d_gx0 * (CIN * 2);
            const unsigned vo = (unsigned)(ix * (CIN * 2) + c * 16);
            char* ld = d_lbase + (iy * C::IWs + 8 * j) * C::PIXB;
            dma16s(rb, vo, ld);
            dma16s(rb + in_plane_b, vo, ld + C::IN_BYTES);
          }
          return;
        }
        const bool ok = needed && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const char* src = reinterpret_cast<const char*>(a.in) + ((long)d_n * a.H + (ok ? gy : 0)) * f_rowpitch + (long)(ok ? gx : 0) * (CIN * 2) + c * 16;
        if (8 * j + 8 <= C::IWs || needed)
          dma2(src, ok, reinterpret_cast<const char*>(a.zeros) + c * 16, d_lbase + (iy * C::IWs + 8 * j) * C::PIXB);
      }
    } else {
      const int slot0 = (wave + 4 * p) * SPW;
      const int pslot = slot0 + lane / C::CPP;
      const int cs = lane % C::CPP;
      if (pslot < C::NSLOT) {
        const int iy = pslot / C::IWs;
        const int rem = pslot - iy * C::IWs;
        const int ix = (S == 2) ? ((rem < C::IWh) ? 2 * rem : 2 * (rem - C::IWh) + 1) : rem;
        const int c = cs ^ ((rem / C::PPR) % C::CPP);
        const int gy = d_gy0 + iy, gx = d_gx0 + ix;
        bool needed = ix < C::IW;
        if (KS == 1 && S == 2) needed = needed && !(ix & 1) && !(iy & 1);
        if (needed) {
          const bool valid = (gy >= 0) && (gy < d_H) && (gx >= 0) && (gx < d_W);
          const char* src = reinterpret_cast<const char*>(d_in + (((size_t)d_n * d_H + (valid ? gy : 0)) * d_W + (valid ? gx : 0)) * CIN + c * 8);
          dma2(src, valid, reinterpret_cast<const char*>(a.zeros + c * 8), d_lbase + slot0 * C::PIXB);
        }
      }
    }
  };
  auto issue_dma = [&](int t, int buf) {
    dma_setup(t, buf);
    for (int p = 0; p < NPIECE; ++p) dma_piece(p);
  };
  // pieces of the next tile issued from inside the contraction (WREG kernels with two buffers): piece i goes in front of
  // k-step (i * SPREAD) / NPIECE, i.e. they are done after ~70 % of the k-steps and have the rest of the tile to land
  constexpr bool DMA_IN_LOOP = WREG && C::NBUF == 2;
  constexpr int SPREAD = (C::NK * 7 + 9) / 10;

  // RES: the identity branch's tile -> LDS in copy-out order (pixel-major lines of this workgroup's channel slice), added
  // where the output lines are formed: no registers held across the contraction, no exposed load latency
  auto issue_res = [&](int n, int ty0, int tx0) {
    if constexpr (RES) {
      constexpr int OCPP_ = NCT * 4, PPI = 64 / OCPP_, NI = C::OPX / PPI;
      const int c = lane % OCPP_;
      for (int j = wave; j < NI; j += 4) {
        const int pl_ = j * PPI + lane / OCPP_;
        const int oy = ty0 * C::TH + pl_ / C::TW, ox = tx0 * C::TW + pl_ % C::TW;
        const bool ok = oy < a.OH && ox < a.OW;
        const _Float16* src = a.res + (((size_t)n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.cout + cog * NCT * 32 + c * 8;
        const _Float16* z = a.zeros + c * 8;
        dma16(ok ? src : z, smem + C::RES_OFF + j * 1024);
        dma16(ok ? src + a.res_plane : z, smem + C::RES_OFF + C::RES_PLANE + j * 1024);
      }
    }
  };

  // OUTM 1: running GroupNorm sums of the values this thread copies out (always chunk threadIdx.x % OCPP = one group of 8
  // channels), in fp64 across tiles; flushed as fixed-point atomics when the walk leaves an image
  double gn_s = 0., gn_q = 0.;
  int gn_n = -1;
  auto gn_flush = [&]() {
    if constexpr (OUTM == 1) {
      constexpr int OCPP = NCT * 4;
      double s = gn_s, q = gn_q;
#pragma unroll
      for (int d = 32; d >= OCPP; d >>= 1) {
        s += __shfl_xor(s, d);
        q += __shfl_xor(q, d);
      }
      // the four waves through LDS, then ONE pair of atomics per group and workgroup, spread over kGnRep replicas of the
      // sums (workgroup b -> replica b % kGnRep).  Every workgroup of a launch adds to the same 16 x 2 words per image:
      // one pair per WAVE onto one replica measured 98 us for the 506-tile level of a single 1080p frame against 65 us
      // for the same level of eight frames -- pure contention on 32 addresses.
      double* red = reinterpret_cast<double*>(smem + C::GNR_OFF);       // [4 waves][OCPP][2]
      static_assert(4 * OCPP * 2 * 8 <= 1024, "reduction scratch");
      if (lane < OCPP) {
        red[(wave * OCPP + lane) * 2] = s;
        red[(wave * OCPP + lane) * 2 + 1] = q;
      }
      block_barrier();
      if (wave == 0 && lane < OCPP && gn_n >= 0) {
        double ts = 0., tq = 0.;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          ts += red[(w4 * OCPP + lane) * 2];
          tq += red[(w4 * OCPP + lane) * 2 + 1];
        }
        const int grp = cog * OCPP + lane;
        const int ngrp = (TAIL ? a.cout2 : a.cout) / 8;
        unsigned long long* dst = a.gn_acc + ((((size_t)(blockIdx.x % kGnRep)) * a.N + gn_n) * ngrp + grp) * 2;
        atomicAdd(dst, (unsigned long long)__double2ll_rn(ts * kGnFix));
        atomicAdd(dst + 1, (unsigned long long)__double2ll_rn(tq * kGnFix));
      }
      block_barrier();
      gn_s = gn_q = 0.;
    }
  };

  int dbg_it = 0; (void)dbg_it;     // tile counter of this workgroup
  // ---- PRODUCE: the first stem pair computed into the input tile
  char* const psm = smem + C::LDS_BYTES;
  half8 pw2h[PROD ? 4 : 1], pw2l[PROD ? 4 : 1];
  (void)psm; (void)pw2h; (void)pw2l;
  if constexpr (PROD) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // k-steps in the order the wave has them: its own conv0 slab's two first, then the other slab's
      const int q = j < 2 ? 2 * ct + j : 2 * (ct ^ 1) + (j - 2);
      pw2h[j] = P.w2[(ct * 4 + q) * 64 + lane];
      pw2l[j] = P.w2[2 * 4 * 64 + (ct * 4 + q) * 64 + lane];
    }
    half8* w1s = reinterpret_cast<half8*>(psm + ProdCfg::W1_OFF);
    for (int i = threadIdx.x; i < 2 * 2 * 2 * 64; i += 256) w1s[i] = P.w1[i];
    half8* w4s = reinterpret_cast<half8*>(psm + ProdCfg::W4_OFF);
    for (int i = threadIdx.x; i < 2 * 2 * 4 * 64; i += 256) w4s[i] = a.w2[i];
    float* pb = reinterpret_cast<float*>(psm + ProdCfg::PB_OFF);
    if (threadIdx.x < 64) pb[threadIdx.x] = P.b2[threadIdx.x];
  }
  // tile t -> image, mid-tensor origin (gy0, gx0) of the 9 x 33 input tile, frame origin of its 19 x 68 pixel patch
  auto prod_origin = [&](int t, int& n, int& gy0, int& gx0) {
    n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    gy0 = ty0 * C::TH * 2 - 1;
    gx0 = tx0 * C::TW * 2 - 1;
  };
  // fp16 frames whose rows are 16-byte aligned (FW % 8 == 0): the raw patch by 16-byte LDS-DMA, two 512-byte rows per
  // instruction.  A row starts ProdCfg::JUNK halfs left of the patch: the byte offset (64 tx0 - 4) * 6 - 8 of that column is a
  // multiple of 16, and so are both image edges (whole lanes are in or out)
  auto frame_dma = [&](int t, int fb) {
    if constexpr (PROD) {
      int n, gy0, gx0;
      prod_origin(t, n, gy0, gx0);
      const int fy0 = 2 * gy0 - 1, fxm = 2 * gx0 - 2;
      const _Float16* fr = reinterpret_cast<const _Float16*>(P.frame);
      char* lbase = psm + ProdCfg::FRAME_OFF + fb * ProdCfg::FRAME_BYTES;
      constexpr int NI = (ProdCfg::FR + 1) / 2;
      constexpr int NL = (ProdCfg::FJ + ProdCfg::JUNK - 3 + 7) / 8;      // 16-byte lanes of a row
      const int hcol = fxm * 3 - ProdCfg::JUNK + 3 + 8 * (lane & 31);
      const bool colok = (lane & 31) < NL && hcol >= 0 && hcol < P.FW * 3;
      for (int i = wave; i < NI; i += 4) {
        const int fy = fy0 + 2 * i + (lane >> 5);
        const bool ok = colok && fy >= 0 && fy < P.FH;
        const _Float16* src = ok ? fr + ((size_t)n * P.FH + fy) * P.FW * 3 + hcol : a.zeros;
        dma16(src, lbase + i * 1024);
      }
    }
  };
  // any format: load, convert (simple_normalize for uint8), split, store the hi and lo patch -- no prefetch (fallback path)
  auto frame_fill = [&](int t) {
    if constexpr (PROD) {
      int n, gy0, gx0;
      prod_origin(t, n, gy0, gx0);
      const int fy0 = 2 * gy0 - 1, fxm = 2 * gx0 - 2;
      _Float16* f0 = reinterpret_cast<_Float16*>(psm + ProdCfg::FRAME_OFF);
      _Float16* f1 = reinterpret_cast<_Float16*>(psm + ProdCfg::FRAME_OFF + ProdCfg::FRAME_BYTES);
      constexpr int NE = ProdCfg::FR * ProdCfg::FJ;
      for (int i = threadIdx.x; i < NE; i += 256) {
        const int r = i / ProdCfg::FJ, j = i - r * ProdCfg::FJ;
        const int px = j / 3, c = j - px * 3;
        const int fy = fy0 + r, fx = fxm + px;
        const bool ok = fy >= 0 && fy < P.FH && fx >= 0 && fx < P.FW;
        const float v = ok ? px_value<PROD ? PFMT : 0>(load_px_raw<PROD ? PFMT : 0>(P.frame, n, P.FH, P.FW, fy, fx, c)) : 0.f;
        const _Float16 hh = (_Float16)v;
        f0[r * ProdCfg::FPITCH + ProdCfg::JUNK - 3 + j] = hh;
        if constexpr (PFMT != IN_NHWC_F16) f1[r * ProdCfg::FPITCH + ProdCfg::JUNK - 3 + j] = (_Float16)((v - (float)hh) * kLo);
      }
    }
  };
  // the 297 pixels of the input tile in rounds of 64; wave = (cout slab) x (32-pixel group), like the consumer.  A wave computes
  // conv0 (two k-steps gathered from the raw patch; the bias rides on a constant-one k-slot) for ITS slab of 32 channels.  The
  // 32x32 accumulator layout hands lane (h, pixel) the channels {8g + 4h + e}: the eight values of two g's, packed to fp16
  // hi / lo, ARE the lane's B fragment of a k-step of the 1x1 whose K runs in that channel order (the host packs the 1x1's
  // filter to match) -- the intermediate never takes the [pixel][channel] form.  The two slab waves of a pixel group swap
  // their two k-steps through LDS in fragment order (4 KB each way, conflict-free b128) and run the 1x1 for their own slab.
  // (As [pixel][channel] planes in LDS the producer was LDS-bound: 44 KB per wave and round on the CU's one 128 B / clk pipe;
  //  with every wave computing both slabs itself it was VALU-bound on the doubled hi / lo splits.)
  auto produce = [&](int t, int fb) {
    if constexpr (PROD) {
      constexpr bool HASLO = PFMT != IN_NHWC_F16;
      constexpr int NR = ProdCfg::NROUND;
      int n, gy0, gx0;
      prod_origin(t, n, gy0, gx0);
      const uint32_t* fh = reinterpret_cast<const uint32_t*>(psm + ProdCfg::FRAME_OFF + (HASLO ? 0 : fb) * ProdCfg::FRAME_BYTES);
      const uint32_t* fl = reinterpret_cast<const uint32_t*>(psm + ProdCfg::FRAME_OFF + ProdCfg::FRAME_BYTES);
      const half8* w1s = reinterpret_cast<const half8*>(psm + ProdCfg::W1_OFF) + lane;
      const float* pb = reinterpret_cast<const float*>(psm + ProdCfg::PB_OFF);
      constexpr int RD = ProdCfg::FPITCH / 2;      // dwords per patch row
      // k-slots: step 0 {h=0: row 0 halfs [junk, e0..e6], h=1: row 2 [junk, e0..e6]}, step 1 {h=0: row 1 [junk, e0..e6],
      // h=1: (row 0 e7 e8, row 1 e7 e8, row 2 e7 e8, ONE, pad)}, e = 3 dx + c; junk / pad slots have zero weights, the slot
      // of the constant one carries the bias
      const int go0 = h ? 4 : RD, go1 = h ? RD + 4 : RD + 1, go2 = h ? 2 * RD + 4 : RD + 2;
      auto gather = [&](const uint32_t* f, int base0, uint32_t one, half8& g0, half8& g1) {
        union { half8 v; uint32_t u[4]; } f0, f1;
        const uint32_t* a0p = f + base0 + (h ? 2 * RD : 0);
        f0.u[0] = a0p[0]; f0.u[1] = a0p[1]; f0.u[2] = a0p[2]; f0.u[3] = a0p[3];
        const uint32_t* b0 = f + base0;
        f1.u[0] = b0[go0]; f1.u[1] = b0[go1]; f1.u[2] = b0[go2];
        const uint32_t last = b0[RD + 3];
        f1.u[3] = h ? one : last;
        g0 = f0.v; g1 = f1.v;
      };
      // (main, corr) accumulators of 32 channels -> ReLU -> the two k-step fragments (hi, lo) they form
      auto to_frag = [&](const f32x16& m, const f32x16& c, half8* xh, half8* xl) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          union { half8 v; uint32_t w[4]; } vh, vl;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y0 = fmaxf(comb(m[8 * u + 2 * k], c[8 * u + 2 * k]), 0.f);
            const float y1 = fmaxf(comb(m[8 * u + 2 * k + 1], c[8 * u + 2 * k + 1]), 0.f);
            split2(y0, y1, vh.w[k], vl.w[k]);
          }
          xh[u] = vh.v; xl[u] = vl.v;
        }
      };
      // this wave's conv0 fragments (registers for the lifetime of the tile: 16)
      const half8 wah = w1s[((0 * 2 + ct) * 2 + 0) * 64], wbh = w1s[((0 * 2 + ct) * 2 + 1) * 64];
      const half8 wal = w1s[((1 * 2 + ct) * 2 + 0) * 64], wbl = w1s[((1 * 2 + ct) * 2 + 1) * 64];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // software pipeline over the rounds -- one wave per SIMD has nobody else to hide its MFMA -> VALU -> LDS chain behind:
      // conv0 of round r + 1 is issued IN FRONT of the 1x1 of round r, so the VALU splits round r + 1's conv0 result while the
      // matrix pipe runs the 1x1, then splits the 1x1's output (a serial chain measured 2.1 k cycles a round)
      f32x16 am, ac;
      half8 xh[4], xl[4];      // k-steps of the 1x1: this wave's conv0 slab first
      auto conv0 = [&](int r) {
        const int p = (2 * r + pg) * 32 + pix;
        const int pc = p < ProdCfg::NPX ? p : ProdCfg::NPX - 1;
        const int my = pc / ProdCfg::MW, mx = pc - my * ProdCfg::MW;
        const int base0 = (2 * my) * RD + 3 * mx + (ProdCfg::JUNK - 1) / 2;
        half8 x0h, x1h, x0l, x1l;
        gather(fh, base0, 0x3c00u, x0h, x1h);
        if constexpr (HASLO) gather(fl, base0, 0u, x0l, x1l);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah, x0h, zero16, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wal, x0h, zero16, 0, 0, 0);
        am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh, x1h, am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbl, x1h, ac, 0, 0, 0);
        if constexpr (HASLO) {
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah, x0l, ac, 0, 0, 0);
          ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wbh, x1l, ac, 0, 0, 0);
        }
      };
      // swap with the other slab's wave of this pixel group (wave = pg * 2 + ct)
      auto exchange = [&](int r, const half8* oh, const half8* ol) {
        half8* xw = reinterpret_cast<half8*>(psm + ProdCfg::XCH_OFF + (r & 1) * ProdCfg::XCH_BYTES) + lane;
        half8* mine = xw + wave * 4 * 64;
        mine[0] = oh[0]; mine[64] = oh[1]; mine[128] = ol[0]; mine[192] = ol[1];
        block_barrier();
        const half8* theirs = xw + (wave ^ 1) * 4 * 64;
        xh[0] = oh[0]; xh[1] = oh[1]; xl[0] = ol[0]; xl[1] = ol[1];
        xh[2] = theirs[0]; xh[3] = theirs[64]; xl[2] = theirs[128]; xl[3] = theirs[192];
      };
      {
        conv0(0);
        half8 oh[2], ol[2];
        to_frag(am, ac, oh, ol);
        exchange(0, oh, ol);
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r == 2) PL_T(1);
        if (r + 1 < NR) conv0(r + 1);
        if (r == 2) PL_T(2);
        // the 1x1 on this wave's slab: bias from LDS into the main accumulators
        f32x16 tm, tc;
        {
          const float* bp = pb + ct * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
            tm[4 * g + 0] = b4.x; tm[4 * g + 1] = b4.y; tm[4 * g + 2] = b4.z; tm[4 * g + 3] = b4.w;
          }
        }
        tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(pw2h[0], xh[0], tm, 0, 0, 0);
        tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pw2h[0], xl[0], zero16, 0, 0, 0);
        tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pw2l[0], xh[0], tc, 0, 0, 0);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          tm = __builtin_amdgcn_mfma_f32_32x32x16_f16(pw2h[q], xh[q], tm, 0, 0, 0);
          tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pw2h[q], xl[q], tc, 0, 0, 0);
          tc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pw2l[q], xh[q], tc, 0, 0, 0);
        }
        half8 oh[2], ol[2];
        if (r + 1 < NR) {
          to_frag(am, ac, oh, ol);
          // a wave's MFMAs issue in order and each waits for the pipe: VALU work overlaps them only from BETWEEN them
          __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (r == 2) PL_T(14);
        // -> ReLU (zero outside the mid tensor: v_med3(y, 0, +inf | 0)) -> the consumer's input tile
        const int p = (2 * r + pg) * 32 + pix;
        const int pc = p < ProdCfg::NPX ? p : ProdCfg::NPX - 1;
        const int my = pc / ProdCfg::MW, mx = pc - my * ProdCfg::MW;
        const int gy = gy0 + my, gx = gx0 + mx;
        const float vmax = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? __builtin_inff() : 0.f;
        // (lanes past the tile's last pixel write the slot no tap reads -- column IWs - 1 -- instead of branching)
        const int rem = p < ProdCfg::NPX ? (mx & 1) * C::IWh + (mx >> 1) : C::IWs - 1;
        const int fk = (rem / C::PPR) % C::CPP;
        char* dst = smem + (my * C::IWs + rem) * C::PIXB + 8 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = __builtin_amdgcn_fmed3f(comb(tm[4 * g + e], tc[4 * g + e]), 0.f, vmax);
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = ((ct * 4 + g) ^ fk) * 16;
          *reinterpret_cast<uint2*>(dst + o) = vh;
          *reinterpret_cast<uint2*>(dst + C::IN_BYTES + o) = vl;
        }
#ifdef LFD_PL_TIMING
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (r == 2) PL_T(15);
        if (r + 1 < NR) exchange(r + 1, oh, ol);
      }
    }
  };

  float gn_a[GNIN ? 8 : 1], gn_b[GNIN ? 8 : 1];
  int gnin_n = -1;
  (void)gn_a; (void)gn_b; (void)gnin_n;
  int n_out32 = 0;          // OUTM 2: output store instructions per 32-pixel tile of this wave
  if constexpr (OUTM == 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) n_out32 += (co_base + 8 * (i >> 2) + (i & 3) < a.f_c0 + a.f_c1) ? 1 : 0;
  }
  (void)n_out32;
  int t = t_begin;
  int buf = 0;
  bool first = true;
  if (C::NBUF == 2 && t < t_end) issue_dma(t, 0);
  if constexpr (PROD) {
    if (P.dma_ok && t < t_end) frame_dma(t, 0);
  }
  for (; t < t_end; t += t_step, buf ^= (C::NBUF - 1), ++dbg_it) {
    bool has_next = false;
    PL_T(0);
    if (C::NBUF == 2) {
      // the VMEM operations issued after tile t's DMA that may still be in flight are the previous tile's copy-out stores:
      // every lane issues exactly NST of them (out-of-image lanes into the trash line), vmcnt retires in order
      constexpr int NST = (OUTM == 2) ? 0 : 2 * ((C::OPX * NCT * 4) / 256);
      static_assert(NST == 0 || NST == 4 || NST == 8, "copy-out stores per thread");
      if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (OUTM == 2) {
        // fp32 outputs: one store instruction per accumulator register that holds an output channel in either lane half
        // (n_out32 of them per 32-pixel tile, fixed for the launch) -- waiting for vmcnt(0) here exposed the full write
        // latency of the previous tile's outputs on every tile (the all-levels output conv: 94 us for 184 MB)
        switch (n_out32 * PT) {
          case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
          case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
          case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
          case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
          case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      }
      else if constexpr (NST == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      first = false;
      PL_T(1);
      block_barrier();
      PL_T(2);
      has_next = t + t_step < t_end;
      if (has_next) {
        if constexpr (DMA_IN_LOOP) dma_setup(t + t_step, buf ^ 1);
        else issue_dma(t + t_step, buf ^ 1);
      }
    } else if constexpr (PROD) {
      // the frame patch of tile t (DMA issued one tile ago, before that tile's copy-out stores: vmcnt retires in order)
      constexpr int NST = 2 * ((C::OPX * NCT * 4) / 256);
      static_assert(NST == 4, "copy-out stores per thread");
      const int fb = dbg_it & 1;
      if (P.dma_ok) {
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
      }
      block_barrier();        // patch landed; the previous tile's staging reads of the input-tile bytes are done
      PL_T(10);
      if (P.dma_ok) {
        if (t + t_step < t_end) frame_dma(t + t_step, fb ^ 1);
      } else {
        frame_fill(t);
        block_barrier();
      }
      PL_T(11);
      produce(t, P.dma_ok ? fb : 0);
      block_barrier();
      PL_T(12);
    } else {
      block_barrier();
      issue_dma(t, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      block_barrier();
    }

    if constexpr (ML) {
      const int l = level_of(t);
      if (l != cur_l) {
        // the walk enters another level: flush the GroupNorm sums of the one it leaves, switch the per-level fields, and --
        // the neck convs -- reload the stationary filter and bias of the new level
        if constexpr (OUTM == 1) {
          gn_flush();
          gn_n = -1;
        }
        const PlLevel& lv = L.lv[l];
        cur_l = l;
        a.out = lv.out; a.out_plane = lv.out_plane; a.H = a.OH = lv.H; a.W = a.OW = lv.W;
        a.tiles_x = lv.tiles_x; a.tiles_y = lv.tiles_y; tiles_per_img = lv.tiles_x * lv.tiles_y; tile0 = lv.tile_start;
        a.gn_acc = lv.gn_acc; a.gnin_acc = lv.gnin_acc; a.f_out0 = lv.f_out0; a.f_out1 = lv.f_out1; a.scale1 = lv.scale1;
        a.gnin_gamma = lv.gnin_gamma; a.gnin_beta = lv.gnin_beta;
        gnin_n = -1;
        if (lv.w != a.w) {
          a.w = lv.w; a.bias = lv.bias; a.w2 = lv.w2; a.bias2 = lv.bias2;
          const half8* ws = a.w + ((size_t)(cog * NCT + ct) * C::NK) * 64 + lane;
#pragma unroll
          for (int k = 0; k < NKR; ++k) {
            wh[k] = ws[(size_t)k * 64];
            wl[k] = ws[a.w_plane + (size_t)k * 64];
          }
          if constexpr (TAIL) {
#pragma unroll
            for (int k = 0; k < C::NK2; ++k) {
              w2h[k] = a.w2[((size_t)ct * C::NK2 + k) * 64 + lane];
              w2l[k] = a.w2[a.w2_plane + ((size_t)ct * C::NK2 + k) * 64 + lane];
            }
          }
          if (threadIdx.x < NCT * 32) {
            sbias[threadIdx.x] = a.bias[cog * NCT * 32 + threadIdx.x];
            if constexpr (TAIL) sbias[256 + threadIdx.x] = a.bias2[threadIdx.x];
          }
          block_barrier();
        }
      }
    }
    const int n = (t - tile0) / tiles_per_img;
    const int tr = (t - tile0) - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const char* xb = smem + buf * 2 * C::IN_BYTES;
    if constexpr (GNIN) {
      // every thread owns one 16-byte chunk position = one GroupNorm group (8 channels) of the pixels it visits
      const int gc = (int)threadIdx.x & 15;
      if (n != gnin_n) {
        gnin_n = n;
        long long s = 0, q = 0;
#pragma unroll
        for (int r = 0; r < kGnRep; ++r) {       // (integer adds: order-independent, the statistics stay bit-reproducible)
          s += (long long)a.gnin_acc[(((size_t)r * a.N + n) * 16 + gc) * 2];
          q += (long long)a.gnin_acc[(((size_t)r * a.N + n) * 16 + gc) * 2 + 1];
        }
        const double cnt = (double)a.H * (double)a.W * 8.0;
        const double m = (double)s / kGnFix / cnt;
        double var = (double)q / kGnFix / cnt - m * m;
        var = var > 0. ? var : 0.;
        const float rstd = (float)(1. / sqrt(var + (double)a.gnin_eps)), mean = (float)m;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          gn_a[e] = rstd * a.gnin_gamma[gc * 8 + e];
          gn_b[e] = a.gnin_beta[gc * 8 + e] - mean * gn_a[e];
        }
      }
      char* xw = smem + buf * 2 * C::IN_BYTES;
      static_assert(!GNIN || (C::NSLOT * 16) % 256 == 0, "whole transform rounds");
#pragma unroll
      for (int r = 0; r < (C::NSLOT * 16) / 256; ++r) {
        const int slot = ((int)threadIdx.x >> 4) + 16 * r;
        const int o = slot * 256 + ((gc ^ (slot & 15)) * 16);
        const lfd_f16x8 hh = *reinterpret_cast<const lfd_f16x8*>(xw + o), ll = *reinterpret_cast<const lfd_f16x8*>(xw + C::IN_BYTES + o);
        union { uint32_t u[4]; uint4 v; } oh, ol;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = fmaxf(fmaf(join1(hh[2 * j], ll[2 * j]), gn_a[2 * j], gn_b[2 * j]), 0.f);
          const float y1 = fmaxf(fmaf(join1(hh[2 * j + 1], ll[2 * j + 1]), gn_a[2 * j + 1], gn_b[2 * j + 1]), 0.f);
          split2(y0, y1, oh.u[j], ol.u[j]);
        }
        *reinterpret_cast<uint4*>(xw + o) = oh.v;
        *reinterpret_cast<uint4*>(xw + C::IN_BYTES + o) = ol.v;
      }
      block_barrier();
    }
    if constexpr (OUTM == 1) {
      if (n != gn_n) {
        gn_flush();
        gn_n = n;
      }
    }

    issue_res(n, ty0, tx0);
    PL_T(3);

    f32x16 accm[PT], accc[PT];
    f32x16 adm[DS ? PT : 1], adc[DS ? PT : 1];
    {
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          accm[pt][4 * g + 0] = b4.x; accm[pt][4 * g + 1] = b4.y; accm[pt][4 * g + 2] = b4.z; accm[pt][4 * g + 3] = b4.w;
          accc[pt][4 * g + 0] = 0.f; accc[pt][4 * g + 1] = 0.f; accc[pt][4 * g + 2] = 0.f; accc[pt][4 * g + 3] = 0.f;
        }
      }
      if constexpr (DS) {
        const float* bd = sbias + 128 + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bd + 8 * g);
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            adm[pt][4 * g + 0] = b4.x; adm[pt][4 * g + 1] = b4.y; adm[pt][4 * g + 2] = b4.z; adm[pt][4 * g + 3] = b4.w;
            adc[pt][4 * g + 0] = 0.f; adc[pt][4 * g + 1] = 0.f; adc[pt][4 * g + 2] = 0.f; adc[pt][4 * g + 3] = 0.f;
          }
        }
      }
    }

    // ---- contraction: 3 MFMAs per (k-step, pixel tile); activation fragments PD k-steps ahead in a register ring
    if constexpr (WREG) {
      auto xaddr = [&](int k, int pt) {
        const int r = k / (KS * C::NQ), s = (k / C::NQ) % KS, q = k % C::NQ;
        const int off = XTAB ? xoff[s][XTAB ? q : 0] : (xbase[s] + (((2 * q) ^ xkey[s]) << 4));
        return xb + off + (r + pt * C::RPT * S) * C::IWs * C::PIXB;
      };
      constexpr int PD = ((RES && C::NK > 20) || KS == 1) ? 2 : 3;    // (the residual prefetch takes 32 registers of the 3x3 64-channel kernel's 512)
      half8 xqh[PD + 1][PT], xql[PD + 1][PT];
#pragma unroll
      for (int k = 0; k < PD && k < C::NK; ++k) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const char* p = xaddr(k, pt);
          xqh[k][pt] = *reinterpret_cast<const half8*>(p);
          xql[k][pt] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
        }
      }
      static_for([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (DMA_IN_LOOP) {
          if (has_next) {
            static_for([&](auto pc) {
              constexpr int p = decltype(pc)::value;
              if constexpr ((p * SPREAD) / NPIECE == k) dma_piece(p);
            }, std::make_integer_sequence<int, NPIECE>{});
          }
        }
        if constexpr (k + PD < C::NK) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const char* p = xaddr(k + PD, pt);
            xqh[(k + PD) % (PD + 1)][pt] = *reinterpret_cast<const half8*>(p);
            xql[(k + PD) % (PD + 1)][pt] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
          }
        }
        PL_SB();
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) accm[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)][pt], accm[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) accc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)][pt], accc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) accc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)][pt], accc[pt], 0, 0, 0);
        if constexpr (DS) {
          if constexpr (k / C::NQ == 4) {     // centre tap = the 1x1 stride-2 branch's input pixel (lfd_resnet.py:458-468)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
              adm[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdh[k % C::NQ], xqh[k % (PD + 1)][pt], adm[pt], 0, 0, 0);
              adc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdh[k % C::NQ], xql[k % (PD + 1)][pt], adc[pt], 0, 0, 0);
              adc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdl[k % C::NQ], xqh[k % (PD + 1)][pt], adc[pt], 0, 0, 0);
            }
          }
        }
        PL_SB();
      }, std::make_integer_sequence<int, C::NK>{});
    } else {
      // weights streamed from L2 (128-channel 3x3 layers on the small last-stage maps): a ring of PW (hi, lo) fragment pairs
      constexpr int RK = KS * C::NQ;
      constexpr int PW = (RK % 12 == 0) ? 12 : 8;
      static_assert(RK % PW == 0, "weight ring must wrap on a tap-row boundary");
      half8 wqh[PW], wql[PW];
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        wqh[i] = wsrc[(size_t)i * 64];
        wql[i] = wsrc[a.w_plane + (size_t)i * 64];
      }
      constexpr int XPD = RK > 3 ? 3 : RK - 1;
      half8 xqh[XPD + 1][PT], xql[XPD + 1][PT];
#pragma unroll 1
      for (int r = 0; r < KS; ++r) {
        const char* xr = xb + r * C::IWs * C::PIXB;
        const int knext = r * RK + PW;
        auto xaddr = [&](int j, int pt) {
          const int s = j / C::NQ, q = j % C::NQ;
          const int off = XTAB ? xoff[s][XTAB ? q : 0] : (xbase[s] + (((2 * q) ^ xkey[s]) << 4));
          return xr + off + (pt * C::RPT * S) * C::IWs * C::PIXB;
        };
#pragma unroll
        for (int j = 0; j < XPD; ++j)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const char* p = xaddr(j, pt);
            xqh[j][pt] = *reinterpret_cast<const half8*>(p);
            xql[j][pt] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
          }
#pragma unroll
        for (int j = 0; j < RK; ++j) {
          const int s = j / C::NQ, q = j % C::NQ;
          (void)s; (void)q;
          const half8 wfh = wqh[j % PW], wfl = wql[j % PW];
          PL_SB();
          {
            const size_t kn = (size_t)((knext + j) < C::NK ? (knext + j) : (C::NK - 1)) * 64;
            wqh[j % PW] = wsrc[kn];
            wql[j % PW] = wsrc[a.w_plane + kn];
          }
          if (j + XPD < RK) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
              const char* p = xaddr(j + XPD, pt);
              xqh[(j + XPD) % (XPD + 1)][pt] = *reinterpret_cast<const half8*>(p);
              xql[(j + XPD) % (XPD + 1)][pt] = *reinterpret_cast<const half8*>(p + C::IN_BYTES);
            }
          }
          PL_SB();
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const half8 xh = xqh[j % (XPD + 1)][pt], xl = xql[j % (XPD + 1)][pt];
            accm[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh, xh, accm[pt], 0, 0, 0);
            accc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh, xl, accc[pt], 0, 0, 0);
            accc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfl, xh, accc[pt], 0, 0, 0);
            if constexpr (DS) {
              if (r == 1 && s == 1) {
                adm[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdh[q], xh, adm[pt], 0, 0, 0);
                adc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdh[q], xl, adc[pt], 0, 0, 0);
                adc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdl[q], xh, adc[pt], 0, 0, 0);
              }
            }
          }
        }
      }
    }

    PL_T(4);
    char* scr = C::ALIAS ? (smem + buf * 2 * C::IN_BYTES) : (smem + C::SCR_OFF);
    if constexpr (TAIL) {
      // main conv's epilogue (bias in accm, ReLU) -> hi / lo operand planes of the chained 1x1 in LDS
      char* mid = scr;
      if constexpr (C::ALIAS) block_barrier();      // every wave is done reading the input tile the scratch overlays
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int pb = (pg * PT + pt) * 32 + pix;
        const int fm = (pb / C::MPPR) % C::MCPP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = comb(accm[pt][4 * g + e], accc[pt][4 * g + e]);
            if (a.relu) y[e] = fmaxf(y[e], 0.f);
          }
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = pb * C::MPIXB + (((ct * 4 + g) ^ fm) * 16) + 8 * h;
          *reinterpret_cast<uint2*>(mid + o) = vh;
          *reinterpret_cast<uint2*>(mid + C::MID_PLANE + o) = vl;
        }
      }
      block_barrier();
      {
        const float* bp = sbias + 256 + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            accm[pt][4 * g + 0] = b4.x; accm[pt][4 * g + 1] = b4.y; accm[pt][4 * g + 2] = b4.z; accm[pt][4 * g + 3] = b4.w;
            accc[pt][4 * g + 0] = 0.f; accc[pt][4 * g + 1] = 0.f; accc[pt][4 * g + 2] = 0.f; accc[pt][4 * g + 3] = 0.f;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < C::NK2; ++q) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const int pb = (pg * PT + pt) * 32 + pix;
          const int fm = (pb / C::MPPR) % C::MCPP;
          const int o = pb * C::MPIXB + (((2 * q + h) ^ fm) * 16);
          const half8 xh = *reinterpret_cast<const half8*>(mid + o);
          const half8 xl = *reinterpret_cast<const half8*>(mid + C::MID_PLANE + o);
          half8 wq, wql;
          if constexpr (TAILW_REG) {
            wq = w2h[q]; wql = w2l[q];
          } else {
            const half8* tws = reinterpret_cast<const half8*>(smem + C::LDS_BYTES + ProdCfg::W4_OFF) + (ct * C::NK2 + q) * 64 + lane;
            wq = tws[0]; wql = tws[2 * C::NK2 * 64];
          }
          accm[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq, xh, accm[pt], 0, 0, 0);
          accc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq, xl, accc[pt], 0, 0, 0);
          accc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wql, xh, accc[pt], 0, 0, 0);
        }
      }
    }

    if constexpr (OUTM == 2) {
      // fp32 outputs straight from the accumulator layout (lane = pixel, registers = channels 8g + 4h + e of the slab)
      // exactly n_out32 store instructions per 32-pixel tile and wave (the counted wait at the loop top): lanes without an
      // output -- outside the image, or a channel past the last -- write the trash line
      const float sc1 = a.scale1 ? *a.scale1 : 1.f;
      const int ctot = a.f_c0 + a.f_c1;
      float* trash = reinterpret_cast<float*>(const_cast<_Float16*>(a.zeros) + 1024) + (threadIdx.x & 127);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int oy = ty0 * C::TH + (pg * PT + pt) * C::RPT + oyl;
        const int ox = tx0 * C::TW + oxl;
        const bool in_img = oy < a.OH && ox < a.OW;
        const size_t pixi = in_img ? (size_t)oy * a.OW + ox : 0;
        float* p0 = a.f_out0 + (size_t)n * a.f_img0 + pixi * a.f_c0;
        float* p1 = a.f_out1 + (size_t)n * a.f_img1 + pixi * a.f_c1 - a.f_c0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (co_base + 8 * g + e < ctot) {          // wave-uniform: the lower lane half's channel of this register
              const int co = co_base + 8 * g + 4 * h + e;
              const float y = comb(accm[pt][4 * g + e], accc[pt][4 * g + e]);
              const bool first_out = co < a.f_c0;
              float* dst = (in_img && co < ctot) ? (first_out ? p0 : p1) + co : trash;
              *dst = first_out ? y : y * sc1;
            }
          }
      }
      continue;
    }

    // ---- epilogue: y = main + 2^-11 corr (+ residual) -> ReLU -> (hi, lo) -> LDS staging planes -> full-line 16-byte stores
    const int cout_out = TAIL ? a.cout2 : a.cout;
    const bool relu_out = TAIL ? a.relu2 : a.relu;
    constexpr int OCPP = NCT * 4;
    constexpr int OPIXB = NCT * 64;
    constexpr int OPPR = (OCPP >= 16) ? 1 : 16 / OCPP;
    constexpr int OPX = C::OPX;
    char* sout = scr;
    // staging overlays the consumed input tile (ALIAS) or the chained 1x1's operand planes (TAIL): every wave must be done
    // reading them (separate scratch without TAIL: the loop-top barrier did that)
    PL_T(5);
    if constexpr (C::ALIAS || TAIL) block_barrier();
    PL_T(6);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int pb = (pg * PT + pt) * 32 + pix;
      const int fo = (pb / OPPR) % OCPP;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = comb(accm[pt][4 * g + e], accc[pt][4 * g + e]);
          if (relu_out && !RES) y[e] = fmaxf(y[e], 0.f);       // RES: the staged value is the pre-activation conv + bias
        }
        uint2 vh, vl;
        split2(y[0], y[1], vh.x, vl.x);
        split2(y[2], y[3], vh.y, vl.y);
        const int o = pb * OPIXB + (((ct * 4 + g) ^ fo) * 16) + 8 * h;
        *reinterpret_cast<uint2*>(sout + o) = vh;
        *reinterpret_cast<uint2*>(sout + C::OUT_PLANE + o) = vl;
      }
    }
    PL_T(7);
    if constexpr (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's residual DMAs landed
    block_barrier();
    PL_T(8);
    {
      const int cslice = TAIL ? 0 : cog * NCT * 32;
      float tile_s = 0.f, tile_q = 0.f;      // OUTM 1: this thread's sums over the tile (<= 64 values), fp64 across tiles
      static_assert((OPX * OCPP) % 256 == 0, "whole copy-out rounds");
#pragma unroll
      for (int k = 0; k < (OPX * OCPP) / 256; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int pb = i / OCPP, c = i - pb * OCPP;
        const int t32 = pb >> 5, p32 = pb & 31;
        const int oy = ty0 * C::TH + t32 * C::RPT + p32 / C::TW;
        const int ox = tx0 * C::TW + p32 % C::TW;
        const int fo = (pb / OPPR) % OCPP;
        uint4 vh = *reinterpret_cast<const uint4*>(sout + pb * OPIXB + ((c ^ fo) * 16));
        uint4 vl = *reinterpret_cast<const uint4*>(sout + C::OUT_PLANE + pb * OPIXB + ((c ^ fo) * 16));
        if constexpr (RES) {
          // y = (conv + bias) + identity -> ReLU -> planes (lfd_resnet.py:151-152), on the 8 channels of this line chunk
          const uint4 rh = *reinterpret_cast<const uint4*>(smem + C::RES_OFF + i * 16);
          const uint4 rl = *reinterpret_cast<const uint4*>(smem + C::RES_OFF + C::RES_PLANE + i * 16);
          const lfd_f16x8 sh = __builtin_bit_cast(lfd_f16x8, vh), sl = __builtin_bit_cast(lfd_f16x8, vl);
          const lfd_f16x8 qh = __builtin_bit_cast(lfd_f16x8, rh), ql = __builtin_bit_cast(lfd_f16x8, rl);
          union { uint32_t u[4]; uint4 v; } oh, ol;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float y0 = join1(sh[2 * j], sl[2 * j]) + join1(qh[2 * j], ql[2 * j]);
            float y1 = join1(sh[2 * j + 1], sl[2 * j + 1]) + join1(qh[2 * j + 1], ql[2 * j + 1]);
            if (relu_out) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            split2(y0, y1, oh.u[j], ol.u[j]);
          }
          vh = oh.v; vl = ol.v;
        }
        const bool ok = oy < a.OH && ox < a.OW;
        // exactly two stores per lane and iteration (the counted wait at the loop top): out-of-image lanes write the trash line
        _Float16* dst = a.out + (((size_t)n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * cout_out + cslice + c * 8;
        _Float16* trash = const_cast<_Float16*>(a.zeros) + 1024 + (threadIdx.x & 127) * 8;
        *reinterpret_cast<uint4*>(ok ? dst : trash) = vh;
        *reinterpret_cast<uint4*>(ok ? dst + a.out_plane : trash) = vl;
        if constexpr (OUTM == 1) {
          if (ok) {
            const lfd_f16x8 hh = __builtin_bit_cast(lfd_f16x8, vh), ll = __builtin_bit_cast(lfd_f16x8, vl);
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = join1(hh[e], ll[e]);
              s += v;
              q += v * v;
            }
            tile_s += s;
            tile_q += q;
          }
        }
      }
      if constexpr (OUTM == 1) {
        gn_s += (double)tile_s;
        gn_q += (double)tile_q;
      }
    }
    PL_T(9);
    if constexpr (DS) {
      // second output: the identity branch (bias in adm, no ReLU) through the same staging planes
      block_barrier();
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int pb = (pg * PT + pt) * 32 + pix;
        const int fo = (pb / OPPR) % OCPP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = comb(adm[pt][4 * g + e], adc[pt][4 * g + e]);
          uint2 vh, vl;
          split2(y[0], y[1], vh.x, vl.x);
          split2(y[2], y[3], vh.y, vl.y);
          const int o = pb * OPIXB + (((ct * 4 + g) ^ fo) * 16) + 8 * h;
          *reinterpret_cast<uint2*>(sout + o) = vh;
          *reinterpret_cast<uint2*>(sout + C::OUT_PLANE + o) = vl;
        }
      }
      block_barrier();
      const int cslice = cog * NCT * 32;
      for (int i = threadIdx.x; i < OPX * OCPP; i += 256) {
        const int pb = i / OCPP, c = i - pb * OCPP;
        const int t32 = pb >> 5, p32 = pb & 31;
        const int oy = ty0 * C::TH + t32 * C::RPT + p32 / C::TW;
        const int ox = tx0 * C::TW + p32 % C::TW;
        if (oy < a.OH && ox < a.OW) {
          const int fo = (pb / OPPR) % OCPP;
          const uint4 vh = *reinterpret_cast<const uint4*>(sout + pb * OPIXB + ((c ^ fo) * 16));
          const uint4 vl = *reinterpret_cast<const uint4*>(sout + C::OUT_PLANE + pb * OPIXB + ((c ^ fo) * 16));
          _Float16* dst = a.out_ds + (((size_t)n * a.OH + oy) * a.OW + ox) * a.cout + cslice + c * 8;
          *reinterpret_cast<uint4*>(dst) = vh;
          *reinterpret_cast<uint4*>(dst + a.ds_plane) = vl;
        }
      }
    }
  }
  gn_flush();
}

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, int OUTM, int PTO, bool GNIN = false>
struct PlHeavy {
  // registers of the stationary hi + lo fragments (main slab, chained 1x1, identity branch); beyond what two waves per SIMD
  // can hold next to 64 accumulator and 64 ring registers: one workgroup per CU, up to 512 registers per wave
  static constexpr int wregs = (WREG ? (KS * KS * CIN / 16) * 8 : 96) + (TAIL ? NCT * 2 * 8 : 0) + (DS ? (CIN / 16) * 8 : 0);
  static constexpr bool value = !WREG || wregs > 96;
};

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, int OUTM, int PTO, bool GNIN = false>
__global__ __launch_bounds__(256, (PlHeavy<CIN, KS, S, NCT, WREG, TAIL, RES, DS, OUTM, PTO>::value ? 1 : 2)) void k_pl_conv(PlArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PlLevels none;
  none.n = 0;
  pl_block<CIN, KS, S, NCT, WREG, TAIL, RES, DS, OUTM, PTO, GNIN, false>(a, none, smem, PlProd{});
}

// the same block over the tiles of several pyramid levels (1x1 convs of the neck / head)
template <int CIN, int NCT, bool TAIL, int OUTM, int PTO, bool GNIN>
__global__ __launch_bounds__(256, (PlHeavy<CIN, 1, 1, NCT, true, TAIL, false, false, OUTM, PTO>::value ? 1 : 2)) void k_pl_conv_ml(PlArgs a, PlLevels L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  pl_block<CIN, 1, 1, NCT, true, TAIL, false, false, OUTM, PTO, GNIN, true>(a, L, smem, PlProd{});
}

template <int CIN, int NCT, bool TAIL, int OUTM, int PTO, bool GNIN>
int launch_pl_ml_(const PlArgs& a0, const PlLevels& L0, hipStream_t st) {
  using C = PCfg<CIN, 1, 1, NCT, TAIL, false, PTO>;
  PlArgs a = a0;
  PlLevels L = L0;
  long nt = 0;
  for (int i = 0; i < L.n; ++i) {
    L.lv[i].tiles_x = (L.lv[i].W + C::TW - 1) / C::TW;
    L.lv[i].tiles_y = (L.lv[i].H + C::TH - 1) / C::TH;
    L.lv[i].tile_start = (int)nt;
    nt += (long)a.N * L.lv[i].tiles_x * L.lv[i].tiles_y;
    if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  }
  a.ntiles = (int)nt;
  // level 0's fields as the initial state (the kernel switches at its first tile: cur_l = -1)
  a.in = L.lv[0].in; a.in_plane = L.lv[0].in_plane; a.out = L.lv[0].out; a.out_plane = L.lv[0].out_plane;
  a.H = a.OH = L.lv[0].H; a.W = a.OW = L.lv[0].W; a.tiles_x = L.lv[0].tiles_x; a.tiles_y = L.lv[0].tiles_y;
  a.w = L.lv[0].w; a.bias = L.lv[0].bias; a.w2 = L.lv[0].w2; a.bias2 = L.lv[0].bias2;
  a.gn_acc = L.lv[0].gn_acc; a.gnin_acc = L.lv[0].gnin_acc; a.gnin_gamma = L.lv[0].gnin_gamma; a.gnin_beta = L.lv[0].gnin_beta;
  a.f_out0 = L.lv[0].f_out0; a.f_out1 = L.lv[0].f_out1; a.scale1 = L.lv[0].scale1;
  const int cgroups = TAIL ? 1 : (a.cout + NCT * 32 - 1) / (NCT * 32);
  constexpr int LDSB = C::LDS_BYTES;
  auto kern = k_pl_conv_ml<CIN, NCT, TAIL, OUTM, PTO, GNIN>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  constexpr bool heavy = PlHeavy<CIN, 1, 1, NCT, true, TAIL, false, false, OUTM, PTO>::value;
  constexpr int per_cu = (heavy || LDSB > 80 * 1024) ? 1 : 2;
  int blocks = (256 * per_cu) / cgroups;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(blocks, cgroups), dim3(256), LDSB, st, a, L);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, int OUTM, int PTO, bool GNIN = false>
int launch_pl_(const PlArgs& a0, hipStream_t st) {
  using C = PCfg<CIN, KS, S, NCT, TAIL, RES, PTO>;
  PlArgs a = a0;
  a.tiles_x = (a.OW + C::TW - 1) / C::TW;
  a.tiles_y = (a.OH + C::TH - 1) / C::TH;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  const int cgroups = TAIL ? 1 : (a.cout + NCT * 32 - 1) / (NCT * 32);
  constexpr int LDSB = C::LDS_BYTES;
  auto kern = k_pl_conv<CIN, KS, S, NCT, WREG, TAIL, RES, DS, OUTM, PTO, GNIN>;
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  constexpr bool heavy = PlHeavy<CIN, KS, S, NCT, WREG, TAIL, RES, DS, OUTM, PTO>::value;
  constexpr int per_cu = (heavy || LDSB > 80 * 1024) ? 1 : 2;
  int blocks = (256 * per_cu) / cgroups;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(kern, dim3(blocks, cgroups), dim3(256), LDSB, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}


}  // namespace pl
