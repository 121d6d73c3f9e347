// csrc/conv_impl.h -- (device templates shared by conv.hip and conv_acc32.hip)
// NHWC fp16 implicit-GEMM convolution (1x1 / 3x3, stride 1 / 2) on gfx950 MFMA.
//
// Replaces the cuDNN-backed nn.Conv2d + nn.BatchNorm2d + ReLU (+ residual add) stacks of the
// reference backbone (lfd/model/backbone/lfd_resnet.py:96-154 FasterBlock, :354-439 stem,
// :458-468 downsample) for inference: BN is folded into the weights/bias on the host, the
// epilogue fuses bias + residual + ReLU, and an optional chained 1x1 conv ("tail") consumes
// the tile through LDS without a trip to HBM (stem 3x3 -> 1x1 pairs).
//
// Design (MI355X-first, not a port of a warp-tiled CUDA kernel):
//   * weights-stationary: each wave64 keeps the whole [32 cout x K] filter slab of its
//     cout tile in VGPRs (K = 9*64 -> 36 fragments = 144 VGPRs) for the lifetime of a
//     persistent workgroup; only the activation operand streams through LDS;
//   * v_mfma_f32_32x32x16_f16 with the roles swapped (A = weights, B = pixels) so that the
//     accumulator layout is lane = pixel, registers = 4-channel groups -> 8-byte NHWC stores;
//   * the input halo tile is fetched by direct global->LDS DMA (global_load_lds_dwordx4,
//     no VGPR staging), double-buffered so tile t+1 lands while tile t is on the MFMA pipe;
//     LDS image is lane-linear, the XOR bank swizzle is applied on the SOURCE address and on
//     the ds_read_b128 side (guide rule 21); out-of-image pixels read a zero line;
//   * workgroups are persistent (grid = 2 x 256 CUs) and walk XCD-contiguous tile ranges so
//     neighbouring halo re-reads hit the same XCD's L2.
#pragma once
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvArgs {
  const _Float16* in;     // [N,H,W,CIN]
  _Float16* out;          // [N,OH,OW,COUT_OUT]
  float* out32;           // ACC32 kernels: raw fp32 accumulators (conv + bias) [N,OH,OW,COUT], instead of `out`
  const half8* w;         // packed main weights [cout_tile][NK][64 lanes] x 8 halfs
  const float* bias;      // [COUT] (BN folded)
  const _Float16* res;    // optional residual [N,OH,OW,COUT_OUT] (added before ReLU)
  int res_px;             // halfs between residual pixels (cout; 0 = every pixel reads the same line, i.e. "no residual"
                          // for the layers of a chained launch that have none: res then points at the zero line)
  const half8* w2;        // tail 1x1 packed weights [cout2_tile][CMID/16][64]
  const float* bias2;     // [COUT2]
  const half8* wds;       // DS: packed 1x1 stride-2 downsample weights [cout/32][CIN/16][64]
  const float* bds;       // DS: its bias [cout]
  _Float16* out_ds;       // DS: identity-branch output [N,OH,OW,cout] (no ReLU)
  const _Float16* zeros;  // 4 KB line: bytes [0,2048) stay zero (source of out-of-image pixels),
                          // bytes [2048,4096) are a write-only trash area for masked stores
  int N, H, W, OH, OW;
  int cout;    // main conv output channels
  int cout2;   // tail output channels (TAIL only)
  int relu, relu2;
  int tiles_x, tiles_y, ntiles;
  float* stat_partials;  // STATS kernels: [gridDim.x][2][cout] per-workgroup sums of the stored outputs and of their squares
  // PRO kernels (1x1 stride 1): `in` is the PRE-normalisation output y of the producing train-mode BatchNorm unit; the
  // activation fragments are normalised + ReLU'd on their way from LDS into the MFMA (the unit's z tensor never exists)
  const float* pro_stats;   // [2][CIN]: batch mean | rstd of the producer (lfd_bn_train_stats_f16 layout)
  const float* pro_gamma;   // [CIN]
  const float* pro_beta;    // [CIN]
  // BSUM kernels (a data-gradient conv whose output dz is the gradient wrt the activation of a train-mode BatchNorm + ReLU
  // unit without residual): the backward sums of that unit -- sum g and sum g * xhat per channel, g = dz * [ReLU passed],
  // k_bn_bwd_partial's arithmetic on the fp16 values stored -- leave through stat_partials like the forward statistics
  const _Float16* bsum_y;   // the unit's pre-normalisation output, same shape as `out`
  const float* bsum_stats;  // [2][cout] mean | rstd
  const float* bsum_gamma;  // [cout]
  const float* bsum_beta;   // [cout]
};

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL>
struct Cfg {
  // k-steps whose weight fragments live in LDS instead of VGPRs (register-pressure relief for
  // the 64-ch 3x3 s1 workhorse: 28 of 36 fragments stay in registers, 2 taps are LDS-resident)
  static constexpr int WL = (WREG && CIN == 64 && KS == 3 && S == 1) ? 8 : 0;
  static constexpr int PT = (S == 1) ? 2 : 1;       // 32-pixel MFMA tiles per wave
  // output tile width.  3x3 s1 64ch: 8 rows x 16 columns (2 x 16 pixels per MFMA tile) -- 135x240 maps needs 2040 tiles =
  // 3.98 rounds of the 512 resident workgroups instead of 2176 = 4.25 -> 5 with 4 x 32, and the halo shrinks 1.59 -> 1.41
  // (same-session A/B against 4 x 32: 27.3-28.3 us vs 31.7-32.0 us per 135x240 launch, 31-32 vs 33-34 with residual)
  static constexpr int TW = (S == 1 && !(CIN == 64 && KS == 3)) ? 32 : 16;
  static constexpr int RPT = 32 / TW;               // output rows per MFMA pixel tile
  static constexpr int PG = 4 / NCT;                // pixel groups (waves along pixels) per block
  static constexpr int TH = PG * PT * RPT;          // output tile height
  static constexpr int PAD = KS / 2;
  static constexpr int IH = (TH - 1) * S + KS;
  static constexpr int IW = (TW - 1) * S + KS;
  static constexpr int IWh = (IW + 1) / 2;
  static constexpr int IWs = (S == 2) ? 2 * IWh : ((IW + 1) & ~1);  // slots per row (even)
  static constexpr int CPP = CIN / 8;               // 16-byte chunks per pixel
  static constexpr int PIXB = CIN * 2;
  static constexpr int PPR = (CPP >= 16) ? 1 : 16 / CPP;  // pixels per 256-B bank row
  static constexpr int NSLOT = IH * IWs;
  static constexpr int OUT_STAGE_BYTES = (4 / NCT) * PT * 32 * NCT * 64;   // epilogue staging tile (fp16)
  static constexpr int IN_RAW = NSLOT * PIXB > OUT_STAGE_BYTES ? NSLOT * PIXB : OUT_STAGE_BYTES;
  static constexpr int IN_BYTES = ((IN_RAW + 1023) / 1024) * 1024;
#ifdef CV_S2_SINGLE
  static constexpr int NBUF = (S == 1) ? 2 : 1;     // S=2 tiles are 4x larger: single buffer, 2 blocks/CU
#else
  // S=2 tiles are 4x larger: single buffer, 2 blocks/CU -- except the 64-channel 3x3 (the HBM-bound first conv of a stage,
  // with or without its downsample branch): its 39 KB tile fits twice and still leaves room for two blocks per CU
  // (2 x 81.4 KB of 160 KB), so the next tile's DMA overlaps this tile's contraction inside the workgroup as well
  static constexpr int NBUF = (S == 1 || (CIN == 64 && KS == 3 && NCT == 2 && !TAIL)) ? 2 : 1;
#endif
  static constexpr int NK = KS * KS * CIN / 16;     // MFMA k-steps of the main conv
  static constexpr int NQ = CIN / 16;
  // tail (1x1 on the main conv's output): CMID = NCT*32 channels
  static constexpr int CMID = NCT * 32;
  static constexpr int MCPP = CMID / 8;
  static constexpr int MPIXB = CMID * 2;
  static constexpr int MPPR = (MCPP >= 16) ? 1 : 16 / MCPP;
  static constexpr int NK2 = CMID / 16;
  static constexpr int MID_BYTES = TAIL ? (PG * PT * 32 * MPIXB) : 0;
  static constexpr int WL_BYTES = WL * NCT * 1024;
  static constexpr int BIAS_OFF = NBUF * IN_BYTES + MID_BYTES + WL_BYTES;   // 3 x 128 floats: bias, ds bias, tail bias
  static constexpr int LDS_BYTES = BIAS_OFF + 3 * 128 * 4;
};

// global -> LDS DMA (16 B per lane; LDS destination = wave-uniform base in M0 + lane * 16).
// Issued through inline asm ON PURPOSE: when the compiler sees the global_load_lds builtin it assumes every
// later LDS read may alias the in-flight DMA and inserts s_waitcnt vmcnt(0) in front of the first one --
// i.e. it waits for the NEXT tile's prefetch before starting this tile's contraction, which silently turns
// the double buffer into a single buffer.  With the DMA opaque to the compiler the hand-placed vmcnt waits
// at the tile boundary are the only synchronisation with it.
__device__ __forceinline__ void dma16(const void* g, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory");
}

// workgroup barrier for LDS hand-offs: LDS operations drained, no fence on global memory (__syncthreads()
// would add s_waitcnt vmcnt(0) and drain the prefetch and the previous tile's stores with it)
__device__ __forceinline__ void block_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifdef LFD_CONV_TIMING
__device__ unsigned long long g_conv_dbg[8 * 16];
#define CV_T(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && dbg_it < 8) g_conv_dbg[dbg_it * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define CV_T(i)
#endif
#ifdef LFD_CONV_TIMING
#define CV_END() do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_conv_dbg[122] = __builtin_readcyclecounter(); g_conv_dbg[123] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define CV_END()
#endif


// STATS (training forward, lfd_resnet.py:96-154 in train mode): the per-channel sums BatchNorm's batch statistics need are
// taken from the fp16 values on their way from the staging tile to memory -- every thread copies the same 16-byte channel
// chunk of every pixel it stores, so 8 sums + 8 sums of squares per thread last the whole persistent tile walk -- and the
// workgroup leaves one row of partials; k_bn_stats_final (train.hip) adds the rows in fp64.  The separate read of the
// whole output (k_bn_stats_partial) is what this replaces.
__device__ __forceinline__ void stats_add(uint4 v, bool ok, float (&s)[8], float (&q)[8]) {
  if (!ok) v = make_uint4(0, 0, 0, 0);
  const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(wd[j] & 0xffffu));
    const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(wd[j] >> 16));
    s[2 * j] += lo;  q[2 * j] += lo * lo;
    s[2 * j + 1] += hi;  q[2 * j + 1] += hi * hi;
  }
}

__device__ __forceinline__ void bsum_add(uint4 v, uint4 yv, bool ok, const float (&m)[8], const float (&r)[8], const float (&ga)[8],
                                         const float (&be)[8], float (&s)[8], float (&q)[8]) {
  const lfd_f16x8 d = __builtin_bit_cast(lfd_f16x8, v), yy = __builtin_bit_cast(lfd_f16x8, yv);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float g = (float)d[e];
    const float xh = ((float)yy[e] - m[e]) * r[e];
    if (!(ga[e] * xh + be[e] > 0.f)) g = 0.f;
    if (!ok) g = 0.f;
    s[e] += g;
    q[e] += g * xh;
  }
}

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, bool ACC32 = false, bool STATS = false, bool PRO = false, bool BSUM = false>
__device__ __forceinline__ void conv_block(const ConvArgs& a, char* smem) {
  static_assert(!PRO || (KS == 1 && S == 1 && WREG && !TAIL && !DS), "PRO: a 1x1 stride-1 conv fed by a train-mode BatchNorm + ReLU unit");
  static_assert(!BSUM || (KS == 1 && S == 1 && !STATS && !TAIL && !DS && !ACC32), "BSUM: the 1x1 data-gradient conv of a stem pair");
  static_assert(!DS || (KS == 3 && S == 2 && !TAIL && !RES), "DS: the residual block's 1x1 s2 downsample rides on its 3x3 s2 conv");
  static_assert(!ACC32 || (!TAIL && !RES && !DS), "ACC32 writes the bare accumulators of ONE conv");
  static_assert(!STATS || (!TAIL && !RES && !DS && !ACC32), "STATS: the bare conv in front of a train-mode BatchNorm");
  float st_s[8], st_q[8];
  if constexpr (STATS || BSUM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) st_s[e] = st_q[e] = 0.f;
  }
  using C = Cfg<CIN, KS, S, NCT, WREG, TAIL>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ct = wave % NCT;         // cout tile of this wave inside the block's cout group
  const int pg = wave / NCT;         // pixel group
  const int h = lane >> 5;           // k half / channel half
  const int pix = lane & 31;
  const int oyl = pix / C::TW, oxl = pix % C::TW;
  const int cog = blockIdx.y;        // cout group (NCT*32 channels each)
#ifdef LFD_CONV_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { g_conv_dbg[120] = __builtin_readcyclecounter(); g_conv_dbg[121] = __builtin_amdgcn_s_memrealtime(); }
#endif
  const int co_base = (cog * NCT + ct) * 32;

  // ---- PRO: the producer's per-channel affine for the 8 channels (16 q + 8 h .. + 7) of each of this lane's fragments,
  //      formed exactly as k_bn_apply forms it (train.hip): a = gamma * rstd, b = beta - mean * a
  constexpr int NQP = PRO ? CIN / 16 : 1;
  float pro_a[NQP][8], pro_b[NQP][8];
  if constexpr (PRO) {
#pragma unroll
    for (int q = 0; q < NQP; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = 16 * q + 8 * h + e;
        pro_a[q][e] = a.pro_gamma[ch] * a.pro_stats[CIN + ch];
        pro_b[q][e] = a.pro_beta[ch] - a.pro_stats[ch] * pro_a[q][e];
      }
  }
  auto pro_apply = [&](half8 v, int q) {
    if constexpr (PRO) {
      return (half8)lfd_affine_relu_f16x8(v, pro_a[q], pro_b[q]);
    } else {
      return v;
    }
  };

  // ---- BSUM: the unit's statistics / affine of the 8 channels this thread copies out (chunk threadIdx.x % (NCT * 4) of every pixel)
  float bs_m[BSUM ? 8 : 1], bs_r[BSUM ? 8 : 1], bs_g[BSUM ? 8 : 1], bs_b[BSUM ? 8 : 1];
  if constexpr (BSUM) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cog * NCT * 32 + ((int)threadIdx.x % (NCT * 4)) * 8 + e;
      bs_m[e] = a.bsum_stats[ch];
      bs_r[e] = a.bsum_stats[a.cout + ch];
      bs_g[e] = a.bsum_gamma[ch];
      bs_b[e] = a.bsum_beta[ch];
    }
  }

  // ---- biases into LDS.  They are re-read for every tile; as global loads the compiler's wait for them
  //      (vmcnt is in-order) would also wait for the just-issued DMA prefetch of the next tile.
  float* sbias = reinterpret_cast<float*>(smem + C::BIAS_OFF);
  if (threadIdx.x < NCT * 32) {
    sbias[threadIdx.x] = a.bias[cog * NCT * 32 + threadIdx.x];
    if constexpr (DS) sbias[128 + threadIdx.x] = a.bds[cog * NCT * 32 + threadIdx.x];
    if constexpr (TAIL) sbias[256 + threadIdx.x] = a.bias2[threadIdx.x];
  }

  const uint32_t lo_mid = a.relu ? LFD_PK_RELU : LFD_PK_NONE;   // activation between the conv and its chained 1x1

  // ---- stationary weights
  constexpr int NKR = WREG ? (C::NK - C::WL) : 1;   // fragments held in VGPRs
  half8 wreg[NKR];
  const half8* wsrc = a.w + ((size_t)(cog * NCT + ct) * C::NK) * 64 + lane;
  half8* wlds = reinterpret_cast<half8*>(smem + C::NBUF * C::IN_BYTES + C::MID_BYTES) + (ct * C::WL) * 64 + lane;
  if (WREG) {
#pragma unroll
    for (int k = 0; k < NKR; ++k) wreg[k] = wsrc[(size_t)k * 64];
#pragma unroll
    for (int k = 0; k < C::WL; ++k) wlds[k * 64] = wsrc[(size_t)(NKR + k) * 64];  // visible after the first barrier
  }
  half8 w2reg[TAIL ? C::NK2 : 1];
  if (TAIL) {
#pragma unroll
    for (int k = 0; k < C::NK2; ++k) w2reg[k] = a.w2[((size_t)ct * C::NK2 + k) * 64 + lane];
  }

  // DS: the identity branch conv1x1 stride 2 (lfd_resnet.py:458-468) reads exactly the centre tap
  // (r = s = 1) of this 3x3 stride-2 conv's window: same LDS fragments, one extra MFMA each.
  half8 wdsr[DS ? C::NQ : 1];
  if constexpr (DS) {
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) wdsr[q] = a.wds[((size_t)(cog * NCT + ct) * C::NQ + q) * 64 + lane];
  }

  // ---- per-lane LDS read offsets: one per (column tap s, 16-channel group q).  For 128 input
  // channels (NQ = 8) the 24-entry table would cost more registers than the weight ring: keep the
  // per-tap pixel base + swizzle key and form the XOR term at the read (2 VALU per ds_read).
  constexpr bool XTAB = C::NQ <= 4;
  int xoff[KS][XTAB ? C::NQ : 1];
  int xbase[KS], xkey[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int ix = oxl * S + s;
    const int rem = (S == 2) ? ((ix & 1) * C::IWh + (ix >> 1)) : ix;
    const int f = (rem / C::PPR) % C::CPP;
    const int rowbase = ((pg * C::PT * C::RPT + oyl) * S) * C::IWs + rem;
    xbase[s] = rowbase * C::PIXB;
    xkey[s] = f ^ h;            // (2q + h) ^ f == (2q) ^ (f ^ h) because bit 0 of 2q is clear
#pragma unroll
    for (int q = 0; q < (XTAB ? C::NQ : 1); ++q) xoff[s][q] = rowbase * C::PIXB + (((2 * q + h) ^ f) * 16);
  }

  // ---- persistent tile walk, XCD-contiguous ranges (block b runs on XCD b % 8)
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;  // blocks living on this XCD
  const int tiles_per_img = a.tiles_x * a.tiles_y;

  // FAST (the 3x3 s1 64-channel workhorse, 8 x 16 tile, 10 x 18 halo): DMA instructions are aligned to halo rows --
  // columns 0..15 of a row are two 64-lane instructions, columns 16..17 one 16-lane instruction -- so that the source
  // address is a SCALAR row base + a per-lane constant and the only per-instruction vector work is the validity
  // select.  (The generic slot walk below costs ~40 dependent VALU instructions per DMA -- integer divide, 64-bit
  // multiply -- which measured 2.6 k cycles per tile, as long as the tile's whole contraction.)
  constexpr bool FAST = (CIN == 64 && KS == 3 && S == 1 && NCT == 2 && !TAIL && !DS);
  const long f_rowpitch = (long)a.W * (CIN * 2);
  auto issue_dma_fast = [&](int t, int buf) {
    int ol = lane;
    asm volatile("" : "+v"(ol));     // opaque: the per-lane constants below are recomputed per tile (5 VALU) instead of
                                     // being hoisted into registers this 256-VGPR kernel does not have
    const int f_lpx = ol >> 3;
    const int f_c0 = ((ol & 7) ^ (f_lpx >> 1)) * 16;                     // chunk offset for columns 0..7 (key = ix >> 1)
    const int f_off0 = f_lpx * 128 + f_c0, f_off1 = 1024 + f_lpx * 128 + (f_c0 ^ 64);   // columns 8..15: key + 4
    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const int gy0 = ty0 * C::TH - 1, gx0 = tx0 * C::TW - 1;
    // address of halo pixel (row 0, column 0) -- outside the image for border tiles, only formed
    const char* p00 = reinterpret_cast<const char*>(a.in) + ((long)n * a.H + gy0) * f_rowpitch + (long)gx0 * (CIN * 2);
    const char* zsrc0 = reinterpret_cast<const char*>(a.zeros) + f_c0;
    const char* zsrc1 = reinterpret_cast<const char*>(a.zeros) + (f_c0 ^ 64);
    const bool xv0 = (gx0 + f_lpx >= 0) && (gx0 + f_lpx < a.W);
    const bool xv1 = (gx0 + 8 + f_lpx < a.W);
    char* lbase = smem + buf * C::IN_BYTES;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int m = wave + 4 * j;                  // 20 main instructions: row m >> 1, column half m & 1
      const int iy = m >> 1, hf = m & 1;
      const int gy = gy0 + iy;
      const bool rv = gy >= 0 && gy < a.H;
      const char* rowp = p00 + iy * f_rowpitch;
      const char* src = hf ? ((rv && xv1) ? rowp + f_off1 : zsrc1) : ((rv && xv0) ? rowp + f_off0 : zsrc0);
      dma16(src, lbase + (iy * C::IWs + 8 * hf) * C::PIXB);
    }
    // columns 16, 17 of row iy: 16 lanes (2 pixels x 8 chunks, key = 0); rows wave, wave + 4, wave + 8
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int iy = wave + 4 * j;
      if (iy < C::IH && ol < 16) {
        const int gy = gy0 + iy, gx = gx0 + 16 + (ol >> 3);
        const bool ok = gy >= 0 && gy < a.H && gx < a.W;
        const char* src = ok ? p00 + iy * f_rowpitch + 2048 + ol * 16 : reinterpret_cast<const char*>(a.zeros) + (ol & 7) * 16;
        dma16(src, lbase + (iy * C::IWs + 16) * C::PIXB);
      }
    }
  };
  // FAST2 (3x3 stride 2, 64 input channels: the first conv of every stage + its 1x1 downsample branch): the same idea for
  // the column-de-interleaved stride-2 tile.  A halo row is IWs = 34 slots (even columns 0..32, then odd columns 1..31):
  // five row-aligned instructions of 8 slots; wave w issues rows w, w+4, ...  The slot -> (column, swizzled chunk) map of
  // the five parts is computed once per tile (30 VALU), each instruction adds a scalar row base and selects the zero line
  // for out-of-image pixels.  The generic slot walk cost 4.5 k cycles of issue per tile against 1.5 k of contraction
  // (integer divisions by 34 and 64-bit multiplies per instruction, tools/timing/conv_s2_phases.py).
  constexpr bool FAST2 = (CIN == 64 && KS == 3 && S == 2 && !TAIL);
  auto issue_dma_s2 = [&](int t, int buf) {
    int ol = lane;
    asm volatile("" : "+v"(ol));     // recomputed per tile, not hoisted (registers)
    const int sl = ol >> 3, cs = ol & 7;
    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const int gy0 = ty0 * C::TH * 2 - 1, gx0 = tx0 * C::TW * 2 - 1;
    const char* p00 = reinterpret_cast<const char*>(a.in) + ((long)n * a.H + gy0) * f_rowpitch + (long)gx0 * (CIN * 2);
    char* lbase = smem + buf * C::IN_BYTES;
    constexpr int NP = (C::IWs + 7) / 8;            // instructions per row (5)
    int off[NP];                                    // byte offset of this lane's chunk inside the row, or -1: not needed / outside
    const char* zsrc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int rem = 8 * j + sl;
      const int ix = rem < C::IWh ? 2 * rem : 2 * rem - (2 * C::IWh - 1);
      const int c = cs ^ ((rem / C::PPR) % C::CPP);
      const int gx = gx0 + ix;
      const bool ok = rem < C::IWs && ix < C::IW && gx >= 0 && gx < a.W;
      off[j] = ok ? ix * (CIN * 2) + c * 16 : -1;
      zsrc[j] = reinterpret_cast<const char*>(a.zeros) + c * 16;
    }
    for (int iy = wave; iy < C::IH; iy += 4) {
      const int gy = gy0 + iy;
      const bool rv = gy >= 0 && gy < a.H;          // scalar
      const char* rowp = p00 + iy * f_rowpitch;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const char* src = (rv && off[j] >= 0) ? rowp + off[j] : zsrc[j];
        if (8 * j + 8 <= C::IWs || (8 * j + sl < C::IWs && 2 * (8 * j + sl) - (2 * C::IWh - 1) < C::IW))
          dma16(src, lbase + (iy * C::IWs + 8 * j) * C::PIXB);
      }
    }
  };
  auto issue_dma = [&](int t, int buf) {
    if constexpr (FAST) { issue_dma_fast(t, buf); return; }
    if constexpr (FAST2) { issue_dma_s2(t, buf); return; }
    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const int gy0 = ty0 * C::TH * S - C::PAD, gx0 = tx0 * C::TW * S - C::PAD;
    constexpr int SPW = 64 / C::CPP;  // pixel slots per wave instruction
    char* lbase = smem + buf * C::IN_BYTES;
    for (int slot0 = wave * SPW; slot0 < C::NSLOT; slot0 += 4 * SPW) {
      const int pslot = slot0 + lane / C::CPP;
      const int cs = lane % C::CPP;
      if (pslot < C::NSLOT) {
        const int iy = pslot / C::IWs;
        const int rem = pslot - iy * C::IWs;
        const int ix = (S == 2) ? ((rem < C::IWh) ? 2 * rem : 2 * (rem - C::IWh) + 1) : rem;
        const int c = cs ^ ((rem / C::PPR) % C::CPP);
        const int gy = gy0 + iy, gx = gx0 + ix;
        bool needed = ix < C::IW;
        if (KS == 1 && S == 2) needed = needed && !(ix & 1) && !(iy & 1);
        if (needed) {
          const bool valid = (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
          const _Float16* src = valid ? a.in + (((size_t)n * a.H + gy) * a.W + gx) * CIN + c * 8
                                      : a.zeros + c * 8;
          dma16(src, lbase + slot0 * C::PIXB);
        }
      }
    }
  };

  int t = t_begin + bix;
  int buf = 0;
  bool first = true;
  if (C::NBUF == 2 && t < t_end) issue_dma(t, 0);
  int dbg_it = 0; (void)dbg_it;
  for (; t < t_end; t += t_step, buf ^= (C::NBUF - 1), ++dbg_it) {
    bool has_next_tile = false; (void)has_next_tile;
    CV_T(0);
    if (C::NBUF == 2) {
      // Tile t's DMA was issued one iteration ago; the only VMEM operations issued after it that can
      // still be in flight are the previous tile's NST copy-out stores (every lane issues exactly NST
      // of them, out-of-image lanes into a trash line).  vmcnt retires in order, so allowing NST
      // outstanding operations waits for the DMA but not for the stores' write latency.
      constexpr int NST = (C::PG * C::PT * 32 * NCT * 4) / 256;
      static_assert(NST == 2 || NST == 4, "copy-out stores per thread");
      // (first tile: no stores behind the DMA yet -> everything must have landed)
      // (ACC32: the accumulator stores below are not counted -> drain everything)
      // (DS: the identity-branch stores that follow the NST copy-out stores are CONDITIONAL -- a wave whose pixels are all
      //  outside the image issues none -- so they must not be counted: with NST allowed, whatever is still in flight is
      //  younger than the DMA in either case)
      if (first || ACC32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (NST == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      first = false;
      CV_T(1);
      block_barrier();  // tile t landed for every wave; everyone is done with buffer buf^1 and `mid`
      CV_T(2);
      has_next_tile = t + t_step < t_end;
      if (has_next_tile) issue_dma(t + t_step, buf ^ 1);
      CV_T(3);
    } else {
      block_barrier();  // everyone is done reading the single buffer / `mid`
      CV_T(1);
      issue_dma(t, 0);
      CV_T(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      block_barrier();
      CV_T(3);
    }

    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const char* xb = smem + buf * C::IN_BYTES;

    // residual (identity branch) of this lane's outputs: requested now, consumed in the epilogue,
    // so its latency hides under the contraction
    // (compile-time RES: a run-time branch around these loads would make the compiler drain vmcnt --
    //  and with it the just-issued DMA of the next tile -- before the contraction starts)
    half4 resv[RES ? C::PT : 1][4];
    if constexpr (RES) {
#pragma unroll
      for (int pt = 0; pt < C::PT; ++pt) {
        const int oy = ty0 * C::TH + (pg * C::PT + pt) * C::RPT + oyl;
        const int ox = tx0 * C::TW + oxl;
        const bool ok = oy < a.OH && ox < a.OW;
        const size_t o = (((size_t)n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.res_px + co_base + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) resv[pt][g] = *reinterpret_cast<const half4*>(a.res + o + 8 * g);
      }
    }

    // BSUM: the unit's y at the pixels / chunk this thread copies out below (same mapping), requested now like the residual
    constexpr int B_OCPP = NCT * 4, B_OPX = C::PG * C::PT * 32, B_NST = (B_OPX * B_OCPP) / 256;
    uint4 bs_y[BSUM ? B_NST : 1];
    if constexpr (BSUM) {
      static_assert(!BSUM || (B_OPX * B_OCPP) % 256 == 0, "whole copy-out rounds");
#pragma unroll
      for (int k = 0; k < B_NST; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int pb = i / B_OCPP, c = i - pb * B_OCPP;
        const int t32 = pb >> 5, p32 = pb & 31;
        const int oy = ty0 * C::TH + t32 * C::RPT + p32 / C::TW;
        const int ox = tx0 * C::TW + p32 % C::TW;
        const bool ok = oy < a.OH && ox < a.OW;
        bs_y[k] = *reinterpret_cast<const uint4*>(a.bsum_y + (((size_t)n * a.OH + (ok ? oy : 0)) * a.OW + (ok ? ox : 0)) * a.cout +
                                                  cog * NCT * 32 + c * 8);
      }
    }

    f32x16 accd[DS ? C::PT : 1];
    if constexpr (DS) {
      const float* bp = sbias + 128 + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < C::PT; ++pt) {
          accd[pt][4 * g + 0] = b4.x; accd[pt][4 * g + 1] = b4.y; accd[pt][4 * g + 2] = b4.z; accd[pt][4 * g + 3] = b4.w;
        }
      }
    }
    f32x16 acc[C::PT];
    {
      // accumulators start at the (BN-folded) bias of this lane's 16 channels
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < C::PT; ++pt) {
          acc[pt][4 * g + 0] = b4.x; acc[pt][4 * g + 1] = b4.y;
          acc[pt][4 * g + 2] = b4.z; acc[pt][4 * g + 3] = b4.w;
        }
      }
    }
    // ---- main contraction.  Flat, fully unrolled k loop with an explicit register ring: the
    // activation fragments (and the LDS-resident weight fragments) of k-step k+PD are requested
    // before the MFMAs of k-step k issue; sched_barrier pins that order so the compiler's counted
    // lgkmcnt waits leave PD k-steps of LDS latency in flight (its own schedule prefetches only one).
    auto xfrag = [&](int k, int pt) {
      const int r = k / (KS * C::NQ), s = (k / C::NQ) % KS, q = k % C::NQ;
      const int off = XTAB ? xoff[s][XTAB ? q : 0] : (xbase[s] + (((2 * q) ^ xkey[s]) << 4));
      return *reinterpret_cast<const half8*>(xb + off + (r + pt * C::RPT * S) * C::IWs * C::PIXB);
    };
    if constexpr (WREG) {
      constexpr int PD = 3;                       // prefetch distance in k-steps
      half8 xq[PD + 1][C::PT];
      half8 wq[PD + 1];
#pragma unroll
      for (int k = 0; k < PD && k < C::NK; ++k) {
#pragma unroll
        for (int pt = 0; pt < C::PT; ++pt) xq[k][pt] = xfrag(k, pt);
        if (k >= NKR) wq[k] = wlds[(k - NKR) * 64];
      }
#pragma unroll
      for (int k = 0; k < C::NK; ++k) {
        if (k + PD < C::NK) {
#pragma unroll
          for (int pt = 0; pt < C::PT; ++pt) xq[(k + PD) % (PD + 1)][pt] = xfrag(k + PD, pt);
          if (k + PD >= NKR) wq[(k + PD) % (PD + 1)] = wlds[(k + PD - NKR) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        const half8 wf = (k < NKR) ? wreg[k < NKR ? k : 0] : wq[k % (PD + 1)];
        if constexpr (PRO) {      // at the USE, not at the fetch: the LDS reads stay PD k-steps ahead
#pragma unroll
          for (int pt = 0; pt < C::PT; ++pt) xq[k % (PD + 1)][pt] = pro_apply(xq[k % (PD + 1)][pt], k % C::NQ);
        }
#pragma unroll
        for (int pt = 0; pt < C::PT; ++pt)
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xq[k % (PD + 1)][pt], acc[pt], 0, 0, 0);
        if constexpr (DS) {
          if (k / C::NQ == 4) {     // centre tap
#pragma unroll
            for (int pt = 0; pt < C::PT; ++pt)
              accd[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdsr[k % C::NQ], xq[k % (PD + 1)][pt], accd[pt], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // weights streamed from L2 (128-channel layers on tiny maps): a ring of PW fragments stays
      // in flight across the (rolled) tap-row loop so the ~1 us L2 latency is paid once, not per row
      constexpr int RK = KS * C::NQ;              // k-steps per tap row
      // (a whole tap row in flight, PW = RK = 24 for 128 channels, measured slower: bs-1 forward 0.236 -> 0.248 ms, same session)
      constexpr int PW = (RK % 12 == 0) ? 12 : 8; // fragments in flight: half a tap row (cold L2/MALL: ~2 us per round trip)
      static_assert(RK % PW == 0, "weight ring must wrap on a tap-row boundary");
      half8 wq[PW];
#pragma unroll
      for (int i = 0; i < PW; ++i) wq[i] = wsrc[(size_t)i * 64];
      // the pixel fragments go through a register ring as well (XPD k-steps ahead, restarted per tap row): read right in
      // front of their MFMA, each of the 72 x PT LDS reads of a 128-channel tile waited its full round trip
      constexpr int XPD = RK > 3 ? 3 : RK - 1;
      half8 xq[XPD + 1][C::PT];
#pragma unroll 1
      for (int r = 0; r < KS; ++r) {
        const char* xr = xb + r * C::IWs * C::PIXB;
        const int knext = r * RK + PW;   // ring refill index; clamped (re-reads the last fragment, branch-free)
        auto xfrag = [&](int j, int pt) {
          const int s = j / C::NQ, q = j % C::NQ;
          const int off = XTAB ? xoff[s][XTAB ? q : 0] : (xbase[s] + (((2 * q) ^ xkey[s]) << 4));
          return *reinterpret_cast<const half8*>(xr + off + (pt * C::RPT * S) * C::IWs * C::PIXB);
        };
#pragma unroll
        for (int j = 0; j < XPD; ++j)
#pragma unroll
          for (int pt = 0; pt < C::PT; ++pt) xq[j][pt] = xfrag(j, pt);
#pragma unroll
        for (int j = 0; j < RK; ++j) {
          const int s = j / C::NQ, q = j % C::NQ;
          (void)s;
          const half8 wf = wq[j % PW];
          __builtin_amdgcn_sched_barrier(0);
          wq[j % PW] = wsrc[(size_t)((knext + j) < C::NK ? (knext + j) : (C::NK - 1)) * 64];
          if (j + XPD < RK) {
#pragma unroll
            for (int pt = 0; pt < C::PT; ++pt) xq[(j + XPD) % (XPD + 1)][pt] = xfrag(j + XPD, pt);
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the refills HERE: left alone, the compiler sinks every
                                               // load next to its use (load -> waitcnt(0) -> MFMA)
#pragma unroll
          for (int pt = 0; pt < C::PT; ++pt) {
            const half8 xf = xq[j % (XPD + 1)][pt];
            acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[pt], 0, 0, 0);
            if constexpr (DS) {
              if (r == 1 && s == 1) accd[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdsr[q], xf, accd[pt], 0, 0, 0);
            }
          }
        }
      }
    }

    if (TAIL) {
      // main conv epilogue -> fp16 -> LDS `mid` tile [pixel][CMID] (swizzled), then 1x1 tail
      char* mid = smem + C::NBUF * C::IN_BYTES;
#pragma unroll
      for (int pt = 0; pt < C::PT; ++pt) {
        const int pb = (pg * C::PT + pt) * 32 + pix;
        const int fm = (pb / C::MPPR) % C::MCPP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = lfd_cvt_pk_max(acc[pt][4 * g + 0], acc[pt][4 * g + 1], lo_mid);
          v.y = lfd_cvt_pk_max(acc[pt][4 * g + 2], acc[pt][4 * g + 3], lo_mid);
          const int cm = ct * 4 + g;
          *reinterpret_cast<uint2*>(mid + pb * C::MPIXB + ((cm ^ fm) * 16) + 8 * h) = v;
        }
      }
      block_barrier();
      {
        const float* bp = sbias + 256 + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
          for (int pt = 0; pt < C::PT; ++pt) {
            acc[pt][4 * g + 0] = b4.x; acc[pt][4 * g + 1] = b4.y;
            acc[pt][4 * g + 2] = b4.z; acc[pt][4 * g + 3] = b4.w;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < C::NK2; ++q) {
#pragma unroll
        for (int pt = 0; pt < C::PT; ++pt) {
          const int pb = (pg * C::PT + pt) * 32 + pix;
          const int fm = (pb / C::MPPR) % C::MCPP;
          const half8 xf = *reinterpret_cast<const half8*>(mid + pb * C::MPIXB + (((2 * q + h) ^ fm) * 16));
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2reg[q], xf, acc[pt], 0, 0, 0);
        }
      }
    }

    if constexpr (ACC32) {
      // debug / parity instrument (lfd_conv2d_nhwc_f16_acc32): the fp32 accumulators (conv + bias) leave the kernel as they
      // are -- no activation, no fp16 rounding -- straight from the MFMA layout (lane = pixel, 4 consecutive channels per
      // register group -> 16-byte stores).  The next loop-top barrier is the only hand-off the input buffers need.
#pragma unroll
      for (int pt = 0; pt < C::PT; ++pt) {
        const int oy = ty0 * C::TH + (pg * C::PT + pt) * C::RPT + oyl;
        const int ox = tx0 * C::TW + oxl;
        if (oy < a.OH && ox < a.OW) {
          float* o = a.out32 + (((size_t)n * a.OH + oy) * a.OW + ox) * a.cout + co_base + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(o + 8 * g) = make_float4(acc[pt][4 * g + 0], acc[pt][4 * g + 1], acc[pt][4 * g + 2], acc[pt][4 * g + 3]);
        }
      }
      continue;
    }
    CV_T(4);
    // ---- epilogue: (+ residual) -> ReLU -> fp16 -> LDS staging -> full-line 16-byte NHWC stores.
    // Straight from the accumulator layout every store instruction would touch 32 pixel lines with
    // 16 bytes each (store-issue bound); staged through LDS, consecutive lanes write consecutive
    // 16-byte chunks of one pixel line.  The staging tile lives in storage that is dead by now: the
    // consumed input buffer (or the `mid` tile of the chained 1x1).
    const int cout_out = TAIL ? a.cout2 : a.cout;
    const uint32_t lo_out = (TAIL ? a.relu2 : a.relu) ? LFD_PK_RELU : LFD_PK_NONE;
    constexpr int OCPP = NCT * 4;                         // 16-byte chunks per pixel of this block's channel slice
    constexpr int OPIXB = NCT * 64;
    constexpr int OPPR = (OCPP >= 16) ? 1 : 16 / OCPP;
    constexpr int OPX = C::PG * C::PT * 32;
    static_assert(OPX * OPIXB <= C::IN_BYTES, "output staging must fit the input buffer");
    char* sout = TAIL ? (smem + C::NBUF * C::IN_BYTES) : (smem + buf * C::IN_BYTES);
    block_barrier();   // every wave is done reading the input buffer / mid tile
    CV_T(5);
#pragma unroll
    for (int pt = 0; pt < C::PT; ++pt) {
      const int pb = (pg * C::PT + pt) * 32 + pix;
      const int fo = (pb / OPPR) % OCPP;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float x0 = acc[pt][4 * g + 0], x1 = acc[pt][4 * g + 1], x2 = acc[pt][4 * g + 2], x3 = acc[pt][4 * g + 3];
        if constexpr (RES) {
          x0 += (float)resv[pt][g][0]; x1 += (float)resv[pt][g][1]; x2 += (float)resv[pt][g][2]; x3 += (float)resv[pt][g][3];
        }
        uint2 v;
        v.x = lfd_cvt_pk_max(x0, x1, lo_out);
        v.y = lfd_cvt_pk_max(x2, x3, lo_out);
        *reinterpret_cast<uint2*>(sout + pb * OPIXB + (((ct * 4 + g) ^ fo) * 16) + 8 * h) = v;
      }
    }
    CV_T(6);
    block_barrier();
    CV_T(7);
    if constexpr (FAST) {
      // iteration k of thread tid: staging pixel (tid >> 3) + 32k = MFMA tile k, pixel tid >> 3 -> output row
      // 8 ty0 + 2k + (tid >> 7), column 16 tx0 + ((tid >> 3) & 15), chunk tid & 7: scalar row base + per-thread constant
      int otid = threadIdx.x;
      asm volatile("" : "+v"(otid));   // (opaque for the same reason as in issue_dma_fast)
      const int trow = otid >> 7, tcol = (otid >> 3) & 15, tc = otid & 7;
      const int lofs = (otid >> 3) * OPIXB + ((tc ^ ((otid >> 4) & 7)) * 16);
      const long gofs = trow * f_rowpitch + tcol * 128 + tc * 16;
      char* obase = reinterpret_cast<char*>(a.out) + ((long)n * a.OH + ty0 * C::TH) * f_rowpitch + (long)tx0 * C::TW * 128;
      const bool colok = tx0 * C::TW + tcol < a.OW;
      char* trash = reinterpret_cast<char*>(const_cast<_Float16*>(a.zeros)) + 2048 + (otid & 127) * 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 v = *reinterpret_cast<const uint4*>(sout + k * 4096 + lofs);
        const bool ok = colok && ty0 * C::TH + 2 * k + trow < a.OH;
        char* dst = ok ? obase + 2 * k * f_rowpitch + gofs : trash;
        *reinterpret_cast<uint4*>(dst) = v;
        if constexpr (STATS) stats_add(v, ok, st_s, st_q);
      }
    } else if constexpr (BSUM) {
      const int cslice = cog * NCT * 32;
#pragma unroll
      for (int k = 0; k < B_NST; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int pb = i / OCPP, c = i - pb * OCPP;
        const int t32 = pb >> 5, p32 = pb & 31;
        const int oy = ty0 * C::TH + t32 * C::RPT + p32 / C::TW;
        const int ox = tx0 * C::TW + p32 % C::TW;
        const int fo = (pb / OPPR) % OCPP;
        const uint4 v = *reinterpret_cast<const uint4*>(sout + pb * OPIXB + ((c ^ fo) * 16));
        const bool ok = oy < a.OH && ox < a.OW;
        _Float16* dst = ok ? a.out + (((size_t)n * a.OH + oy) * a.OW + ox) * cout_out + cslice + c * 8
                           : const_cast<_Float16*>(a.zeros) + 1024 + (threadIdx.x & 127) * 8;
        *reinterpret_cast<uint4*>(dst) = v;
        bsum_add(v, bs_y[k], ok, bs_m, bs_r, bs_g, bs_b, st_s, st_q);
      }
    } else {
      const int cslice = TAIL ? 0 : cog * NCT * 32;
      for (int i = threadIdx.x; i < OPX * OCPP; i += 256) {
        const int pb = i / OCPP, c = i - pb * OCPP;
        const int t32 = pb >> 5, p32 = pb & 31;
        const int oy = ty0 * C::TH + t32 * C::RPT + p32 / C::TW;
        const int ox = tx0 * C::TW + p32 % C::TW;
        const int fo = (pb / OPPR) % OCPP;
        const uint4 v = *reinterpret_cast<const uint4*>(sout + pb * OPIXB + ((c ^ fo) * 16));
        // exactly one store per lane and iteration (see the counted wait at the loop top): lanes of
        // out-of-image pixels write into the trash half of the `zeros` line
        const bool ok = oy < a.OH && ox < a.OW;
        _Float16* dst = ok ? a.out + (((size_t)n * a.OH + oy) * a.OW + ox) * cout_out + cslice + c * 8
                           : const_cast<_Float16*>(a.zeros) + 1024 + (threadIdx.x & 127) * 8;
        *reinterpret_cast<uint4*>(dst) = v;
        if constexpr (STATS) stats_add(v, ok, st_s, st_q);   // (c == threadIdx.x % OCPP in every iteration)
      }
    }
    CV_T(8);
    if constexpr (DS) {
      // second output: the identity branch (bias already in accd, no ReLU), same staging tile
      block_barrier();
#pragma unroll
      for (int pt = 0; pt < C::PT; ++pt) {
        const int pb = (pg * C::PT + pt) * 32 + pix;
        const int fo = (pb / OPPR) % OCPP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (_Float16)accd[pt][4 * g + j];
          *reinterpret_cast<half4*>(sout + pb * OPIXB + (((ct * 4 + g) ^ fo) * 16) + 8 * h) = v;
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
          const uint4 v = *reinterpret_cast<const uint4*>(sout + pb * OPIXB + ((c ^ fo) * 16));
          *reinterpret_cast<uint4*>(a.out_ds + (((size_t)n * a.OH + oy) * a.OW + ox) * a.cout + cslice + c * 8) = v;
        }
      }
    }
  }
  if constexpr (STATS || BSUM) {
    // lanes l, l + OCPP, ... of a wave hold the same chunk: butterfly over them, then the four waves through LDS in wave order
    constexpr int OCPP = NCT * 4, CB = NCT * 32;
    static_assert(256 % OCPP == 0 && 64 % OCPP == 0, "one channel chunk per thread");
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int d = 32; d >= OCPP; d >>= 1) {
        st_s[e] += __shfl_xor(st_s[e], d);
        st_q[e] += __shfl_xor(st_q[e], d);
      }
    }
    block_barrier();      // the staging tile is dead for every wave
    float* red = reinterpret_cast<float*>(smem);          // [4 waves][2][CB]
    static_assert(4 * 2 * CB * 4 <= C::IN_BYTES, "partials fit the input buffer");
    if (lane < OCPP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * CB + lane * 8 + e] = st_s[e];
        red[(wave * 2 + 1) * CB + lane * 8 + e] = st_q[e];
      }
    }
    block_barrier();
    for (int o = threadIdx.x; o < 2 * CB; o += 256) {
      const int q = o / CB, chl = o - q * CB;
      const float t = ((red[(0 * 2 + q) * CB + chl] + red[(1 * 2 + q) * CB + chl]) + red[(2 * 2 + q) * CB + chl]) + red[(3 * 2 + q) * CB + chl];
      a.stat_partials[(size_t)blockIdx.x * 2 * a.cout + (size_t)q * a.cout + cog * CB + chl] = t;
    }
  }
  CV_END();
}

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, bool ACC32 = false, bool STATS = false, bool PRO = false, bool BSUM = false>
__global__ __launch_bounds__(256, 2) void k_conv(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_block<CIN, KS, S, NCT, WREG, TAIL, RES, DS, ACC32, STATS, PRO, BSUM>(a, smem);
}

template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL, bool RES, bool DS, bool ACC32 = false, bool STATS = false, bool PRO = false, bool BSUM = false>
int launch_conv_(const ConvArgs& a0, hipStream_t st, int* blocks_out = nullptr) {
  using C = Cfg<CIN, KS, S, NCT, WREG, TAIL>;
  ConvArgs a = a0;
  a.tiles_x = (a.OW + C::TW - 1) / C::TW;
  a.tiles_y = (a.OH + C::TH - 1) / C::TH;
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  const int cgroups = TAIL ? 1 : a.cout / (NCT * 32);
  constexpr int LDSB = C::LDS_BYTES;
  // the inference shapes all fit two workgroups per CU; a few data-gradient shapes of the training path (few output
  // channels -> tall tiles) only fit one
  static_assert(LDSB <= 160 * 1024, "LDS capacity");
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv<CIN, KS, S, NCT, WREG, TAIL, RES, DS, ACC32, STATS, PRO, BSUM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  // workgroup b works on XCD b % 8's contiguous tile range [xcd * ceil(ntiles / 8), ...): a small launch needs
  // 8 * ceil(ntiles / 8) workgroups for every tile to have its own (17 tiles on 17 workgroups = two rounds on five XCDs)
  int blocks = 512 / cgroups;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  if (blocks < 1) blocks = 1;
  if (blocks_out) *blocks_out = blocks;
  hipLaunchKernelGGL((k_conv<CIN, KS, S, NCT, WREG, TAIL, RES, DS, ACC32, STATS, PRO, BSUM>), dim3(blocks, cgroups), dim3(256), LDSB, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

// residual variants are only instantiated for stride-1 convs: the 3x3 that closes a residual block, and (training) the
// data-gradient convs, whose "residual" is the gradient already collected for the same activation
template <int CIN, int KS, int S, int NCT, bool WREG, bool TAIL>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  if constexpr (S == 1 && !TAIL) {
    if (a.res) return launch_conv_<CIN, KS, S, NCT, WREG, TAIL, true, false>(a, st);
  } else {
    if (a.res) return LFD_ERR_UNSUPPORTED;
  }
  if constexpr (KS == 3 && S == 2 && !TAIL) {
    if (a.wds) return launch_conv_<CIN, KS, S, NCT, WREG, TAIL, false, true>(a, st);
  } else {
    if (a.wds) return LFD_ERR_UNSUPPORTED;
  }
  return launch_conv_<CIN, KS, S, NCT, WREG, TAIL, false, false>(a, st);
}

}  // namespace
