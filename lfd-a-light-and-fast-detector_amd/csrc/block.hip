// csrc/block.hip -- lfd_fasterblock_fused_f16: one launch per residual block of the LFD backbone.
//
// Reference: FasterBlock.forward (lfd/model/backbone/lfd_resnet.py:96-154) for the blocks WITHOUT a downsample branch
// (every block of a stage but the first):      y = ReLU( BN2(conv3x3(ReLU(BN1(conv3x3(x))))) + x ),   64 -> 64 -> 64 channels.
// The reference runs two cuDNN convolutions plus BatchNorm / ReLU / add kernels with the 64-channel intermediate map going
// through memory; the two-launch path of this library (conv_impl.h) still writes and re-reads it once.  Here the
// intermediate never leaves the CU: per 8 x 16 output tile its 10 x 18 halo tile is produced into LDS and consumed in place
// (HBM traffic of a block = one read + one write of the map + the residual re-read served by L2).
//
// MI355X-first structure -- PRODUCER / CONSUMER WAVE SPECIALISATION on the register file:
//   * one 512-thread workgroup per CU = two waves per SIMD.  Waves 0-3 are conv1 PRODUCERS, waves 4-7 conv2 CONSUMERS; the
//     hardware places wave w and wave w+4 on the same SIMD, so every SIMD's matrix pipe is shared by one producer and
//     one consumer whose non-MFMA phases (LDS-DMA issue, epilogues, stores) hide under the partner's MFMAs.
//   * both filters are STATIONARY: a wave owns a 32-output-channel slab of ONE of the two convolutions (36 MFMA fragments =
//     36 KB) for the lifetime of the persistent workgroup -- the 512 KB register file of a CU is what holds the 144 KB of
//     filters; nothing is re-fetched per tile.  (Producers keep 25 fragments in VGPRs and 11 in LDS -- they carry three
//     accumulator tiles -- consumers all 36 in VGPRs.)
//   * software pipeline over tiles with ONE s_barrier per step: in step s the producers contract tile s (input halo tile
//     fetched by global->LDS DMA one step ahead, double buffered) into mid[s & 1] while the consumers contract tile s-1 from
//     mid[(s-1) & 1], add the identity (fp16 loads of x, L2-hot: the producers' DMA just pulled the same lines), and store.
//   * v_mfma_f32_32x32x16_f16, A = filters, B = pixels (lane = pixel): bias / residual / ReLU / fp16 pack are lane-local;
//     the mid tile is written pixel-major with a 144-byte pitch (conflict-free ds_write_b64 and ds_read_b128, no swizzle
//     arithmetic on the consumer side: one address register, every tap an immediate offset).
//   * conv1 covers the 180 halo pixels of the mid tile as 6 x 32 LINEARLY numbered pixels (m -> row m / 18, column m % 18):
//     93.75 % of the MFMA slots do useful work; halo recompute factor (192 + 128) / (2 x 128) = 1.25.
//   * results are BIT-IDENTICAL to the two-launch path: same operand rounding points (fp16 intermediate after ReLU), same
//     k order, bias in the accumulator, one rounding of acc + residual (tests/test_gpu_block.py).
#include "conv_impl.h"

namespace {

struct BlockArgs {
  const _Float16* in;    // [N,H,W,64]
  _Float16* out;         // [N,H,W,64]
  const half8* w1;       // packed [2][36][64] half8 (ops.pack_conv_weight of the BN-folded conv1)
  const float* b1;       // [64]
  const half8* w2;
  const float* b2;
  const _Float16* zeros;  // 4 KB line: [0,2048) zero, [2048,4096) trash
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
};

struct BK {
  static constexpr int TH = 8, TW = 16;                 // output tile
  static constexpr int MH = TH + 2, MW = TW + 2;        // mid tile (conv1 output with conv2's halo): 10 x 18 = 180 pixels
  static constexpr int MPIX = MH * MW;
  static constexpr int IH = TH + 4, IW = TW + 4;        // input halo tile 12 x 20
  static constexpr int IN_PIXB = 144;                   // input tile: same padded pixel pitch as the mid tile (see issue_dma)
  static constexpr int IN_ROWB = 3104;                  // 20 * 144 = 2880 padded so that ROWB / 16 = 2 (mod 16): with the 18-wide linear
                                                        // pixel numbering of conv1 the step from column 17 to column 0 of the next row is
                                                        // then +9 sixteen-byte units like every other step -> conflict-free ds_read_b128
  static constexpr int IN_BYTES = IH * IN_ROWB;         // 37248
  static constexpr int MID_PIXB = 144;                  // 128 + 16: pixel pitch that spreads 16 consecutive pixels over all banks
  static constexpr int MID_ROWB = 2816;                 // 18 * 144 = 2592 padded to a multiple of 256 (row pairs stay conflict-free)
  static constexpr int MID_BYTES = MH * MID_ROWB;       // 28160
  static constexpr int STG_WAVE = 32 * 64;              // consumer-private staging: 32 pixels x 32 channels fp16 (one MFMA tile at a time)
  static constexpr int NK = 36;                         // k-steps of a 3x3x64 contraction
  static constexpr int WLP = 11;                        // producer weight fragments living in LDS (per cout tile)
  static constexpr int NKRP = NK - WLP;                 // ... and in VGPRs
  static constexpr int OFF_IN = 0;
  static constexpr int OFF_MID = OFF_IN + 2 * IN_BYTES;          // 74496
  static constexpr int OFF_STG = OFF_MID + 2 * MID_BYTES;        // 130816
  static constexpr int OFF_WP = OFF_STG + 4 * STG_WAVE;          // 139008
  static constexpr int OFF_BIAS = OFF_WP + 2 * WLP * 1024;       // 161536
  static constexpr int LDS_BYTES = OFF_BIAS + 2 * 64 * 4;        // 162048 <= 163840
};
static_assert(BK::LDS_BYTES <= 160 * 1024, "LDS capacity");

#ifdef LFD_BLOCK_TIMING
// phase stamps of workgroup 0: [role 0 = producer wave 0, 1 = consumer wave 4][step < 16][stamp < 8] (shader clock)
__device__ unsigned long long g_blk_dbg[2 * 16 * 8];
#define BT(role, i) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && dbg_step < 16) { g_blk_dbg[((role) * 16 + dbg_step) * 8 + (i)] = __builtin_readcyclecounter(); \
    if ((i) == 0) g_blk_dbg[((role) * 16 + dbg_step) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BT(role, i)
#endif

// (image, tile row, tile column) of a workgroup's tiles t_first, t_first + t_step, ...: the integer divisions are done once,
// every further tile is three scalar add-and-wrap steps (a division per tile and wave measured ~250 cycles of dependent SALU).
struct TileWalk {
  int n, ty, tx;        // current tile
  int dn, dty, dtx;     // t_step decomposed the same way
  int tiles_x, tiles_y;
  __device__ __forceinline__ void init(const BlockArgs& a, int t, int t_step) {
    tiles_x = a.tiles_x; tiles_y = a.tiles_y;
    const int per_img = a.tiles_x * a.tiles_y;
    n = t / per_img;
    int tr = t - n * per_img;
    ty = tr / a.tiles_x;
    tx = tr - ty * a.tiles_x;
    dn = t_step / per_img;
    tr = t_step - dn * per_img;
    dty = tr / a.tiles_x;
    dtx = tr - dty * a.tiles_x;
  }
  __device__ __forceinline__ void advance() {
    tx += dtx;
    if (tx >= tiles_x) { tx -= tiles_x; ty += 1; }
    ty += dty;
    if (ty >= tiles_y) { ty -= tiles_y; n += 1; }
    n += dn;
  }
};

// ---- input halo DMA: 12 rows x 20 pixels, pixel pitch 144 B in LDS (8 chunks of 16 B + one 16-byte gap), so that the
// MFMA B-fragment reads are conflict-free WITHOUT a swizzle and every tap / k-step is an immediate offset from one address
// register per pixel tile.  A DMA instruction writes lane-linear (LDS byte 16 L of its 1 KB window), so the gaps are made on
// the SOURCE side: lane L carries chunk L % 9 of pixel L / 9 (chunk 8 = the gap: masked off together with lane 63, whose
// 16 bytes would land in the next window).  A row is three windows of 7 + 7 + 6 pixels; rows rw, rw + 4, rw + 8 belong to
// wave rw (0..3) of the issuing role.  Out-of-image pixels come from the zero line: a lane whose COLUMN is outside the image
// points at the zero line with a row pitch of 0, a ROW outside the image is a wave-uniform case -- one 64-bit multiply-add
// per DMA.
__device__ __forceinline__ void issue_dma(const BlockArgs& a, char* smem, int rw, const TileWalk& tw, int buf) {
  const long rowpitch = (long)a.W * 128;
  int ol = threadIdx.x & 63;
  asm volatile("" : "+v"(ol));          // recompute the per-lane constants per tile instead of pinning registers
  const int lpx = (ol * 57) >> 9;       // ol / 9 for ol < 64
  const int ck = ol - 9 * lpx;
  const int n = tw.n;
  const int gy0 = tw.ty * BK::TH - 2, gx0 = tw.tx * BK::TW - 2;
  const char* img = reinterpret_cast<const char*>(a.in) + (long)n * a.H * rowpitch;
  const char* zsrc = reinterpret_cast<const char*>(a.zeros) + (ck & 7) * 16;
  char* lbase = smem + BK::OFF_IN + buf * BK::IN_BYTES;
  if (ck < 8 && ol < 63) {
#pragma unroll
    for (int seg = 0; seg < 3; ++seg) {
      const int col = 7 * seg + lpx;
      const int gx = gx0 + col;
      const bool xv = (gx >= 0) && (gx < a.W);
      const char* cbase_p = xv ? img + (long)gx * 128 + ck * 16 : zsrc;      // row 0 of the image at this lane's column
      const unsigned rp = xv ? (unsigned)rowpitch : 0u;
      if (seg < 2 || col < BK::IW) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int iy = rw + 4 * i;
          const int gy = gy0 + iy;                                           // wave-uniform
          const char* src = (gy >= 0 && gy < a.H) ? cbase_p + (unsigned long)rp * (unsigned)gy : zsrc;
          dma16(src, lbase + iy * BK::IN_ROWB + seg * 1008);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ producer (conv1)
__device__ __forceinline__ void producer(const BlockArgs& a, char* smem, int pw, int t_first, int t_end, int t_step) {
  const int lane = threadIdx.x & 63;
  const int ct = pw & 1, pgp = pw >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + BK::OFF_BIAS);

  // ---- stationary conv1 filter slab: 25 fragments in VGPRs, 11 in LDS (shared by the two producers of this cout tile)
  half8 wreg[BK::NKRP];
  const half8* wsrc = a.w1 + (size_t)ct * BK::NK * 64 + lane;
#pragma unroll
  for (int k = 0; k < BK::NKRP; ++k) wreg[k] = wsrc[(size_t)k * 64];
  half8* wlds = reinterpret_cast<half8*>(smem + BK::OFF_WP) + (ct * BK::WLP) * 64 + lane;
  if (pgp == 0) {
#pragma unroll
    for (int k = 0; k < BK::WLP; ++k) wlds[k * 64] = wsrc[(size_t)(BK::NKRP + k) * 64];   // visible after the first barrier
  }

  // ---- per-lane geometry of this wave's three 32-pixel MFMA tiles: mid pixel m = (3 pgp + pt) * 32 + pix, linear over 10 x 18
  int pbase[3], mwoff[3], mrc[3];
  bool mval[3];
#pragma unroll
  for (int pt = 0; pt < 3; ++pt) {
    const int m = (pgp * 3 + pt) * 32 + pix;
    mval[pt] = m < BK::MPIX;
    const int mm = mval[pt] ? m : BK::MPIX - 1;
    const int row = mm / BK::MW, col = mm - row * BK::MW;
    pbase[pt] = row * BK::IN_ROWB + col * BK::IN_PIXB + h * 16;
    mwoff[pt] = row * BK::MID_ROWB + col * BK::MID_PIXB + ct * 64 + h * 8;
    mrc[pt] = (row << 8) | col;
  }

  int t = t_first;
  int buf = 0;
  TileWalk cur;
  cur.init(a, t_first, t_step);
#ifndef BK_DMA_CONS
  TileWalk nxt = cur;
  nxt.advance();
#endif
  int dbg_step = 0; (void)dbg_step;
#ifndef BK_DMA_CONS
  for (;; t += t_step, buf ^= 1, ++dbg_step, cur = nxt, nxt.advance()) {
#else
  for (;; t += t_step, buf ^= 1, ++dbg_step, cur.advance()) {
#endif
    const bool active = t < t_end;
    BT(0, 0);
#ifndef BK_DMA_CONS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of tile t's input (issued a step ago) has landed
#endif
    block_barrier();       // in[buf] landed (every issuing wave awaited its DMA); the consumers are done with mid[buf] (tile t - 2 t_step)
    BT(0, 2);
    if (!active) break;    // (the consumers run one more step and meet nobody: the workgroup's barrier count stays equal, see k_block)
#ifndef BK_DMA_CONS
    // The PRODUCERS feed themselves (round 3): in[buf ^ 1] is free from this barrier on (they read it in the previous step), the
    // tile has a whole step to land, and the consumer wave of this SIMD starts its 72 MFMAs right away while this wave is busy
    // with addresses -- phase stamps of round 2's arrangement showed the consumer chain (DMA issue 2300 + MFMAs 4550 + epilogue
    // 1550 cycles) as the critical path of a 9300-cycle step with the producers idle for the last 3600 of it.  Measured:
    // 8300 cycles per step, but the chip gives most of it back as clock (it runs k_block64 at its 1.4 kW power cap, 2.1 GHz:
    // profiles/r03_power_trace.json): -1.2 % at 8 x 135 x 240, -3.8 % at 32 x 135 x 240, +0.8 % at 8 x 68 x 120.
    if (t + t_step < t_end) issue_dma(a, smem, pw, nxt, buf ^ 1);
#endif
    BT(0, 3);

    const int ty0 = cur.ty, tx0 = cur.tx;
    const char* xin = smem + BK::OFF_IN;
    const int pb0 = pbase[0] + buf * BK::IN_BYTES, pb1 = pbase[1] + buf * BK::IN_BYTES, pb2 = pbase[2] + buf * BK::IN_BYTES;
    f32x16 acc[3];
    {
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
          acc[pt][4 * g + 0] = b4.x; acc[pt][4 * g + 1] = b4.y; acc[pt][4 * g + 2] = b4.z; acc[pt][4 * g + 3] = b4.w;
        }
      }
    }
    auto xfrag = [&](int k, int pt) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return *reinterpret_cast<const half8*>(xin + (pt == 0 ? pb0 : (pt == 1 ? pb1 : pb2)) + (r * BK::IN_ROWB + s * BK::IN_PIXB + q * 32));
    };
    constexpr int PD = 2;
    half8 xq[PD + 1][3];
    half8 wq[PD + 1];
#pragma unroll
    for (int k = 0; k < PD; ++k) {
#pragma unroll
      for (int pt = 0; pt < 3; ++pt) xq[k][pt] = xfrag(k, pt);
      if (k >= BK::NKRP) wq[k] = wlds[(k - BK::NKRP) * 64];
    }
#pragma unroll
    for (int k = 0; k < BK::NK; ++k) {
      if (k + PD < BK::NK) {
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) xq[(k + PD) % (PD + 1)][pt] = xfrag(k + PD, pt);
        if (k + PD >= BK::NKRP) wq[(k + PD) % (PD + 1)] = wlds[(k + PD - BK::NKRP) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      const half8 wf = (k < BK::NKRP) ? wreg[k < BK::NKRP ? k : 0] : wq[k % (PD + 1)];
#pragma unroll
      for (int pt = 0; pt < 3; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xq[k % (PD + 1)][pt], acc[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    BT(0, 4);
    // ---- epilogue: ReLU -> fp16 -> mid[buf]; halo pixels OUTSIDE THE IMAGE are conv2's zero padding, not conv1 outputs
    char* mid = smem + BK::OFF_MID + buf * BK::MID_BYTES;
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
      if (mval[pt]) {
        const int my = ty0 * BK::TH - 1 + (mrc[pt] >> 8), mx = tx0 * BK::TW - 1 + (mrc[pt] & 255);
        const bool inimg = my >= 0 && my < a.H && mx >= 0 && mx < a.W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = lfd_cvt_pk_max(acc[pt][4 * g + 0], acc[pt][4 * g + 1], LFD_PK_RELU);
          v.y = lfd_cvt_pk_max(acc[pt][4 * g + 2], acc[pt][4 * g + 3], LFD_PK_RELU);
          if (!inimg) { v.x = 0u; v.y = 0u; }
          *reinterpret_cast<uint2*>(mid + mwoff[pt] + 16 * g) = v;
        }
      }
    }
    BT(0, 5);
  }
}

// ------------------------------------------------------------------------------------------------ consumer (conv2)
// The consumers also feed the producers: they carry a third fewer MFMAs (72 vs 108 per step), so the LDS-DMA of the NEXT
// input halo tile is issued by them at the top of every step (measured on the first version, where the producers issued
// it: ~2000 cycles of address arithmetic + 9 DMA instructions per wave and step in front of the longer MFMA chain).
__device__ __forceinline__ void consumer(const BlockArgs& a, char* smem, int cw, int t_first, int t_end, int t_step) {
  const int lane = threadIdx.x & 63;
  const int ct = cw & 1, pgc = cw >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + BK::OFF_BIAS) + 64;
  const int co_base = ct * 32;

  half8 wreg[BK::NK];
  const half8* wsrc = a.w2 + (size_t)ct * BK::NK * 64 + lane;
#pragma unroll
  for (int k = 0; k < BK::NK; ++k) wreg[k] = wsrc[(size_t)k * 64];

  // mid pixel of output pixel (row 4 pgc + 2 pt + (pix >> 4), column pix & 15) at tap (0, 0); taps / pt / q are immediates
  const int cbase = (pgc * 4 + (pix >> 4)) * BK::MID_ROWB + (pix & 15) * BK::MID_PIXB + h * 16;
  char* stg = smem + BK::OFF_STG + cw * BK::STG_WAVE;
  const long rowpitch = (long)a.W * 128;

  // ---- the two halves of a consumer step.  contract(): conv2 of one tile from mid[] into the accumulators (+ request of the
  // identity values); epilogue(): + identity -> ReLU -> fp16 -> stores.  Default order: contract, then epilogue of the SAME
  // tile.  -DBK_EPI_EARLY runs the epilogue at the START of the next step instead, so that both non-MFMA phases of this wave
  // fall into the producer's MFMA window -- measured round 3 (phase stamps, tools/probe_block.py): the epilogue takes 5100
  // instead of 1550 cycles beside a contracting partner (the older wave wins every issue arbitration and the LDS round trips
  // queue behind its B-fragment reads), the two waves' MFMA phases no longer overlap, and a lone wave only reaches 45-56
  // cycles per MFMA: 47.0 vs 45.3 us at 8 x 135 x 240.  A negative result, kept for A/B.
  f32x16 acc[2];
  half4 resv[2][4];
  auto contract = [&](const TileWalk& tw, const char* mid) {
    const int n = tw.n, ty0 = tw.ty, tx0 = tw.tx;
    {
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          acc[pt][4 * g + 0] = b4.x; acc[pt][4 * g + 1] = b4.y; acc[pt][4 * g + 2] = b4.z; acc[pt][4 * g + 3] = b4.w;
        }
      }
    }
    auto xfrag = [&](int k, int pt) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return *reinterpret_cast<const half8*>(mid + cbase + (pt * 2 * BK::MID_ROWB + r * BK::MID_ROWB + s * BK::MID_PIXB + q * 32));
    };
    // identity branch: requested two thirds into the contraction (its ~L2 latency hides under the last 12 k-steps)
    constexpr int RES_K = 22;
    constexpr int PD = 3;
    half8 xq[PD + 1][2];
#pragma unroll
    for (int k = 0; k < PD; ++k) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) xq[k][pt] = xfrag(k, pt);
    }
#pragma unroll
    for (int k = 0; k < BK::NK; ++k) {
      if (k + PD < BK::NK) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) xq[(k + PD) % (PD + 1)][pt] = xfrag(k + PD, pt);
      }
      if (k == RES_K) {
#ifdef BK_NO_RES
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
          for (int g = 0; g < 4; ++g) resv[pt][g] = half4{0, 0, 0, 0};
#else
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const int oy = ty0 * BK::TH + pgc * 4 + pt * 2 + (pix >> 4);
          const int ox = tx0 * BK::TW + (pix & 15);
          const bool ok = oy < a.H && ox < a.W;
          const _Float16* rp = a.in + (((size_t)n * a.H + (ok ? oy : 0)) * a.W + (ok ? ox : 0)) * 64 + co_base + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) resv[pt][g] = *reinterpret_cast<const half4*>(rp + 8 * g);
        }
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[k], xq[k % (PD + 1)][pt], acc[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // one MFMA tile (32 pixels = 2 output rows x 16) at a time: + identity -> ReLU -> fp16 -> wave-private staging (32 pixels x
  // 64 B, chunk XOR (p >> 2) & 3) -> 16-byte stores; four lanes write the 64-byte half line of one pixel (the other cout
  // tile's wave writes the other half).  LDS operations of one wave execute in order: the second tile may overwrite the
  // staging area right after the first tile's reads were issued.
  auto epilogue = [&](const TileWalk& tw) {
    const int n = tw.n, ty0 = tw.ty, tx0 = tw.tx;
    char* obase = reinterpret_cast<char*>(a.out) + ((long)n * a.H + ty0 * BK::TH + pgc * 4) * rowpitch + (long)tx0 * BK::TW * 128 + ct * 64;
    char* trash = reinterpret_cast<char*>(const_cast<_Float16*>(a.zeros)) + 2048 + (threadIdx.x & 127) * 16;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float x0 = acc[pt][4 * g + 0] + (float)resv[pt][g][0], x1 = acc[pt][4 * g + 1] + (float)resv[pt][g][1];
        const float x2 = acc[pt][4 * g + 2] + (float)resv[pt][g][2], x3 = acc[pt][4 * g + 3] + (float)resv[pt][g][3];
        uint2 v;
        v.x = lfd_cvt_pk_max(x0, x1, LFD_PK_RELU);
        v.y = lfd_cvt_pk_max(x2, x3, LFD_PK_RELU);
        *reinterpret_cast<uint2*>(stg + pix * 64 + ((g ^ ((pix >> 2) & 3)) << 4) + 8 * h) = v;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = j * 64 + lane;
        const int p = idx >> 2, c = idx & 3;                              // p = pixel of this MFMA tile
        const uint4 v = *reinterpret_cast<const uint4*>(stg + p * 64 + ((c ^ ((p >> 2) & 3)) << 4));
        const int orow = pt * 2 + (p >> 4), ocol = p & 15;
        const bool ok = (ty0 * BK::TH + pgc * 4 + orow < a.H) && (tx0 * BK::TW + ocol < a.W);
        char* dst = ok ? obase + orow * rowpitch + ocol * 128 + c * 16 : trash;
#ifdef BK_NO_STORE
        if (a.N < 0)
#endif
        *reinterpret_cast<uint4*>(dst) = v;
      }
    }
  };

  int t = t_first;        // the tile the PRODUCERS work on in this step; this wave contracts the previous one
  int buf = 0;
  int tp = -1;
  bool pending = false; (void)pending;    // acc / resv hold a contracted tile (coordinates `pnd`) whose epilogue has not run yet
  TileWalk pnd, prv, cur, nxt;  // tiles: awaiting its epilogue, tp (contracted here), t (the producers'), t + t_step (fetched now)
  cur.init(a, t_first, t_step);
  prv = cur;
  pnd = cur;
  nxt = cur;
  nxt.advance();
  bool first_step = true; (void)first_step;
  if (t < t_end) issue_dma(a, smem, cw, cur, 0);     // the FIRST tile's input is always fetched by the consumers: the producers' prologue is the longer one
  int dbg_step = 0; (void)dbg_step;
  for (;; t += t_step, buf ^= 1, ++dbg_step, prv = cur, cur = nxt, nxt.advance()) {
    BT(1, 0);
    // -DBK_DMA_CONS (rounds 1-2): this wave issued the DMA of tile t one step ago and awaits it here.  Default (round 3): the
    // producers fetch and await every tile but the first.
#ifndef BK_DMA_CONS
    if (first_step) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the prologue DMA (later tiles: the producers fetch and await them)
    first_step = false;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    BT(1, 1);
    block_barrier();       // in[buf] landed for everybody; mid[buf ^ 1] (tile tp) is complete; the producers are done with in[buf ^ 1]
    BT(1, 2);
#ifndef BK_EPI_EARLY
#ifdef BK_DMA_CONS
    if (t + t_step < t_end) issue_dma(a, smem, cw, nxt, buf ^ 1);
#endif
    BT(1, 3);
    if (tp >= 0) {
      contract(prv, smem + BK::OFF_MID + (buf ^ 1) * BK::MID_BYTES);
      BT(1, 4);
      epilogue(prv);
      BT(1, 5);
    }
#else
    // the epilogue comes FIRST: its identity values are compiler-visible loads, and a compiler-counted vmcnt in front of their
    // first use would also wait for DMA instructions issued after them (VMEM retires in order, the inline-asm DMA is not counted)
    if (pending) epilogue(pnd);
    pending = false;
    BT(1, 3);
#if !defined(BK_NO_DMA) && defined(BK_DMA_CONS)
    if (t + t_step < t_end) issue_dma(a, smem, cw, nxt, buf ^ 1);
#endif
    BT(1, 4);
    if (tp >= 0) {
      contract(prv, smem + BK::OFF_MID + (buf ^ 1) * BK::MID_BYTES);
      pnd = prv;
      pending = true;
    }
    BT(1, 5);
#endif
    if (t >= t_end) break;   // the producers had no tile in this step: tp was the last one
    tp = t;
  }
#ifdef BK_EPI_EARLY
  if (pending) epilogue(pnd);
#endif
}

__global__ __launch_bounds__(512) void k_block64(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x < 128) {
    float* sb = reinterpret_cast<float*>(smem + BK::OFF_BIAS);
    sb[threadIdx.x] = threadIdx.x < 64 ? a.b1[threadIdx.x] : a.b2[threadIdx.x - 64];   // visible after the first barrier
  }
  // persistent tile walk, XCD-contiguous ranges (block b runs on XCD b % 8): same mapping as conv_impl.h
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  // Both roles execute exactly (tiles of this workgroup + 1) barriers: the producers one per tile plus the one they leave
  // on, the consumers one per step of the producers plus the step in which they drain the last tile.
#ifdef BK_SETPRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // the second-dispatched half loses every arbitration otherwise
#endif
#ifdef BK_SETPRIO3
  if (wave >= 4) __builtin_amdgcn_s_setprio(3);
#endif
#ifdef BK_CONS_FIRST
  // (A/B: the consumers as the OLDER half of the workgroup -- the older wave of a SIMD wins the issue arbitration)
  if (wave >= 4) producer(a, smem, wave - 4, t_begin + bix, t_end, t_step);
  else consumer(a, smem, wave, t_begin + bix, t_end, t_step);
#else
  if (wave < 4) producer(a, smem, wave, t_begin + bix, t_end, t_step);
  else consumer(a, smem, wave - 4, t_begin + bix, t_end, t_step);
#endif
}

}  // namespace

#ifdef LFD_BLOCK_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_block_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_blk_dbg), sizeof(unsigned long long) * 2 * 16 * 8);
}
#endif

// block_rows.hip: the row-streaming form of the same operator (bit-identical), for large maps
int lfd_block64_rows_launch(const _Float16* in, _Float16* out, const void* w1, const float* b1, const void* w2, const float* b2,
                            const _Float16* zeros, int n, int h, int w, hipStream_t st);
// LFD_BLOCK_ROWS: '0' = always the 8 x 16 tile kernel below, '1' = always the row-streaming kernel, unset = by map size
static int block_rows_mode() {
  const int m = lfd_tune(LFD_TUNE_BLOCK_ROWS);
  return m;
}
// measured per shape (tools/timing/block_rows_sweep.py, same session, rows vs tiles): 8 x 135 x 240 -12 %, 32 x 135 x 240 -17 %,
// 8 x 68 x 120 -16 %, 1 x 540 x 960 -11 %, 4 x 180 x 320 -7 %, 2 x 135 x 240 -4 %; 8 x 34 x 60 +-0, 1 x 135 x 240 +10 % (a segment of
// five rows is mostly prologue), 32 x 160 x 160 +5 % (six strips of 30 columns for 160: 12 % of the lanes idle)
#ifndef LFD_BLOCK_ROWS_MIN_PIXELS
#define LFD_BLOCK_ROWS_MIN_PIXELS 60000L
#endif
static bool block_rows_suits(int n, int h, int w) {
  const int strips = (w + 29) / 30;
  return (long)n * h * w >= LFD_BLOCK_ROWS_MIN_PIXELS && strips * 30 * 10 <= w * 11;
}

extern "C" int lfd_fasterblock_fused_f16(int32_t n, int32_t h, int32_t w, const void* in, void* out, const void* w1_packed,
                                         const float* b1, const void* w2_packed, const float* b2, const void* zeros,
                                         lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || !w1_packed || !b1 || !w2_packed || !b2 || !zeros || in == out) return LFD_ERR_INVALID_ARGUMENT;
  if (n < 1 || h < 1 || w < 1 || !lfd_aligned16(in) || !lfd_aligned16(out)) return LFD_ERR_INVALID_ARGUMENT;
  {
    const int mode = block_rows_mode();
    if (mode == 1 || (mode < 0 && block_rows_suits(n, h, w)))
      return lfd_block64_rows_launch((const _Float16*)in, (_Float16*)out, w1_packed, b1, w2_packed, b2, (const _Float16*)zeros, n, h, w, st);
  }
  BlockArgs a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out;
  a.w1 = (const half8*)w1_packed; a.b1 = b1; a.w2 = (const half8*)w2_packed; a.b2 = b2;
  a.zeros = (const _Float16*)zeros;
  a.N = n; a.H = h; a.W = w;
  a.tiles_x = (w + BK::TW - 1) / BK::TW;
  a.tiles_y = (h + BK::TH - 1) / BK::TH;
  a.ntiles = n * a.tiles_x * a.tiles_y;
  static int cus_of[64] = {};
  const int dev_ = lfd_device_ordinal();
  int& cus = cus_of[dev_];
  if (!cus) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block64), hipFuncAttributeMaxDynamicSharedMemorySize,
                            BK::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1)
      return LFD_ERR_LAUNCH_FAILED;
    cus = c;
  }
  // one 512-thread workgroup per CU (LDS: 159.5 KB each); small launches: one workgroup per tile of every XCD's range
  int blocks = 8 * ((a.ntiles + 7) / 8) < cus ? 8 * ((a.ntiles + 7) / 8) : cus;
  hipLaunchKernelGGL(k_block64, dim3(blocks), dim3(512), BK::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
