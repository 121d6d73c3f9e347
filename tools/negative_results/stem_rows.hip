// csrc/stem_rows.hip -- the ROW-STREAMING form of lfd_stem_faster_fused_f16 for NHWC fp16 frames:
//   conv3x3 s2 (3->64)+BN+ReLU -> conv1x1+BN+ReLU -> conv3x3 s2 (64->64)+BN+ReLU -> conv1x1+BN+ReLU
// (reference lfd/model/backbone/lfd_resnet.py:376-413), the same operator as stem_fused.hip's k_stem2x.
//
// k_stem2x runs ONE wave per SIMD (all four filters in 512 registers): its two phases -- intermediate tile from the raw frame
// (VALU-heavy: gathers, packs, LDS writes), then the 3x3 s2 contraction -- alternate inside that one wave and nothing overlaps
// them: SQ MFMA-pipe busy 0.46, VALU active 0.45, 0.33-0.37 of the MFMA peak.  Here the two phases belong to DIFFERENT waves of
// the same SIMD (the producer / consumer split of block.hip, down.hip) and rows stream through LDS rings (down.hip):
//   * a 512-thread workgroup owns a strip of 31 output columns (one MFMA pixel tile per output row, lane 31 idle) and walks down
//     a segment of output rows, two per step;
//   * waves 0-3 PRODUCE four rows of the 64-channel stride-2 intermediate per step (4 x 64 pixels = 8 MFMA tiles, two per wave): conv 1 as two K = 16 MFMAs per 32-channel slab on a K layout of 3 rows x (1 junk + 9 values) -- each lane's
//     fragments are EIGHT aligned dword reads from the raw ring, no packing arithmetic --, ReLU -> fp16 in registers, which
//     ARE the B fragments of the 1x1 under a K permutation of its filter (gathered once per wave from the standard pack),
//     ReLU -> fp16 -> the intermediate ring in the de-interleaved, XOR-swizzled layout of the stride-2 contraction;
//   * waves 4-7 CONSUME: conv 3 (3x3 s2, 36 k-steps, filter slab register-stationary) for output rows (2j, 2j+1) x two slabs
//     two steps behind, ReLU -> fp16 -> a small LDS tile; one step later a PRODUCER wave runs the chained 1x1 on the full 64
//     channels of a row -> ReLU -> fp16 -> stores (the consumers are the longer chain of a step).  The consumers also issue the
//     LDS-DMA of the raw rows (one 16-byte-per-lane instruction per row, two rows per wave and step, counted vmcnt);
//   * one s_barrier per step.
// STATUS (round 3): within one fp16 ulp of k_stem2x and of the two-kernel stem on every tested frame.  A strip is 31 output columns
// = 64 intermediate columns = eight producer tiles per step, two per producer wave (a first version with 32 / 65 columns had nine:
// the wave with three was the critical path of a 5600-cycle step; now ~4500, producers 3700, consumers 4300), 256 workgroups at
// 8 x 1080p.  Against k_stem2x (tools/timing/stem_rows_time.py): -7 % at 1 x 1080p, -11 % at 2 x 1080p, equal at 8 x 1080p, where
// both run at the power limit -- opt-in (LFD_STEM_ROWS=1, stem_fused.hip): the gain at batch 1 is 2 us of a 220 us forward.  Tried without effect: two producer
// tiles in lock step (registers), biases in registers / as the first MFMA's C operand, s_setprio on the producers.
// Requirements (else lfd_stem_faster_fused_f16 keeps k_stem2x): 64 channels, NHWC fp16 frame, 16-byte aligned base, W % 4 == 0 (raw
// rows 8-byte aligned).  Rounding points as in k_stem2x (fp16 after every ReLU); summation order differs:
// the two agree to ~1 fp16 ulp (tests/test_gpu_conv.py).
#include "conv_impl.h"
#include <type_traits>

namespace {

struct SRArgs {
  const _Float16* in;    // [N, H, W, 3]
  _Float16* out;         // [N, H2, W2, 64]
  const half8* w1;       // [2][2][64]  standard stem pack (engine.pack_stem_weight)
  const float* b1;
  const half8* w2;       // [2][4][64]  standard 1x1 pack
  const float* b2;
  const half8* w3;       // [2][36][64]
  const float* b3;
  const half8* w4;       // [2][4][64]
  const float* b4;
  int N, H, W, H1, W1, H2, W2;
  int strips, segs, SH, nwork;
};

struct SR {
  static constexpr int TW = 31;                        // output columns of a strip (one MFMA pixel tile, lane 31 idle)
  static constexpr int IW = 64;                        // intermediate columns: 2 ox0 - 1 .. 2 ox0 + 62
  static constexpr int IWh = 33, IWs = 66;             // ring row layout of down.hip (65 columns wide: column 64 unused)
  static constexpr int IN_ROWB = IWs * 128;            // 8448: intermediate ring row (layout of down.hip's input ring)
  static constexpr int NI = 10;                        // intermediate ring rows (9 live)
  static constexpr int RAW_LANES = 49;                 // raw columns 4 ox0 - 4 .. 4 ox0 + 126 (784 bytes = 49 lanes x 16)
  static constexpr int RAW_ROWB = 816;
  static constexpr int NRAW = 32;                      // raw ring rows
  static constexpr int PIXB = 144;
  static constexpr int Y3_ROWB = 32 * PIXB;            // 4608 (all 32 lanes of an MFMA tile write, the 32nd column is not an output)
  static constexpr int NK = 36;
  static constexpr int OFF_I = 0;
  static constexpr int OFF_RAW = OFF_I + NI * IN_ROWB;                 // 84480
  static constexpr int OFF_Y3 = OFF_RAW + NRAW * RAW_ROWB + 1024;      // (+ dummy DMA window / zero words)
  static constexpr int OFF_BIAS = OFF_Y3 + 2 * 2 * Y3_ROWB;            // [step parity][row][32 px][144]
  static constexpr int LDS_BYTES = OFF_BIAS + 4 * 64 * 4;
  static constexpr int OFF_DUMMY = OFF_RAW + NRAW * RAW_ROWB;          // 1024 bytes: filler DMA target (first 16: kept zero? no: see OFF_ZERO)
  static constexpr int M_STEP = 4 * IW;                // 256 intermediate pixels per step = 8 tiles
};
static_assert(SR::LDS_BYTES <= 160 * 1024, "LDS capacity");

#ifdef LFD_SROWS_TIMING
__device__ unsigned long long g_sr_dbg[2 * 16 * 8];
#define ST(role, i) do { if (blockIdx.x == 8 && (threadIdx.x & 255) == 0 && s >= 8 && s < 24) { g_sr_dbg[((role) * 16 + s - 8) * 8 + (i)] = __builtin_readcyclecounter(); \
    if ((i) == 0) g_sr_dbg[((role) * 16 + s - 8) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define ST(role, i)
#endif

struct SSeg { int n, oy0, ox0, rows, JC, T; };

__device__ __forceinline__ int swrap(int v, int m) { return v >= m ? v - m : v; }
template <int N>
__device__ __forceinline__ void sr_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------ producer (conv 1 + 1x1)
__device__ __forceinline__ void sr_producer(const SRArgs& a, char* smem, int pw, const SSeg& sg) {
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + SR::OFF_BIAS);

  // ---- conv 1 filter in the K layout of the raw window: K = 10 row + slot, slot 0 = junk (weight 0), slot 1 + e = element
  // e = 3 s + c of tap row `row`; K 30, 31 = 0.  Gathered from the standard pack: step0 = {row0 e0..7 | row1 e0..7},
  // step1 = {row2 e0..7 | row0 e8, row1 e8, row2 e8, 0 x 5}  (engine.pack_stem_weight).
  half8 w1f[2][2];
  {
    const _Float16* w1s = reinterpret_cast<const _Float16*>(a.w1);
    const int m = lane & 31, hA = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int K = 16 * ks + 8 * hA + i;
          const int row = K / 10, slot = K - 10 * row;
          _Float16 v = (_Float16)0.f;
          if (K < 30 && slot > 0) {
            const int e = slot - 1;
            // std index: ((ct * 2 + step) * 64 + half * 32 + m) * 8 + j
            const int step = e < 8 ? (row == 2) : 1, half = e < 8 ? (row == 1) : 1, j = e < 8 ? e : row;
            v = w1s[((ct * 2 + step) * 64 + half * 32 + m) * 8 + j];
          }
          w1f[ct][ks][i] = v;
        }
  }
  // ---- 1x1 filter under the K permutation that makes conv 1's accumulators its B fragments: k-step (ct, j) of output slab so,
  // lane (m, hA): elements 0..3 = W2[32 so + m][32 ct + 16 j + 4 hA + 0..3], 4..7 = W2[..][32 ct + 16 j + 8 + 4 hA + 0..3]
  half8 w2f[2][4];
  {
    const _Float16* w2s = reinterpret_cast<const _Float16*>(a.w2);
    const int m = lane & 31, hA = lane >> 5;
#pragma unroll
    for (int so = 0; so < 2; ++so)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const half4 lo = *reinterpret_cast<const half4*>(w2s + ((so * 4 + kk) * 64 + m) * 8 + 4 * hA);
        const half4 hi = *reinterpret_cast<const half4*>(w2s + ((so * 4 + kk) * 64 + 32 + m) * 8 + 4 * hA);
#pragma unroll
        for (int i = 0; i < 4; ++i) { w2f[so][kk][i] = lo[i]; w2f[so][kk][4 + i] = hi[i]; }
      }
  }

  // ---- the eight dword reads of a pixel's window: (tap row, dword) per k-step half; index 3 = the zero word
  //   h0: k-step 0 = row0 d0..3             k-step 1 = row1 d3, d4, row2 d0, d1
  //   h1: k-step 0 = row0 d4, row1 d0..2    k-step 1 = row2 d2, d3, d4, ZERO
  int rsel[8], dcst[8];
  {
    const int r0[8] = {0, 0, 0, 0, 1, 1, 2, 2}, d0[8] = {0, 1, 2, 3, 3, 4, 0, 1};
    const int r1[8] = {0, 1, 1, 1, 2, 2, 2, 3}, d1[8] = {4, 0, 1, 2, 2, 3, 4, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) { rsel[i] = h ? r1[i] : r0[i]; dcst[i] = 4 * (h ? d1[i] : d0[i]); }
  }
  // this wave's tiles of the step's 260 intermediate pixels: pw, pw + 4 and (wave 0) 8
  const int ntile = 2;
  int t_irow[3], t_icol[3], t_woff[3];
  bool t_ok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int tt = k < 2 ? pw + 4 * k : 8;
    const int m = 32 * tt + pix;
    t_ok[k] = m < SR::M_STEP;
    const int mm = t_ok[k] ? m : 0;
    t_irow[k] = mm / SR::IW;
    t_icol[k] = mm - t_irow[k] * SR::IW;
    const int rem = (t_icol[k] & 1) * SR::IWh + (t_icol[k] >> 1);
    t_woff[k] = rem * 128 + 8 * h;         // + ((chunk ^ f) << 4) at the write
  }
  const int zero_addr = SR::OFF_DUMMY + 512;            // a dword that is never written: zeroed below
  if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(smem + zero_addr) = 0u;

  // ---- the chained 1x1 of conv 3 (moved here from the consumers: they are the longer chain of a step): wave pw finishes
  // output row rc = pw >> 1, channel slab ct = pw & 1 of the rows the consumers contracted one step ago
  const int tct = pw & 1, trc = pw >> 1;
  half8 w4r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w4r[k] = a.w4[((size_t)tct * 4 + k) * 64 + lane];
  const int ox = sg.ox0 + pix;
  const bool colok = pix < SR::TW && ox < a.W2;
  _Float16* obase = a.out + (((size_t)sg.n * a.H2 + sg.oy0) * a.W2 + (colok ? ox : 0)) * 64 + tct * 32 + 4 * h;
  const int y3r = trc * SR::Y3_ROWB + pix * SR::PIXB + h * 16;

  int s_raw = swrap(SR::NRAW - 6, SR::NRAW);            // ring slot of raw row 8 s - 6
  int s_int = SR::NI - 3;                               // ring slot of intermediate row 4 s - 3

  // the biases of conv 1 and the 1x1 in the accumulator layout, in REGISTERS: as float4 reads from the LDS table they were 16 KB of
  // LDS traffic per tile -- as much as the consumers' B fragments
  f32x16 bias1[2], bias2[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bias1[ct][4 * g + e] = a.b1[ct * 32 + 8 * g + 4 * h + e];
        bias2[ct][4 * g + e] = a.b2[ct * 32 + 8 * g + 4 * h + e];
      }

  // NT tiles (k0 .. k0 + NT - 1 of this wave) in lock step: their dependent chains (window reads -> 2 MFMAs -> pack -> 4 MFMAs ->
  // pack -> ring writes, ~1800 cycles per tile when run one after the other beside a contracting consumer) interleave
  auto tiles = [&](auto ntag, int k0, int s) {
    constexpr int NT = decltype(ntag)::value;
    uint32_t wd[NT][8];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int k = k0 + u;
      int radr[4];
#pragma unroll
      for (int r = 0; r < 3; ++r) radr[r] = SR::OFF_RAW + swrap(swrap(s_raw + 2 * t_irow[k] + r, SR::NRAW), SR::NRAW) * SR::RAW_ROWB + 4 + 12 * t_icol[k];
      radr[3] = zero_addr;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int base = rsel[i] == 0 ? radr[0] : (rsel[i] == 1 ? radr[1] : (rsel[i] == 2 ? radr[2] : radr[3]));
        wd[u][i] = *reinterpret_cast<const uint32_t*>(smem + base + (rsel[i] == 3 ? 0 : dcst[i]));
      }
    }
    // ---- conv 1 (both slabs) -> ReLU -> fp16: the 1x1's B fragments
    uint32_t y1[NT][2][8];
    {
      f32x16 acc[NT][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          union { uint32_t w[4]; half8 v; } c;
#pragma unroll
          for (int i = 0; i < 4; ++i) c.w[i] = wd[u][4 * ks + i];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)      // (the bias registers are the C operand of the first MFMA: no copies)
            acc[u][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[ct][ks], c.v, ks == 0 ? bias1[ct] : acc[u][ct], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 8; ++q) y1[u][ct][q] = lfd_cvt_pk_max(acc[u][ct][2 * q], acc[u][ct][2 * q + 1], LFD_PK_RELU);
    }
    // ---- 1x1 (both output slabs) -> ReLU -> fp16 -> intermediate ring
    f32x16 acc2[NT][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        union { uint32_t w[4]; half8 v; } bb;
#pragma unroll
        for (int i = 0; i < 4; ++i) bb.w[i] = y1[u][kk >> 1][4 * (kk & 1) + i];
#pragma unroll
        for (int so = 0; so < 2; ++so)
          acc2[u][so] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[so][kk], bb.v, kk == 0 ? bias2[so] : acc2[u][so], 0, 0, 0);
      }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int k = k0 + u;
      const int il = 4 * s - 3 + t_irow[k];
      const int gi = 2 * sg.oy0 - 1 + il, gx = 2 * sg.ox0 - 1 + t_icol[k];
      const bool inimg = gi >= 0 && gi < a.H1 && gx >= 0 && gx < a.W1;
      const bool wr = t_ok[k] && il >= 0;
      const int rem = (t_icol[k] & 1) * SR::IWh + (t_icol[k] >> 1);
      const int f = (rem >> 1) & 7;
      char* dst = smem + SR::OFF_I + swrap(swrap(s_int + t_irow[k], SR::NI), SR::NI) * SR::IN_ROWB + t_woff[k];
#pragma unroll
      for (int so = 0; so < 2; ++so)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = lfd_cvt_pk_max(acc2[u][so][4 * g + 0], acc2[u][so][4 * g + 1], LFD_PK_RELU);
          v.y = lfd_cvt_pk_max(acc2[u][so][4 * g + 2], acc2[u][so][4 * g + 3], LFD_PK_RELU);
          if (!inimg) { v.x = 0u; v.y = 0u; }
          if (wr) *reinterpret_cast<uint2*>(dst + (((4 * so + g) ^ f) << 4)) = v;
        }
    }
  };

  __builtin_amdgcn_s_barrier();                         // biases + zero word visible
  for (int s = 0; s < sg.T; ++s) {
    ST(0, 0);
    block_barrier();       // the consumers awaited the raw rows of this step before they arrived here
    ST(0, 1);
    // ---- tail of C-step j = s - 3 (y3 tile written in step s - 1)
    {
      const int j = s - 3;
      const int ol = 2 * j + trc;
      if (j >= 0 && ol < sg.rows) {
        f32x16 acc;
        const float* bp = sbias + 192 + tct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        }
        const char* y3 = smem + SR::OFF_Y3 + (j & 1) * 2 * SR::Y3_ROWB + y3r;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4r[q], *reinterpret_cast<const half8*>(y3 + q * 32), acc, 0, 0, 0);
        _Float16* o = obase + (size_t)ol * a.W2 * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = lfd_cvt_pk_max(acc[4 * g + 0], acc[4 * g + 1], LFD_PK_RELU);
          v.y = lfd_cvt_pk_max(acc[4 * g + 2], acc[4 * g + 3], LFD_PK_RELU);
          if (colok) *reinterpret_cast<uint2*>(o + 8 * g) = v;
        }
      }
    }
    if (s <= sg.JC) {
#ifdef SR_PAIR_TILES      // (A/B: two tiles in lock step -- needs more registers than the wave has next to the biases: spills)
      tiles(std::integral_constant<int, 2>{}, 0, s);
      if (ntile == 3) tiles(std::integral_constant<int, 1>{}, 2, s);
#else
      tiles(std::integral_constant<int, 1>{}, 0, s);
      tiles(std::integral_constant<int, 1>{}, 1, s);
      if (ntile == 3) tiles(std::integral_constant<int, 1>{}, 2, s);
#endif
    }
    ST(0, 2);
    s_raw = swrap(s_raw + 8, SR::NRAW);
    s_int = swrap(s_int + 4, SR::NI);
  }
}

// ------------------------------------------------------------------------------------------------ consumer (conv 3 + 1x1)
// INTERIOR is a template parameter: with both raw-row paths in ONE loop the compiler waits (vmcnt 0) for the border path's register
// loads at the join behind the branch -- i.e. for the DMA just issued on the interior path (2800 cycles per step)
template <bool INTERIOR>
__device__ __forceinline__ void sr_consumer(const SRArgs& a, char* smem, int cw, const SSeg& sg) {
  const int lane = threadIdx.x & 63;
  const int ct = cw & 1, rc = cw >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + SR::OFF_BIAS);

  // ---- raw rows: batch b = raw rows 8 b - 5 .. 8 b + 2 (row 8 b - 6 came with batch b - 1), two rows per consumer wave.  A row of
  // the strip = raw columns 4 ox0 - 4 .. 4 ox0 + 126, 784 bytes = 49 chunks of 16 (8-byte aligned in memory).
  //   INTERIOR strips (every chunk inside the frame's row): one 16-byte-per-lane LDS-DMA per row, two steps ahead, counted vmcnt;
  //   a row above / below the frame is written as zeros by ds_write and still issues a DMA (into a dummy window) to keep the count.
  //   BORDER strips (the first: columns -4 .. -1 are conv 1's zero padding and end in the middle of a chunk; the last ones: the
  //   row ends inside the strip): the chunks go through registers -- requested in step s, written to LDS in step s + 1 -- with
  //   chunks that straddle the frame's edge assembled from their valid halves.  No DMA, no counted wait in these workgroups.
  const unsigned rowpitch = (unsigned)a.W * 6u;
  const char* img = reinterpret_cast<const char*>(a.in) + (long)sg.n * a.H * (long)rowpitch;
  const long b0 = ((long)4 * sg.ox0 - 4) * 6;                        // byte offset of the strip's first column in a raw row
  const long c0 = b0 + lane * 16;                                    // this lane's chunk
  const bool lane_in = lane < SR::RAW_LANES;
  const bool chunk_full = c0 >= 0 && c0 + 16 <= (long)rowpitch;
  const char* lane_src = img + (chunk_full ? c0 : 0);
  auto slot_of = [](int rl) { return (rl + 4 * SR::NRAW) & (SR::NRAW - 1); };
  auto row_valid = [&](int rl) { const int gy = 4 * sg.oy0 - 3 + rl; return rl >= 0 && gy >= 0 && gy < a.H; };
  auto issue_raw = [&](int rl) {                                     // interior strips
    char* ldst = smem + SR::OFF_RAW + slot_of(rl) * SR::RAW_ROWB;
    if (row_valid(rl)) {
      if (lane_in) dma16(lane_src + (unsigned long)rowpitch * (unsigned)(4 * sg.oy0 - 3 + rl), ldst);
    } else {
      if (lane_in) *reinterpret_cast<uint4*>(ldst + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
      if (lane < 16) dma16(img + lane * 16, smem + SR::OFF_DUMMY);     // any valid address: the data goes to the dummy window
    }
  };
  // border strips: this lane's chunk as two 8-byte pieces (rows and the strip's first byte are 8-byte aligned: a piece is inside the
  // frame's row or outside it, never across its edge).  load_chunk only LOADS (from a clamped address); zero-filling happens at
  // store time, one step later, from the positions alone -- any arithmetic on the loaded values here would put the load's latency
  // on this step's critical path.
  const bool v_lo = c0 >= 0 && c0 + 8 <= (long)rowpitch, v_hi = c0 + 8 >= 0 && c0 + 16 <= (long)rowpitch;
  const char* src_lo = img + (v_lo ? c0 : 0);
  const char* src_hi = img + (v_hi ? c0 + 8 : 0);
  auto load_chunk = [&](int rl) {
    const int gy = 4 * sg.oy0 - 3 + rl;
    const unsigned long ro = (unsigned long)rowpitch * (unsigned)(row_valid(rl) ? gy : 0);
    const uint2 lo = *reinterpret_cast<const uint2*>(src_lo + ro), hi = *reinterpret_cast<const uint2*>(src_hi + ro);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  auto store_chunk = [&](int rl, uint4 v) {
    const bool rv = row_valid(rl);
    if (!(rv && v_lo)) { v.x = 0u; v.y = 0u; }
    if (!(rv && v_hi)) { v.z = 0u; v.w = 0u; }
    if (lane_in) *reinterpret_cast<uint4*>(smem + SR::OFF_RAW + slot_of(rl) * SR::RAW_ROWB + lane * 16) = v;
  };
  const int myrow0 = -5 + 2 * cw;                                    // this wave's first row of batch 0
  uint4 pend[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  auto issue_batch = [&](int b) {
    if constexpr (INTERIOR) { issue_raw(8 * b + myrow0); issue_raw(8 * b + myrow0 + 1); }
  };
  if constexpr (!INTERIOR) {       // batches 0 and 1 straight into the ring
#pragma unroll
    for (int b = 0; b < 2; ++b) { store_chunk(8 * b + myrow0, load_chunk(8 * b + myrow0)); store_chunk(8 * b + myrow0 + 1, load_chunk(8 * b + myrow0 + 1)); }
  }
  issue_batch(0);
  issue_batch(1);

  // conv 3 slab + the chained 1x1's slab: requested after the prologue DMA, forced to have arrived here (see down.hip)
  half8 wreg[SR::NK];
  {
    const half8* wsrc = a.w3 + (size_t)ct * SR::NK * 64 + lane;
#pragma unroll
    for (int k = 0; k < SR::NK; ++k) wreg[k] = wsrc[(size_t)k * 64];
  }
#pragma unroll
  for (int k = 0; k < SR::NK; ++k) asm volatile("" : "+v"(wreg[k]));

  int xoff[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = 2 * pix + s;
    const int rem = (ix & 1) * SR::IWh + (ix >> 1);
    const int f = (rem >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) xoff[s][q] = rem * 128 + (((2 * q + h) ^ f) << 4);
  }
  const int y3w = rc * SR::Y3_ROWB + pix * SR::PIXB + ct * 64 + h * 8;     // this wave's writes into a y3 tile

  int s_int = rc * 2;                                   // ring slot of intermediate row 4 j + 2 rc (tap row 0), j = s - 2
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < sg.T; ++s) {
    ST(1, 0);
    if constexpr (INTERIOR) sr_wait_vmcnt<2>();    // batch s landed (batch s + 1, two instructions, may be in flight); no other VMEM in this loop
    ST(1, 1);
    block_barrier();
    ST(1, 2);
    if constexpr (INTERIOR) {
      issue_batch(s + 2);
    } else {
      // batch s + 1 (requested a step ago) -> ring; batch s + 2 -> registers.  (Step 0 writes nothing: batch 1 is in the ring.)
      if (s > 0) { store_chunk(8 * (s + 1) + myrow0, pend[0]); store_chunk(8 * (s + 1) + myrow0 + 1, pend[1]); }
      pend[0] = load_chunk(8 * (s + 2) + myrow0);
      pend[1] = load_chunk(8 * (s + 2) + myrow0 + 1);
    }
    ST(1, 3);
    ST(1, 4);
    // ---- conv 3 for output row 2 j + rc, j = s - 2: intermediate rows 4 j + 2 rc + r
    if (s >= 2) {
      const int j = s - 2;
      const int ol = 2 * j + rc;
      if (ol < sg.rows) {
        f32x16 acc;
        const float* bp = sbias + 128 + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        }
        const char* xin = smem + SR::OFF_I;
        int rb[3];
        rb[0] = s_int * SR::IN_ROWB;
        rb[1] = swrap(s_int + 1, SR::NI) * SR::IN_ROWB;
        rb[2] = swrap(s_int + 2, SR::NI) * SR::IN_ROWB;
        auto xfrag = [&](int k) {
          const int r = k / 12, sx = (k / 4) % 3, q = k % 4;
          return *reinterpret_cast<const half8*>(xin + rb[r] + xoff[sx][q]);
        };
        constexpr int PD = 3;
        half8 xq[PD + 1];
#pragma unroll
        for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
#pragma unroll
        for (int k = 0; k < SR::NK; ++k) {
          if (k + PD < SR::NK) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
          __builtin_amdgcn_sched_barrier(0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[k], xq[k % (PD + 1)], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        char* y3 = smem + SR::OFF_Y3 + (j & 1) * 2 * SR::Y3_ROWB + y3w;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = lfd_cvt_pk_max(acc[4 * g + 0], acc[4 * g + 1], LFD_PK_RELU);
          v.y = lfd_cvt_pk_max(acc[4 * g + 2], acc[4 * g + 3], LFD_PK_RELU);
          *reinterpret_cast<uint2*>(y3 + 16 * g) = v;
        }
      }
      s_int = swrap(s_int + 4, SR::NI);
    }
    ST(1, 5);
  }
}

__global__ __launch_bounds__(512) void k_stem_rows(SRArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x < 256) {
    float* sb = reinterpret_cast<float*>(smem + SR::OFF_BIAS);
    const int i = threadIdx.x & 63, w = threadIdx.x >> 6;
    sb[threadIdx.x] = w == 0 ? a.b1[i] : (w == 1 ? a.b2[i] : (w == 2 ? a.b3[i] : a.b4[i]));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.nwork + 7) / 8;
  const int w = xcd * per_xcd + bix;
  if (!(bix < per_xcd && w < a.nwork)) return;
  SSeg sg;
  {
    const int per_img = a.segs * a.strips;
    sg.n = w / per_img;
    const int r = w - sg.n * per_img;
    const int seg = r / a.strips, strip = r - seg * a.strips;
    sg.oy0 = seg * a.SH;
    sg.ox0 = strip * SR::TW;
    sg.rows = (a.H2 - sg.oy0) < a.SH ? (a.H2 - sg.oy0) : a.SH;
    sg.JC = (sg.rows + 1) / 2;       // consumer steps; the producers run steps 0 .. JC
    sg.T = sg.JC + 3;
  }
#ifdef SR_PRIO_P
  if (wave < 4) __builtin_amdgcn_s_setprio(2);
#endif
  if (wave < 4) sr_producer(a, smem, wave, sg);
  else {
    const long b0 = ((long)4 * sg.ox0 - 4) * 6, rowb = (long)a.W * 6;      // the strip's raw columns lie inside the frame's row?
    if (b0 >= 0 && b0 + SR::RAW_LANES * 16 <= rowb) sr_consumer<true>(a, smem, wave - 4, sg);
    else sr_consumer<false>(a, smem, wave - 4, sg);
  }
}

}  // namespace

#ifdef LFD_SROWS_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_srows_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sr_dbg), sizeof(unsigned long long) * 2 * 16 * 8);
}
#endif

// called by lfd_stem_faster_fused_f16 (stem_fused.hip); LFD_ERR_UNSUPPORTED = the shape does not suit this kernel
int lfd_stem_rows_launch(const void* in, void* out, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                         const float* b3, const void* w4, const float* b4, int n, int h, int w, hipStream_t st) {
  if ((w % 4) != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return LFD_ERR_UNSUPPORTED;      // raw rows 8-byte aligned
  SRArgs a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out;
  a.w1 = (const half8*)w1; a.b1 = b1; a.w2 = (const half8*)w2; a.b2 = b2; a.w3 = (const half8*)w3; a.b3 = b3; a.w4 = (const half8*)w4; a.b4 = b4;
  a.N = n; a.H = h; a.W = w;
  a.H1 = (h - 1) / 2 + 1; a.W1 = (w - 1) / 2 + 1;
  a.H2 = (a.H1 - 1) / 2 + 1; a.W2 = (a.W1 - 1) / 2 + 1;
  static int cus = 0;
  if (!cus) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                            SR::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1)
      return LFD_ERR_LAUNCH_FAILED;
    cus = c;
  }
  a.strips = (a.W2 + SR::TW - 1) / SR::TW;
  const long cols = (long)n * a.strips;
  int segs = (int)(cus / cols);
  if (segs < 1) segs = 1;
  int sh = (a.H2 + segs - 1) / segs;
  if (sh < 4) sh = 4;
  if (sh > a.H2) sh = a.H2;
  a.SH = sh;
  a.segs = (a.H2 + sh - 1) / sh;
  const long nwork = cols * a.segs;
  if (nwork > 0x3fffffffL) return LFD_ERR_UNSUPPORTED;
  a.nwork = (int)nwork;
  const int blocks = 8 * ((a.nwork + 7) / 8);
  hipLaunchKernelGGL(k_stem_rows, dim3(blocks), dim3(512), SR::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
