// csrc/block_rows.hip -- the ROW-STREAMING form of lfd_fasterblock_fused_f16 for large maps.
//
// Reference: FasterBlock.forward (lfd/model/backbone/lfd_resnet.py:96-154) without a downsample branch,
//        y = ReLU( BN2(conv3x3(ReLU(BN1(conv3x3(x))))) + x ),   64 -> 64 -> 64 channels,
// the same operator as block.hip's k_block64 (8 x 16 output tiles: conv1 covers the 10 x 18 halo tile as 6 x 32 linearly
// numbered pixels -- halo recompute 1.25 at 93.75 % MFMA slot use, i.e. 0.80 useful slots -- and re-reads x from L2 for
// the identity).  Here a workgroup owns a STRIP of 30 output columns and walks down a segment of rows (the scheme of
// down.hip, which needed it for its stride-2 halo):
//   * one MFMA pixel tile = one 32-pixel row of the strip: conv1 produces a mid row with no wasted slot, conv2 uses 30 of 32;
//     no vertical halo recompute inside a segment (2 extra mid rows per segment): 0.88-0.91 useful slots;
//   * input ring: 12 rows x 34 pixels of x, 144-byte pixel pitch (the DMA writes lane-linear: 7 pixels x 9 chunks per
//     instruction, the 9th chunk is the gap), fetched three steps ahead; the IDENTITY is read from the same ring two steps after
//     conv1 consumed the row -- x is read from HBM/L2 exactly once;
//   * mid ring: 6 rows x 32 (+2) pixels of ReLU(conv1) as fp16 (the rounding point of the two-launch path);
//   * 512 threads: waves 0-3 = conv1 for mid rows (2j, 2j+1) x two 32-channel slabs and the DMA issue (5 instructions per row,
//     10 per step, split 3/2/3/2 over the four waves, three steps ahead, counted vmcnt: every step issues a fixed number per
//     wave), waves 4-7 = conv2 + identity + stores for output rows (2j-4, 2j-3);
//   * one s_barrier per step of two rows; both filters register-stationary.
// Bit-identical to k_block64 and to the two-launch path (same k order, bias in the accumulator, one rounding of acc + x).
#include "conv_impl.h"

namespace {

struct RowsArgs {
  const _Float16* in;    // [N,H,W,64]
  _Float16* out;         // [N,H,W,64]
  const half8* w1;       // packed [2][36][64]
  const float* b1;
  const half8* w2;
  const float* b2;
  const _Float16* zeros; // 4 KB line: [0,2048) zero, [2048,4096) trash
  int N, H, W;
  int strips, segs, SH, nwork;
};

struct RB {
  static constexpr int TW = 30;                       // output columns of a strip
  static constexpr int MW = 32;                       // mid columns = one MFMA pixel tile
  static constexpr int IW = MW + 2;                   // 34 input columns
  static constexpr int PIXB = 144;                    // 128 + 16: conflict-free ds_read_b128 over 32 consecutive pixels
  static constexpr int IN_ROWB = IW * PIXB;           // 4896
  static constexpr int NDMA = (IW + 6) / 7;           // 5 instructions of 7 pixels per row
  static constexpr int NIN = 12;                      // input ring: rows 2j-2 .. 2j+3 live in step j, 2j+4 .. 2j+9 in flight
  static constexpr int MID_ROWB = (MW + 2) * PIXB;    // 4896 (conv2's lanes 30, 31 read two columns past the row)
  static constexpr int NMID = 6;
  static constexpr int NK = 36;
  static constexpr int OFF_IN = 0;
  static constexpr int OFF_MID = OFF_IN + NIN * IN_ROWB + 256;    // (+ slack: the last DMA window of a row is 1008 B from byte 4032)
  static constexpr int OFF_BIAS = OFF_MID + NMID * MID_ROWB;
  static constexpr int LDS_BYTES = OFF_BIAS + 2 * 64 * 4;
};
static_assert(RB::LDS_BYTES <= 160 * 1024, "LDS capacity");

#ifdef LFD_ROWS_TIMING
__device__ unsigned long long g_rows_dbg[2 * 16 * 8];
#define RT(role, i) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && j < 16) { g_rows_dbg[((role) * 16 + j) * 8 + (i)] = __builtin_readcyclecounter(); \
    if ((i) == 0) g_rows_dbg[((role) * 16 + j) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define RT(role, i)
#endif

struct RSeg { int n, oy0, ox0, rows, J, T; };

__device__ __forceinline__ int rwrap(int v, int m) { return v >= m ? v - m : v; }
template <int N>
__device__ __forceinline__ void rows_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// scalar base + per-lane 32-bit offset + immediate: the five instructions of a row differ only in the immediate.  The
// instruction offset of an LDS-DMA load is added to the LDS address as well as to the memory address (as for MUBUF with
// LDS = 1): M0 carries the destination minus the immediate.
template <int IMM>
__device__ __forceinline__ void dma16_si(unsigned voff, const void* sbase, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base) - (unsigned)IMM;
  const unsigned long long b = (unsigned long long)(uintptr_t)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sb), "s"(m0v), "n"(IMM) : "memory");
}

// Instruction `seg` of input row il (il < 0: a filler row of zeros -- keeps the per-step instruction count constant): lane L
// carries chunk L % 9 of pixel 7 seg + L / 9 (chunk 8 = the pitch gap and lane 63 are masked off); columns / rows outside the
// image come from the zero line (conv1's zero padding).
struct RowsDma {
  unsigned voff;        // (L / 9) * 128 + (L % 9) * 16: byte offset of this lane's chunk from the first pixel of its instruction
  bool lane_on;         // L % 9 < 8 && L < 63
  int px;               // L / 9
  unsigned full;        // bit seg: every pixel of instruction seg lies inside the image's columns
};
__device__ __forceinline__ void rows_dma_init(const RowsArgs& a, const RSeg& sg, RowsDma& d) {
  const int ol = threadIdx.x & 63;
  d.px = (ol * 57) >> 9;                    // ol / 9 for ol < 64
  const int ck = ol - 9 * d.px;
  d.voff = (unsigned)(d.px * 128 + (ck & 7) * 16);
  d.lane_on = ck < 8 && ol < 63;
  d.full = 0u;
  const int gx0 = sg.ox0 - 2;
#pragma unroll
  for (int seg = 0; seg < RB::NDMA; ++seg) {
    const int c0 = 7 * seg, c1 = (7 * seg + 6 < RB::IW - 1) ? 7 * seg + 6 : RB::IW - 1;
    if (gx0 + c0 >= 0 && gx0 + c1 < a.W) d.full |= 1u << seg;
  }
}
template <int SEG>
__device__ __forceinline__ void rows_issue_one(const RowsArgs& a, char* smem, const RSeg& sg, const RowsDma& d, int il, int slot) {
  const int gy = sg.oy0 - 2 + il;                               // wave-uniform
  const long rowpitch = (long)a.W * 128;
  const bool rv = il >= 0 && gy >= 0 && gy < a.H;
  const char* rowp = reinterpret_cast<const char*>(a.in) + ((long)sg.n * a.H + (rv ? gy : 0)) * rowpitch + (long)(sg.ox0 - 2) * 128;
  const char* zp = reinterpret_cast<const char*>(a.zeros);
  char* ldst = smem + RB::OFF_IN + slot * RB::IN_ROWB + SEG * 1008;
  const int col = 7 * SEG + d.px;
  const bool act = d.lane_on && col < RB::IW;
  if (!rv) {
    if (act) dma16_si<0>(d.voff & 127u, zp, ldst);              // chunk offset inside the zero line
  } else if ((d.full >> SEG) & 1u) {
    if (act) dma16_si<SEG * 7 * 128>(d.voff, rowp, ldst);
  } else {
    const int gx = sg.ox0 - 2 + col;
    const char* src = (gx >= 0 && gx < a.W) ? rowp + SEG * 7 * 128 + d.voff : zp + (d.voff & 127u);
    if (act) dma16(src, ldst);
  }
}
// wave cw's share of a batch (two rows): row cw >> 1, instructions 0-2 (even cw) or 3-4 (odd cw)
__device__ __forceinline__ void rows_issue_share(const RowsArgs& a, char* smem, const RSeg& sg, const RowsDma& d, int cw, int il, int slot) {
  if (cw & 1) {
    rows_issue_one<3>(a, smem, sg, d, il, slot);
    rows_issue_one<4>(a, smem, sg, d, il, slot);
  } else {
    rows_issue_one<0>(a, smem, sg, d, il, slot);
    rows_issue_one<1>(a, smem, sg, d, il, slot);
    rows_issue_one<2>(a, smem, sg, d, il, slot);
  }
}

// ------------------------------------------------------------------------------------------------ producer (conv1)
template <int NC>      // this wave's DMA instructions per batch (3 or 2)
__device__ __forceinline__ void rows_producer(const RowsArgs& a, char* smem, int pw, const RSeg& sg) {
  const int lane = threadIdx.x & 63;
  const int ct = pw & 1, rp = pw >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + RB::OFF_BIAS);

  // ---- DMA: batch b = input rows 2b + 2, 2b + 3 (the rows step b adds to conv1's window); rows 0, 1 and batches 0 .. 2 in
  // the prologue, batch j + 3 at the END of step j -- in the 1700 cycles this wave waited for the consumers when THEY issued it
  // (stamps: consumer chain = 800 issue + 1800 contraction + 400 epilogue, producer chain = 1400 + 360).  Row (pw >> 1) of a
  // batch, instructions 0-2 (even pw) or 3-4 (odd pw); three steps of lead: the loaded HBM latency is about two (down.hip).
  RowsDma dm;
  rows_dma_init(a, sg, dm);
  const int myrow = pw >> 1;
  rows_issue_share(a, smem, sg, dm, pw, myrow, myrow);
  rows_issue_share(a, smem, sg, dm, pw, 2 + myrow, 2 + myrow);
  rows_issue_share(a, smem, sg, dm, pw, 1 < sg.J ? 4 + myrow : -1, 4 + myrow);
  rows_issue_share(a, smem, sg, dm, pw, 2 < sg.J ? 6 + myrow : -1, 6 + myrow);
  int s_dma = rwrap(8 + myrow, RB::NIN);            // batch j + 3's row for this wave: 2 (j + 3) + 2 + myrow

  // the conv1 slab: requested after the prologue DMA (HBM rows and L2 filter travel together) and forced to have ARRIVED here,
  // which drains that DMA too (needed in step 0 anyway); see down.hip for why the compiler must not place this wait itself
  half8 wreg[RB::NK];
  {
    const half8* wsrc = a.w1 + (size_t)ct * RB::NK * 64 + lane;
#pragma unroll
    for (int k = 0; k < RB::NK; ++k) wreg[k] = wsrc[(size_t)k * 64];
  }
#pragma unroll
  for (int k = 0; k < RB::NK; ++k) asm volatile("" : "+v"(wreg[k]));
  const int xbase = pix * RB::PIXB + h * 16;        // mid column pix, tap s reads input column pix + s: immediates
  const int mx = sg.ox0 - 1 + pix;
  const bool colin = mx >= 0 && mx < a.W;
  const int mwoff = pix * RB::PIXB + ct * 64 + h * 8;

  int s_in = rwrap(rp, RB::NIN), s_mid = rp;        // input row 2j + rp (tap row 0), mid row 2j + rp
  __builtin_amdgcn_s_barrier();                     // biases visible
  for (int j = 0; j < sg.T; ++j) {
    RT(0, 0);
    rows_wait_vmcnt<2 * NC>();     // this wave's share of batch j landed (batches j + 1, j + 2 may be in flight)
    RT(0, 1);
    block_barrier();
    RT(0, 2);
#ifdef RB_DMA_AT_TOP
    // (A/B) batch j + 3 FIRST: while this wave is blocked on its DMA instructions the consumer of the SIMD has the matrix pipe to
    // itself, and this wave's longer tail (epilogue into the mid ring) no longer follows a contraction that started late
    rows_issue_share(a, smem, sg, dm, pw, j + 3 < sg.J ? 2 * (j + 3) + 2 + myrow : -1, s_dma);     // (filler rows keep the count)
    s_dma = rwrap(s_dma + 2, RB::NIN);
    RT(0, 3);
    if (j >= sg.J) continue;
#else
    if (j >= sg.J) {
      rows_issue_share(a, smem, sg, dm, pw, -1, s_dma);
      s_dma = rwrap(s_dma + 2, RB::NIN);
      continue;
    }
#endif
    f32x16 acc;
    {
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
        acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
      }
    }
    const char* xin = smem + RB::OFF_IN + xbase;
    int rb[3];
    rb[0] = s_in * RB::IN_ROWB;
    rb[1] = rwrap(s_in + 1, RB::NIN) * RB::IN_ROWB;
    rb[2] = rwrap(s_in + 2, RB::NIN) * RB::IN_ROWB;
    auto xfrag = [&](int k) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return *reinterpret_cast<const half8*>(xin + rb[r] + (s * RB::PIXB + q * 32));
    };
    constexpr int PD = 3;
    half8 xq[PD + 1];
#pragma unroll
    for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
#pragma unroll
    for (int k = 0; k < RB::NK; ++k) {
#ifdef RB_PROBE_HALF_LDS
      if (k + PD < RB::NK) xq[(k + PD) % (PD + 1)] = ((k + PD) & 1) ? xq[(k + PD - 1) % (PD + 1)] : xfrag(k + PD);
#else
      if (k + PD < RB::NK) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
#endif
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[k], xq[k % (PD + 1)], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    RT(0, 4);
    // ---- epilogue: ReLU -> fp16 -> mid ring; mid pixels outside the map are conv2's zero padding
    const int my = sg.oy0 - 1 + 2 * j + rp;
    const bool inimg = colin && my >= 0 && my < a.H;
    char* mid = smem + RB::OFF_MID + s_mid * RB::MID_ROWB + mwoff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 v;
      v.x = lfd_cvt_pk_max(acc[4 * g + 0], acc[4 * g + 1], LFD_PK_RELU);
      v.y = lfd_cvt_pk_max(acc[4 * g + 2], acc[4 * g + 3], LFD_PK_RELU);
      if (!inimg) { v.x = 0u; v.y = 0u; }
      *reinterpret_cast<uint2*>(mid + 16 * g) = v;
    }
    s_in = rwrap(s_in + 2, RB::NIN);
    s_mid = rwrap(s_mid + 2, RB::NMID);
    RT(0, 5);
#ifndef RB_DMA_AT_TOP
    rows_issue_share(a, smem, sg, dm, pw, j + 3 < sg.J ? 2 * (j + 3) + 2 + myrow : -1, s_dma);
    s_dma = rwrap(s_dma + 2, RB::NIN);
#endif
  }
}

// ------------------------------------------------------------------------------------------------ consumer (conv2 + identity)
__device__ __forceinline__ void rows_consumer(const RowsArgs& a, char* smem, int cw, const RSeg& sg) {
  const int lane = threadIdx.x & 63;
  const int ct = cw & 1, rc = cw >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + RB::OFF_BIAS) + 64;
  const int cbase = pix * RB::PIXB + h * 16;
  const int ox = sg.ox0 + pix;
  const bool colok = pix < RB::TW && ox < a.W;
  const int idoff = ((pix < RB::TW ? pix : 0) + 2) * RB::PIXB + ct * 64 + h * 8;      // x[oy, ox] = input column pix + 2
  _Float16* obase = a.out + (((size_t)sg.n * a.H + sg.oy0) * a.W + (colok ? ox : 0)) * 64 + ct * 32 + 4 * h;

  half8 wreg[RB::NK];
  {
    const half8* wsrc = a.w2 + (size_t)ct * RB::NK * 64 + lane;
#pragma unroll
    for (int k = 0; k < RB::NK; ++k) wreg[k] = wsrc[(size_t)k * 64];
  }

  // output row ol = 2 (j - 2) + rc reads mid rows ol .. ol + 2 and, for the identity, input row ol + 2
  int s_mid = rc, s_idr = rwrap(rc + 2, RB::NIN);
  // (Tried: finishing the row of step j BETWEEN the MFMAs of step j + 1 -- a software-pipelined epilogue.  The contraction grew
  //  by exactly the epilogue's length, 3480 -> 3700 cycles per step: the issue slots between dependent MFMAs are not free when
  //  the other wave of the SIMD contracts as well.)
#ifdef RB_EPI_DEFER
  // (A/B) DEFERRED EPILOGUE: the row contracted in step j is finished at the START of step j + 1, while the producer of this
  // SIMD contracts.  Measured with -DRB_DMA_AT_TOP / -DRB_EPI_DEFER and the stamps (tools/probe_rows.py): the step is 3480-3800
  // CYCLES depending on the arrangement (the deferred epilogue takes 1700 cycles beside a contracting partner instead of 400),
  // but 1.85-1.95 us of WALL time in every one of them -- the clock moves between 1.98 and 2.15 GHz to the same power.  What
  // this kernel gains over the 8 x 16 tiles (7-19 %) is the work it does not do: 0.89 instead of 0.80 useful MFMA slots and B
  // fragments, no second read of x.
  f32x16 accp;
  half4 idvp[4];
  _Float16* op = obase;
  bool pend = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) accp[r] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) idvp[g] = half4{0, 0, 0, 0};
  auto finish_all = [&]() {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float x0 = accp[4 * g + 0] + (float)idvp[g][0], x1 = accp[4 * g + 1] + (float)idvp[g][1];
      const float x2 = accp[4 * g + 2] + (float)idvp[g][2], x3 = accp[4 * g + 3] + (float)idvp[g][3];
      uint2 v;
      v.x = lfd_cvt_pk_max(x0, x1, LFD_PK_RELU);
      v.y = lfd_cvt_pk_max(x2, x3, LFD_PK_RELU);
      if (colok) *reinterpret_cast<uint2*>(op + 8 * g) = v;
    }
  };
#endif
  __builtin_amdgcn_s_barrier();
  for (int j = 0; j < sg.T; ++j) {
    RT(1, 0);
    block_barrier();       // the producers awaited the DMA of this step's rows (identity: two steps old) before they arrived here
    RT(1, 2);
#ifdef RB_EPI_DEFER
    if (pend) finish_all();
    pend = false;
    RT(1, 3);
#endif
    const int ol = 2 * (j - 2) + rc;
    if (j < 2) continue;
    if (ol < sg.rows) {
      f32x16 acc;
      {
        const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
          acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        }
      }
      const char* mid = smem + RB::OFF_MID + cbase;
      int rb[3];
      rb[0] = s_mid * RB::MID_ROWB;
      rb[1] = rwrap(s_mid + 1, RB::NMID) * RB::MID_ROWB;
      rb[2] = rwrap(s_mid + 2, RB::NMID) * RB::MID_ROWB;
      auto xfrag = [&](int k) {
        const int r = k / 12, s = (k / 4) % 3, q = k % 4;
        return *reinterpret_cast<const half8*>(mid + rb[r] + (s * RB::PIXB + q * 32));
      };
      constexpr int PD = 3;
      half8 xq[PD + 1];
#pragma unroll
      for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
      half4 idv[4];
      const char* idb = smem + RB::OFF_IN + s_idr * RB::IN_ROWB + idoff;
#pragma unroll
      for (int k = 0; k < RB::NK; ++k) {
#ifdef RB_PROBE_HALF_LDS
        if (k + PD < RB::NK) xq[(k + PD) % (PD + 1)] = ((k + PD) & 1) ? xq[(k + PD - 1) % (PD + 1)] : xfrag(k + PD);
#else
        if (k + PD < RB::NK) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
#endif
        if (k == 30) {
#pragma unroll
          for (int g = 0; g < 4; ++g) idv[g] = *reinterpret_cast<const half4*>(idb + 16 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[k], xq[k % (PD + 1)], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      RT(1, 4);
#ifdef RB_EPI_DEFER
      accp = acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) idvp[g] = idv[g];
      op = obase + (size_t)ol * a.W * 64;
      pend = true;
#else
      _Float16* o = obase + (size_t)ol * a.W * 64;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float x0 = acc[4 * g + 0] + (float)idv[g][0], x1 = acc[4 * g + 1] + (float)idv[g][1];
        const float x2 = acc[4 * g + 2] + (float)idv[g][2], x3 = acc[4 * g + 3] + (float)idv[g][3];
        uint2 v;
        v.x = lfd_cvt_pk_max(x0, x1, LFD_PK_RELU);
        v.y = lfd_cvt_pk_max(x2, x3, LFD_PK_RELU);
        if (colok) *reinterpret_cast<uint2*>(o + 8 * g) = v;
      }
#endif
    }
    s_mid = rwrap(s_mid + 2, RB::NMID);
    s_idr = rwrap(s_idr + 2, RB::NIN);
    RT(1, 5);
  }
#ifdef RB_EPI_DEFER
  if (pend) finish_all();
#endif
}

__global__ __launch_bounds__(512) void k_block64_rows(RowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x < 128) {
    float* sb = reinterpret_cast<float*>(smem + RB::OFF_BIAS);
    sb[threadIdx.x] = threadIdx.x < 64 ? a.b1[threadIdx.x] : a.b2[threadIdx.x - 64];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.nwork + 7) / 8;
  const int w = xcd * per_xcd + bix;
  if (!(bix < per_xcd && w < a.nwork)) return;      // (the whole workgroup)
  RSeg sg;
  {
    const int per_img = a.segs * a.strips;
    sg.n = w / per_img;
    const int r = w - sg.n * per_img;
    const int seg = r / a.strips, strip = r - seg * a.strips;
    sg.oy0 = seg * a.SH;
    sg.ox0 = strip * RB::TW;
    sg.rows = (a.H - sg.oy0) < a.SH ? (a.H - sg.oy0) : a.SH;
    sg.J = (sg.rows + 3) / 2;
    sg.T = (sg.rows + 1) / 2 + 2;
  }
  if (wave >= 4) rows_consumer(a, smem, wave - 4, sg);
  else if (wave & 1) rows_producer<2>(a, smem, wave, sg);
  else rows_producer<3>(a, smem, wave, sg);
}

}  // namespace

#ifdef LFD_ROWS_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_rows_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rows_dbg), sizeof(unsigned long long) * 2 * 16 * 8);
}
#endif

// called by lfd_fasterblock_fused_f16 (block.hip) for large maps; returns LFD_ERR_UNSUPPORTED when the shape does not suit it
int lfd_block64_rows_launch(const _Float16* in, _Float16* out, const void* w1, const float* b1, const void* w2, const float* b2,
                            const _Float16* zeros, int n, int h, int w, hipStream_t st) {
  RowsArgs a{};
  a.in = in; a.out = out; a.w1 = (const half8*)w1; a.b1 = b1; a.w2 = (const half8*)w2; a.b2 = b2; a.zeros = zeros;
  a.N = n; a.H = h; a.W = w;
  static int cus_of[64] = {};
  const int dev_ = lfd_device_ordinal();
  int& cus = cus_of[dev_];
  if (!cus) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block64_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                            RB::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1)
      return LFD_ERR_LAUNCH_FAILED;
    cus = c;
  }
  a.strips = (w + RB::TW - 1) / RB::TW;
  const long cols = (long)n * a.strips;
  const int target = lfd_tune(LFD_TUNE_ROWS_WGS);     // (A/B: workgroups per launch)
  int segs = (int)((target > 0 ? target : cus) / cols);
  if (segs < 1) segs = 1;
  int sh = (h + segs - 1) / segs;
  if (sh < 4) sh = 4;
  if (sh > h) sh = h;
  a.SH = sh;
  a.segs = (h + sh - 1) / sh;
  const long nwork = cols * a.segs;
  if (nwork > 0x3fffffffL) return LFD_ERR_UNSUPPORTED;
  a.nwork = (int)nwork;
  const int blocks = 8 * ((a.nwork + 7) / 8);
  hipLaunchKernelGGL(k_block64_rows, dim3(blocks), dim3(512), RB::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
