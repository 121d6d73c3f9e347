// csrc/down.hip -- lfd_downblock_fused_f16: the FIRST block of a backbone stage in one launch.
//
// Reference: FasterBlock.forward with a downsample branch (lfd/model/backbone/lfd_resnet.py:96-154, the branch itself
// :458-468), 64 -> 64 channels:
//        y1  = ReLU(BN1(conv3x3 stride 2 (x)))          ident = BN_d(conv1x1 stride 2 (x))
//        out = ReLU(BN2(conv3x3 stride 1 (y1)) + ident)
// The two-launch path (conv_impl.h: k_conv<64,3,2,..,DS> then k_conv<64,3,1,..,RES>) writes y1 and ident to HBM and reads
// them back; its first launch is bound by the read of x (the largest map of the stage), its second runs at 0.22 of the
// MFMA roof.  Here y1 and ident never leave the CU.
//
// MI355X-first structure -- ROW STREAMING instead of 2-D tiles.  A stride-2 halo tile of an 8 x 16 output tile is
// 21 x 37 pixels = 112 KB: it cannot be double-buffered in 160 KB of LDS.  A workgroup therefore owns a
// column STRIP of 30 output columns and walks DOWN a segment of SH output rows with three ring buffers of rows in LDS:
//   * input ring   13 rows x 65 pixels (x rows, column-de-interleaved + XOR-swizzled exactly like the stride-2 tile of
//                  conv_impl.h: conflict-free ds_read_b128 for the stride-2 B fragments), filled by global->LDS DMA two
//                  steps ahead (~60 KB in flight per CU: what 5 TB/s x 2 us of latency needs);
//   * mid ring      6 rows x 32 (+2) pixels of y1 (fp16 after ReLU -- the same rounding point as the two-launch path),
//                  144-byte pixel pitch (conflict-free reads for 32 consecutive pixels);
//   * ident ring    5 rows x 30 pixels of the downsample branch (fp16, no ReLU).
// One MFMA pixel tile (32 pixels) = ONE ROW of the strip: conv1 produces a 32-pixel mid row with no wasted MFMA slot, conv2
// consumes it for 30 outputs (94 % slot use); there is no vertical halo recompute inside a segment (2 extra mid rows per
// segment), the horizontal one is 32 / 30.  [k_block64's 8 x 16 tiles: 1.25 x recompute at 93.75 % slot use.]
//   * 512 threads = 8 waves, PRODUCER / CONSUMER specialisation as in block.hip: waves 0-3 run conv1 (+ the 1x1 branch on
//     the centre tap's fragments) for mid rows (2j, 2j+1) x two 32-channel slabs in step j, waves 4-7 run conv2 for
//     output rows (2j-4, 2j-3) x two slabs; wave w and w+4 share a SIMD.  All three filters are register-stationary.
//   * ONE s_barrier per step; the CONSUMERS issue the DMA of the rows two steps ahead right after it (conv2 has fewer MFMAs
//     per step than conv1 + branch) and wait for their own share with a counted vmcnt (instructions retire in order: every
//     step issues exactly 9 DMA + 4 store instructions per wave).
// Results are BIT-IDENTICAL to the two-launch path: same k order (tap-major, 16-channel group minor), bias in the
// accumulator, the same fp16 rounding of y1 / ident, one rounding of acc + ident (tests/test_gpu_down.py).
#include "conv_impl.h"

namespace {

struct DownArgs {
  const _Float16* in;    // [N,H,W,64]
  _Float16* out;         // [N,OH,OW,64]
  const half8* w1;       // packed [2][36][64] half8: BN-folded conv1 (3x3 stride 2)
  const float* b1;
  const half8* wd;       // packed [2][4][64]: BN-folded downsample conv (1x1 stride 2)
  const float* bd;
  const half8* w2;       // packed [2][36][64]: BN-folded conv2 (3x3 stride 1)
  const float* b2;
  const _Float16* zeros; // 4 KB line: [0,2048) zero
  int N, H, W, OH, OW;
  int strips, segs, SH;  // strips of DN::TW columns per row, segments of SH rows per column
  int nwork;             // N * segs * strips
};

struct DN {
  static constexpr int TW = 30;                       // output columns of a strip
  static constexpr int MW = 32;                       // mid columns (TW + 2) = one MFMA pixel tile
  static constexpr int IW = 2 * MW + 1;               // 65 input columns
  static constexpr int IWh = (IW + 1) / 2;            // 33 even columns, then 32 odd ones
  static constexpr int IWs = 2 * IWh;                 // 66 slots per row
  static constexpr int IN_ROWB = IWs * 128;           // 8448
  static constexpr int NIN = 13;                      // input ring rows: 5 live + 8 in flight
  static constexpr int NDMA = (IWs + 7) / 8;          // 9 DMA instructions per row
  static constexpr int MID_PIXB = 144;
  static constexpr int MID_ROWB = (MW + 2) * MID_PIXB;   // 4896: lanes 30, 31 of conv2 read two columns past the row (unused results)
  static constexpr int NMID = 6;
  static constexpr int ID_PIXB = 144;
  static constexpr int ID_ROWB = TW * ID_PIXB;        // 4320
  static constexpr int NID = 5;
  static constexpr int NK = 36;
  static constexpr int OFF_IN = 0;
  static constexpr int OFF_MID = OFF_IN + NIN * IN_ROWB;        // 109824
  static constexpr int OFF_ID = OFF_MID + NMID * MID_ROWB;      // 139200
  static constexpr int OFF_BIAS = OFF_ID + NID * ID_ROWB;       // 160800
  static constexpr int LDS_BYTES = OFF_BIAS + 3 * 64 * 4;       // 161568
};
static_assert(DN::LDS_BYTES <= 160 * 1024, "LDS capacity");

#ifdef LFD_DOWN_TIMING
// phase stamps of workgroup 0: [role 0 = producer wave 0, 1 = consumer wave 4][step < 16][stamp < 8] (shader clock; [7] = 100 MHz)
__device__ unsigned long long g_down_dbg[2 * 16 * 8];
#define DT(role, i) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && j < 16) { g_down_dbg[((role) * 16 + j) * 8 + (i)] = __builtin_readcyclecounter(); \
    if ((i) == 0) g_down_dbg[((role) * 16 + j) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define DT(role, i)
#endif

struct Seg {          // this workgroup's segment
  int n, oy0, ox0, rows;   // image, first output row / column, output rows (<= SH)
  int J, T;                // producer steps, total steps
};

// ---- one input row -> ring slot: NDMA instructions of 8 slots (64 lanes x 16 B).  Slot `rem` of a row holds input column
// ix = 2 rem (rem < 33) or 2 (rem - 33) + 1; LDS chunk position cs of a slot holds source chunk cs ^ ((rem >> 1) & 7)
// (the swizzle is applied on the SOURCE side: a DMA instruction writes lane-linear).  Columns / rows outside the image
// come from the zero line (conv1's zero padding).  `off[j]` = byte offset of this lane's chunk from the row's first
// strip column (or -1: column outside the image), computed once per workgroup (row_offsets); interior strips -- all but the
// first and the last of a row -- take the path without per-lane selects (first version: ~240 cycles per instruction beside a
// contracting partner wave, 2200 per row: the longest phase of the step).
// saddr form of the DMA: 64-bit SCALAR base + 32-bit per-lane byte offset -- no per-lane 64-bit address arithmetic
__device__ __forceinline__ void dma16_s(unsigned voff, const void* sbase, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  const unsigned long long b = (unsigned long long)(uintptr_t)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(m0v) : "memory");
}
struct RowOff {
  int off[DN::NDMA];     // byte offset of this lane's chunk from the strip's first input column, -1: column outside the image
  unsigned full;         // bit j: every lane of instruction j reads inside the image (all instructions of an interior strip)
};
__device__ __forceinline__ void row_offsets(const DownArgs& a, const Seg& sg, RowOff& ro, bool opaque = false) {
  int ol = threadIdx.x & 63;
  if (opaque) asm volatile("" : "+v"(ol));     // producers: recomputed per row -- they have no registers to pin
  const int sl = ol >> 3, cs = ol & 7;
  const int gx0 = 2 * sg.ox0 - 3;
  ro.full = 0u;
#pragma unroll
  for (int j = 0; j < DN::NDMA; ++j) {
    const int rem = 8 * j + sl;
    const int ix = rem < DN::IWh ? 2 * rem : 2 * rem - (2 * DN::IWh - 1);
    const int c = cs ^ ((rem >> 1) & 7);
    const int gx = gx0 + ix;
    const bool used = rem < DN::IWs - 1;                        // slot 65 does not exist (65 columns in 66 slots)
    const bool ok = gx >= 0 && gx < a.W;
    ro.off[j] = (used && ok) ? ix * 128 + c * 16 : (used ? -1 : 3 * 128);   // (unused slot: any address inside the image -- column 2 ox0)
    if (__ballot(used && !ok) == 0ull) ro.full |= 1u << j;
  }
}
// One row = NDMA instructions.  il < 0: a filler row (every lane reads the zero line) -- keeps the number of DMA instructions
// per step constant, which the counted vmcnt waits rely on.  RowJob holds what is common to a row's instructions; issue_one
// emits instruction j (a compile-time index after unrolling) -- the consumers place them BETWEEN the MFMAs of their k loop.
struct RowJob { bool rv; const char* rowp; char* lbase; };
__device__ __forceinline__ RowJob row_job(const DownArgs& a, char* smem, const Seg& sg, int il, int slot) {
  RowJob r;
  const int gy = 2 * sg.oy0 - 3 + il;                           // wave-uniform
  const long rowpitch = (long)a.W * 128;
  r.rv = il >= 0 && gy >= 0 && gy < a.H;
  r.rowp = reinterpret_cast<const char*>(a.in) + ((long)sg.n * a.H + (r.rv ? gy : 0)) * rowpitch + (long)(2 * sg.ox0 - 3) * 128;
  r.lbase = smem + DN::OFF_IN + slot * DN::IN_ROWB;
  return r;
}
__device__ __forceinline__ void issue_one(const DownArgs& a, const RowOff& ro, const RowJob& r, int j) {
  const int ol = threadIdx.x & 63;
  const int sl = ol >> 3, cs = ol & 7;
  const char* zp = reinterpret_cast<const char*>(a.zeros);
  const int o = ro.off[j];
  const bool act = 8 * j + 8 <= DN::IWs || sl < DN::IWs - 8 * j;
  if (!r.rv) {                                                  // a whole row of zeros (above / below the image, filler rows)
    const unsigned zc = (cs ^ (((8 * j + sl) >> 1) & 7)) * 16;
    if (act) dma16_s(zc, zp, r.lbase + j * 1024);
  } else if ((ro.full >> j) & 1u) {
    if (act) dma16_s((unsigned)o, r.rowp, r.lbase + j * 1024);
  } else {                                                      // first / last strip of a row: some lanes are conv1's padding
    const unsigned zc = (cs ^ (((8 * j + sl) >> 1) & 7)) * 16;
    const char* src = o >= 0 ? r.rowp + o : zp + zc;
    if (act) dma16(src, r.lbase + j * 1024);
  }
}
// Instructions 0 .. DN_SPLIT-1 of every row are issued by the producer wave, the rest by the consumer wave.  Measured (8 x 270 x
// 480, same session): 0 -> 57.6 us, 2 -> 62.2, 4 -> 62.6, 5 -> 59.8: the time a wave spends blocked on a DMA instruction (110-250
// cycles) is back-pressure of the memory system, not issue work that more waves would share -- the kernel moves 187 MB at
// 3.3 TB/s, and the stand-alone stride-2 conv (same DMA pattern, eight issuing waves per CU) tops out at 4.3 TB/s.
#ifndef DN_SPLIT
#define DN_SPLIT 0
#endif
template <int J0, int J1>
__device__ __forceinline__ void issue_row(const DownArgs& a, char* smem, const Seg& sg, const RowOff& ro, int il, int slot) {
  const RowJob r = row_job(a, smem, sg, il, slot);
#pragma unroll
  for (int j = J0; j < J1; ++j) issue_one(a, ro, r, j);
}

__device__ __forceinline__ int wrap(int v, int m) { return v >= m ? v - m : v; }
template <int N>
__device__ __forceinline__ void lfd_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------ producer (conv1 + branch)
__device__ __forceinline__ void down_producer(const DownArgs& a, char* smem, int pw, const Seg& sg) {
  const int lane = threadIdx.x & 63;
  const int ct = pw & 1, rp = pw >> 1;            // cout slab, row of the step's pair
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + DN::OFF_BIAS);

  half8 wreg[DN::NK];
  {
    const half8* wsrc = a.w1 + (size_t)ct * DN::NK * 64 + lane;
#pragma unroll
    for (int k = 0; k < DN::NK; ++k) wreg[k] = wsrc[(size_t)k * 64];
  }
  half8 wdr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wdr[q] = a.wd[((size_t)ct * 4 + q) * 64 + lane];
  // the filters must have ARRIVED before the first DMA is issued: the compiler waits for them at their first use -- inside the
  // step loop, with a vmcnt that knows nothing of the DMA instructions issued in between and would drain them every step
#pragma unroll
  for (int k = 0; k < DN::NK; ++k) asm volatile("" : "+v"(wreg[k]));
#pragma unroll
  for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(wdr[q]));

  // this wave's share of the input DMA: instructions 0 .. DN_SPLIT-1 of row 4b + 1 + pw of every batch b, the consumer wave
  // pw issues the rest of the same row.  An LDS-DMA instruction blocks its wave for 110-250 cycles (measured, with or without a
  // contracting partner, interleaved with MFMAs or not): a row is 9 of them, and the step is as long as its longest chain.
  auto issue_share = [&](int il, int slot) {
    if constexpr (DN_SPLIT > 0) {
      RowOff ro;
      row_offsets(a, sg, ro, true);
      issue_row<0, DN_SPLIT>(a, smem, sg, ro, il, slot);
    }
  };
  if (pw == 0) issue_share(0, 0);
  issue_share(1 + pw, 1 + pw);
  issue_share(1 < sg.J ? 5 + pw : -1, 5 + pw);
  int s_dma = wrap(9 + pw, DN::NIN);              // batch j + 2's row for this wave: 4 (j + 2) + 1 + pw

  // B-fragment offsets inside an input row: mid column pix, tap s reads input column 2 pix + s
  int xoff[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = 2 * pix + s;
    const int rem = (ix & 1) * DN::IWh + (ix >> 1);
    const int f = (rem >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) xoff[s][q] = rem * 128 + (((2 * q + h) ^ f) << 4);
  }
  const int mx = sg.ox0 - 1 + pix;
  const bool colin = mx >= 0 && mx < a.OW;
  const int mwoff = pix * DN::MID_PIXB + ct * 64 + h * 8;
  const bool idcol = pix >= 1 && pix <= DN::TW;
  const int idoff = (idcol ? pix - 1 : 0) * DN::ID_PIXB + ct * 64 + h * 8;

  // ring positions of this wave's rows, advanced per step: input row 4j + 2rp (slot of tap row 0), mid / ident row 2j + rp
  int s_in = wrap(2 * rp, DN::NIN), s_mid = rp, s_id = rp;
  __builtin_amdgcn_s_barrier();                   // biases visible (written by k_down64 before the roles split)
  for (int j = 0; j < sg.T; ++j) {
    DT(0, 0);
    if constexpr (DN_SPLIT > 0) {
      // this wave's share of batch j landed (batch j + 1's may be in flight); the consumers await the rest
      if constexpr (DN_SPLIT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if constexpr (DN_SPLIT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if constexpr (DN_SPLIT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if constexpr (DN_SPLIT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if constexpr (DN_SPLIT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      if constexpr (DN_SPLIT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    DT(0, 1);
    block_barrier();
    DT(0, 2);
    issue_share(j + 2 < sg.J ? 4 * (j + 2) + 1 + pw : -1, s_dma);     // (filler rows keep the count)
    s_dma = wrap(s_dma + 4, DN::NIN);
    if (j >= sg.J) continue;
    DT(0, 3);

    f32x16 acc, accd;
    {
      const float* bp = sbias + ct * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
        acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        const float4 d4 = *reinterpret_cast<const float4*>(bp + 64 + 8 * g);
        accd[4 * g + 0] = d4.x; accd[4 * g + 1] = d4.y; accd[4 * g + 2] = d4.z; accd[4 * g + 3] = d4.w;
      }
    }
    const char* xin = smem + DN::OFF_IN;
    int rb[3];
    rb[0] = s_in * DN::IN_ROWB;
    rb[1] = wrap(s_in + 1, DN::NIN) * DN::IN_ROWB;
    rb[2] = wrap(s_in + 2, DN::NIN) * DN::IN_ROWB;
    auto xfrag = [&](int k) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return *reinterpret_cast<const half8*>(xin + rb[r] + xoff[s][q]);
    };
    constexpr int PD = 3;
    half8 xq[PD + 1];
#pragma unroll
    for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
#pragma unroll
    for (int k = 0; k < DN::NK; ++k) {
      if (k + PD < DN::NK) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[k], xq[k % (PD + 1)], acc, 0, 0, 0);
      if (k / 4 == 4) accd = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdr[k % 4], xq[k % (PD + 1)], accd, 0, 0, 0);   // centre tap
      __builtin_amdgcn_sched_barrier(0);
    }
    DT(0, 4);
    // ---- epilogue: y1 -> mid ring (mid pixels outside the map are conv2's zero padding), branch -> ident ring
    const int ml = 2 * j + rp;
    const int my = sg.oy0 - 1 + ml;
    const bool inimg = colin && my >= 0 && my < a.OH;
    char* mid = smem + DN::OFF_MID + s_mid * DN::MID_ROWB + mwoff;
    char* idb = smem + DN::OFF_ID + s_id * DN::ID_ROWB + idoff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 v;
      v.x = lfd_cvt_pk_max(acc[4 * g + 0], acc[4 * g + 1], LFD_PK_RELU);
      v.y = lfd_cvt_pk_max(acc[4 * g + 2], acc[4 * g + 3], LFD_PK_RELU);
      if (!inimg) { v.x = 0u; v.y = 0u; }
      *reinterpret_cast<uint2*>(mid + 16 * g) = v;
      half4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = (_Float16)accd[4 * g + e];
      if (idcol) *reinterpret_cast<half4*>(idb + 16 * g) = d;
    }
    s_in = wrap(s_in + 4, DN::NIN);
    s_mid = wrap(s_mid + 2, DN::NMID);
    s_id = wrap(s_id + 2, DN::NID);
    DT(0, 5);
  }
}

// ------------------------------------------------------------------------------------------------ consumer (conv2 + add)
__device__ __forceinline__ void down_consumer(const DownArgs& a, char* smem, int cw, const Seg& sg) {
  const int lane = threadIdx.x & 63;
  const int ct = cw & 1, rc = cw >> 1;
  const int h = lane >> 5, pix = lane & 31;
  const float* sbias = reinterpret_cast<const float*>(smem + DN::OFF_BIAS) + 128;

  const int cbase = pix * DN::MID_PIXB + h * 16;
  const int ox = sg.ox0 + pix;
  const bool colok = pix < DN::TW && ox < a.OW;
  const int idoff = (pix < DN::TW ? pix : 0) * DN::ID_PIXB + ct * 64 + h * 8;
  _Float16* obase = a.out + (((size_t)sg.n * a.OH + sg.oy0) * a.OW + (colok ? ox : 0)) * 64 + ct * 32 + 4 * h;

  char* trash = reinterpret_cast<char*>(const_cast<_Float16*>(a.zeros)) + 2048 + (threadIdx.x & 63) * 16;

  // ---- the consumers feed the producers: conv2 has fewer MFMAs per step than conv1 + branch, and in the first version (DMA
  // issued by the producers) these waves idled for 3000 of a step's 5000 cycles.  Batch b = input rows 4b+1 .. 4b+4, one row
  // per consumer wave and step, two steps ahead; row 0 rides with wave 0's prologue.  Every step issues EXACTLY NDMA DMA
  // instructions and then 4 stores per wave (filler rows from the zero line / stores into the trash line when there is
  // nothing to do), so that "batch j has landed" is a counted vmcnt: VMEM operations retire in order.
  RowOff ro;
  row_offsets(a, sg, ro);
  if (cw == 0) issue_row<DN_SPLIT, DN::NDMA>(a, smem, sg, ro, 0, 0);
  issue_row<DN_SPLIT, DN::NDMA>(a, smem, sg, ro, 1 + cw, 1 + cw);
  issue_row<DN_SPLIT, DN::NDMA>(a, smem, sg, ro, 1 < sg.J ? 5 + cw : -1, 5 + cw);
  int s_dma = wrap(9 + cw, DN::NIN);              // batch j + 2's row for this wave: 4 (j + 2) + 1 + cw

  // the stationary conv2 slab -- requested AFTER the prologue DMA, so that the first input rows travel from HBM while the filter
  // comes from L2, and forced to have ARRIVED here (which, VMEM retiring in order, drains that DMA too: it is needed in step 0
  // anyway): left to the compiler, the wait for the filter sits at its first use inside the step loop, with a vmcnt that knows
  // nothing of the DMA instructions issued in between and drains them every step
  half8 wreg[DN::NK];
  {
    const half8* wsrc = a.w2 + (size_t)ct * DN::NK * 64 + lane;
#pragma unroll
    for (int k = 0; k < DN::NK; ++k) wreg[k] = wsrc[(size_t)k * 64];
  }
#pragma unroll
  for (int k = 0; k < DN::NK; ++k) asm volatile("" : "+v"(wreg[k]));

  // ring positions: output row ol = 2 (j - 2) + rc reads mid rows ol .. ol + 2 and ident row ol + 1
  int s_mid = rc, s_id = wrap(rc + 1, DN::NID);
  __builtin_amdgcn_s_barrier();
  for (int j = 0; j < sg.T; ++j) {
    DT(1, 0);
    // behind batch j's DMA in this wave's queue: [4 stores, batch j + 1, 4 stores] from step 2 on
    static_assert(DN_SPLIT >= 0 && DN_SPLIT <= 6, "producer share");
    constexpr int NC = DN::NDMA - DN_SPLIT;       // this wave's DMA instructions per row
    if (j == 0) lfd_wait_vmcnt<NC>();
    else if (j == 1) lfd_wait_vmcnt<NC + 4>();
    else lfd_wait_vmcnt<NC + 8>();
    DT(1, 1);
    block_barrier();
    DT(1, 2);
    // this step's DMA (batch j + 2): its NDMA instructions ride between the MFMAs of the k loop below, where they cost nothing
    // (issued in one piece beside the contracting producer wave of this SIMD they took 1000-1700 cycles: the longest phase)
    issue_row<DN_SPLIT, DN::NDMA>(a, smem, sg, ro, j + 2 < sg.J ? 4 * (j + 2) + 1 + cw : -1, s_dma);
    s_dma = wrap(s_dma + 4, DN::NIN);
    DT(1, 3);
    const int ol = 2 * (j - 2) + rc;
    if (j < 2 || ol >= sg.rows) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {        // four separate stores (volatile: not merged)
        volatile uint32_t* tp = reinterpret_cast<volatile uint32_t*>(trash + 8 * g);
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(tp), "v"(make_uint2(0u, 0u)) : "memory");
      }
    }
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
      const char* mid = smem + DN::OFF_MID + cbase;
      int rb[3];
      rb[0] = s_mid * DN::MID_ROWB;
      rb[1] = wrap(s_mid + 1, DN::NMID) * DN::MID_ROWB;
      rb[2] = wrap(s_mid + 2, DN::NMID) * DN::MID_ROWB;
      auto xfrag = [&](int k) {
        const int r = k / 12, s = (k / 4) % 3, q = k % 4;
        return *reinterpret_cast<const half8*>(mid + rb[r] + (s * DN::MID_PIXB + q * 32));
      };
      constexpr int PD = 3;
      half8 xq[PD + 1];
#pragma unroll
      for (int k = 0; k < PD; ++k) xq[k] = xfrag(k);
      half4 idv[4];
      const char* idb = smem + DN::OFF_ID + s_id * DN::ID_ROWB + idoff;
#pragma unroll
      for (int k = 0; k < DN::NK; ++k) {
        if (k + PD < DN::NK) xq[(k + PD) % (PD + 1)] = xfrag(k + PD);
        if (k == 30) {
#pragma unroll
          for (int g = 0; g < 4; ++g) idv[g] = *reinterpret_cast<const half4*>(idb + 16 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[k], xq[k % (PD + 1)], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      DT(1, 4);
      _Float16* o = obase + (size_t)ol * a.OW * 64;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float x0 = acc[4 * g + 0] + (float)idv[g][0], x1 = acc[4 * g + 1] + (float)idv[g][1];
        const float x2 = acc[4 * g + 2] + (float)idv[g][2], x3 = acc[4 * g + 3] + (float)idv[g][3];
        uint2 v;
        v.x = lfd_cvt_pk_max(x0, x1, LFD_PK_RELU);
        v.y = lfd_cvt_pk_max(x2, x3, LFD_PK_RELU);
        *reinterpret_cast<uint2*>(colok ? reinterpret_cast<char*>(o + 8 * g) : trash) = v;
      }
    }
    s_mid = wrap(s_mid + 2, DN::NMID);
    s_id = wrap(s_id + 2, DN::NID);
    DT(1, 5);
  }
}

__global__ __launch_bounds__(512) void k_down64(DownArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x < 192) {
    float* sb = reinterpret_cast<float*>(smem + DN::OFF_BIAS);
    sb[threadIdx.x] = threadIdx.x < 64 ? a.b1[threadIdx.x] : (threadIdx.x < 128 ? a.bd[threadIdx.x - 64] : a.b2[threadIdx.x - 128]);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // XCD-contiguous work ranges (workgroup b runs on XCD b % 8): neighbouring strips / segments share halo lines in one L2
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.nwork + 7) / 8;
  const int w = xcd * per_xcd + bix;
  const bool live = bix < per_xcd && w < a.nwork;
  Seg sg;
  {
    const int ww = live ? w : 0;
    const int per_img = a.segs * a.strips;
    sg.n = ww / per_img;
    const int r = ww - sg.n * per_img;
    const int seg = r / a.strips, strip = r - seg * a.strips;
    sg.oy0 = seg * a.SH;
    sg.ox0 = strip * DN::TW;
    sg.rows = (a.OH - sg.oy0) < a.SH ? (a.OH - sg.oy0) : a.SH;
    if (!live) sg.rows = 0;
    sg.J = live ? (sg.rows + 3) / 2 : 0;            // mid rows 0 .. rows + 1
    sg.T = live ? (sg.rows + 1) / 2 + 2 : 0;
  }
  if (!live) return;                                // (the whole workgroup: no barrier is left waiting)
  if (wave < 4) down_producer(a, smem, wave, sg);
  else down_consumer(a, smem, wave - 4, sg);
}

}  // namespace

#ifdef LFD_DOWN_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_down_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_down_dbg), sizeof(unsigned long long) * 2 * 16 * 8);
}
#endif

extern "C" int lfd_downblock_fused_f16(int32_t n, int32_t h, int32_t w, const void* in, void* out, const void* w1_packed,
                                       const float* b1, const void* wd_packed, const float* bd, const void* w2_packed,
                                       const float* b2, const void* zeros, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || !w1_packed || !b1 || !wd_packed || !bd || !w2_packed || !b2 || !zeros || in == out)
    return LFD_ERR_INVALID_ARGUMENT;
  if (n < 1 || h < 1 || w < 1 || !lfd_aligned16(in) || !lfd_aligned16(out)) return LFD_ERR_INVALID_ARGUMENT;
  DownArgs a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out;
  a.w1 = (const half8*)w1_packed; a.b1 = b1; a.wd = (const half8*)wd_packed; a.bd = bd; a.w2 = (const half8*)w2_packed; a.b2 = b2;
  a.zeros = (const _Float16*)zeros;
  a.N = n; a.H = h; a.W = w;
  a.OH = (h - 1) / 2 + 1;
  a.OW = (w - 1) / 2 + 1;
  static int cus_of[64] = {};
  const int dev_ = lfd_device_ordinal();
  int& cus = cus_of[dev_];
  if (!cus) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_down64), hipFuncAttributeMaxDynamicSharedMemorySize,
                            DN::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1)
      return LFD_ERR_LAUNCH_FAILED;
    cus = c;
  }
  a.strips = (a.OW + DN::TW - 1) / DN::TW;
  // one workgroup per CU (LDS): cut every column into as many segments as it takes to give each CU one, at least 4 rows each
  const long cols = (long)n * a.strips;
  int segs = (int)(cus / cols);
  if (segs < 1) segs = 1;
  int sh = (a.OH + segs - 1) / segs;
  if (sh < 4) sh = 4;
  if (sh > a.OH) sh = a.OH;
  a.SH = sh;
  a.segs = (a.OH + sh - 1) / sh;
  const long nwork = cols * a.segs;
  if (nwork > 0x3fffffffL) return LFD_ERR_UNSUPPORTED;
  a.nwork = (int)nwork;
  const int blocks = 8 * ((a.nwork + 7) / 8);
  hipLaunchKernelGGL(k_down64, dim3(blocks), dim3(512), DN::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
