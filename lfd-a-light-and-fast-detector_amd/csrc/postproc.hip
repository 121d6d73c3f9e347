// csrc/postproc.hip -- per-location decode + score threshold + NMS on gfx950 (wave64).
//
// Replaces, behind include/lfd_hip.h:
//   * nms_ext.nms                 (reference nms_ext.cpp:18-27, nms_kernel.cu:24-138, nms_cpu.cpp:7-66)
//   * batched_nms/multiclass_nms  (reference lfd/model/utils/nms.py:119-220)
//   * LFD._get_results_for_single_image decode (reference lfd/model/lfd.py:434-509, 261-282)
//
// Pipeline, all on the device, per image ("segment"), no host round trip:
//   count  : per 256-point block, number of (point,class) pairs with score > thr
//   scatter: ordered compaction (point-major, class-minor == torch.nonzero order) of the
//            candidates {decoded box, score, label, point}, + max coordinate (for the
//            reference's class-offset trick)
//   sort   : rank sort on the unique 64-bit key (~ord(score), index) -> stable,
//            score-descending order; writes the offset-shifted boxes in sorted order
//   mask   : 64x64-tile suppression bitmask, one wave64 per tile, lane = row box,
//            64-bit row masks built by one lane each (wave == the reference's 64-thread block)
//   scan   : greedy scan on the device: one workgroup per image; wave 0 resolves each
//            64-box diagonal block with ballot/readlane on SALU, all waves OR the kept
//            rows' masks into the LDS-resident `remv` words
//
// This TU is compiled with -ffp-contract=off: IoU and the class-offset arithmetic must be
// evaluated exactly as written (separate fp32 mul/add/sub, IEEE divide) to be bit-identical
// with the reference.
#include "common.h"
#include "decode_impl.h"

namespace {

constexpr int kBlock = 256;

struct SegBuffers {
  int cap;            // candidate capacity per segment
  int words;          // ceil(cap/64): mask row stride in 64-bit words
  const int* counts;  // [nseg,4] device counts (slot 0 = K) or nullptr
  int k_host;         // used when counts == nullptr
  int* total;         // appended segments (a producer kernel filled cand_* through LfdAppendTarget): [nseg] candidates
                      // seen, K = min(total, cap); sort key (score, point * nclass + label); reset by k_scan.  Else nullptr
  int nclass;
  int* total_ws;      // the workspace's counter array (total == total_ws when the segment is an appended one)
  int class_agnostic; // nms_cfg['class_agnostic']: no coordinate offsets (nms.py:145-146)
  float iou_thr;
  float4* cand_box;   // [nseg,cap]
  float* cand_score;  // [nseg,cap]
  int* cand_label;    // [nseg,cap]
  int* cand_point;    // [nseg,cap] (may be nullptr)
  uint32_t* maxord;   // [nseg] ordered-uint max coordinate of the candidates
  float4* s_box;      // [nseg,cap] sorted, offset-shifted boxes
  float* s_area;      // [nseg,cap]
  float* s_score;     // [nseg,cap]
  int* s_label;       // [nseg,cap]
  int* s_idx;         // [nseg,cap] candidate ordinal of sorted position
  int* s_point;       // [nseg,cap] point index of sorted position (nullptr when cand_point is)
  unsigned long long* mask;  // [nseg,cap,words]
};

__device__ __forceinline__ int seg_k(const SegBuffers& b, int seg) {
  int k = b.total ? b.total[seg] : (b.counts ? b.counts[seg * 4] : b.k_host);
  return k < b.cap ? k : b.cap;
}

// ------------------------------------------------------------------ block scan helpers
__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = lfd_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan over a 256-thread block; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* smem /*>=5 ints*/) {
  const int lane = lfd_lane(), w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) smem[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / 64; ++i) {
    int s = smem[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ------------------------------------------------------------------ decode helpers
struct LevelTable {
  int n;
  int start[LFD_MAX_LEVELS + 1];
  int w[LFD_MAX_LEVELS];
  int stride[LFD_MAX_LEVELS];
  float rmax[LFD_MAX_LEVELS];  // max(range) -- 'sigmoid' mode (lfd.py:484)
  float rhi[LFD_MAX_LEVELS];   // range[1]   -- 'independent' mode (lfd.py:469)
};

struct DecodeParams {
  LevelTable lv;
  int P, C, Cc;  // points, foreground classes, classification channels
  int score_mode, decode_mode, in_dtype;
  float score_thr;
  const void* cls;
  const void* reg;
  const float* meta;  // [N,3] = clampW, clampH, resize_scale
  int lds_stride;     // > 0: softmax rows of a block are staged through LDS with this (odd) float stride
  // sibling meta-architectures (FCOS / LFDv2 get_results, lfd_detect_batched_ex); all zero on the LFD path
  const void* ctr;        // [N,P] centerness logits (in_dtype): scores are multiplied by sigmoid(ctr) (fcos.py:376,403-409)
  const uint32_t* keys;   // [N,P] top-k key of the point (fp32 bits of a value >= 0) | bit 31 = selected
  uint32_t sel_levels;    // bit l: level l keeps only its pre_nms_limit best points (fcos.py:383-390, lfdv2.py:620-627)
};

__device__ __forceinline__ float sigmoidf_ref(float x) { return lfd_sigmoidf_ref(x); }

// score of (point p, class c); for softmax mode `mx`/`inv` come from softmax_stats().
// eight logits of a row, requested together (channels past the end re-read the last one; the caller skips them)
__device__ __forceinline__ void load_row8(const DecodeParams& d, int64_t row, int c0, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = lfd_load_f(d.cls, row * d.Cc + (c0 + j < d.Cc ? c0 + j : d.Cc - 1), d.in_dtype);
}
// (same operations in the same order as the rolled loops `for c: m = fmaxf(m, x[c])`, `for c: s += expf(x[c] - m)`; the
// loads come eight at a time instead of one dependent round trip per channel)
__device__ __forceinline__ void softmax_stats(const DecodeParams& d, int64_t row, float* mx, float* sum) {
  float m = -INFINITY;
  for (int c0 = 0; c0 < d.Cc; c0 += 8) {
    float v[8];
    load_row8(d, row, c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (c0 + j < d.Cc) m = fmaxf(m, v[j]);
  }
  float s = 0.f;
  for (int c0 = 0; c0 < d.Cc; c0 += 8) {
    float v[8];
    load_row8(d, row, c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (c0 + j < d.Cc) s += expf(v[j] - m);
  }
  *mx = m;
  *sum = s;
}

__device__ __forceinline__ float score_of(const DecodeParams& d, int64_t row, int c, float mx, float sum) {
  float x = lfd_load_f(d.cls, row * d.Cc + c, d.in_dtype);
  if (d.score_mode == 1) return expf(x - mx) / sum;
  return sigmoidf_ref(x);
}

// decoded, clamped, rescaled box of point p of image n (lfd.py:468-499, 261-282)
__device__ __forceinline__ float4 decode_box(const DecodeParams& d, int n, int p) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < LFD_MAX_LEVELS; ++i)
    if (i < d.lv.n && p >= d.lv.start[i]) l = i;
  const int q = p - d.lv.start[l];
  const int iy = q / d.lv.w[l], ix = q - iy * d.lv.w[l];
  const float px = (float)(ix * d.lv.stride[l]);
  const float py = (float)(iy * d.lv.stride[l]);
  const int64_t row = (int64_t)n * d.P + p;
  float r0 = lfd_load_f(d.reg, row * 4 + 0, d.in_dtype);
  float r1 = lfd_load_f(d.reg, row * 4 + 1, d.in_dtype);
  float r2 = lfd_load_f(d.reg, row * 4 + 2, d.in_dtype);
  float r3 = lfd_load_f(d.reg, row * 4 + 3, d.in_dtype);
  const float W = d.meta[n * 3 + 0], H = d.meta[n * 3 + 1], sc = d.meta[n * 3 + 2];
  return lfd_decode_core(d.decode_mode, r0, r1, r2, r3, px, py, d.decode_mode == 0 ? d.lv.rmax[l] : d.lv.rhi[l], W, H, sc);
}

__device__ __forceinline__ int level_of(const LevelTable& lv, int p) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < LFD_MAX_LEVELS; ++i)
    if (i < lv.n && p >= lv.start[i]) l = i;
  return l;
}
// score factor of a point (1 when the model has no centerness branch: s * 1.f == s bit for bit)
__device__ __forceinline__ float ex_factor(const DecodeParams& d, int64_t row) {
  return d.ctr ? sigmoidf_ref(lfd_load_f(d.ctr, row, d.in_dtype)) : 1.f;
}
// true when the point's level runs a pre-NMS top-k and the point did not make it
__device__ __forceinline__ bool ex_dropped(const DecodeParams& d, int64_t row, int p) {
  if (!d.sel_levels) return false;
  return ((d.sel_levels >> level_of(d.lv, p)) & 1u) && !(d.keys[row] >> 31);
}

// Softmax models (46 channels for TT100K): a thread-per-point sweep reads its 184-byte row with a 184-byte lane stride --
// every load instruction touches 64 cache lines.  The rows of a block's 256 points are one contiguous region: load it
// coalesced into LDS (odd float stride: conflict-free thread-per-row reads), then evaluate exactly the expressions of
// softmax_stats / score_of (same order, same expf / divide) from there.
__device__ __forceinline__ const float* stage_rows(const DecodeParams& d, int n, int blk, float* s_rows) {
  const int base_p = blk * kBlock;
  const int rows = (d.P - base_p) < kBlock ? (d.P - base_p) : kBlock;
  const int64_t g0 = ((int64_t)n * d.P + base_p) * d.Cc;
  const int total = rows * d.Cc;
  // (row, channel) of element i advance by constants from one iteration to the next: one division per thread instead of one per
  // element; EIGHT loads are requested before the first LDS store -- rolled, the loop is a chain of dependent round trips
  // (load, wait, store, next: 46 of them per thread for TT100K's 46 channels = most of the kernel's 66 us)
  const int dr = kBlock / d.Cc, dc = kBlock - dr * d.Cc;
  int r = threadIdx.x / d.Cc, c = threadIdx.x - r * d.Cc;
  for (int i0 = threadIdx.x; i0 < total; i0 += 8 * kBlock) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * kBlock;
      v[j] = lfd_load_f(d.cls, g0 + (i < total ? i : total - 1), d.in_dtype);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (i0 + j * kBlock < total) s_rows[r * d.lds_stride + c] = v[j];
      r += dr;
      c += dc;
      if (c >= d.Cc) { c -= d.Cc; r += 1; }
    }
  }
  __syncthreads();
  return s_rows + threadIdx.x * d.lds_stride;
}
__device__ __forceinline__ void softmax_stats_row(const DecodeParams& d, const float* row, float* mx, float* sum) {
  float m = -INFINITY;
  for (int c = 0; c < d.Cc; ++c) m = fmaxf(m, row[c]);
  float s = 0.f;
  for (int c = 0; c < d.Cc; ++c) s += expf(row[c] - m);
  *mx = m;
  *sum = s;
}

// ------------------------------------------------------------------ count / scatter
// EX: the sibling meta-architectures' extras (centerness factor, per-level top-k survivors only)
// softmax score of class c > thr, decided exactly as `expf(x - mx) / sum > thr` but without the divide for the (vast)
// majority of classes: e / sum > thr needs e > thr * sum * (1 - 2^-20) -- the quotient and the product are each within
// an ulp (2^-23 relative) of the exact values -- so anything below that bound is a sure reject; the rest is decided by
// the divide itself.  Never changes a decision; saves ~44 of 45 IEEE divides per point (TT100K: 46 channels).
__device__ __forceinline__ bool softmax_above(float e, float sum, float lo_bound, float thr) {
  if (e < lo_bound) return false;
  return e / sum > thr;
}

// pcnt (nullable): per-point number of candidates, read back by k_scatter so that only points that HAVE candidates are
// evaluated a second time (round 3: k_scatter re-staged and re-evaluated every row -- 76 us of the 865 us TT100K step)
template <bool EX>
__global__ __launch_bounds__(kBlock) void k_count(DecodeParams d, int* blockcounts, int nblk,
                                                  uint32_t* maxord, int* counts, unsigned char* pcnt) {
  extern __shared__ float s_rows[];
  __shared__ int smem[8];
  const int n = blockIdx.y, blk = blockIdx.x;
  const int p = blk * kBlock + threadIdx.x;
  int cnt = 0;
  float f = 1.f;
  bool live = p < d.P;
  if (d.lds_stride > 0) {
    const float* rowp = stage_rows(d, n, blk, s_rows);
    if (EX && live) {
      const int64_t row = (int64_t)n * d.P + p;
      live = !ex_dropped(d, row, p);
      f = ex_factor(d, row);
    }
    if (live) {
      float mx, sum;
      softmax_stats_row(d, rowp, &mx, &sum);
      if (EX) {
        for (int c = 0; c < d.C; ++c) {
          const float s = expf(rowp[c] - mx) / sum;
          cnt += s * f > d.score_thr;
        }
      } else {
        // every e = expf(x - mx) has x - mx <= 0, i.e. e <= 1 (+ the function's few ulp): when even that is below
        // softmax_above's sure-reject bound no class of this point can pass -- the usual case (background wins), and the
        // second sweep of expf (half of this ALU-bound kernel) is skipped.  Written so that a NaN sum takes the loop.
        const float lo_bound = d.score_thr * sum * 0.999999f;
        if (!(lo_bound > 1.000001f))
          for (int c = 0; c < d.C; ++c) cnt += softmax_above(expf(rowp[c] - mx), sum, lo_bound, d.score_thr);
      }
    }
  } else if (live) {
    const int64_t row = (int64_t)n * d.P + p;
    if (EX) {
      live = !ex_dropped(d, row, p);
      f = ex_factor(d, row);
    }
    if (live) {
      float mx = 0.f, sum = 1.f;
      if (d.score_mode == 1) softmax_stats(d, row, &mx, &sum);
      for (int c = 0; c < d.C; ++c) {
        const float s = score_of(d, row, c, mx, sum);
        cnt += (EX ? s * f : s) > d.score_thr;
      }
    }
  }
  if (pcnt && p < d.P) pcnt[(int64_t)n * d.P + p] = (unsigned char)(cnt < 255 ? cnt : 255);   // 255 = "255 or more": recount
  int total;
  block_excl_scan(cnt, &total, smem);
  if (threadIdx.x == 0) {
    blockcounts[n * nblk + blk] = total;
    if (blk == 0) {
      maxord[n] = 0u;
      counts[n * 4 + 1] = 0;
    }
  }
}

template <bool EX>
__global__ __launch_bounds__(kBlock) void k_scatter(DecodeParams d, const int* blockcounts, int nblk,
                                                    SegBuffers b, int* counts, const unsigned char* pcnt) {
  extern __shared__ float s_rows[];
  __shared__ int smem[8];
  __shared__ uint32_t smax[kBlock / 64];
  __shared__ int s_list[kBlock], s_off[kBlock], s_n;
  const int n = blockIdx.y, blk = blockIdx.x;
  // softmax rows of at most 64 channels: a WAVE finishes each candidate point (lane = channel), see below
  const bool by_wave = !EX && pcnt && d.score_mode == 1 && d.Cc <= 64;
  // a block whose points have no candidate writes nothing (block 0 stays: it publishes the image's totals) -- with a
  // usable threshold that is nearly every block, and the three scans below were 15 us of nothing for TT100K
  if (pcnt && blk != 0 && blockcounts[n * nblk + blk] == 0) return;
  // base = sum of the counts of the preceding blocks of this image (fixed order -> deterministic)
  int part = 0, all = 0;
  for (int i0 = threadIdx.x; i0 < nblk; i0 += 8 * kBlock) {       // eight loads in flight, not one round trip per iteration
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * kBlock;
      v[j] = blockcounts[n * nblk + (i < nblk ? i : 0)];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j * kBlock;
      if (i < nblk) {
        all += v[j];
        if (i < blk) part += v[j];
      }
    }
  }
  // two block reductions via the scan helper's totals
  block_excl_scan(part, &part, smem);
  block_excl_scan(all, &all, smem);
  const int base = part;
  if (blk == 0 && threadIdx.x == 0) {
    counts[n * 4 + 0] = all < b.cap ? all : b.cap;
    counts[n * 4 + 2] = all > b.cap;
    counts[n * 4 + 3] = all;
  }
  const int p = blk * kBlock + threadIdx.x;
  int cnt = 0;
  float mx = 0.f, sum = 1.f, f = 1.f;
  int64_t row = (int64_t)n * d.P + p;
  const float* rowp = nullptr;
  bool live = p < d.P;
  if (pcnt) {
    // the counts k_count recorded: only points that have candidates (a few hundred of ~1e5) look at their row again, straight
    // from global memory (softmax_stats / score_of evaluate the same expressions in the same order as the staged path)
    cnt = live ? (int)pcnt[row] : 0;
    if (EX && cnt > 0) f = ex_factor(d, row);
    if (cnt > 0 && d.score_mode == 1 && !by_wave) softmax_stats(d, row, &mx, &sum);
    // (by_wave implies Cc <= 64 channels -- see its definition -- so a by_wave point can never reach the saturation value 255
    //  and the recount below, which needs mx / sum, only runs where softmax_stats did: ADVICE r3)
    if (cnt == 255 && !by_wave) {          // saturated counter (>= 255 classes above the threshold at one point): recount
      cnt = 0;
      for (int c = 0; c < d.C; ++c) {
        const float s = score_of(d, row, c, mx, sum);
        cnt += (EX ? s * f : s) > d.score_thr;
      }
    }
  } else {
    if (d.lds_stride > 0) rowp = stage_rows(d, n, blk, s_rows);
    if (EX && live) {
      live = !ex_dropped(d, row, p);
      f = ex_factor(d, row);
    }
    if (live) {
      if (rowp) softmax_stats_row(d, rowp, &mx, &sum);
      else if (d.score_mode == 1) softmax_stats(d, row, &mx, &sum);
      for (int c = 0; c < d.C; ++c) {
        const float s = rowp ? expf(rowp[c] - mx) / sum : score_of(d, row, c, mx, sum);
        cnt += (EX ? s * f : s) > d.score_thr;
      }
    }
  }
  int total;
  int off = base + block_excl_scan(cnt, &total, smem);
  uint32_t mo = 0u;
  if (by_wave) {
    // A thread that walks its point's row alone runs ~140 accurate expf / divides as ONE dependent chain (25 us per block
    // that holds a candidate; a few hundred candidates among 3e5 points, so the other 63 lanes idle).  Instead the block
    // lists its candidate points (with their output offsets -- the list order is irrelevant) and each wave takes them in
    // turn with lane = channel: one coalesced load of the row, one expf per lane, the max by butterfly (exact, order-free),
    // the sum by a lane-ordered chain of adds (the SAME fp32 additions in the SAME order as softmax_stats), one divide per
    // lane, ballot -> output slots in class order.  Same values as k_count saw, so the counts agree.
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (cnt > 0) {
      const int i = atomicAdd(&s_n, 1);
      s_list[i] = threadIdx.x;
      s_off[i] = off;
    }
    __syncthreads();
    const int nl = s_n, lane = lfd_lane();
    for (int i = threadIdx.x >> 6; i < nl; i += kBlock / 64) {
      const int pp = blk * kBlock + s_list[i];
      const int o0 = s_off[i];
      const int64_t r = (int64_t)n * d.P + pp;
      const float x = lfd_load_f(d.cls, r * d.Cc + (lane < d.Cc ? lane : d.Cc - 1), d.in_dtype);
      const float4 box = decode_box(d, n, pp);     // its loads are requested with the row's, not after the softmax chain
      float m = lane < d.Cc ? x : -INFINITY;
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
      const float e = lane < d.Cc ? expf(x - m) : 0.f;
      float sm = 0.f;
      for (int c = 0; c < d.Cc; ++c) sm += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), c));
      const float sc = e / sm;
      const bool pass = lane < d.C && sc > d.score_thr;
      const unsigned long long bal = __ballot(pass);
      const int o = o0 + __popcll(bal & ((1ull << lane) - 1ull));
      if (pass && o < b.cap) {
        const int64_t oo = (int64_t)n * b.cap + o;
        b.cand_box[oo] = box;
        b.cand_score[oo] = sc;
        b.cand_label[oo] = lane;
        if (b.cand_point) b.cand_point[oo] = pp;
        const uint32_t q = lfd_float_ord(fmaxf(fmaxf(box.x, box.y), fmaxf(box.z, box.w)));
        mo = q > mo ? q : mo;
      }
    }
  } else if (cnt > 0) {
    const float4 box = decode_box(d, n, p);
    bool wrote = false;
    for (int c0 = 0; c0 < d.C; c0 += 8) {
      float xv[8];
      if (!rowp) load_row8(d, row, c0, xv);        // eight logits in flight (score_of's one-load-per-class is a dependent chain)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        if (c >= d.C) break;
        const float x = rowp ? rowp[c] : xv[j];
        float s = d.score_mode == 1 ? expf(x - mx) / sum : sigmoidf_ref(x);     // == score_of()
        if (EX) s = s * f;
        if (s > d.score_thr) {
          if (off < b.cap) {
            const int64_t o = (int64_t)n * b.cap + off;
            b.cand_box[o] = box;
            b.cand_score[o] = s;
            b.cand_label[o] = c;
            if (b.cand_point) b.cand_point[o] = p;
            wrote = true;
          }
          ++off;
        }
      }
    }
    if (wrote) mo = lfd_float_ord(fmaxf(fmaxf(box.x, box.y), fmaxf(box.z, box.w)));
  }
  // block max -> one atomic
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    uint32_t o = __shfl_xor(mo, s, 64);
    mo = o > mo ? o : mo;
  }
  if (lfd_lane() == 0) smax[threadIdx.x >> 6] = mo;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = smax[0];
    for (int i = 1; i < kBlock / 64; ++i) m = smax[i] > m ? smax[i] : m;
    if (m) atomicMax(&b.maxord[n], m);
  }
}

// ------------------------------------------------------------------ per-level pre-NMS top-k (sibling meta-architectures)
// key of a point = max over classes of its final score (fcos.py:384 sorts after the centerness factor, lfdv2.py:621 has none)
__global__ __launch_bounds__(kBlock) void k_ex_keys(DecodeParams d, uint32_t* keys) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= d.P) return;
  if (!((d.sel_levels >> level_of(d.lv, p)) & 1u)) return;
  const int64_t row = (int64_t)n * d.P + p;
  const float f = ex_factor(d, row);
  float mx = 0.f, sum = 1.f;
  if (d.score_mode == 1) softmax_stats(d, row, &mx, &sum);
  float best = 0.f;     // scores are >= 0: their fp32 bit patterns order like the values
  for (int c = 0; c < d.C; ++c) best = fmaxf(best, score_of(d, row, c, mx, sum) * f);
  keys[row] = __float_as_uint(best) & 0x7fffffffu;
}

constexpr int kSelThreads = 1024;
constexpr int kSelBins = 4096;
// One workgroup per (level, image): radix-select the k-th largest key (12 + 10 + 9 bits, LDS histograms), then mark the k
// survivors with bit 31 of their key: every key above the k-th value, and -- in index order -- as many of the keys EQUAL to
// it as are needed to reach k (torch.topk leaves the choice among equal values open; lowest index first here).
// Each pass streams the level's keys (L2-resident, written by k_ex_keys) with four loads in flight per thread; the bin of
// the k-th key is found by a block-wide suffix scan over the histogram (thread t owns the four bins 4t..4t+3 from the top).
__global__ __launch_bounds__(kSelThreads) void k_ex_select(DecodeParams d, uint32_t* keys, int limit) {
  __shared__ int hist[kSelBins];
  __shared__ int s_wsum[kSelThreads / 64];
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining, s_running, s_ties;
  const int l = blockIdx.x, n = blockIdx.y;
  if (l >= d.lv.n || !((d.sel_levels >> l) & 1u)) return;
  const int p0 = d.lv.start[l], np = d.lv.start[l + 1] - p0;
  uint32_t* kk = keys + (int64_t)n * d.P + p0;
  const int lane = lfd_lane(), w = threadIdx.x >> 6;
  uint32_t prefix = 0u, mask = 0u;
  int remaining = limit;                 // 0 < limit < np (host)
  constexpr int kShift[3] = {19, 9, 0};
  constexpr int kBits[3] = {12, 10, 9};
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = kShift[pass], nb = 1 << kBits[pass];
    const uint32_t bm = (uint32_t)nb - 1u;
    for (int i = threadIdx.x; i < nb; i += kSelThreads) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < np; i += 4 * kSelThreads) {
      uint32_t k[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {       // unconditional loads (index clamped) + select: a conditional load is a branch per load
        const int idx = i + j * kSelThreads;
        const uint32_t raw = kk[idx < np ? idx : np - 1];
        k[j] = idx < np ? raw : 0xffffffffu;                    // bit 31 set: never matches
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((k[j] & (mask | 0x80000000u)) == prefix) atomicAdd(&hist[(k[j] >> shift) & bm], 1);
    }
    __syncthreads();
    // descending position q = 4 * tid + j  <->  bin nb - 1 - q
    int c[4], sum4 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = 4 * (int)threadIdx.x + j;
      c[j] = q < nb ? hist[nb - 1 - q] : 0;
      sum4 += c[j];
    }
    const int inc = wave_incl_scan(sum4);
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    int before = 0;
    for (int j = 0; j < w; ++j) before += s_wsum[j];
    int acc = before + inc - sum4;          // keys in the bins above this thread's four
    if (acc < remaining && remaining <= acc + sum4) {      // exactly one thread: the k-th key is in one of its bins
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (acc < remaining && remaining <= acc + c[j]) {
          s_prefix = prefix | ((uint32_t)(nb - 1 - (4 * (int)threadIdx.x + j)) << shift);
          s_remaining = remaining - acc;
          s_ties = c[j];
        }
        acc += c[j];
      }
    }
    __syncthreads();
    prefix = s_prefix;
    remaining = s_remaining;
    mask |= bm << shift;
    __syncthreads();
  }
  // prefix = the k-th largest key; `remaining` of the s_ties keys equal to it survive
  if (s_ties == remaining) {             // (the usual case: distinct keys) every tie survives, order does not matter
    for (int i = threadIdx.x; i < np; i += kSelThreads) {
      const uint32_t k = kk[i];
      if (k >= prefix) kk[i] = k | 0x80000000u;
    }
    return;
  }
  if (threadIdx.x == 0) s_running = 0;
  __syncthreads();
  for (int base = 0; base < np; base += kSelThreads) {
    const int i = base + threadIdx.x;
    const uint32_t k = i < np ? kk[i] : 0u;
    const int tie = (i < np && k == prefix) ? 1 : 0;
    const int inc = wave_incl_scan(tie);
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    int before = s_running, tot = 0;
    for (int j = 0; j < kSelThreads / 64; ++j) {
      const int v = s_wsum[j];
      if (j < w) before += v;
      tot += v;
    }
    const bool sel = i < np && (k > prefix || (tie && before + inc - 1 < remaining));
    if (sel) kk[i] = k | 0x80000000u;
    __syncthreads();
    if (threadIdx.x == 0) s_running += tot;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void k_decode_all(DecodeParams d, float* out_boxes, float* out_scores) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= d.P) return;
  const int64_t row = (int64_t)n * d.P + p;
  float4 box = decode_box(d, n, p);
  reinterpret_cast<float4*>(out_boxes)[row] = box;
  float mx = 0.f, sum = 1.f;
  if (d.score_mode == 1) softmax_stats(d, row, &mx, &sum);
  for (int c = 0; c < d.C; ++c) out_scores[row * d.C + c] = score_of(d, row, c, mx, sum);
}

// ------------------------------------------------------------------ API-path prepare kernels
// nms_ext.nms input: dets [n,5] -> candidate arrays (label 0, class agnostic).
__global__ void k_prepare_dets(const float* dets, int n, SegBuffers b) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  b.cand_box[i] = make_float4(dets[i * 5 + 0], dets[i * 5 + 1], dets[i * 5 + 2], dets[i * 5 + 3]);
  b.cand_score[i] = dets[i * 5 + 4];
  b.cand_label[i] = 0;
}

// batched_nms inputs -> candidate arrays + max coordinate (nms.py:148)
__global__ void k_prepare_batched(const float* boxes, const float* scores, const int64_t* labels, int k,
                                  SegBuffers b) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mo = 0u;
  if (i < k) {
    float4 bx = reinterpret_cast<const float4*>(boxes)[i];
    b.cand_box[i] = bx;
    b.cand_score[i] = scores[i];
    b.cand_label[i] = (int)labels[i];
    mo = lfd_float_ord(fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    uint32_t o = __shfl_xor(mo, s, 64);
    mo = o > mo ? o : mo;
  }
  if (lfd_lane() == 0 && mo) atomicMax(&b.maxord[0], mo);
}

// ------------------------------------------------------------------ sort (rank sort on unique keys)
__device__ __forceinline__ unsigned long long sort_key(float score, int idx) {
  if (score == 0.f) score = 0.f;  // -0 == +0 (ties broken by index, as a stable sort would)
  return ((unsigned long long)(~lfd_float_ord(score)) << 32) | (unsigned)idx;
}

__global__ __launch_bounds__(kBlock) void k_rank_sort(SegBuffers b) {
  __shared__ unsigned long long keys[kBlock];
  const int seg = blockIdx.y;
  const int K = seg_k(b, seg);
  if ((int)(blockIdx.x * kBlock) >= K) return;  // uniform per block
  const int64_t so = (int64_t)seg * b.cap;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < K;
  const float my_score = live ? b.cand_score[so + i] : 0.f;
  // appended segments arrive in any order: their tie-break is the reference's nonzero() order, point-major class-minor
  auto tie = [&](int j) { return b.total ? b.cand_point[so + j] * b.nclass + b.cand_label[so + j] : j; };
  const unsigned long long my = live ? sort_key(my_score, tie(i)) : ~0ull;
  int rank = 0;
  for (int j0 = 0; j0 < K; j0 += kBlock) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    keys[threadIdx.x] = j < K ? sort_key(b.cand_score[so + j], tie(j)) : ~0ull;
    __syncthreads();
    const int lim = (K - j0) < kBlock ? (K - j0) : kBlock;
#pragma unroll 8
    for (int t = 0; t < lim; ++t) rank += keys[t] < my;
  }
  if (!live) return;
  const float4 bx = b.cand_box[so + i];
  const int label = b.cand_label[so + i];
  float4 sb = bx;
  if (!b.class_agnostic) {
    // offsets = label.to(f32) * (max_coordinate + 1); boxes + offsets   (nms.py:148-150)
    const float step = lfd_ord_float(b.maxord[seg]) + 1.0f;
    const float off = (float)label * step;
    sb.x = bx.x + off; sb.y = bx.y + off; sb.z = bx.z + off; sb.w = bx.w + off;
  }
  b.s_idx[so + rank] = i;
  if (b.s_point) b.s_point[so + rank] = b.cand_point[so + i];
  b.s_score[so + rank] = my_score;
  b.s_label[so + rank] = label;
  b.s_box[so + rank] = sb;
  b.s_area[so + rank] = (sb.z - sb.x) * (sb.w - sb.y);   // nms_cpu.cpp:21 / nms_kernel.cu:19-20
}

// ------------------------------------------------------------------ suppression bitmask
// One wave64 per 64x64 tile (row block r, col block c >= r).  Lane t owns row box r*64+t and
// builds its 64-bit mask of column boxes with IoU > thr (only j > i inside the diagonal tile),
// exactly nms_kernel.cu:24-68 with threadsPerBlock == wave size.  Column boxes are broadcast
// with v_readlane (constant lane after unrolling) instead of LDS.
__device__ __forceinline__ float bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(kBlock) void k_mask(SegBuffers b) {
  // One 256-thread workgroup per 64x64 tile: lane = row box, each of the 4 waves tests 16 of the 64
  // column boxes (4x shorter dependent chain than one wave per tile); the 16-bit partial masks are
  // merged through LDS.
  __shared__ unsigned short part[4][64];
  const int seg = blockIdx.y;
  const int K = seg_k(b, seg);
  const int nb = (K + 63) >> 6;
  const int64_t so = (int64_t)seg * b.cap;
  const int lane = lfd_lane();
  const int cg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // column group 0..3
  const int ntiles = nb * nb;
  const float thr = b.iou_thr;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int r = t / nb, c = t - r * nb;
    if (c < r) continue;
    const int ri = r * 64 + lane, ci = c * 64 + cg * 16 + (lane & 15);
    float4 rb = make_float4(0.f, 0.f, 0.f, 0.f), cb = rb;
    float ra = 0.f, ca = 0.f;
    if (ri < K) { rb = b.s_box[so + ri]; ra = b.s_area[so + ri]; }
    if (ci < K) { cb = b.s_box[so + ci]; ca = b.s_area[so + ci]; }
    const int ncol = K - c * 64;   // valid columns of this tile (>= 64 except in the last one)
    const int start = (r == c) ? lane + 1 : 0;
    unsigned m = 0u;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = cg * 16 + jj;
      const float bx = bcast(cb.x, jj), by = bcast(cb.y, jj), bz = bcast(cb.z, jj), bw = bcast(cb.w, jj);
      const float sb = bcast(ca, jj);
      const float left = fmaxf(rb.x, bx), right = fminf(rb.z, bz);
      const float top = fmaxf(rb.y, by), bottom = fminf(rb.w, bw);
      const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
      const float inter = w * h;
      const float iou = inter / (ra + sb - inter);
      if (j >= start && j < ncol && iou > thr) m |= 1u << jj;
    }
    part[cg][lane] = (unsigned short)m;
    __syncthreads();
    if (cg == 0 && ri < K) {
      const unsigned long long full = (unsigned long long)part[0][lane] | ((unsigned long long)part[1][lane] << 16) |
                                      ((unsigned long long)part[2][lane] << 32) | ((unsigned long long)part[3][lane] << 48);
      b.mask[(so + ri) * b.words + c] = full;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ greedy scan + outputs
struct ScanOut {
  float* dets;     // [nseg,cap,5] or nullptr
  int* labels;     // [nseg,cap] or nullptr
  int* cand;       // [nseg,cap] or nullptr
  int* point;      // [nseg,cap] or nullptr
  int64_t* keep64; // [cap] (API path, single segment) or nullptr
  int* counts;     // [nseg,4]: slot 1 <- number kept      (or nullptr)
  int* num_keep;   // [1] (API path)                       (or nullptr)
  int max_keep;    // > 0: multiclass_nms max_num -- only the first max_keep kept rows count (nms.py:217-219)
};

constexpr int kScanThreads = 512;
constexpr int kScanMaxWords = 4096;  // cap <= 262144 candidates per segment
constexpr int kScanPfWords = 32;     // chunk-row prefetch through LDS when K <= 2048

// dynamic LDS: remv[words] | pad | rowblk[4096 words]
//   K <= 512  (nb <= 8):  rowblk = the WHOLE mask [K rows][8 words], loaded in one round trip
//   K <= 2048 (nb <= 32): rowblk = two [64 rows][32 words] chunk blocks, filled one chunk ahead
//   larger K: mask rows are read from global memory as needed
// The diagonal-block resolve of chunk c runs on wave c % 8, which fetched that chunk's row payload
// (box, score, label, ordinal, point) and diagonal word 8 chunks earlier, so no dependent global
// load sits on the serial path.
__global__ __launch_bounds__(kScanThreads) void k_scan(SegBuffers b, ScanOut o) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long scan_smem[];
  unsigned long long* remv = scan_smem;
  unsigned long long* rowblk = scan_smem + b.words + 2;
  __shared__ unsigned long long s_keep;
  __shared__ int s_nkept;
  __shared__ int s_list[64];
  constexpr int NW = kScanThreads / 64;
  const int seg = blockIdx.x;
  const int K = seg_k(b, seg);
  const int nb = (K + 63) >> 6;
  const int64_t so = (int64_t)seg * b.cap;
  const int lane = lfd_lane();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mode = nb <= 8 ? 0 : (nb <= kScanPfWords ? 1 : 2);
  for (int w = threadIdx.x; w < nb; w += kScanThreads) remv[w] = 0ull;
  if (threadIdx.x == 0) s_nkept = 0;
  float step = 0.f;
  if (!b.class_agnostic && K > 0) step = lfd_ord_float(b.maxord[seg]) + 1.0f;

  constexpr int PFN = (64 * kScanPfWords) / kScanThreads;   // 4 words per thread per chunk block
  unsigned long long pfv[PFN];
  auto pf_load = [&](int c) {
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int idx = i * kScanThreads + threadIdx.x;
      const int r = idx / kScanPfWords, w = idx % kScanPfWords;
      const int row = c * 64 + r;
      pfv[i] = (row < K && w < nb) ? b.mask[(so + row) * b.words + w] : 0ull;
    }
  };
  auto pf_store = [&](int c) {
#pragma unroll
    for (int i = 0; i < PFN; ++i) rowblk[(c & 1) * 64 * kScanPfWords + i * kScanThreads + threadIdx.x] = pfv[i];
  };
  if (mode == 0) {
    unsigned long long v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {       // [512 rows][8 words] = 4096 words, 8 per thread, all in flight
      const int idx = i * kScanThreads + threadIdx.x;
      const int row = idx >> 3, w = idx & 7;
      v[i] = (row < K && w < nb) ? b.mask[(so + row) * b.words + w] : 0ull;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) rowblk[i * kScanThreads + threadIdx.x] = v[i];
  } else if (mode == 1) {
    pf_load(0);
    pf_store(0);
  }
  // row payload + diagonal word of this wave's next chunk
  float4 nbx = make_float4(0.f, 0.f, 0.f, 0.f);
  float nsc = 0.f;
  int nlab = 0, nci = 0, npt = 0;
  unsigned long long ndiag = 0ull;
  auto row_load = [&](int c) {
    const int row = c * 64 + lane;
    if (c < nb && row < K) {
      nbx = b.s_box[so + row]; nsc = b.s_score[so + row]; nlab = b.s_label[so + row]; nci = b.s_idx[so + row];
      npt = b.s_point ? b.s_point[so + row] : 0;
      if (mode != 0) ndiag = b.mask[(so + row) * b.words + c];
    }
  };
  row_load(wave);
  __syncthreads();
  for (int c = 0; c < nb; ++c) {
    if (mode == 1 && c + 1 < nb) pf_load(c + 1);
    const unsigned long long* blk = rowblk + (c & 1) * 64 * kScanPfWords;
    if (wave == (c % NW)) {
      const int row = c * 64 + lane;
      const unsigned long long diag = mode == 0 ? rowblk[row * 8 + c] : ndiag;
      const float4 cbx = nbx;
      const float csc = nsc;
      const int clab = nlab, cci = nci, cpt = npt;
      const unsigned long long r = remv[c];
      // Greedy resolve of the 64x64 diagonal block on the scalar unit.  Only rows whose mask is
      // non-zero can suppress anything, so the serial walk visits just those (typically a handful):
      // a row still alive when the walk reaches it is kept and clears its victims; every row that
      // survives is kept.  Same result as visiting all 64 rows in order (nms_kernel.cu:117-127).
      unsigned long long kb = __ballot(row < K && !((r >> lane) & 1ull));
      unsigned long long pending = kb & __ballot(diag != 0ull);
      const int dlo = (int)(diag & 0xffffffffull), dhi = (int)(diag >> 32);
      while (pending) {
        const int bit = __builtin_ctzll(pending);
        const unsigned long long m =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, bit) << 32) |
            (unsigned)__builtin_amdgcn_readlane(dlo, bit);
        kb &= ~m;                 // victims of a kept row (m only has bits above `bit`)
        pending &= kb;            // suppressed rows no longer suppress
        pending &= ~(1ull << bit);
      }
      const int base = s_nkept;
      if ((kb >> lane) & 1ull) {       // append the kept rows of this block, in order
        const int pos = base + __popcll(kb & ((1ull << lane) - 1ull));
        const int64_t oo = so + pos;
        if (o.dets) {
          float4 bx = cbx;
          if (!b.class_agnostic) {   // nms_bboxes[:, :4] - offsets[kept]   (nms.py:156)
            const float off = (float)clab * step;
            bx.x = bx.x - off; bx.y = bx.y - off; bx.z = bx.z - off; bx.w = bx.w - off;
          }
          float* d = o.dets + oo * 5;
          d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w; d[4] = csc;
        }
        if (o.labels) o.labels[oo] = clab;
        if (o.cand) o.cand[oo] = cci;
        if (o.point && b.s_point) o.point[oo] = cpt;
        if (o.keep64) o.keep64[pos] = (int64_t)cci;
      }
      if ((kb >> lane) & 1ull) s_list[__popcll(kb & ((1ull << lane) - 1ull))] = lane;   // kept rows, compacted
      if (lane == 0) {
        s_keep = kb;
        s_nkept = base + __popcll(kb);
      }
      row_load(c + NW);                // this wave's next turn
    }
    __syncthreads();
    const unsigned long long kb = s_keep;
    const int nw = nb - c - 1;  // words to the right of the diagonal
    if (nw > 0 && kb) {
      const int nk = __popcll(kb);
      const int total = nk * nw;
      for (int q = threadIdx.x; q < total; q += kScanThreads) {
        const int ki = q / nw, w = c + 1 + (q - ki * nw);
        const int bit = s_list[ki];
        unsigned long long v;
        if (mode == 0) v = rowblk[(c * 64 + bit) * 8 + w];
        else if (mode == 1) v = blk[bit * kScanPfWords + w];
        else v = b.mask[(so + c * 64 + bit) * b.words + w];
        if (v) atomicOr(&remv[w], v);
      }
    }
    if (mode == 1 && c + 1 < nb) pf_store(c + 1);   // other buffer: nobody reads it during this iteration
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int nkept = (o.max_keep > 0 && s_nkept > o.max_keep) ? o.max_keep : s_nkept;   // rows are score-descending
    if (o.counts) o.counts[seg * 4 + 1] = nkept;
    if (o.num_keep) o.num_keep[0] = nkept;
    if (b.total) {       // appended segment: publish the counts k_scatter would have written, re-arm the counters
      const int all = b.total[seg];
      if (o.counts) {
        o.counts[seg * 4 + 0] = all < b.cap ? all : b.cap;
        o.counts[seg * 4 + 2] = all > b.cap;
        o.counts[seg * 4 + 3] = all;
      }
      b.total[seg] = 0;
      b.maxord[seg] = 0u;
    }
  }
}

// ------------------------------------------------------------------ host-side plumbing
struct SegLayout {
  size_t bytes;
};

SegBuffers carve_seg(LfdCarver& cv, int nseg, int cap, bool need_point) {
  SegBuffers b{};
  b.cap = cap;
  b.words = (cap + 63) / 64;
  const size_t tot = (size_t)nseg * cap;
  b.cand_box = cv.take<float4>(tot);
  b.cand_score = cv.take<float>(tot);
  b.cand_label = cv.take<int>(tot);
  b.cand_point = need_point ? cv.take<int>(tot) : nullptr;
  b.maxord = cv.take<uint32_t>(nseg);
  b.total_ws = cv.take<int>(nseg);      // (appended segments only; see lfd_detect_bind_append)
  b.total = nullptr;
  b.nclass = 1;
  b.s_box = cv.take<float4>(tot);
  b.s_area = cv.take<float>(tot);
  b.s_score = cv.take<float>(tot);
  b.s_label = cv.take<int>(tot);
  b.s_idx = cv.take<int>(tot);
  b.s_point = need_point ? cv.take<int>(tot) : nullptr;
  b.mask = cv.take<unsigned long long>(tot * b.words);
  return b;
}

size_t seg_bytes(int nseg, int cap, bool need_point) {
  LfdCarver cv(nullptr);
  carve_seg(cv, nseg, cap, need_point);
  return cv.used();
}

int run_sort_mask_scan(const SegBuffers& b, const ScanOut& o, int nseg, hipStream_t st) {
  if (b.words > kScanMaxWords) return LFD_ERR_UNSUPPORTED;
  dim3 gs((b.cap + kBlock - 1) / kBlock, nseg);
  hipLaunchKernelGGL(k_rank_sort, gs, dim3(kBlock), 0, st, b);
  LFD_CHECK_LAUNCH();
  // enough waves to cover the tile list of the capacity, capped: waves loop over tiles
  // waves loop over the (data-dependent) tile list: a modest fixed grid per segment, sized so that the
  // whole batch fills the chip once (launching capacity-many idle workgroups costs more than the work)
  long long nb = b.words, tiles = nb * nb;
  long long blocks = tiles;
  const long long per_seg = nseg >= 8 ? 128 : 1024 / (nseg > 0 ? nseg : 1);
  if (blocks > per_seg) blocks = per_seg;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_mask, dim3((unsigned)blocks, nseg), dim3(kBlock), 0, st, b);
  LFD_CHECK_LAUNCH();
  const size_t scan_lds = ((size_t)b.words + 2 + 2 * 64 * kScanPfWords) * sizeof(unsigned long long);
  static unsigned long long attr_done_mask = 0;
  const int attr_done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(attr_done_mask, attr_done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((kScanMaxWords + 2 + 2 * 64 * kScanPfWords) * sizeof(unsigned long long))) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(attr_done_mask, attr_done_dev);
  }
  hipLaunchKernelGGL(k_scan, dim3(nseg), dim3(kScanThreads), scan_lds, st, b, o);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

int fill_decode_params(const lfd_detect_desc_t* desc, int32_t in_dtype, const void* cls, const void* reg,
                       const float* meta, DecodeParams* d) {
  if (!desc || desc->num_levels < 1 || desc->num_levels > LFD_MAX_LEVELS) return LFD_ERR_INVALID_ARGUMENT;
  if (desc->num_classes < 1 || desc->num_cls_channels < desc->num_classes) return LFD_ERR_INVALID_ARGUMENT;
  if (in_dtype != LFD_F32 && in_dtype != LFD_F16) return LFD_ERR_INVALID_ARGUMENT;
  d->lv.n = desc->num_levels;
  int p = 0;
  for (int i = 0; i < desc->num_levels; ++i) {
    if (desc->level_h[i] < 0 || desc->level_w[i] < 0) return LFD_ERR_INVALID_ARGUMENT;
    d->lv.start[i] = p;
    d->lv.w[i] = desc->level_w[i] > 0 ? desc->level_w[i] : 1;
    d->lv.stride[i] = desc->level_stride[i];
    d->lv.rmax[i] = desc->level_range_lo[i] > desc->level_range_hi[i] ? desc->level_range_lo[i]
                                                                      : desc->level_range_hi[i];
    d->lv.rhi[i] = desc->level_range_hi[i];
    p += desc->level_h[i] * desc->level_w[i];
  }
  for (int i = desc->num_levels; i <= LFD_MAX_LEVELS; ++i) d->lv.start[i] = p;
  d->P = p;
  d->C = desc->num_classes;
  d->Cc = desc->num_cls_channels;
  d->score_mode = desc->score_mode;
  d->decode_mode = desc->decode_mode;
  d->in_dtype = in_dtype;
  d->score_thr = desc->score_thr;
  d->cls = cls;
  d->reg = reg;
  d->meta = meta;
  // softmax score maps: stage the class rows of a block through LDS (odd stride; 256 rows must fit 64 KB)
  d->lds_stride = (desc->score_mode == 1 && (desc->num_cls_channels | 1) * kBlock * 4 <= 64 * 1024) ? (desc->num_cls_channels | 1) : 0;
  return LFD_OK;
}

int desc_points(const lfd_detect_desc_t* desc) {
  int p = 0;
  for (int i = 0; i < desc->num_levels && i < LFD_MAX_LEVELS; ++i) p += desc->level_h[i] * desc->level_w[i];
  return p;
}

}  // namespace

// the append target of a producer kernel (head.hip): same carving as lfd_detect_from_candidates
int lfd_detect_bind_append(const lfd_detect_desc_t* desc, int32_t batch, const float* img_meta, void* workspace,
                           size_t workspace_bytes, LfdAppendTarget* out) {
  if (!desc || !out || !workspace || !img_meta || batch < 1 || desc->max_candidates < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (desc->num_levels < 1 || desc->num_levels > LFD_MAX_LEVELS) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_detect_workspace_bytes(desc, batch)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int P = desc_points(desc);
  const int nblk = (P + kBlock - 1) / kBlock;
  LfdCarver cv(workspace);
  cv.take<int>((size_t)batch * (nblk > 0 ? nblk : 1));
  SegBuffers b = carve_seg(cv, batch, desc->max_candidates, true);
  out->cand_box = b.cand_box; out->cand_score = b.cand_score; out->cand_label = b.cand_label; out->cand_point = b.cand_point;
  out->maxord = b.maxord; out->total = b.total_ws; out->cap = b.cap;
  out->decode_mode = desc->decode_mode; out->score_thr = desc->score_thr; out->meta = img_meta;
  {
    // sigma(x) > thr  =>  x > logit(thr) up to the rounding of the device's expf / divide: widen generously
    const double thr = (double)desc->score_thr;
    double lo = -INFINITY;
    if (thr >= 1.0) lo = INFINITY;                       // sigma(x) <= 1: nothing can pass
    else if (thr > 0.0) { lo = log(thr / (1.0 - thr)); lo -= 1e-3 * (fabs(lo) > 1.0 ? fabs(lo) : 1.0); }
    out->logit_lo = (float)lo;
  }
  for (int i = 0; i < LFD_MAX_LEVELS; ++i) {
    const bool on = i < desc->num_levels;
    out->w[i] = on && desc->level_w[i] > 0 ? desc->level_w[i] : 1;
    out->stride[i] = on ? desc->level_stride[i] : 0;
    const float rmax = on ? (desc->level_range_lo[i] > desc->level_range_hi[i] ? desc->level_range_lo[i] : desc->level_range_hi[i]) : 0.f;
    out->m[i] = desc->decode_mode == 0 ? rmax : (on ? desc->level_range_hi[i] : 0.f);
  }
  return LFD_OK;
}

// =================================================================== C ABI
extern "C" {

size_t lfd_nms_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return seg_bytes(1, (int)n, false) + 256;
}

int lfd_nms_f32(const float* dets, int64_t n, float iou_thr, int64_t* keep, int32_t* num_keep,
                void* workspace, size_t workspace_bytes, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n < 0 || !num_keep) return LFD_ERR_INVALID_ARGUMENT;
  if (n == 0) {
    if (hipMemsetAsync(num_keep, 0, sizeof(int32_t), st) != hipSuccess) return LFD_ERR_LAUNCH_FAILED;
    return LFD_OK;
  }
  if (!dets || !keep || !workspace || n > (1 << 18)) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_nms_workspace_bytes(n)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  LfdCarver cv(workspace);
  SegBuffers b = carve_seg(cv, 1, (int)n, false);
  b.counts = nullptr;
  b.k_host = (int)n;
  b.class_agnostic = 1;
  b.iou_thr = iou_thr;
  hipLaunchKernelGGL(k_prepare_dets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dets, (int)n, b);
  LFD_CHECK_LAUNCH();
  ScanOut o{};
  o.keep64 = keep;
  o.num_keep = num_keep;
  return run_sort_mask_scan(b, o, 1, st);
}

size_t lfd_batched_nms_workspace_bytes(int64_t k) { return lfd_nms_workspace_bytes(k); }

int lfd_batched_nms_f32(const float* boxes, const float* scores, const int64_t* labels, int64_t k,
                        float iou_thr, int32_t class_agnostic, float* out_dets, int64_t* keep,
                        int32_t* num_keep, void* workspace, size_t workspace_bytes, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (k < 0 || !num_keep) return LFD_ERR_INVALID_ARGUMENT;
  if (k == 0) {
    if (hipMemsetAsync(num_keep, 0, sizeof(int32_t), st) != hipSuccess) return LFD_ERR_LAUNCH_FAILED;
    return LFD_OK;
  }
  if (!boxes || !scores || !labels || !workspace || k > (1 << 18)) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_batched_nms_workspace_bytes(k)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  LfdCarver cv(workspace);
  SegBuffers b = carve_seg(cv, 1, (int)k, false);
  b.counts = nullptr;
  b.k_host = (int)k;
  b.class_agnostic = class_agnostic ? 1 : 0;
  b.iou_thr = iou_thr;
  if (hipMemsetAsync(b.maxord, 0, sizeof(uint32_t), st) != hipSuccess) return LFD_ERR_LAUNCH_FAILED;
  hipLaunchKernelGGL(k_prepare_batched, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, boxes, scores,
                     labels, (int)k, b);
  LFD_CHECK_LAUNCH();
  ScanOut o{};
  o.dets = out_dets;
  o.keep64 = keep;
  o.num_keep = num_keep;
  return run_sort_mask_scan(b, o, 1, st);
}

size_t lfd_detect_workspace_bytes(const lfd_detect_desc_t* desc, int32_t batch) {
  if (!desc || batch < 1 || desc->max_candidates < 1) return 0;
  const int P = desc_points(desc);
  const int nblk = (P + kBlock - 1) / kBlock;
  LfdCarver cv(nullptr);
  cv.take<int>((size_t)batch * (nblk > 0 ? nblk : 1));
  carve_seg(cv, batch, desc->max_candidates, true);
  cv.take<unsigned char>((size_t)batch * (P > 0 ? P : 1));      // per-point candidate counts (k_count -> k_scatter); carved LAST:
  return cv.used() + 256;                                       // the offsets lfd_detect_bind_append relies on do not move
}

static int detect_batched_impl(const lfd_detect_desc_t* desc, const lfd_detect_ext_t* ext, int32_t batch, const void* cls,
                               const void* reg, const void* centerness, int32_t in_dtype, const float* img_meta,
                               float* out_dets, int32_t* out_labels, int32_t* out_cand, int32_t* out_point,
                               int32_t* out_counts, void* workspace, size_t workspace_bytes, hipStream_t st) {
  DecodeParams d{};
  int rc = fill_decode_params(desc, in_dtype, cls, reg, img_meta, &d);
  if (rc != LFD_OK) return rc;
  if (batch < 1 || !cls || !reg || !img_meta || !out_counts || !workspace || desc->max_candidates < 1)
    return LFD_ERR_INVALID_ARGUMENT;
  const size_t need = ext ? lfd_detect_ex_workspace_bytes(desc, batch) : lfd_detect_workspace_bytes(desc, batch);
  if (workspace_bytes < need) return LFD_ERR_WORKSPACE_TOO_SMALL;
  if (d.P == 0) {
    if (hipMemsetAsync(out_counts, 0, sizeof(int32_t) * 4 * batch, st) != hipSuccess) return LFD_ERR_LAUNCH_FAILED;
    return LFD_OK;
  }
  const int nblk = (d.P + kBlock - 1) / kBlock;
  LfdCarver cv(workspace);
  int* blockcounts = cv.take<int>((size_t)batch * nblk);
  SegBuffers b = carve_seg(cv, batch, desc->max_candidates, true);
  unsigned char* pcnt = cv.take<unsigned char>((size_t)batch * d.P);
  b.counts = out_counts;
  b.k_host = 0;
  b.class_agnostic = desc->class_agnostic ? 1 : 0;
  b.iou_thr = desc->iou_thr;
  const size_t rows_lds = d.lds_stride > 0 ? (size_t)d.lds_stride * kBlock * sizeof(float) : 0;
  ScanOut o{};
  if (ext) {
    // FCOS / LFDv2: centerness factor, per-level top-k before the threshold, post-NMS cap
    d.ctr = centerness;
    uint32_t* keys = cv.take<uint32_t>((size_t)batch * d.P);
    if (ext->pre_nms_limit > 0)
      for (int i = 0; i < desc->num_levels; ++i)
        if (ext->pre_nms_limit < desc->level_h[i] * desc->level_w[i]) d.sel_levels |= 1u << i;
    if (d.sel_levels) {
      d.keys = keys;
      hipLaunchKernelGGL(k_ex_keys, dim3(nblk, batch), dim3(kBlock), 0, st, d, keys);
      LFD_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_ex_select, dim3(desc->num_levels, batch), dim3(kSelThreads), 0, st, d, keys, ext->pre_nms_limit);
      LFD_CHECK_LAUNCH();
    }
    o.max_keep = ext->post_nms_limit > 0 ? ext->post_nms_limit : 0;
    hipLaunchKernelGGL(k_count<true>, dim3(nblk, batch), dim3(kBlock), rows_lds, st, d, blockcounts, nblk, b.maxord, out_counts, pcnt);
    LFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_scatter<true>, dim3(nblk, batch), dim3(kBlock), 0, st, d, blockcounts, nblk, b, out_counts, pcnt);
    LFD_CHECK_LAUNCH();
  } else {
    hipLaunchKernelGGL(k_count<false>, dim3(nblk, batch), dim3(kBlock), rows_lds, st, d, blockcounts, nblk, b.maxord, out_counts, pcnt);
    LFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_scatter<false>, dim3(nblk, batch), dim3(kBlock), 0, st, d, blockcounts, nblk, b, out_counts, pcnt);
    LFD_CHECK_LAUNCH();
  }
  o.dets = out_dets;
  o.labels = out_labels;
  o.cand = out_cand;
  o.point = out_point;
  o.counts = out_counts;
  return run_sort_mask_scan(b, o, batch, st);
}

int lfd_detect_batched(const lfd_detect_desc_t* desc, int32_t batch, const void* cls, const void* reg,
                       int32_t in_dtype, const float* img_meta, float* out_dets, int32_t* out_labels,
                       int32_t* out_cand, int32_t* out_point, int32_t* out_counts, void* workspace,
                       size_t workspace_bytes, lfd_stream_t stream) {
  return detect_batched_impl(desc, nullptr, batch, cls, reg, nullptr, in_dtype, img_meta, out_dets, out_labels, out_cand,
                             out_point, out_counts, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

size_t lfd_detect_ex_workspace_bytes(const lfd_detect_desc_t* desc, int32_t batch) {
  const size_t base = lfd_detect_workspace_bytes(desc, batch);
  if (!base) return 0;
  return base + (size_t)batch * desc_points(desc) * sizeof(uint32_t) + 256;
}

int lfd_detect_batched_ex(const lfd_detect_desc_t* desc, const lfd_detect_ext_t* ext, int32_t batch, const void* cls,
                          const void* reg, const void* centerness, int32_t in_dtype, const float* img_meta,
                          float* out_dets, int32_t* out_labels, int32_t* out_cand, int32_t* out_point,
                          int32_t* out_counts, void* workspace, size_t workspace_bytes, lfd_stream_t stream) {
  if (!ext) return LFD_ERR_INVALID_ARGUMENT;
  return detect_batched_impl(desc, ext, batch, cls, reg, centerness, in_dtype, img_meta, out_dets, out_labels, out_cand,
                             out_point, out_counts, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int lfd_detect_workspace_reset(const lfd_detect_desc_t* desc, int32_t batch, void* workspace, size_t workspace_bytes,
                               lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!desc || !workspace || batch < 1 || desc->max_candidates < 1) return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_detect_workspace_bytes(desc, batch)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int P = desc_points(desc);
  const int nblk = (P + kBlock - 1) / kBlock;
  LfdCarver cv(workspace);
  cv.take<int>((size_t)batch * (nblk > 0 ? nblk : 1));
  SegBuffers b = carve_seg(cv, batch, desc->max_candidates, true);
  if (hipMemsetAsync(b.maxord, 0, sizeof(uint32_t) * batch, st) != hipSuccess ||
      hipMemsetAsync(b.total_ws, 0, sizeof(int) * batch, st) != hipSuccess)
    return LFD_ERR_LAUNCH_FAILED;
  return LFD_OK;
}

int lfd_detect_from_candidates(const lfd_detect_desc_t* desc, int32_t batch, float* out_dets, int32_t* out_labels,
                               int32_t* out_cand, int32_t* out_point, int32_t* out_counts, void* workspace,
                               size_t workspace_bytes, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!desc || batch < 1 || !out_counts || !workspace || desc->max_candidates < 1 || desc->num_classes < 1)
    return LFD_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < lfd_detect_workspace_bytes(desc, batch)) return LFD_ERR_WORKSPACE_TOO_SMALL;
  const int P = desc_points(desc);
  const int nblk = (P + kBlock - 1) / kBlock;
  LfdCarver cv(workspace);
  cv.take<int>((size_t)batch * (nblk > 0 ? nblk : 1));
  SegBuffers b = carve_seg(cv, batch, desc->max_candidates, true);
  b.counts = out_counts;
  b.k_host = 0;
  b.total = b.total_ws;
  b.nclass = desc->num_classes;
  // one class: the class offsets are label 0 * (max + 1) = +0.0, adding them changes no bit -- skip them, and with them
  // the max-coordinate reduction the producer would have to do
  b.class_agnostic = (desc->class_agnostic || desc->num_classes == 1) ? 1 : 0;
  b.iou_thr = desc->iou_thr;
  ScanOut o{};
  o.dets = out_dets;
  o.labels = out_labels;
  o.cand = out_cand;
  o.point = out_point;
  o.counts = out_counts;
  return run_sort_mask_scan(b, o, batch, st);
}

int lfd_decode_all(const lfd_detect_desc_t* desc, int32_t batch, const void* cls, const void* reg,
                   int32_t in_dtype, const float* img_meta, float* out_boxes, float* out_scores,
                   lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DecodeParams d{};
  int rc = fill_decode_params(desc, in_dtype, cls, reg, img_meta, &d);
  if (rc != LFD_OK) return rc;
  if (batch < 1 || !cls || !reg || !img_meta || !out_boxes || !out_scores) return LFD_ERR_INVALID_ARGUMENT;
  if (d.P == 0) return LFD_OK;
  hipLaunchKernelGGL(k_decode_all, dim3((d.P + kBlock - 1) / kBlock, batch), dim3(kBlock), 0, st, d, out_boxes,
                     out_scores);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // extern "C"
