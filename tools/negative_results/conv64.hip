// csrc/conv64.hip -- the workhorse of the LFD backbone: conv3x3 stride 1, 64 -> 64 channels, NHWC
// fp16, fused bias (+ residual) + ReLU (reference lfd/model/backbone/lfd_resnet.py:96-154: 7 of
// the 11 FasterBlock convs of WIDERFACE_LFD_S run at 135x240 and carry 38 % of the network's FLOPs).
//
// Second-generation kernel for exactly this shape ("one wave per SIMD, whole register file"):
//   * ONE 256-thread workgroup per CU, __launch_bounds__(256, 1): each wave owns its SIMD's 512
//     registers and keeps the COMPLETE filter (64 cout x 576 k = 72 fragments = 288 registers)
//     resident, so every activation fragment read from LDS feeds TWO MFMAs (both 32-channel halves)
//     -- half the LDS traffic of the 2-waves-per-SIMD kernel in conv.hip -- and a wave produces whole
//     128-byte pixel lines, so the epilogue is wave-private (no workgroup barriers);
//   * 8 x 32-pixel output tile per workgroup (halo overhead 1.33x instead of 1.59x), input halo tiles
//     by global->LDS DMA in a 3-deep ring (two tiles in flight), counted vmcnt at the tile boundary;
//   * filter staged once per workgroup through LDS (72 KB from L2 instead of 4 x 72 KB);
//   * persistent workgroups over XCD-contiguous tile ranges.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct C64Args {
  const _Float16* in;     // [N,H,W,64]
  _Float16* out;          // [N,H,W,64]
  const half8* w;         // packed [2 cout tiles][36 k-steps][64 lanes]
  const float* bias;      // [64]
  const _Float16* res;    // residual [N,H,W,64] (RES)
  const _Float16* zeros;  // 4 KB line: [0,2048) zero, [2048,4096) trash for masked stores
  int N, H, W;
  int relu;
  int tiles_x, tiles_y, ntiles;
};

constexpr int TH = 8, TW = 32, PT = 2;
constexpr int IH = TH + 2, IW = TW + 2;          // 10 x 34 halo tile
constexpr int NSLOT = IH * IW;                   // 340 pixel slots of 128 B
constexpr int IN_BYTES = NSLOT * 128;            // 43,520
constexpr int NBUF = 3;
constexpr int STAGE_BYTES = 4 * 64 * 128;        // per wave: its two 32-pixel output rows as full 128-B lines
constexpr int BIAS_OFF = NBUF * IN_BYTES + STAGE_BYTES;
constexpr int LDS_BYTES = BIAS_OFF + 256;               // 163,584 B of the CU's 163,840 (one workgroup per CU)
constexpr int NK = 36;

// global -> LDS DMA (16 B per lane, LDS destination = wave-uniform base in M0 + lane * 16).
// Issued through inline asm ON PURPOSE: when the compiler sees the global_load_lds builtin it assumes every
// later LDS read may alias the in-flight DMA and inserts s_waitcnt vmcnt(0) in front of it -- which drains the
// prefetch ring before the contraction starts and defeats the multi-buffering.  With the DMA opaque, the
// hand-placed counted vmcnt waits at the tile boundary are the only synchronisation with it.
__device__ __forceinline__ void dma16(const void* g, const void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory");   // (m0 is reserved: the compiler never keeps a value in it across statements)
}

__device__ __forceinline__ void block_barrier() {   // s_barrier without the fence's vmcnt(0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifdef LFD_C64_TIMING
__device__ unsigned long long g_c64_dbg[8 * 64];
__device__ unsigned long long g_c64_span[256 * 2];
#define C64_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && it < 8) g_c64_dbg[it * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define C64_T(i)
#endif

template <bool RES>
__global__ __launch_bounds__(256, 1) void k_conv3x3_c64(C64Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, pix = lane & 31;
#ifdef LFD_C64_TIMING
  if (threadIdx.x == 0) g_c64_span[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const long rowpitch = (long)a.W * 128;

  // ---- input halo tile [10 rows][34 pixels][128 B] -> ring slot.  With one wave per SIMD every VALU
  // instruction sits in the MFMA stream, so the per-instruction address work is reduced to one select:
  //   * columns 0..31: DMA instruction j of wave w covers halo row j, columns 8w..8w+7 -- the lane's offset
  //     within the row is a kernel-lifetime constant, the row base is scalar;
  //   * columns 32..33 (2 x 10 pixels = 160 chunks): one ordinary 16-byte load per thread and tile, written
  //     to LDS with ds_write a few k-steps later.
  // Every wave issues exactly 10 DMAs per tile (a tile index past the end reads the zero line into the free
  // ring slot), which keeps the counted vmcnt at the tile boundary a constant.
  const int dl_px = wave * 8 + (lane >> 3);
  const int dl_c = (lane & 7) ^ ((dl_px >> 1) & 7);
  const unsigned dl_off = dl_px * 128 + dl_c * 16;
  const char* zsrc = reinterpret_cast<const char*>(a.zeros) + dl_c * 16;
  const int e_iy = threadIdx.x >> 4, e_px = 32 + ((threadIdx.x >> 3) & 1);
  const int e_c = (threadIdx.x & 7) ^ ((e_px >> 1) & 7);
  const bool e_act = threadIdx.x < 160;
  const unsigned e_lds = (e_iy * IW + e_px) * 128 + (threadIdx.x & 7) * 16;

  struct TileSrc { const char* row0; int gy0, gx0; bool live; };
  auto tile_src = [&](int t) {
    TileSrc ts;
    ts.live = t < t_end;
    const int tt = ts.live ? t : t_begin;
    const int n = tt / tiles_per_img;
    const int tr = tt - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    ts.gy0 = ty0 * TH - 1; ts.gx0 = tx0 * TW - 1;
    // address of halo pixel (row 0, column 0); outside the image for border tiles, only formed, never read
    ts.row0 = reinterpret_cast<const char*>(a.in) + ((long)n * a.H + ts.gy0) * rowpitch + (long)ts.gx0 * 128;
    return ts;
  };
  auto dma_row = [&](int j, const TileSrc& ts, bool xvalid, int buf) {
    const int gy = ts.gy0 + j;
    const bool rv = ts.live && gy >= 0 && gy < a.H;
    const char* rowp = ts.row0 + j * rowpitch;
    const char* src = (rv && xvalid) ? rowp + dl_off : zsrc;
    dma16(src, smem + buf * IN_BYTES + (j * IW + wave * 8) * 128);
  };
  auto extra_load = [&](const TileSrc& ts) {
    const int gy = ts.gy0 + e_iy, gx = ts.gx0 + e_px;
    const bool ok = e_act && ts.live && gy >= 0 && gy < a.H && gx < a.W;   // gx >= 31 always
    const char* src = ok ? ts.row0 + e_iy * rowpitch + e_px * 128 + e_c * 16 : zsrc;
    return *reinterpret_cast<const uint4*>(src);
  };
  auto extra_store = [&](const uint4& v, int buf) {
    if (e_act) *reinterpret_cast<uint4*>(smem + buf * IN_BYTES + e_lds) = v;
  };
  auto xvalid_of = [&](const TileSrc& ts) { const int gx = ts.gx0 + dl_px; return gx >= 0 && gx < a.W; };

  int t = t_begin + bix;
  // ring slot of iteration `it` is (it + 2) % 3, so the first tile lands in slot 2 while the filter is
  // staged through slots 0-1
  {
    const TileSrc ts0 = tile_src(t);
    const bool xv = xvalid_of(ts0);
#pragma unroll
    for (int j = 0; j < IH; ++j) dma_row(j, ts0, xv, 2);
    extra_store(extra_load(ts0), 2);
  }

  // ---- filter: global -> LDS once per workgroup (coalesced), then every wave copies all 72 fragments
  //      into its registers (all four waves hold the same 64 x 576 filter)
  {
    half8* wl = reinterpret_cast<half8*>(smem);
    for (int i = threadIdx.x; i < 2 * NK * 64; i += 256) wl[i] = a.w[i];
    if (threadIdx.x < 64) reinterpret_cast<float*>(smem + BIAS_OFF)[threadIdx.x] = a.bias[threadIdx.x];
  }
  __syncthreads();
  half8 wreg[2][NK];
  {
    const half8* wl = reinterpret_cast<const half8*>(smem);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < NK; ++k) wreg[c][k] = wl[(c * NK + k) * 64 + lane];
  }
  __syncthreads();   // filter staging area is free for the input ring
  {
    const TileSrc ts1 = tile_src(t + t_step);
    const bool xv = xvalid_of(ts1);
#pragma unroll
    for (int j = 0; j < IH; ++j) dma_row(j, ts1, xv, 0);
    extra_store(extra_load(ts1), 0);
  }

  // ---- per-lane LDS read offsets: (column tap s, 16-channel group q); pixel row = 2*wave + pt
  int xoff[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = pix + s;
    const int f = (ix >> 1) & 7;
    const int rowbase = (wave * PT) * IW + ix;
#pragma unroll
    for (int q = 0; q < 4; ++q) xoff[s][q] = rowbase * 128 + (((2 * q + h) ^ f) * 16);
  }

  // wave-private staging: this wave's two output rows x 32 pixels x 128 B (whole lines, all 64 channels).
  // Pixel p (0..63), logical 16-byte chunk c lives at p*128 + ((c ^ ((p >> 1) & 7)) * 16).
  char* wst = smem + NBUF * IN_BYTES + wave * (64 * 128);
  // copy-out of the PREVIOUS tile (instruction j = pixels 8j..8j+7 x 8 chunks) is spread over the current
  // tile's contraction; lane constants: LDS offset for even / odd j, global offset lane*16
  int cl_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) cl_off[e] = (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 4)) ^ (4 * e)) * 16);
  char* trash = reinterpret_cast<char*>(const_cast<_Float16*>(a.zeros)) + 2048 + lane * 16 + (wave & 1) * 1024;
  char* p_base = reinterpret_cast<char*>(a.out);   // previous tile: address of its (row oy0, column 32*tx0) pixel
  int p_oy0 = 1 << 28, p_ox0 = 0;
  auto copy_load = [&](int j) { return *reinterpret_cast<const uint4*>(wst + j * 1024 + cl_off[j & 1]); };
  auto copy_store = [&](int j, const uint4& v) {
    const int oy = p_oy0 + (j >> 2), ox = p_ox0 + (j & 3) * 8 + (lane >> 3);
    char* dst = (oy < a.H && ox < a.W) ? p_base + (j >> 2) * rowpitch + (j & 3) * 1024 + lane * 16 : trash;
    *reinterpret_cast<uint4*>(dst) = v;
  };
  // staging write offsets of this lane's 8 (channel tile c, group g) half4s for pixel `pix` (+ 4096 for pt 1)
  const int fo = (pix >> 1) & 7;

  const float lo = (a.relu & 1) ? 0.f : -__builtin_inff();   // branch-free optional ReLU
  int it = 0;
  for (; t < t_end; t += t_step, ++it) {
    const int buf = (it + 2) % NBUF;
    // This tile's DMA was issued one whole iteration ago.  Everything this wave issued after it -- the next
    // tile's 10 DMAs and, from the second iteration on, 8 copy-out stores -- may still be in flight; vmcnt
    // retires in order, so allowing that many outstanding operations waits for exactly this tile.
    C64_T(0);
    if (it == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    C64_T(1);
    block_barrier();   // tile landed for every wave; the ring slot of tile t+2 was consumed last iteration

    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const char* xb = smem + buf * IN_BYTES;
    const int oy0 = ty0 * TH + wave * PT, ox = tx0 * TW + pix;
    const TileSrc ts2 = tile_src(t + 2 * t_step);
    const bool xv2 = xvalid_of(ts2);
    const int buf2 = (it + 1) % NBUF;

    half4 resv[RES ? PT : 1][2][4];
    const _Float16* resp[PT];
    if constexpr (RES) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + pt;
        const bool ok = oy < a.H && ox < a.W;
        resp[pt] = a.res + (((size_t)n * a.H + (ok ? oy : 0)) * a.W + (ok ? ox : 0)) * 64 + 4 * h;
      }
    }

    f32x16 acc[2][PT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* bp = reinterpret_cast<const float*>(smem + BIAS_OFF) + c * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          acc[c][pt][4 * g + 0] = b4.x; acc[c][pt][4 * g + 1] = b4.y;
          acc[c][pt][4 * g + 2] = b4.z; acc[c][pt][4 * g + 3] = b4.w;
        }
      }
    }

    C64_T(2);
    auto xfrag = [&](int k, int pt) {
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return *reinterpret_cast<const half8*>(xb + xoff[s][q] + (r + pt) * IW * 128);
    };
    {
      // Contraction, 36 k-steps x 4 MFMAs, with all of the tile's memory traffic spread through it:
      //   k = 3j+1     DMA row j of tile t+2                       (10 per wave)
      //   k = 0 / 33   columns 32..33 of tile t+2: global load / ds_write
      //   k = 4j, 4j+2 copy-out of tile t-1: LDS read, then the 16-byte global store (8 per lane)
      //   k = 2i+1     residual load i of this tile (RES)           (16 per lane)
      constexpr int PD = 3;
      half8 xq[PD + 1][PT];
      uint4 cv, ev;
#pragma unroll
      for (int k = 0; k < PD; ++k)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xq[k][pt] = xfrag(k, pt);
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (k + PD < NK) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) xq[(k + PD) % (PD + 1)][pt] = xfrag(k + PD, pt);
        }
#ifdef LFD_C64_SKIP
        if (!(a.relu & 2))
#endif
        if (k % 3 == 1 && k / 3 < IH) dma_row(k / 3, ts2, xv2, buf2);
        if (k == 0) ev = extra_load(ts2);
        if (k == 33) extra_store(ev, buf2);
        if constexpr (RES) {
          if ((k & 1) && (k >> 1) < 16) {
            const int i = k >> 1;
            resv[i >> 3][(i >> 2) & 1][i & 3] = *reinterpret_cast<const half4*>(resp[i >> 3] + ((i >> 2) & 1) * 32 + 8 * (i & 3));
          }
        }
        if (k < 32) {
          if ((k & 3) == 0) cv = copy_load(k >> 2);
          else if ((k & 3) == 2) {
#ifdef LFD_C64_SKIP
            if (!(a.relu & 4))
#endif
            copy_store(k >> 2, cv);
          }
        }
        // no scheduling fence between this k-step's memory/address work and its MFMAs: with a single wave
        // per SIMD the only place that work can hide is in the 32-cycle shadow of each MFMA, so the
        // scheduler must be free to interleave it with the four MFMAs (a fence per k-step keeps the
        // prefetch loads from sinking to their use)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
            acc[c][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[c][k], xq[k % (PD + 1)][pt], acc[c][pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    C64_T(3);
    // ---- epilogue: bias is already in the accumulators; + residual, ReLU, fp16, into the wave's staging
    //      rows (the previous tile's copy-out finished reading them during the contraction above)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float x0 = acc[c][pt][4 * g + 0], x1 = acc[c][pt][4 * g + 1], x2 = acc[c][pt][4 * g + 2], x3 = acc[c][pt][4 * g + 3];
          if constexpr (RES) {
            x0 += (float)resv[pt][c][g][0]; x1 += (float)resv[pt][c][g][1];
            x2 += (float)resv[pt][c][g][2]; x3 += (float)resv[pt][c][g][3];
          }
          x0 = fmaxf(x0, lo); x1 = fmaxf(x1, lo); x2 = fmaxf(x2, lo); x3 = fmaxf(x3, lo);
          half4 v;
          v[0] = (_Float16)x0; v[1] = (_Float16)x1; v[2] = (_Float16)x2; v[3] = (_Float16)x3;
          *reinterpret_cast<half4*>(wst + pt * 4096 + pix * 128 + (((c * 4 + g) ^ fo) * 16) + 8 * h) = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    p_base = reinterpret_cast<char*>(a.out) + ((long)n * a.H + oy0) * rowpitch + (long)tx0 * TW * 128;
    p_oy0 = oy0; p_ox0 = tx0 * TW;
    C64_T(4);
  }
  // flush the last tile's copy-out
  if (it > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) copy_store(j, copy_load(j));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS DMA may outlive the workgroup
#ifdef LFD_C64_TIMING
  if (threadIdx.x == 0) g_c64_span[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <bool RES>
int launch_c64(C64Args a, hipStream_t st) {
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c64<RES>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    done = true;
  }
  int blocks = a.ntiles < 256 ? a.ntiles : 256;
  if (blocks < 1) return LFD_OK;
  hipLaunchKernelGGL((k_conv3x3_c64<RES>), dim3(blocks), dim3(256), LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}

}  // namespace

#ifdef LFD_C64_TIMING
extern "C" __attribute__((visibility("default"))) int lfd_debug_c64_timing(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c64_dbg), sizeof(unsigned long long) * 8 * 64);
}
extern "C" __attribute__((visibility("default"))) int lfd_debug_c64_span(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c64_span), sizeof(unsigned long long) * 512);
}
#endif

// internal entry used by conv.hip's dispatcher (same argument meaning as lfd_conv2d_nhwc_f16)
int lfd_conv3x3_c64_launch(const void* in, void* out, const void* w_packed, const float* bias, const void* residual,
                           const void* zeros, int n, int h, int w, int relu, hipStream_t st) {
  C64Args a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out; a.w = (const half8*)w_packed; a.bias = bias;
  a.res = (const _Float16*)residual; a.zeros = (const _Float16*)zeros;
  a.N = n; a.H = h; a.W = w; a.relu = relu;
  return residual ? launch_c64<true>(a, st) : launch_c64<false>(a, st);
}
