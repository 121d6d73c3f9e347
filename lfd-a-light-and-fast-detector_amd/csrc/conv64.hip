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
constexpr int NDMA = (NSLOT * 8 + 63) / 64;      // 43 wave-level DMA instructions per tile

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

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int tiles_per_img = a.tiles_x * a.tiles_y;

  // One DMA instruction (8 halo pixels x 8 chunks) of a tile into ring slot `buf`.  Instruction j of this
  // wave is global instruction i = wave + 4*j of the tile's 43.  Always issued, so that every wave's
  // VMEM-operation count per tile is a constant: a tile index past the end reads the zero line into the
  // (free) ring slot instead.  The tile's scalars are computed once per tile (TileSrc); the per-lane part
  // is recomputed per instruction from an opaque copy of the lane id -- letting the compiler hoist it out
  // of the tile loop would cost 30+ registers that the 288-register filter does not leave.
  struct TileSrc { const _Float16* img; int gy0, gx0; bool live; };
  auto tile_src = [&](int t) {
    TileSrc ts;
    ts.live = t < t_end;
    const int tt = ts.live ? t : t_begin;
    const int n = tt / tiles_per_img;
    const int tr = tt - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    ts.img = a.in + (size_t)n * a.H * a.W * 64;
    ts.gy0 = ty0 * TH - 1; ts.gx0 = tx0 * TW - 1;
    return ts;
  };
  auto dma_instr = [&](int j, const TileSrc& ts, int buf) {
    const int i = wave + 4 * j;
    if (i < NDMA) {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int pslot = i * 8 + (l >> 3);
      if (pslot < NSLOT) {
        const int iy = (pslot * 241) >> 13, ix = pslot - iy * IW;   // pslot / 34 for pslot < 4096
        const int c = (l & 7) ^ ((ix >> 1) & 7);
        const int gy = ts.gy0 + iy, gx = ts.gx0 + ix;
        const bool valid = ts.live && (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
        const unsigned off = ((unsigned)(gy * a.W + gx) * 64u + (unsigned)c * 8u) * 2u;   // bytes within the image
        const char* src = valid ? reinterpret_cast<const char*>(ts.img) + off : reinterpret_cast<const char*>(a.zeros) + c * 16;
        dma16(src, smem + buf * IN_BYTES + i * 8 * 128);
      }
    }
  };
  constexpr int NDJ = (NDMA + 3) / 4;   // 11 DMA instructions per wave per tile (wave 3: 10)

  int t = t_begin + bix;
  // ring slot of iteration `it` is (it + 2) % 3, so the first tile lands in slot 2 while the filter is
  // staged through slots 0-1
  {
    const TileSrc ts0 = tile_src(t);
#pragma unroll 1
    for (int j = 0; j < NDJ; ++j) dma_instr(j, ts0, 2);
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
#pragma unroll 1
    for (int j = 0; j < NDJ; ++j) dma_instr(j, ts1, 0);
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

  // wave-private staging: this wave's two output rows x 32 pixels x 128 B (whole lines, all 64 channels)
  char* wst = smem + NBUF * IN_BYTES + wave * (64 * 128);
  // copy-out of the PREVIOUS tile (instruction j = 8 pixels x 8 chunks) is spread over the current tile's
  // contraction; stores of masked pixels (and of "no previous tile") go to the trash line
  _Float16* p_img = a.out;
  int p_oy0 = 1 << 28, p_tx0 = 0;
  auto copy_load = [&](int j) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int p64 = j * 8 + (l >> 3), c8 = l & 7;
    return *reinterpret_cast<const uint4*>(wst + p64 * 128 + ((c8 ^ ((p64 >> 1) & 7)) * 16));
  };
  auto copy_store = [&](int j, const uint4& v) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int p64 = j * 8 + (l >> 3), c8 = l & 7;
    const int oy = p_oy0 + (p64 >> 5), oxp = p_tx0 * TW + (p64 & 31);
    const unsigned off = ((unsigned)(oy * a.W + oxp) * 64u + (unsigned)c8 * 8u) * 2u;
    char* dst = (oy < a.H && oxp < a.W) ? reinterpret_cast<char*>(p_img) + off
                                        : reinterpret_cast<char*>(const_cast<_Float16*>(a.zeros)) + 2048 + (l & 63) * 16 + (wave & 1) * 1024;
    *reinterpret_cast<uint4*>(dst) = v;
  };

  const float lo = a.relu ? 0.f : -__builtin_inff();   // branch-free optional ReLU
  int it = 0;
  for (; t < t_end; t += t_step, ++it) {
    const int buf = (it + 2) % NBUF;
    // This tile's DMA was issued one whole iteration ago.  Everything this wave issued after it -- the next
    // tile's DMA (>= 10 instructions) and, from the second iteration on, 8 copy-out stores -- may still be in
    // flight; vmcnt retires in order, so allowing that many outstanding operations waits for exactly this tile.
    C64_T(0);
    if (it == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    C64_T(1);
    block_barrier();   // tile landed for every wave; ring slot (it+3)%3 of tile t+2 was consumed last iteration

    const int n = t / tiles_per_img;
    const int tr = t - n * tiles_per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const char* xb = smem + buf * IN_BYTES;
    const int oy0 = ty0 * TH + wave * PT, ox = tx0 * TW + pix;
    const TileSrc ts2 = tile_src(t + 2 * t_step);
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
      // Contraction, 36 k-steps x 4 MFMAs, with all of the tile's memory traffic spread through it so that
      // no wave ever issues a burst that fills the CU's memory queue (a wave stalled on VMEM issue cannot
      // issue MFMAs either, and with one wave per SIMD nobody else would):
      //   k  0..10  one DMA instruction of tile t+2 (and, RES, k 0..15: one residual load of this tile)
      //   k 16..31  copy-out of tile t-1: LDS read on even k, 16-byte global store on the following odd k
      constexpr int PD = 3;
      half8 xq[PD + 1][PT];
      uint4 cv;
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
        if (k < NDJ) dma_instr(k, ts2, buf2);
        if constexpr (RES) {
          if (k < 16) resv[k >> 3][(k >> 2) & 1][k & 3] = *reinterpret_cast<const half4*>(resp[k >> 3] + ((k >> 2) & 1) * 32 + 8 * (k & 3));
        }
        if (k >= 16 && k < 32) {
          if ((k & 1) == 0) cv = copy_load((k - 16) >> 1);
          else copy_store((k - 16) >> 1, cv);
        }
        __builtin_amdgcn_sched_barrier(0);
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
      const int p64 = pt * 32 + pix;
      const int fo = (p64 >> 1) & 7;
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
          *reinterpret_cast<half4*>(wst + p64 * 128 + (((c * 4 + g) ^ fo) * 16) + 8 * h) = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    p_img = a.out + (size_t)n * a.H * a.W * 64; p_oy0 = oy0; p_tx0 = tx0;
    C64_T(4);
  }
  // flush the last tile's copy-out
  if (it > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) copy_store(j, copy_load(j));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS DMA may outlive the workgroup
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
