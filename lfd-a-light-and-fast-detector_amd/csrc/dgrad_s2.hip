// csrc/dgrad_s2.hip -- lfd_conv3x3s2_dgrad_nhwc_f16: data gradient of a 3x3 stride-2 convolution (64 -> 64 channels) WITHOUT
// the zero-inserted tensor.
//
// Training path of the LFD backbone (lfd_resnet.py:354-439 stem, :96-154 / :458-468 first block of a stage; autograd's
// conv backward in the reference).  Round 2 computed  dx = conv3x3_s1( zero_insert2(dy), W^T flipped )  with the forward conv
// kernel: the 4 x larger zero-inserted tensor is written and read back (k_zero_insert2: 0.21 ms per iteration), and three of
// four products of the contraction multiply zeros (the stem's gradient alone: 202 us + 121 us of an 8.35 ms iteration).
//
// Here dx is computed per output PARITY: with dyz[2i, 2j] = dy[i, j] the terms of
//        dx[R, C] = sum_{r,s} Wd[r][s] . dyz[R - 1 + r, C - 1 + s]
// that are not identically zero are r = 1 for even R and r in {0, 2} for odd R (likewise s for C): 1, 2, 2 and 4 taps for the
// four parities -- 9 tap-products per 2 x 2 output pixels instead of 36.  A workgroup owns an 8 x 16 tile of dy (+ one halo
// row / column), keeps the whole packed filter Wd stationary in registers (the same fragments the forward conv kernel
// consumes: ops.pack_conv_weight_train(..., data_gradient=True)), and for each of its 32-pixel MFMA tiles runs the four
// parities one after the other: contraction over that parity's taps IN THE ORDER THE ZERO-INSERTED CONV VISITS THEM (the
// skipped products are exact zeros: the result is bit-identical), + the gradient already collected for the same
// activation (`res`), fp16, staged through LDS into 16-byte stores.
#include "conv_impl.h"

namespace {

struct DgArgs {
  const _Float16* dy;    // [N, OH, OW, 64]
  _Float16* dx;          // [N, H, W, 64]
  const half8* w;        // packed [2][36][64]: the data-gradient filter (roles swapped, taps flipped)
  const _Float16* res;   // [N, H, W, 64] or null: added before the fp16 rounding
  int N, OH, OW, H, W;
  int tiles_x, tiles_y, ntiles;
};

struct DG {
  static constexpr int TH = 8, TW = 16;                 // dy tile
  static constexpr int IH = TH + 1, IW = TW + 1;        // + halo (row i + 1, column j + 1)
  static constexpr int PIXB = 144;
  static constexpr int ROWB = 2560;                     // 17 * 144 = 2448 padded to 0 mod 256: an MFMA tile spans two rows
  static constexpr int TILE_BYTES = IH * ROWB;          // 23040
  static constexpr int NLD = (IH * IW * 8 + 255) / 256; // 16-byte loads per thread and tile (5)
  static constexpr int OFF_STG = 2 * TILE_BYTES;
  static constexpr int LDS_BYTES = OFF_STG + 4 * 2048;
};

// taps of parity P (0: even coordinate, 1: odd) in ascending order: (filter index r, dy offset)
template <int P> struct Taps;
template <> struct Taps<0> { static constexpr int N = 1; static constexpr int r[2] = {1, 1}; static constexpr int d[2] = {0, 0}; };
template <> struct Taps<1> { static constexpr int N = 2; static constexpr int r[2] = {0, 2}; static constexpr int d[2] = {0, 1}; };

template <int PR, int PC>
__device__ __forceinline__ void phase(const DgArgs& a, const half8 (&wreg)[36], const char* tile, int lbase, char* stg, int n, int oy, int ox,
                                      int ct, int h, int pix, int lane) {
  using TR = Taps<PR>;
  using TC = Taps<PC>;
  constexpr int NT = TR::N * TC::N;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // residual (the gradient already collected for this activation): requested first, used after the contraction
  const int R = 2 * oy + PR, Cc = 2 * ox + PC;
  const bool ok = R < a.H && Cc < a.W;
  half4 rv[4];
  if (a.res) {
    const _Float16* rp = a.res + (((size_t)n * a.H + (ok ? R : 0)) * a.W + (ok ? Cc : 0)) * 64 + ct * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) rv[g] = *reinterpret_cast<const half4*>(rp + 8 * g);
  }
  // B fragments: one tap (four 16-channel groups) ahead of the MFMAs
  half8 xq[2][4];
  auto fetch = [&](int t, half8 (&dst)[4]) {
    const int dr = TR::d[t / TC::N], dc = TC::d[t % TC::N];
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const half8*>(tile + lbase + dr * DG::ROWB + dc * DG::PIXB + q * 32);
  };
  fetch(0, xq[0]);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t + 1 < NT) fetch(t + 1, xq[(t + 1) & 1]);
    const int r = TR::r[t / TC::N], s = TC::r[t % TC::N];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(r * 3 + s) * 4 + q], xq[t & 1][q], acc, 0, 0, 0);
  }
  // ---- + residual -> fp16 -> wave-private staging (32 pixels x 64 B, chunk XOR (p >> 2) & 3) -> 16-byte stores
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float x0 = acc[4 * g + 0], x1 = acc[4 * g + 1], x2 = acc[4 * g + 2], x3 = acc[4 * g + 3];
    if (a.res) { x0 += (float)rv[g][0]; x1 += (float)rv[g][1]; x2 += (float)rv[g][2]; x3 += (float)rv[g][3]; }
    uint2 v;
    v.x = lfd_cvt_pk_max(x0, x1, LFD_PK_NONE);
    v.y = lfd_cvt_pk_max(x2, x3, LFD_PK_NONE);
    *reinterpret_cast<uint2*>(stg + pix * 64 + ((g ^ ((pix >> 2) & 3)) << 4) + 8 * h) = v;
  }
  // lane idx -> staging pixel p = idx >> 2 (its dy pixel: row (p >> 4), column p & 15 of this MFMA tile), chunk idx & 3
  const int oy0 = oy - (pix >> 4), ox0 = ox - (pix & 15);        // dy coordinates of the MFMA tile's first pixel
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = j * 64 + lane;
    const int p = idx >> 2, c = idx & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(stg + p * 64 + ((c ^ ((p >> 2) & 3)) << 4));
    const int Rp = 2 * (oy0 + (p >> 4)) + PR, Cp = 2 * (ox0 + (p & 15)) + PC;
    if (Rp < a.H && Cp < a.W)
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.dx) + ((((size_t)n * a.H + Rp) * a.W + Cp) * 128 + ct * 64 + c * 16)) = v;
  }
}

__global__ __launch_bounds__(256, 2) void k_dgrad_s2_64(DgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ct = wave & 1, pg = wave >> 1;
  const int h = lane >> 5, pix = lane & 31;
  char* stg = smem + DG::OFF_STG + wave * 2048;

  half8 wreg[36];
  {
    const half8* wsrc = a.w + (size_t)ct * 36 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 36; ++k) wreg[k] = wsrc[(size_t)k * 64];
  }

  // persistent tile walk, XCD-contiguous ranges (workgroup b runs on XCD b % 8), as in conv_impl.h
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per_xcd = (a.ntiles + 7) / 8;
  const int t_begin = xcd * per_xcd;
  const int t_end = (t_begin + per_xcd) < a.ntiles ? (t_begin + per_xcd) : a.ntiles;
  const int t_step = (nblk + 7 - xcd) / 8;
  const int per_img = a.tiles_x * a.tiles_y;

  // one tile of dy (9 x 17 pixels, zero outside the map) as NLD 16-byte loads per thread
  uint4 ld[DG::NLD];
  auto load_tile = [&](int t) {
    const int n = t / per_img;
    const int tr = t - n * per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
#pragma unroll
    for (int i = 0; i < DG::NLD; ++i) {
      const int id = threadIdx.x + 256 * i;
      const int px = (id < DG::IH * DG::IW * 8 ? id : 0) >> 3, ck = id & 7;
      const int row = px / DG::IW, col = px - row * DG::IW;
      const int gy = ty0 * DG::TH + row, gx = tx0 * DG::TW + col;
      const bool ok = gy < a.OH && gx < a.OW;
      ld[i] = reinterpret_cast<const uint4*>(a.dy + (((size_t)n * a.OH + (ok ? gy : 0)) * a.OW + (ok ? gx : 0)) * 64)[ck];
      if (!ok) ld[i] = make_uint4(0u, 0u, 0u, 0u);       // unconditional load + select
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < DG::NLD; ++i) {
      const int id = threadIdx.x + 256 * i;
      if (id < DG::IH * DG::IW * 8) {
        const int px = id >> 3, ck = id & 7;
        const int row = px / DG::IW, col = px - row * DG::IW;
        *reinterpret_cast<uint4*>(smem + buf * DG::TILE_BYTES + row * DG::ROWB + col * DG::PIXB + ck * 16) = ld[i];
      }
    }
  };

  int t = t_begin + bix;
  int buf = 0;
  if (t < t_end) { load_tile(t); store_tile(0); }
  __syncthreads();
  for (; t < t_end; t += t_step, buf ^= 1) {
    const bool more = t + t_step < t_end;
    if (more) load_tile(t + t_step);                     // in flight during this tile's contraction
    const int n = t / per_img;
    const int tr = t - n * per_img;
    const int ty0 = tr / a.tiles_x, tx0 = tr - ty0 * a.tiles_x;
    const char* tile = smem + buf * DG::TILE_BYTES;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int row = (pg * 2 + pt) * 2 + (pix >> 4), col = pix & 15;
      const int lbase = row * DG::ROWB + col * DG::PIXB + h * 16;
      const int oy = ty0 * DG::TH + row, ox = tx0 * DG::TW + col;
      phase<0, 0>(a, wreg, tile, lbase, stg, n, oy, ox, ct, h, pix, lane);
      phase<0, 1>(a, wreg, tile, lbase, stg, n, oy, ox, ct, h, pix, lane);
      phase<1, 0>(a, wreg, tile, lbase, stg, n, oy, ox, ct, h, pix, lane);
      phase<1, 1>(a, wreg, tile, lbase, stg, n, oy, ox, ct, h, pix, lane);
    }
    if (more) store_tile(buf ^ 1);                       // (buf ^ 1 was last read before the previous barrier)
    __syncthreads();
  }
}

}  // namespace

extern "C" int lfd_conv3x3s2_dgrad_nhwc_f16(int32_t n, int32_t h, int32_t w, const void* dy, void* dx, const void* w_packed,
                                            const void* residual, lfd_stream_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!dy || !dx || !w_packed || dy == dx) return LFD_ERR_INVALID_ARGUMENT;
  if (n < 1 || h < 1 || w < 1 || !lfd_aligned16(dy) || !lfd_aligned16(dx) || !lfd_aligned16(residual)) return LFD_ERR_INVALID_ARGUMENT;
  DgArgs a{};
  a.dy = (const _Float16*)dy; a.dx = (_Float16*)dx; a.w = (const half8*)w_packed; a.res = (const _Float16*)residual;
  a.N = n; a.H = h; a.W = w;
  a.OH = (h - 1) / 2 + 1;
  a.OW = (w - 1) / 2 + 1;
  a.tiles_x = (a.OW + DG::TW - 1) / DG::TW;
  a.tiles_y = (a.OH + DG::TH - 1) / DG::TH;
  const long nt = (long)n * a.tiles_x * a.tiles_y;
  if (nt > 0x3fffffffL) return LFD_ERR_UNSUPPORTED;
  a.ntiles = (int)nt;
  static unsigned long long done_mask = 0;
  const int done_dev = lfd_device_ordinal();
  if (LFD_ONCE_PER_DEVICE(done_mask, done_dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dgrad_s2_64), hipFuncAttributeMaxDynamicSharedMemorySize,
                            DG::LDS_BYTES) != hipSuccess)
      return LFD_ERR_LAUNCH_FAILED;
    LFD_DONE_ON_DEVICE(done_mask, done_dev);
  }
  int blocks = 512;
  if (blocks > 8 * ((a.ntiles + 7) / 8)) blocks = 8 * ((a.ntiles + 7) / 8);
  hipLaunchKernelGGL(k_dgrad_s2_64, dim3(blocks), dim3(256), DG::LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
