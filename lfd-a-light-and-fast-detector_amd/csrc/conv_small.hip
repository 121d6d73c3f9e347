// csrc/conv_small.hip -- conv3x3 stride 1, 128 -> 128 channels, NHWC fp16, fused bias (+ residual) + ReLU on SMALL maps:
// the last stage of the LFD backbones (FasterBlock convs at 17 x 30 for WIDERFACE_LFD_S @1080p, lfd_resnet.py:96-154;
// 5 of the 6 launches of that stage).
//
// Why a second kernel for this shape: the map is 510 pixels per image -- 1.2 GFLOP per launch at batch 8 -- and the 295 KB
// filter is larger than the activations.  The generic streamed-weight kernel (conv_impl.h, k_conv<128,3,1,4,false>) gives a
// workgroup 64 pixels x all 128 output channels, i.e. every workgroup streams the WHOLE filter from L2 (64 workgroups x
// 288 KB at batch 8, 8 workgroups at batch 1) and every wave contracts all 72 k-steps: 8.0 us per launch at batch 8, 7.8 us
// at batch 1 -- latency of one long dependent chain, 0.06 of the MFMA roof.  Here the work is cut the
// other way:
//   * a workgroup owns 64 pixels (4 rows x 16 columns) and ONE 32-channel slab of the output (grid.y = 4 slabs): it
//     streams 72 KB of filter, and 4 x as many workgroups are in flight (320 at batch 8: the whole chip; 40 at batch 1);
//   * SPLIT-K over the four waves: wave w contracts k-steps 18 w .. 18 w + 17 (a quarter of the 9 taps x 8 channel groups)
//     for both 32-pixel MFMA tiles -- 36 MFMAs per wave instead of 144, its 18 filter fragments (18 KB) requested up front;
//   * the partial accumulators meet in LDS (32 KB) and are added in a FIXED order ((w0 + w1) + w2) + w3, so results do not
//     depend on scheduling; wave w finishes a quarter of the outputs: bias (+ residual) -> ReLU -> fp16 -> stores.
// Numerics: fp32 accumulation like the generic kernel, but the sum over k is associated per quarter -- outputs can differ
// from k_conv's by one fp16 ulp; both stay within the per-launch tolerance of tests/test_gpu_parity_fullsize.py.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SKArgs {
  const _Float16* in;   // [N,H,W,128]
  _Float16* out;        // [N,H,W,128]
  const half8* w;       // packed [4 slabs][72 k-steps][64 lanes] (ops.pack_conv_weight)
  const float* bias;    // [128]
  const _Float16* res;  // [N,H,W,128] or null
  int N, H, W, relu;
  int tiles_x, tiles_y;
};

constexpr int TH = 4, TW = 16, IH = TH + 2, IW = TW + 2;
constexpr int PITCH = 272;                              // bytes per pixel: 128 channels x 2 B + 16 B (bank spread)
constexpr int IN_BYTES = ((IH * IW * PITCH + 255) / 256) * 256;
constexpr int PART_BYTES = 4 * 2 * 16 * 64 * 4;
constexpr int LDS_BYTES = IN_BYTES + PART_BYTES;

__global__ __launch_bounds__(256) void k_conv128_splitk(const SKArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = reinterpret_cast<float*>(smem + IN_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 31, kh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.y;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, n = t / a.tiles_y;

  // ---- this wave's quarter of the filter slab: 18 fragments requested before anything else
  half8 wf[18];
  const half8* wsrc = a.w + ((size_t)slab * 72 + wv * 18) * 64 + lane;
#pragma unroll
  for (int k = 0; k < 18; ++k) wf[k] = wsrc[(size_t)k * 64];

  // ---- what the epilogue needs from global memory, requested now: its latency hides under everything below
  const int ept = wv >> 1, ehf = wv & 1;
  const int oy = ty * TH + ept * 2 + (p >> 4), ox = tx * TW + (p & 15);
  const bool ook = oy < a.H && ox < a.W;
  const size_t opix = (((size_t)n * a.H + (ook ? oy : 0)) * a.W + (ook ? ox : 0)) * 128;
  float4 bv[2];
  half4 rv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c0 = slab * 32 + 8 * (2 * ehf + j) + 4 * kh;
    bv[j] = *reinterpret_cast<const float4*>(a.bias + c0);
    rv[j] = a.res ? *reinterpret_cast<const half4*>(a.res + opix + c0) : half4{0, 0, 0, 0};
  }

  // ---- input halo tile (6 x 18 pixels x 128 channels) -> LDS; zero outside the image (the conv's padding).  All seven
  //      16-byte loads of a thread are requested before the first LDS store (a rolled loop is a chain of round trips).
  {
    constexpr int NIT = (IH * IW * 16 + 255) / 256;
    uint4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      const int pix = (i < IH * IW * 16 ? i : 0) >> 4, ck = i & 15;
      const int iy = pix / IW, ix = pix - iy * IW;
      const int gy = ty * TH - 1 + iy, gx = tx * TW - 1 + ix;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      v[it] = reinterpret_cast<const uint4*>(a.in + (((size_t)n * a.H + (ok ? gy : 0)) * a.W + (ok ? gx : 0)) * 128)[ck];
      if (!ok) v[it] = make_uint4(0u, 0u, 0u, 0u);       // unconditional load + select (a conditional load is a branch)
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      if (i < IH * IW * 16) *reinterpret_cast<uint4*>(smem + (i >> 4) * PITCH + (i & 15) * 16) = v[it];
    }
  }
  __syncthreads();

  f32x16 acc[2];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
  // pixel of lane p in MFMA tile pt: row 2 pt + (p >> 4), column p & 15; tap / channel group offsets are wave-uniform
  const int lbase = ((p >> 4) * IW + (p & 15)) * PITCH + kh * 16;
#pragma unroll
  for (int kk = 0; kk < 18; ++kk) {
    const int k = wv * 18 + kk;                         // k-step = tap * 8 + q
    const int tap = k >> 3, q = k & 7;
    const int dy = tap / 3, dx = tap - dy * 3;
    const int koff = (dy * IW + dx) * PITCH + q * 32;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const half8 b = *reinterpret_cast<const half8*>(smem + lbase + pt * 2 * IW * PITCH + koff);
      acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], b, acc[pt], 0, 0, 0);
    }
  }
  // ---- partial sums -> LDS [wave][pt][register][lane]
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[((wv * 2 + pt) * 16 + r) * 64 + lane] = acc[pt][r];
  __syncthreads();

  // ---- wave w finishes MFMA tile w >> 1, registers 8 (w & 1) .. + 7: fixed-order sum over the four K quarters
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c0 = slab * 32 + 8 * (2 * ehf + j) + 4 * kh;      // register 4 (2 hf + j) + e of lane (kh, p) = channel c0 + e
    const float bb[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 8 * ehf + 4 * j + e;
      const float s01 = part[((0 * 2 + ept) * 16 + r) * 64 + lane] + part[((1 * 2 + ept) * 16 + r) * 64 + lane];
      const float s012 = s01 + part[((2 * 2 + ept) * 16 + r) * 64 + lane];
      v[e] = (s012 + part[((3 * 2 + ept) * 16 + r) * 64 + lane]) + bb[e];
      v[e] += (float)rv[j][e];
    }
    const uint32_t lo2 = a.relu ? LFD_PK_RELU : LFD_PK_NONE;
    uint2 o;
    o.x = lfd_cvt_pk_max(v[0], v[1], lo2);
    o.y = lfd_cvt_pk_max(v[2], v[3], lo2);
    if (ook) *reinterpret_cast<uint2*>(a.out + opix + c0) = o;
  }
}

}  // namespace

// called by conv_dispatch (conv.hip) for cin = cout = 128, 3x3 stride 1, no chained 1x1, maps of at most kSmallPixels pixels
int lfd_conv128_splitk_launch(const _Float16* in, _Float16* out, const void* w_packed, const float* bias, const _Float16* res,
                              int n, int h, int w, int relu, hipStream_t st) {
  SKArgs a{};
  a.in = in; a.out = out; a.w = (const half8*)w_packed; a.bias = bias; a.res = res;
  a.N = n; a.H = h; a.W = w; a.relu = relu;
  a.tiles_x = (w + TW - 1) / TW;
  a.tiles_y = (h + TH - 1) / TH;
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  if (tiles < 1 || tiles > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_conv128_splitk, dim3((unsigned)tiles, 4), dim3(256), LDS_BYTES, st, a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
