// csrc/block128.hip -- a whole 128-channel FasterBlock without downsample branch (lfd_resnet.py:96-154) on the SMALL maps
// of the last backbone stage in ONE launch:   out = relu( conv3x3(relu(conv3x3(in, w1) + b1), w2) + b2 + in ).
//
// Why: the two launches of such a block are k_conv128_splitk (conv_small.hip), ~6.7 us each at 8 x 17 x 30 and ~5.3 us at
// 1 x 17 x 30 -- launch + one round trip for the filter quarter + 36 MFMAs + the partial-sum exchange: latency, not work.
// Here a workgroup owns 4 x 8 output pixels and ALL 128 channels: it recomputes the 6 x 10 halo of the intermediate map
// (60 pixels = two 32-pixel MFMA tiles) into LDS and contracts conv2 from there, so the intermediate map never reaches HBM
// and the second launch disappears.  The price is filter traffic: every wave streams its 32-channel slab of BOTH filters
// (2 x 72 KB) through a ring of R = 48 fragments in AccVGPRs, R loads in flight per wave from the first instruction on; the
// L2 -> CU path (64 B / clk: 576 KB per workgroup = 4.4 us) is the floor of a workgroup, the 216 MFMAs per wave hide under it.
//
// Measured (round 4): in a graphed chain of blocks 9.4 vs 10.8 us per block at 1 x 17 x 30, 10.2 vs 13.4 at 8 x 17 x 30, 20.5 vs
// 30.0 at 16 x 23 x 40 (tools/timing/block128_in_graph.py); inside the network the forward of one 1080p frame gains what
// the chain predicts (0.2213 -> 0.2197 ms, two blocks), the batch of eight ~1 us instead of 6.  (Mapping the split-K
// kernel's slabs onto XCDs so that an XCD streams one slab instead of four changed nothing either.)
//
// Numerics: bit-identical to two k_conv128_splitk launches.  That kernel splits K into four quarters of 18 k-steps and adds
// the quarters as ((q0 + q1) + q2) + q3, then + bias, then + residual; a wave here contracts the quarters one after the other
// (conv1: both pixel tiles per fragment; conv2: two quarters interleaved so that consecutive MFMAs are independent) and adds
// them in the same order with the same operations.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct B128Args {
  const _Float16* in;   // [N,H,W,128]
  _Float16* out;        // [N,H,W,128]
  const half8* w1;      // packed [4 slabs][72 k-steps][64 lanes] (ops.pack_conv_weight)
  const float* b1;
  const half8* w2;
  const float* b2;
  int N, H, W;
  int tiles_x, tiles_y;
};

constexpr int TH = 4, TW = 8;                 // output pixels of a workgroup
constexpr int MW = TW + 2;                    // intermediate halo: 6 x 10 pixels, slot = row * 10 + column
constexpr int MSLOTS = 64;
constexpr int IH = TH + 4, IW = TW + 4;       // input halo: 8 x 12 pixels
constexpr int PITCH = 272;                    // bytes per pixel: 128 channels x 2 B + 16 B (bank spread)
constexpr int IN_BYTES = IH * IW * PITCH;     // 26112
constexpr int MID_BYTES = MSLOTS * PITCH;     // 17408
constexpr int BIAS_BYTES = 256 * 4;            // b1 | b2
constexpr int LDS_BYTES = IN_BYTES + MID_BYTES + BIAS_BYTES;
#ifndef B128_RING
#define B128_RING 48
#endif
constexpr int R = B128_RING;                  // filter fragments in flight per wave (1 KB each per wave)
constexpr int NSEQ = 144;                     // 72 k-steps of conv1, then 72 of conv2

// compile-time repetition: the body sees its index as a constant expression (inline-asm immediates, register-array indices)
#define B128_REP4(M, b) M((b)) M((b) + 1) M((b) + 2) M((b) + 3)
#define B128_REP12(M, b) B128_REP4(M, (b)) B128_REP4(M, (b) + 4) B128_REP4(M, (b) + 8)
#define B128_REP24(M, b) B128_REP12(M, (b)) B128_REP12(M, (b) + 12)
#define B128_REP72(M, b) B128_REP24(M, (b)) B128_REP24(M, (b) + 24) B128_REP24(M, (b) + 48)

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// k-step of the i-th fragment a wave uses (i < 72: conv1 in order; then conv2 with two quarters interleaved)
__device__ __forceinline__ constexpr int seq_k(int i) {
  return i < 72 ? i : 36 * ((i - 72) / 36) + 18 * ((i - 72) & 1) + ((i - 72) % 36) / 2;
}

__global__ __launch_bounds__(256) void k_block128(const B128Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_mid = smem + IN_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 31, kh = lane >> 5;
  const int slab = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, n = t / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;

  // ---- input halo tile (8 x 12 pixels x 128 channels = 1536 16-byte chunks, 6 per thread) and the biases: requested first,
  //      by inline asm like the filter ring below, so that the wait in front of the LDS stores can be COUNTED (the R ring
  //      loads issued after them stay in flight; a compiler-visible load here would be awaited with vmcnt(0): 192 KB per CU
  //      before the first MFMA)
  u32x4 iv[6];
  bool iok[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int i = tid + 256 * it;
    const int pix = i >> 4, ck = i & 15;
    const int iy = pix / IW, ix = pix - iy * IW;
    const int gy = y0 - 2 + iy, gx = x0 - 2 + ix;
    iok[it] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const uint4* src = reinterpret_cast<const uint4*>(a.in + (((size_t)n * a.H + (iok[it] ? gy : 0)) * a.W + (iok[it] ? gx : 0)) * 128) + ck;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(iv[it]) : "v"(src));
  }
  float bias_v;
  {
    const float* src = tid < 128 ? a.b1 + tid : a.b2 + (tid - 128);
    asm volatile("global_load_dword %0, %1, off" : "=v"(bias_v) : "v"(src));
  }
  // ---- the filter ring: R fragments (1 KB per wave each) of this wave's slabs in flight from here on.  The loads are inline
  //      asm straight into AccVGPRs (the MFMAs read them there), in the order the fragments are used, so `vmcnt` can be
  //      counted: before fragment i is used, at most min(R - 1, 143 - i) younger ring loads may be outstanding.  (Left to the
  //      compiler the re-loads sink to just before their use: vmcnt(1) everywhere, one round trip per k-step.)  Loads the
  //      compiler issues itself only make these waits stricter; none are pending inside the loops (biases live in LDS).
  const char* w1u = reinterpret_cast<const char*>(a.w1 + (size_t)slab * 72 * 64);
  const char* w2u = reinterpret_cast<const char*>(a.w2 + (size_t)slab * 72 * 64);
  const int lane16 = lane * 16;
  u32x4 wf[R];
#define B128_LOAD(i_)                                                                                          \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(wf[(i_) % R])                                           \
               : "v"(lane16), "s"(((i_) < 72 ? w1u : w2u) + (size_t)seq_k(i_) * 1024))
#define B128_WAIT(i_)                                                                                          \
  asm volatile("s_waitcnt vmcnt(%1)" : "+a"(wf[(i_) % R]) : "n"((R - 1) < (NSEQ - 1 - (i_)) ? (R - 1) : (NSEQ - 1 - (i_))))
#define B128_FIRST(i_) if constexpr ((i_) < R) B128_LOAD(i_);
  B128_REP72(B128_FIRST, 0)
  asm volatile("s_waitcnt vmcnt(%7)"
               : "+v"(iv[0]), "+v"(iv[1]), "+v"(iv[2]), "+v"(iv[3]), "+v"(iv[4]), "+v"(iv[5]), "+v"(bias_v) : "n"(R));
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int i = tid + 256 * it;
    *reinterpret_cast<u32x4*>(s_in + (i >> 4) * PITCH + (i & 15) * 16) = iok[it] ? iv[it] : u32x4{0u, 0u, 0u, 0u};
  }
  reinterpret_cast<float*>(smem + IN_BYTES + MID_BYTES)[tid] = bias_v;
  const float* s_b1 = reinterpret_cast<const float*>(smem + IN_BYTES + MID_BYTES);
  const float* s_b2 = s_b1 + 128;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- conv1 on the 6 x 10 halo of the intermediate map: lane p of MFMA tile pt = slot 32 pt + p
  int lb1[2];
  bool mid_ok[2];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int slot = 32 * pt + p;
    const int r = slot / MW, c = slot - r * MW;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    mid_ok[pt] = slot < 6 * MW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    lb1[pt] = (slot < 6 * MW ? r * IW + c : 0) * PITCH + kh * 16;
  }
  // (the B operands of k-step i + 1 are requested before the MFMAs of k-step i: LDS latency under the matrix pipe)
  auto koff1 = [](int k) {
    const int tap = k >> 3, cq = k & 7, dy = tap / 3, dx = tap - dy * 3;
    return (dy * IW + dx) * PITCH + cq * 32;
  };
  f32x16 S[2], A[2];
  half8 bn[2];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) bn[pt] = *reinterpret_cast<const half8*>(s_in + lb1[pt] + koff1(0));
#define B128_STEP1(i_) {                                                                                          \
    constexpr int i = (i_);                                                                                       \
    const half8 b0 = bn[0], b1 = bn[1];                                                                           \
    if constexpr (i + 1 < 72) {                                                                                   \
      bn[0] = *reinterpret_cast<const half8*>(s_in + lb1[0] + koff1(i + 1));                                      \
      bn[1] = *reinterpret_cast<const half8*>(s_in + lb1[1] + koff1(i + 1));                                      \
    }                                                                                                             \
    B128_WAIT(i);                                                                                                 \
    if constexpr (i % 18 == 0) { A[0] = zero16(); A[1] = zero16(); }                                              \
    A[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wf[i % R]), b0, A[0], 0, 0, 0);       \
    A[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wf[i % R]), b1, A[1], 0, 0, 0);       \
    if constexpr (i + R < NSEQ) B128_LOAD(i + R);                                                                 \
    if constexpr (i == 17) { S[0] = A[0]; S[1] = A[1]; }                                                          \
    else if constexpr (i % 18 == 17) { S[0] = S[0] + A[0]; S[1] = S[1] + A[1]; }                                  \
  }
  B128_REP72(B128_STEP1, 0)
  // bias -> ReLU -> fp16 -> LDS; zero outside the image (conv2's padding)
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b4 = *reinterpret_cast<const float4*>(s_b1 + slab * 32 + 8 * j + 4 * kh);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = S[pt][4 * j + e] + bb[e];
        v[e] += 0.f;                                  // k_conv128_splitk adds its (absent) residual: -0 -> +0
      }
      uint2 o;
      o.x = lfd_cvt_pk_max(v[0], v[1], LFD_PK_RELU);
      o.y = lfd_cvt_pk_max(v[2], v[3], LFD_PK_RELU);
      if (!mid_ok[pt]) o = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(s_mid + (32 * pt + p) * PITCH + (slab * 32 + 8 * j + 4 * kh) * 2) = o;
    }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ---- conv2 on the 4 x 8 output pixels: lane p = row p >> 3, column p & 7
  const int orow = p >> 3, ocol = p & 7;
  const int lb2 = (orow * MW + ocol) * PITCH + kh * 16;
  auto koff2 = [](int k) {
    const int tap = k >> 3, cq = k & 7, dy = tap / 3, dx = tap - dy * 3;
    return (dy * MW + dx) * PITCH + cq * 32;
  };
  f32x16 S2;
  half8 bq = *reinterpret_cast<const half8*>(s_mid + lb2 + koff2(seq_k(72)));
#define B128_STEP2(i_) {                                                                                          \
    constexpr int i = (i_);                                                                                       \
    constexpr int jj = (i - 72) % 36, h = jj & 1;                                                                 \
    const half8 b = bq;                                                                                           \
    if constexpr (i + 1 < NSEQ) bq = *reinterpret_cast<const half8*>(s_mid + lb2 + koff2(seq_k(i + 1)));          \
    B128_WAIT(i);                                                                                                 \
    if constexpr (jj < 2) A[h] = zero16();                                                                        \
    A[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wf[i % R]), b, A[h], 0, 0, 0);        \
    if constexpr (i + R < NSEQ) B128_LOAD(i + R);                                                                 \
    if constexpr (i == 107) S2 = A[0] + A[1];                                                                     \
    else if constexpr (i == 143) S2 = (S2 + A[0]) + A[1];                                                         \
  }
  B128_REP72(B128_STEP2, 72)
  const int oy = y0 + orow, ox = x0 + ocol;
  const bool ook = oy < a.H && ox < a.W;
  _Float16* op = a.out + (((size_t)n * a.H + (ook ? oy : 0)) * a.W + (ook ? ox : 0)) * 128;
  const char* idp = s_in + ((orow + 2) * IW + ocol + 2) * PITCH;     // the identity: the input's centre pixels
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c0 = slab * 32 + 8 * j + 4 * kh;
    const half4 rv = *reinterpret_cast<const half4*>(idp + c0 * 2);
    const float4 b4 = *reinterpret_cast<const float4*>(s_b2 + c0);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = S2[4 * j + e] + bb[e];
      v[e] += (float)rv[e];
    }
    uint2 o;
    o.x = lfd_cvt_pk_max(v[0], v[1], LFD_PK_RELU);
    o.y = lfd_cvt_pk_max(v[2], v[3], LFD_PK_RELU);
    if (ook) *reinterpret_cast<uint2*>(op + c0) = o;
  }
}

}  // namespace

extern "C" int lfd_fasterblock128_fused_f16(int32_t n, int32_t h, int32_t w, const void* in, void* out, const void* w1_packed,
                                            const float* b1, const void* w2_packed, const float* b2, lfd_stream_t stream) {
  if (n < 1 || h < 1 || w < 1 || !in || !out || !w1_packed || !b1 || !w2_packed || !b2) return LFD_ERR_INVALID_ARGUMENT;
  if (in == out) return LFD_ERR_INVALID_ARGUMENT;
  if ((((uintptr_t)in) | ((uintptr_t)out) | ((uintptr_t)w1_packed) | ((uintptr_t)w2_packed) | ((uintptr_t)b1) | ((uintptr_t)b2)) & 15)
    return LFD_ERR_INVALID_ARGUMENT;
  B128Args a{};
  a.in = (const _Float16*)in; a.out = (_Float16*)out;
  a.w1 = (const half8*)w1_packed; a.b1 = b1; a.w2 = (const half8*)w2_packed; a.b2 = b2;
  a.N = n; a.H = h; a.W = w;
  a.tiles_x = (w + TW - 1) / TW;
  a.tiles_y = (h + TH - 1) / TH;
  const long tiles = (long)a.tiles_x * a.tiles_y * n;
  if (tiles > 0x7fffffffL) return LFD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_block128, dim3((unsigned)tiles), dim3(256), LDS_BYTES, reinterpret_cast<hipStream_t>(stream), a);
  LFD_CHECK_LAUNCH();
  return LFD_OK;
}
