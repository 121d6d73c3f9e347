// Pure MFMA loop with CHANGING, random-looking fp16 operands (tools/ub/mfma_long.hip multiplies the same two fragments for ever:
// the multiplier arrays barely toggle).  Eight A and eight B fragments per lane, hashed from (thread, index) to values in
// [-1, 1), rotated so that consecutive MFMAs of a wave see different operands on both inputs; four independent accumulators
// as in mfma_long.hip.  Same instruction stream rate -- what differs is the power the chip needs for it, i.e. the clock it can
// hold under its 1.4 kW cap.  MODE 0: constant operands (control), 1: random operands, 2: random operands with the A fragment HELD for four consecutive MFMAs
// (weights-stationary order: one filter fragment against four pixel tiles), 3: the B fragment held for four.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float hashf(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (float)(x & 0xffff) / 32768.0f - 1.0f;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* dbg, int n) {
  half8 a[8], b[8];
  for (int j = 0; j < 8; ++j)
    for (int i = 0; i < 8; ++i) {
      const unsigned t = blockIdx.x * 512 + threadIdx.x;
      a[j][i] = MODE ? (_Float16)hashf(t * 131u + j * 17u + i) : (_Float16)(threadIdx.x * 0.001f + i);
      b[j][i] = MODE ? (_Float16)hashf(t * 257u + j * 29u + i + 7777u) : (_Float16)(i * 0.5f);
    }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; i += 2) {
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
      if (MODE == 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 3) & 7], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 6) & 7], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 1) & 7], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 4) & 7], c3, 0, 0, 0);
      } else if (MODE == 3) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 1], b[j], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 2], b[j], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 3], b[j], c3, 0, 0, 0);
      } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 3) & 7], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 1], b[(j + 6) & 7], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 2], b[(j + 1) & 7], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 3], b[(j + 4) & 7], c3, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { dbg[0] = t1 - t0; dbg[1] = r1 - r0; }
}
template <int MODE>
void run(float* out, unsigned long long* dbg) {
  const int blocks = 256, thr = 512, n = 2000000;
  for (int rep = 0; rep < 6; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<blocks, thr>>>(out, dbg, n); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, dbg, 16, hipMemcpyDeviceToHost);
    double fl = (double)blocks * (thr / 64) * n * 4 * 32768.0;
    printf("mode %d (%s operands): %8.1f us  %7.1f TFLOP/s  clock %.3f GHz (%.2f cycles per MFMA)\n", MODE, MODE == 0 ? "constant" : MODE == 1 ? "random" : MODE == 2 ? "random, A held x4" : "random, B held x4", ms * 1e3,
           fl / ms / 1e9, h[0] / (h[1] / 100.0) / 1e3, (double)h[0] / (4.0 * n));
  }
}
int main(int argc, char** argv) {
  float* out; unsigned long long* dbg; hipMalloc(&out, 1 << 22); hipMalloc(&dbg, 64);
  const int mode = argc > 1 ? atoi(argv[1]) : 1;
  if (mode == 1) run<1>(out, dbg); else if (mode == 2) run<2>(out, dbg); else if (mode == 3) run<3>(out, dbg); else run<0>(out, dbg);
  return 0;
}
