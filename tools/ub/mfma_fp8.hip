// v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands, unit scales) back to back on random operands, beside tools/ub/mfma_data.hip's
// fp16 loop: cycles per instruction, the clock the chip holds under its power cap, and therefore the energy of a K = 64 fp8
// product against four K = 16 fp16 ones -- the price list for computing the 2^-11 correction terms of the hi/lo planes mode
// (w_hi x_lo + w_lo x_hi) in fp8.   MODE 0: fp16 32x32x16 random (control), 1: fp8 32x32x64 random.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned hashu(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* dbg, int n) {
  const unsigned t = blockIdx.x * 512 + threadIdx.x;
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned long long t0, r0;
  if constexpr (MODE == 0) {
    half8 a[8], b[8];
    for (int j = 0; j < 8; ++j)
      for (int i = 0; i < 8; ++i) {
        a[j][i] = (_Float16)((float)(hashu(t * 131u + j * 17u + i) & 0xffff) / 32768.0f - 1.0f);
        b[j][i] = (_Float16)((float)(hashu(t * 257u + j * 29u + i + 7777u) & 0xffff) / 32768.0f - 1.0f);
      }
    t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i += 2) {
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + 3) & 7], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 1], b[(j + 6) & 7], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 2], b[(j + 1) & 7], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 3], b[(j + 4) & 7], c3, 0, 0, 0);
      }
    }
  } else {
    v8i a[8], b[8];
    for (int j = 0; j < 8; ++j)
      for (int i = 0; i < 8; ++i) {
        // random e4m3 bytes with the exponent kept off the NaN code (0x7f / 0xff) and small enough not to overflow the sums
        a[j][i] = (int)(hashu(t * 131u + j * 17u + i) & 0xb7b7b7b7u);
        b[j][i] = (int)(hashu(t * 257u + j * 29u + i + 7777u) & 0xb7b7b7b7u);
      }
    t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i += 2) {
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j], b[(j + 3) & 7], c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j + 1], b[(j + 6) & 7], c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j + 2], b[(j + 1) & 7], c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j + 3], b[(j + 4) & 7], c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
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
  const int blocks = 256, thr = 512, n = MODE ? 1000000 : 2000000;
  for (int rep = 0; rep < 6; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<MODE><<<blocks, thr>>>(out, dbg, n); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, dbg, 16, hipMemcpyDeviceToHost);
    const double kk = MODE ? 64.0 : 16.0;
    double fl = (double)blocks * (thr / 64) * n * 4 * 2.0 * 32 * 32 * kk;
    printf("mode %d (%s): %8.1f us  %7.1f TFLOP/s  clock %.3f GHz (%.2f cycles per MFMA per wave, two waves per SIMD)\n", MODE,
           MODE ? "fp8 32x32x64, random" : "fp16 32x32x16, random", ms * 1e3, fl / ms / 1e9, h[0] / (h[1] / 100.0) / 1e3, (double)h[0] / (4.0 * n));
  }
}
int main(int argc, char** argv) {
  float* out; unsigned long long* dbg; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&dbg, 64);
  const int mode = argc > 1 ? atoi(argv[1]) : 1;
  if (mode) run<1>(out, dbg); else run<0>(out, dbg);
  return 0;
}
