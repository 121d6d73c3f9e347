// What the LDS operand reads of the weights-stationary kernels cost in POWER: the fp16 MFMA loop of tools/ub/mfma_data.hip on random
// operands with the B fragments (a) rotating through registers, (b) read from LDS at the planes kernels' ratio -- two ds_read_b128
// per three MFMAs (x_hi is used twice, x_lo once) -- (c) one read per three MFMAs (what a wave that contracts BOTH 32-row slabs of a
// 64-channel conv against one fragment would need).  Same instruction count in the matrix pipe; the clock the chip holds under its
// 1.4 kW cap is the measurement.     mfma_lds <mode 0|1|2>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float hashf(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (float)(x & 0xffff) / 32768.0f - 1.0f;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* dbg, int n) {
  __shared__ __attribute__((aligned(16))) _Float16 tile[2 * 72 * 512];     // 72 fragments x 64 lanes x 8 halfs, two "planes" (144 KB)
  const unsigned t = blockIdx.x * 512 + threadIdx.x;
  for (int i = threadIdx.x; i < 2 * 72 * 512; i += 512) tile[i] = (_Float16)hashf(t * 31u + i);
  half8 a[12];
  for (int j = 0; j < 12; ++j)
    for (int i = 0; i < 8; ++i) a[j][i] = (_Float16)hashf(t * 131u + j * 17u + i);
  half8 b[6];
  for (int j = 0; j < 6; ++j)
    for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)hashf(t * 257u + j * 29u + i + 7777u);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const half8* lh = reinterpret_cast<const half8*>(tile) + lane;                 // "hi plane": fragment f at lh[f * 64]
  const half8* ll = reinterpret_cast<const half8*>(tile) + 72 * 64 + lane;       // "lo plane"
  f32x16 m0 = {0}, c0 = {0}, m1 = {0}, c1 = {0};
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int f = 0; f < 12; f += 2) {
      // two k-steps: (w_hi x_hi | w_hi x_lo + w_lo x_hi) each = 6 MFMAs
      half8 xh0, xl0, xh1, xl1;
      const int ff = (f + 12 * (i & 3) + 6 * (wave & 1)) % 72;
      if (MODE == 0) {
        xh0 = b[f % 6]; xl0 = b[(f + 1) % 6]; xh1 = b[(f + 2) % 6]; xl1 = b[(f + 3) % 6];
      } else if (MODE == 1) {
        xh0 = lh[ff * 64]; xl0 = ll[ff * 64]; xh1 = lh[(ff + 1) * 64]; xl1 = ll[(ff + 1) * 64];
      } else {
        xh0 = lh[ff * 64]; xl0 = ll[ff * 64]; xh1 = b[(f + 2) % 6]; xl1 = b[(f + 3) % 6];
      }
      m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[f], xh0, m0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[f], xl0, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(f + 5) % 12], xh0, c0, 0, 0, 0);
      m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[f + 1], xh1, m1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[f + 1], xl1, c1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(f + 7) % 12], xh1, c1, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += m0[i] + c0[i] + m1[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { dbg[0] = t1 - t0; dbg[1] = r1 - r0; }
}
template <int MODE>
void run(float* out, unsigned long long* dbg) {
  const int blocks = 256, thr = 512, n = 160000;
  for (int rep = 0; rep < 5; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<MODE><<<blocks, thr>>>(out, dbg, n); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, dbg, 16, hipMemcpyDeviceToHost);
    const double nm = (double)blocks * (thr / 64) * n * 36.0;
    printf("mode %d (%s): %8.1f us  %6.1f G MFMA/s  %7.1f TFLOP/s  clock %.3f GHz (%.1f cycles per MFMA per wave)\n", MODE,
           MODE == 0 ? "B fragments in registers" : MODE == 1 ? "2 ds_read_b128 per 3 MFMAs" : "1 ds_read_b128 per 3 MFMAs", ms * 1e3, nm / ms / 1e6,
           nm * 32768.0 / ms / 1e9, h[0] / (h[1] / 100.0) / 1e3, (double)h[0] / (36.0 * n));
  }
}
int main(int argc, char** argv) {
  float* out; unsigned long long* dbg; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&dbg, 64);
  const int mode = argc > 1 ? atoi(argv[1]) : 1;
  if (mode == 0) run<0>(out, dbg); else if (mode == 1) run<1>(out, dbg); else run<2>(out, dbg);
  return 0;
}
