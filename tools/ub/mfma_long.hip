#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* dbg, int n) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { dbg[0] = t1 - t0; dbg[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* dbg; hipMalloc(&out, 1 << 22); hipMalloc(&dbg, 64);
  for (int thr : {512}) for (int blocks : {256}) for (int rep = 0; rep < 6; ++rep) {
    int n = 2000000;   // ~0.5 s per launch: long enough for rocm-smi to sample the steady state
    k<<<blocks, thr>>>(out, dbg, n); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<<<blocks, thr>>>(out, dbg, n); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, dbg, 16, hipMemcpyDeviceToHost);
    double fl = (double)blocks * (thr / 64) * n * 4 * 32768.0;
    printf("blocks %3d thr %3d: %8.1f us  %7.1f TFLOP/s | memtime ticks %llu (%.2f per MFMA), realtime ticks %llu (=%.1f us @100MHz) -> memtime %.3f GHz\n", blocks, thr, ms * 1e3, fl / ms / 1e9,
           h[0], (double)h[0] / (4.0 * n), h[1], h[1] / 100.0, h[0] / (h[1] / 100.0) / 1e3);
  }
  return 0;
}
