#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* dbg, float seed) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = seed * (threadIdx.x + i);
  uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int rep = 0; rep < 32; ++rep) {
    if (MODE == 0) {          // 8 independent cvt_pk + pk_max per rep (16 instrs)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        uint32_t v;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v) : "v"(x[2 * e]), "v"(x[2 * e + 1]));
        asm volatile("v_pk_max_f16 %0, %1, 0" : "=v"(v) : "v"(v));
        r[e] ^= v;
      }
    } else if (MODE == 1) {   // 16 independent v_add_f32
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[e]) : "v"(x[e]), "v"(seed));
    } else if (MODE == 2) {   // dependent chain of 16 v_add_f32
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[0]) : "v"(x[0]), "v"(seed));
    } else if (MODE == 3) {   // 16 independent v_and_b32
#pragma unroll
      for (int e = 0; e < 8; ++e) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(r[e]) : "v"(r[e]), "v"(0x7fffffff)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r[e]) : "v"(r[e]), "v"(0x55)); }
    } else if (MODE == 4) {   // 16 v_mul_lo_u32
#pragma unroll
      for (int e = 0; e < 8; ++e) { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r[e]) : "v"(r[e]), "v"(12345)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r[e]) : "v"(r[e]), "v"(777)); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) dbg[0] = t1 - t0;
}
template <int MODE> void run(const char* n, int thr, float* out, unsigned long long* dbg) {
  k<MODE><<<256, thr>>>(out, dbg, 1.5f); hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, dbg, 8, hipMemcpyDeviceToHost);
  printf("%-28s thr %3d: %6llu cycles for 512 instrs -> %.2f cyc/instr\n", n, thr, h, h / 512.0);
}
int main() {
  float* out; unsigned long long* dbg; hipMalloc(&out, 1 << 20); hipMalloc(&dbg, 64);
  for (int thr : {64, 256, 512}) {
    run<0>("cvt_pk+pk_max independent", thr, out, dbg);
    run<1>("v_add_f32 independent", thr, out, dbg);
    run<2>("v_add_f32 dependent chain", thr, out, dbg);
    run<3>("v_and/v_xor (pairs dep)", thr, out, dbg);
    run<4>("v_mul_lo_u32 (pairs dep)", thr, out, dbg);
  }
  return 0;
}
