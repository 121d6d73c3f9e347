#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// clean issue-cost microbenchmark: NI back-to-back 16 B/lane VMEM instructions per wave, addresses = base + i KB
template <int MODE, int NI>
__global__ __launch_bounds__(256) void k(const char* in, char* out, unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * 4 + wave) * (size_t)NI * 1024 + lane * 16;
  const char* p = in + base; char* q = out + base;
  u4 acc = {0, 0, 0, 0};
  u4 v = {1u, 2u, 3u, 4u};
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (MODE == 0) {
      unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(smem + wave * 16384 + (i & 15) * 1024));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p + i * 1024), "s"(m0v) : "memory");
    } else if (MODE == 1) {
      acc ^= *reinterpret_cast<const u4*>(p + i * 1024);
    } else if (MODE == 2) {
      *reinterpret_cast<u4*>(q + i * 1024) = v;
    } else {   // LDS: unaligned 16-B writes
      *reinterpret_cast<u4 __attribute__((aligned(2)))*>(smem + wave * 16384 + i * 1024 + lane * 16 + 2) = v;
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) { dbg[0] = t1 - t0; dbg[1] = t2 - t0; }
  if (acc.x == 12345) out[0] = 1;
}
template <int MODE, int NI>
void run(const char* name, int blocks, int threads, char* in, char* out, unsigned long long* dbg) {
  hipFuncSetAttribute((const void*)&k<MODE, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<MODE, NI><<<blocks, threads, 65536>>>(in, out, dbg);
  hipDeviceSynchronize();
  k<MODE, NI><<<blocks, threads, 65536>>>(in, out, dbg);
  hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-8s blocks %4d thr %4d NI %3d: issue %6llu cyc (%6.1f/instr)  complete %6llu cyc\n", name, blocks, threads, NI, h[0], (double)h[0] / NI, h[1]);
}
int main() {
  size_t cap = (size_t)1 << 28;
  char *in, *out; unsigned long long* dbg;
  hipMalloc(&in, cap); hipMalloc(&out, cap); hipMalloc(&dbg, 4096);
  hipMemset(in, 1, cap); hipMemset(out, 0, cap);
  for (int thr : {64, 256}) for (int blocks : {1, 256}) {
    run<0, 4>("dma", blocks, thr, in, out, dbg);   run<0, 16>("dma", blocks, thr, in, out, dbg);
    run<1, 4>("load", blocks, thr, in, out, dbg);  run<1, 16>("load", blocks, thr, in, out, dbg);
    run<2, 4>("store", blocks, thr, in, out, dbg); run<2, 16>("store", blocks, thr, in, out, dbg);
    run<3, 4>("ldsw_un", blocks, thr, in, out, dbg); run<3, 16>("ldsw_un", blocks, thr, in, out, dbg);
  }
  return 0;
}
