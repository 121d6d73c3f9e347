// tools/ub/pair_kloop.hip -- how fast can the contraction of one 4 x 16-pixel tile of the 3x3 64-channel planes conv run
// (csrc/planes_c3.hip: 108 MFMAs per wave, weights in registers, operand fragments through a register ring from LDS) when
//   NW = 4: one wave per SIMD does all 36 k-steps (the shipped kernel's structure: 512 registers per wave), or
//   NW = 8: two waves per SIMD do 18 k-steps each (144 weight registers per wave, <= 256 in total)
// -- nothing but the k loop and one barrier per tile: the floor of either structure.   hipcc --offload-arch=gfx950 -O3 -o pair_kloop pair_kloop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k(const half8* w, float* out, unsigned long long* dbg, int ntiles) {
  constexpr int NKW = 36 * 4 / NW;            // k-steps per wave
  constexpr int IN_BYTES = 14336;
  __shared__ __attribute__((aligned(16))) char smem[2 * IN_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = wave & 1, pg = (wave >> 1) & 1, kpart = wave >> 2;
  const int h = lane >> 5, pix = lane & 31, oyl = pix >> 4, oxl = pix & 15;
  for (int i = threadIdx.x; i < 2 * IN_BYTES / 4; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
  half8 wh[NKW], wl[NKW];
#pragma unroll
  for (int k = 0; k < NKW; ++k) { wh[k] = w[(ct * 72 + kpart * NKW + k) * 64 + lane]; wl[k] = w[(ct * 72 + 36 + kpart * NKW + k) * 64 + lane]; }
  int xoff[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int ix = oxl + s, f = (ix >> 1) & 7, rowbase = (pg * 2 + oyl) * 18 + ix;
#pragma unroll
    for (int q = 0; q < 4; ++q) xoff[s][q] = rowbase * 128 + (((2 * q + h) ^ f) * 16);
  }
  __syncthreads();
  f32x16 am = {0}, ac = {0};
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto xaddr = [&](int kk) {
      const int k = kpart * NKW + kk;
      const int r = k / 12, s = (k / 4) % 3, q = k % 4;
      return smem + xoff[s][q] + r * 18 * 128;
    };
    constexpr int PD = 3;
    half8 xqh[PD + 1], xql[PD + 1];
#pragma unroll
    for (int k = 0; k < PD; ++k) { xqh[k] = *reinterpret_cast<const half8*>(xaddr(k)); xql[k] = *reinterpret_cast<const half8*>(xaddr(k) + IN_BYTES); }
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      if (k + PD < NKW) {
        xqh[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(xaddr(k + PD));
        xql[(k + PD) % (PD + 1)] = *reinterpret_cast<const half8*>(xaddr(k + PD) + IN_BYTES);
      }
      __builtin_amdgcn_sched_barrier(0);
      am = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xqh[k % (PD + 1)], am, 0, 0, 0);
      ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[k], xql[k % (PD + 1)], ac, 0, 0, 0);
      ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[k], xqh[k % (PD + 1)], ac, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += am[i] + ac[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 100) { dbg[0] = t1 - t0; dbg[1] = r1 - r0; }
}

int main() {
  half8* w; float* out; unsigned long long* dbg;
  hipMalloc(&w, 2 * 72 * 64 * 16); hipMalloc(&out, 1 << 22); hipMalloc(&dbg, 64);
  hipMemset(w, 0x3c, 2 * 72 * 64 * 16);
  const int ntiles = 400;
  for (int nw : {4, 8}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (nw == 4) k<4><<<256, 256>>>(w, out, dbg, ntiles); else k<8><<<256, 512>>>(w, out, dbg, ntiles);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hh[2]; hipMemcpy(hh, dbg, 16, hipMemcpyDeviceToHost);
      double fl = 256.0 * 4 * 108 * 32768.0 * ntiles;
      printf("NW %d: %8.1f us, %6.1f TFLOP/s, %7.1f cycles per tile (MFMA-bound: 3456), clock %.2f GHz\n", nw, ms * 1e3, fl / ms / 1e9,
             (double)hh[0] / ntiles, hh[0] / (hh[1] / 100.0) / 1e3);
    }
  }
  return 0;
}
