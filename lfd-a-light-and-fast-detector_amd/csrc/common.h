// csrc/common.h -- shared helpers for the gfx950 kernels of liblfd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/lfd_hip.h"

#define LFD_WAVE 64

#define LFD_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return LFD_ERR_LAUNCH_FAILED;     \
  } while (0)

static inline size_t lfd_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves sub-buffers out of a caller-owned workspace (256-B aligned).
struct LfdCarver {
  char* base;
  size_t off;
  explicit LfdCarver(void* p) : base(reinterpret_cast<char*>(p)), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = lfd_align_up(off, 256);
    T* r = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return r;
  }
  size_t used() const { return lfd_align_up(off, 256); }
};

__device__ __forceinline__ int lfd_lane() { return threadIdx.x & 63; }

// Monotone float -> uint32 map (a < b  <=>  ord(a) < ord(b) for non-NaN floats).
__device__ __forceinline__ uint32_t lfd_float_ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float lfd_ord_float(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__device__ __forceinline__ float lfd_load_f(const void* p, int64_t i, int dtype) {
  return dtype == LFD_F16 ? __half2float(reinterpret_cast<const __half*>(p)[i])
                          : reinterpret_cast<const float*>(p)[i];
}
