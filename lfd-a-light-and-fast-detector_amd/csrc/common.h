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

// NHWC fp16 tensors cross the ABI 16-byte aligned (include/lfd_hip.h, conventions): the kernels move them as 16-byte vectors
// and LDS-DMA lines; a misaligned pointer is a status code, not a memory fault
static inline bool lfd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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

// Epilogue helper: two fp32 -> packed fp16 (round to nearest even, v_cvt_pk_f16_f32), then max with a packed
// lower bound (v_pk_max_f16): lo2 = 0x00000000 is ReLU, 0xfc00fc00 (-inf, -inf) is the identity.  Rounding is
// monotone and 0 is exact, so max(round(x), 0) == round(max(x, 0)); two instructions per PAIR instead of the
// three per ELEMENT that fmaxf + a scalar conversion compile to (fmaxf also canonicalises its input).
typedef float lfd_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 lfd_f16x2 __attribute__((ext_vector_type(2)));
#define LFD_PK_RELU 0x00000000u
#define LFD_PK_NONE 0xfc00fc00u
__device__ __forceinline__ uint32_t lfd_cvt_pk_max(float x, float y, uint32_t lo2) {
  lfd_f32x2 f; f[0] = x; f[1] = y;
  union { lfd_f16x2 v; uint32_t u; } r, l;
  r.v = __builtin_convertvector(f, lfd_f16x2);
  l.u = lo2;
  r.v = __builtin_elementwise_max(r.v, l.v);
  return r.u;
}

// relu(x * a + b) of 8 fp16 values with per-element fp32 (a, b) -> 8 fp16: the arithmetic of k_bn_apply (train.hip) --
// fp32 multiply, then add (two roundings, no fma), ReLU, round to nearest even -- written on 2-vectors so that the compiler
// emits v_pk_mul_f32 / v_pk_add_f32 / v_cvt_pk_f16_f32 / v_pk_max_f16: 24 VALU instructions instead of ~40 scalar ones.
typedef _Float16 lfd_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ lfd_f16x8 lfd_affine_relu_f16x8(lfd_f16x8 v, const float (&a)[8], const float (&b)[8]) {
  union { lfd_f16x8 h; uint32_t u[4]; } o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lfd_f32x2 x, aa, bb;
    x[0] = (float)v[2 * j]; x[1] = (float)v[2 * j + 1];
    aa[0] = a[2 * j]; aa[1] = a[2 * j + 1];
    bb[0] = b[2 * j]; bb[1] = b[2 * j + 1];
    const lfd_f32x2 f = x * aa + bb;
    o.u[j] = lfd_cvt_pk_max(f[0], f[1], LFD_PK_RELU);
  }
  return o.h;
}

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

// One-time per-DEVICE setup of a launcher (hipFuncSetAttribute and the CU count belong to a device: a process that drives
// several GPUs must repeat them on each -- ADVICE r3): `mask` is a function-local static, bit = device ordinal.
static inline int lfd_device_ordinal() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 ? dev : 0;
}
#define LFD_ONCE_PER_DEVICE(mask, dev) (!(((mask) >> (dev)) & 1ull))
#define LFD_DONE_ON_DEVICE(mask, dev) ((mask) |= (1ull << (dev)))

// Tuning knobs (include/lfd_hip.h: lfd_tuning_set / lfd_tuning_get) -- the library's only mutable global state; every kernel
// variant a test or an A/B timing selects goes through here instead of a getenv() read once behind the caller's back.
int lfd_tune(int key);
