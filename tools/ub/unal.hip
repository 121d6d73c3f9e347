#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t __attribute__((aligned(2))) u32a2;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U4A2 { uint32_t x, y, z, w; };
__global__ void k(const uint16_t* in, uint32_t* out, int sh) {
  __shared__ __attribute__((aligned(16))) uint16_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = in[i];
  __syncthreads();
  // misaligned LDS dword read
  const u32a2* p = reinterpret_cast<const u32a2*>(s + sh + 6 * threadIdx.x);
  uint32_t a = p[0], b = p[1];
  // misaligned global 16-B read
  const U4A2* g = reinterpret_cast<const U4A2*>(in + sh + 8 * threadIdx.x);
  U4A2 v = *g;
  out[threadIdx.x * 6 + 0] = a; out[threadIdx.x * 6 + 1] = b;
  out[threadIdx.x * 6 + 2] = v.x; out[threadIdx.x * 6 + 3] = v.y; out[threadIdx.x * 6 + 4] = v.z; out[threadIdx.x * 6 + 5] = v.w;
}
int main() {
  uint16_t h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (uint16_t)i;
  uint16_t* d; uint32_t* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 6 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int sh : {0, 1, 3}) {
    k<<<1, 64>>>(d, o, sh);
    uint32_t r[64 * 6]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
      uint32_t e0 = (sh + 6 * t) | ((sh + 6 * t + 1) << 16), e1 = (sh + 6 * t + 2) | ((sh + 6 * t + 3) << 16);
      if (r[t * 6] != e0 || r[t * 6 + 1] != e1) ++bad;
      for (int q = 0; q < 4; ++q) { uint32_t e = (sh + 8 * t + 2 * q) | ((sh + 8 * t + 2 * q + 1) << 16); if (r[t * 6 + 2 + q] != e) ++bad; }
    }
    printf("shift %d halfs: %s (%d bad)  sample %08x %08x\n", sh, bad ? "MISMATCH" : "ok", bad, r[6], r[8]);
  }
  return 0;
}
