// What the gfx950 fp8 instructions the c8 kernels rely on actually do (run on the GPU box; prints PASS / FAIL lines):
//   1. v_cvt_scalef32_pk_fp8_f16: rounding, the meaning of the scale operand, behaviour above 448 and below 2^-9;
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3 operands: lane l of A holds row l % 32, lane l of B holds column l % 32, and
//      byte p of lane l of A meets byte p of lane (n, l / 32) of B (the K index is SOME fixed function of (l / 32, p), the same for
//      both operands -- all the kernels need); accumulator layout = the 32x32 fp32 layout of the fp16 instruction; the E8M0 scale
//      operands multiply the product by 2^(sa - 127) 2^(sb - 127).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_cvt(unsigned* o, const _Float16* in, float scale, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  v2h a = {in[2 * i], in[2 * i + 1]};
  v2s r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, a, scale, false);
  o[i] = __builtin_bit_cast(unsigned, r) & 0xffff;
}
template <int SA, int SB>
__global__ void k_mfma(float* c_out, const unsigned char* a, const unsigned char* b) {
  const int l = threadIdx.x;
  v8i av, bv;
  memcpy(&av, a + l * 32, 32);
  memcpy(&bv, b + l * 32, 32);
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, SA * 0x01010101, 0, SB * 0x01010101);
  for (int r = 0; r < 16; ++r) c_out[l * 16 + r] = c[r];
}
static float dec(unsigned char v) {   // e4m3fn
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  if (e == 15 && m == 7) f = NAN;
  return s ? -f : f;
}
int main() {
  int fails = 0;
  {
    const float vals[] = {0.f, 0.0009765625f, 0.0015f, 0.001953125f, 0.003f, 0.0156f, 0.3f, 1.f, 1.0625f, 1.07f, 1.1875f, 100.f, 448.f, 460.f, 480.f, 1000.f, -1000.f, 65504.f, -0.3f, 17.f};
    const int n = sizeof(vals) / sizeof(float);
    _Float16 h[64]; for (int i = 0; i < n; ++i) h[i] = (_Float16)vals[i];
    _Float16* d; unsigned* o; (void)hipMalloc(&d, 256); (void)hipMalloc(&o, 256);
    (void)hipMemcpy(d, h, 128, hipMemcpyHostToDevice);
    for (float sc : {1.0f, 2.0f, 0.5f}) {
      k_cvt<<<1, 64>>>(o, d, sc, n / 2); (void)hipDeviceSynchronize();
      unsigned r[32]; (void)hipMemcpy(r, o, 128, hipMemcpyDeviceToHost);
      printf("scale %.1f:", sc);
      for (int i = 0; i < n; ++i) { unsigned char b = (r[i / 2] >> (8 * (i & 1))) & 0xff; printf("  %g->%g(0x%02x)", vals[i], dec(b), b); }
      printf("\n");
    }
  }
  {
    unsigned char a[64 * 32], b[64 * 32];
    srand(3);
    // small exactly representable values: {-2,-1.5,...,2} subset; bytes chosen from a table
    const unsigned char tab[8] = {0x00, 0x38, 0x40, 0x3c, 0xb8, 0xc0, 0x30, 0xb0};   // 0, 1, 2, 1.5, -1, -2, 0.5, -0.5
    for (int i = 0; i < 64 * 32; ++i) { a[i] = tab[rand() & 7]; b[i] = tab[rand() & 7]; }
    unsigned char *da, *db; float* dc; (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dc, 64 * 16 * 4);
    (void)hipMemcpy(da, a, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b, 2048, hipMemcpyHostToDevice);
    float c[64 * 16];
    for (int variant = 0; variant < 3; ++variant) {
      if (variant == 0) k_mfma<127, 127><<<1, 64>>>(dc, da, db);
      if (variant == 1) k_mfma<120, 127><<<1, 64>>>(dc, da, db);
      if (variant == 2) k_mfma<127, 130><<<1, 64>>>(dc, da, db);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
      const float mul = variant == 0 ? 1.f : variant == 1 ? ldexpf(1.f, -7) : 8.f;
      double maxerr = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
          const int n = l & 31, kb = l >> 5, g = r >> 2, e = r & 3;
          const int row = 8 * g + 4 * kb + e;                 // the fp16 instruction's accumulator layout
          double ref = 0;
          for (int kk = 0; kk < 2; ++kk)
            for (int p = 0; p < 32; ++p) ref += (double)dec(a[(row + 32 * kk) * 32 + p]) * dec(b[(n + 32 * kk) * 32 + p]);
          maxerr = fmax(maxerr, fabs(c[l * 16 + r] - ref * mul));
        }
      printf("mfma 32x32x64 fp8 variant %d (scale product %g): max |c - ref| = %g  %s\n", variant, mul, maxerr, maxerr == 0 ? "PASS" : "FAIL");
      fails += maxerr != 0;
    }
  }
  return fails;
}
