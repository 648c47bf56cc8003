// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 operand / scale layout on gfx950 (fp8 e4m3 x fp8 e4m3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C, int opsel) {
  // A [16][128] fp8 row-major, B [16 cols][128 k] fp8 (i.e. W[n][k]), sa [16][4] e8m0, sb [16][4]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(A + r * 128 + g * 32 + i * 4);
    b[i] = *reinterpret_cast<const int*>(B + r * 128 + g * 32 + i * 4);
  }
  // scale operand: 4 bytes per lane; put the wanted scale in byte `opsel`, garbage elsewhere
  unsigned sav = 0x7f7f7f7fu, sbv = 0x7f7f7f7fu;
  sav = (sav & ~(0xffu << (8 * opsel))) | ((unsigned)sa[r * 4 + g] << (8 * opsel));
  sbv = (sbv & ~(0xffu << (8 * opsel))) | ((unsigned)sb[r * 4 + g] << (8 * opsel));
  f32x4 c = {0, 0, 0, 0};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sav, 0, (int)sbv);
  else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, (int)sav, 1, (int)sbv);
  else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, (int)sav, 2, (int)sbv);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, (int)sav, 3, (int)sbv);
  for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];  // assume C[row=(l>>4)*4+i][col=l&15]
}

static float e4m3(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 0) x = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) x = NAN;
  else x = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<uint8_t> A(16 * 128), B(16 * 128), sa(64), sb(64);
  uint32_t st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (auto& v : A) { v = rnd() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; if (mode & 1) v &= 0xbf; }
  for (auto& v : B) { v = rnd() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; if (mode & 1) v &= 0xbf; }
  for (auto& v : sa) v = (mode & 2) ? 127 : 124 + rnd() % 6;
  for (auto& v : sb) v = (mode & 2) ? 127 : 125 + rnd() % 5;
  uint8_t *dA, *dB, *dsa, *dsb; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 1024);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
  std::vector<float> ref(256, 0.f);
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double acc = 0;
    for (int kb = 0; kb < 4; ++kb) {
      double s = 0;
      for (int kk = 0; kk < 32; ++kk) s += (double)e4m3(A[m * 128 + kb * 32 + kk]) * e4m3(B[n * 128 + kb * 32 + kk]);
      acc += s * ldexp(1.0, (int)sa[m * 4 + kb] - 127) * ldexp(1.0, (int)sb[n * 4 + kb] - 127);
    }
    ref[m * 16 + n] = (float)acc;
  }
  for (int opsel = 0; opsel < 4; ++opsel) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, opsel);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 256; ++i) { maxerr = fmax(maxerr, fabs(C[i] - ref[i])); maxref = fmax(maxref, fabs(ref[i])); }
    printf("opsel %d: max err %g (max ref %g)  C[0..3] %g %g %g %g ref %g %g %g %g\n", opsel, maxerr, maxref, C[0], C[1], C[2], C[3], ref[0], ref[1], ref[2], ref[3]);
  }
  return 0;
}
