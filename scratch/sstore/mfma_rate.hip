// sustained MFMA rate under full-chip load: v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16, random vs zero operands,
// 1 / 2 waves per SIMD.  Inline asm on fixed accumulators (the compiler rotates C++ accumulators into dependent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE>
__global__ __launch_bounds__(256) void heavy(float* sink, int iters, unsigned long long* out, float scale) {
  bf16x8 a[4], b[4];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u; a[j][i] = (__bf16)(scale * ((int)(s >> 9) % 2001 - 1000) * 1e-3f);
      s = s * 1664525u + 1013904223u; b[j][i] = (__bf16)(scale * ((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    }
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float acc_sum = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 c0_ = {0, 0, 0, 0}, c1 = c0_, c2 = c0_, c3 = c0_, c4 = c0_, c5 = c0_, c6 = c0_, c7 = c0_;
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %8, %12, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %9, %13, %1\n\t"
                   "v_mfma_f32_16x16x32_bf16 %2, %10, %14, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %11, %15, %3\n\t"
                   "v_mfma_f32_16x16x32_bf16 %4, %8, %13, %4\n\tv_mfma_f32_16x16x32_bf16 %5, %9, %14, %5\n\t"
                   "v_mfma_f32_16x16x32_bf16 %6, %10, %15, %6\n\tv_mfma_f32_16x16x32_bf16 %7, %11, %12, %7"
                   : "+v"(c0_), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    }
    acc_sum = c0_[0] + c1[0] + c2[0] + c3[0] + c4[0] + c5[0] + c6[0] + c7[0];
  } else {
    f32x16 c0_, c1, c2, c3;
    for (int e = 0; e < 16; ++e) { c0_[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %8, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %5, %9, %1\n\t"
                   "v_mfma_f32_32x32x16_bf16 %2, %6, %10, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %11, %3"
                   : "+v"(c0_), "+v"(c1), "+v"(c2), "+v"(c3)
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    }
    acc_sum = c0_[0] + c1[0] + c2[0] + c3[0];
  }
  unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc_sum == 12345.f) sink[0] = acc_sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1_ - c0; out[1] = r1 - r0; }
}
template <int SHAPE>
void run(const char* name, float scale, int waves_per_simd) {
  unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
  unsigned long long h[2];
  const int iters = 100000;
  const double fl_per_iter = SHAPE == 16 ? 8.0 * 16 * 16 * 32 * 2 : 4.0 * 32 * 32 * 16 * 2;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    heavy<SHAPE><<<256 * waves_per_simd, 256>>>(sink, iters, d, scale);
    hipEventRecord(e1); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * waves_per_simd * 4 * iters * fl_per_iter;
    printf("%s %s operands, %d wave(s)/SIMD: %.1f ms  %.0f TFLOP/s  shader clock %.0f MHz  -> %.1f cycles per MFMA per SIMD\n", name, scale ? "random" : "zero  ",
           waves_per_simd, ms, fl / ms / 1e9, 100.0 * h[0] / h[1], (double)h[0] / (iters * (SHAPE == 16 ? 8.0 : 4.0) * waves_per_simd));
  }
}
int main() {
  run<16>("16x16x32", 1.f, 2); run<32>("32x32x16", 1.f, 2);
  run<16>("16x16x32", 0.f, 2); run<32>("32x32x16", 0.f, 2);
  run<16>("16x16x32", 1.f, 1); run<32>("32x32x16", 1.f, 1);
  return 0;
}
