// Dependent-accumulation latency of the bf16 MFMAs on gfx950: N independent accumulator chains issued round-robin by ONE wave per
// SIMD (256 blocks x 256 threads), cycles per MFMA as a function of N.  N = 1 is a fully dependent chain (D of one MFMA is C of
// the next).  Also: the same with the chains of TWO waves per SIMD (512-thread blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int N, int THREADS>
__global__ __launch_bounds__(THREADS) void chain(float* sink, int iters, unsigned long long* out) {
  bf16x8 a, b;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
  }
  float acc = 0.f;
  unsigned long long t0, t1;
  if constexpr (SHAPE == 32) {
    f32x16 c[N];
    for (int n = 0; n < N; ++n) for (int e = 0; e < 16; ++e) c[n][e] = 0.f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 24 / N; ++r)
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[n], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < N; ++n) acc += c[n][0];
  } else {
    f32x4 c[N];
    for (int n = 0; n < N; ++n) for (int e = 0; e < 4; ++e) c[n][e] = 0.f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 24 / N; ++r)
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[n], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int n = 0; n < N; ++n) acc += c[n][0];
  }
  if (acc == 12345.f) sink[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) out[1] = t1 - t0;
}
template <int SHAPE, int N, int THREADS>
void run() {
  unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
  unsigned long long h[2] = {0, 0};
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(d, 0, 16);
    hipEventRecord(e0);
    chain<SHAPE, N, THREADS><<<256, THREADS>>>(sink, iters, d);
    hipEventRecord(e1); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double per = (double)h[0] / iters / 24.0;
  printf("%dx%d  %d chain(s)  %d wave(s)/SIMD: %6.1f cycles per MFMA (wave 0)  -> same-accumulator distance %6.1f cycles   %.2f ms  wave4 %.1f\n",
         SHAPE, SHAPE, N, THREADS / 256, per, per * N, ms, (double)h[1] / iters / 24.0);
  hipFree(d); hipFree(sink);
}
int main() {
  run<32, 1, 256>(); run<32, 2, 256>(); run<32, 3, 256>(); run<32, 4, 256>(); run<32, 6, 256>(); run<32, 8, 256>();
  run<16, 1, 256>(); run<16, 2, 256>(); run<16, 3, 256>(); run<16, 4, 256>(); run<16, 6, 256>(); run<16, 8, 256>(); run<16, 12, 256>();
  run<32, 1, 512>(); run<32, 2, 512>(); run<32, 3, 512>(); run<16, 1, 512>(); run<16, 2, 512>(); run<16, 4, 512>(); run<16, 8, 512>();
  return 0;
}
