// which clock does s_memtime count?  compare against s_memrealtime (100 MHz) on a light kernel and beside a heavy MFMA kernel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(unsigned long long* out, int spin) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)v; }
}
__global__ __launch_bounds__(256) void heavy(float* sink, int iters, unsigned long long* out) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.01f); }
  f32x4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int j = 0; j < 8; ++j) s += acc[j][0];
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 64); hipMalloc(&sink, 64);
  unsigned long long h[3];
  probe<<<1, 64>>>(d, 2000000);
  hipDeviceSynchronize(); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("light: memtime %llu  memrealtime %llu  ratio %.2f -> memtime at %.0f MHz if realtime is 100 MHz\n", h[0], h[1], (double)h[0] / h[1], 100.0 * h[0] / h[1]);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    heavy<<<256 * 8, 256>>>(sink, 200000, d);
    hipEventRecord(e1);
    hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * 8 * 4 * 200000.0 * 8 * 16 * 16 * 32 * 2;
    printf("heavy (all CUs, 8 waves/CU of back-to-back MFMA): %.1f ms  %.0f TFLOP/s  memtime/realtime %.2f -> %.0f MHz\n", ms, fl / ms / 1e9, (double)h[0] / h[1], 100.0 * h[0] / h[1]);
  }
  return 0;
}
