// Second co-issue probe (gfx950): (a) two waves per SIMD GUARANTEED (512-thread blocks, one block per CU): does a wave's MFMA issue
// overlap the other wave's VALU?  (b) what an LDS read / an LDS-DMA piece costs as a filler beside 32x32x16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))

#define M32 "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
#define M32b "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
#define M32c "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\t"
#define M32d "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\t"
#define M16 "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"
#define M16b "v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
#define M16c "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\t"
#define M16d "v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n\t"
#define F1 "v_fma_f32 %6, %6, %14, %15\n\t"
#define F2 F1 "v_fma_f32 %7, %7, %14, %15\n\t"
#define F4 F2 "v_fma_f32 %8, %8, %14, %15\n\tv_fma_f32 %9, %9, %14, %15\n\t"
#define F6 F4 "v_fma_f32 %10, %10, %14, %15\n\tv_fma_f32 %11, %11, %14, %15\n\t"
#define F8 F6 "v_fma_f32 %12, %12, %14, %15\n\tv_fma_f32 %13, %13, %14, %15\n\t"
#define E1 "v_exp_f32 %6, %6\n\t"
#define E2 E1 "v_exp_f32 %7, %7\n\t"
// the softmax mix per 32x32 MFMA at dh = 96: 1.33 exp + 1.33 add + 0.67 cvt  ->  per 3 MFMAs: 4 exp, 4 add, 2 cvt
#define SMX_A "v_exp_f32 %6, %6\n\tv_add_f32 %10, %10, %6\n\tv_exp_f32 %7, %7\n\t"
#define SMX_B "v_add_f32 %11, %11, %7\n\tv_cvt_pk_bf16_f32 %12, %6, %7\n\tv_exp_f32 %8, %8\n\t"
#define SMX_C "v_add_f32 %10, %10, %8\n\tv_exp_f32 %9, %9\n\tv_add_f32 %11, %11, %9\n\tv_cvt_pk_bf16_f32 %13, %8, %9\n\t"

#define KERNEL(NAME, ACC, NACC, BODY, THREADS)                                                                              \
  __global__ __launch_bounds__(THREADS) void NAME(float* sink, int iters, unsigned long long* out) {                       \
    bf16x8 a, b;                                                                                                            \
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;                                                  \
    for (int i = 0; i < 8; ++i) {                                                                                           \
      s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);                               \
      s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);                               \
    }                                                                                                                       \
    ACC c0, c1, c2, c3;                                                                                                     \
    for (int e = 0; e < NACC; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }                                  \
    float v0 = -0.001f * threadIdx.x, v1 = v0 - 1, v2 = v0 - 2, v3 = v0 - 3, v4 = v0 - 4, v5 = v0 - 5, v6 = v0 - 6, v7 = v0 - 7; \
    float k1 = 0.999f, k2 = -1e-4f;                                                                                         \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                   \
    for (int i = 0; i < iters; ++i) {                                                                                       \
      asm volatile(BODY : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)                                                            \
                   : "v"(a), "v"(b), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(k1), "v"(k2)); \
    }                                                                                                                       \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                   \
    float acc = c0[0] + c1[0] + c2[0] + c3[0] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                      \
    if (acc == 12345.f) sink[0] = acc;                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                              \
    if (threadIdx.x == 64 * 4 && blockIdx.x == 0) out[1] = t1 - t0;                                                         \
  }

#define BOTH(NAME, ACC, NACC, BODY) KERNEL(NAME##_1, ACC, NACC, BODY, 256) KERNEL(NAME##_2, ACC, NACC, BODY, 512)

BOTH(m32_f0, f32x16, 16, M32 M32b M32c M32d)
BOTH(m32_f4, f32x16, 16, M32 F4 M32b F4 M32c F4 M32d F4)
BOTH(m32_f8, f32x16, 16, M32 F8 M32b F8 M32c F8 M32d F8)
BOTH(m32_e2, f32x16, 16, M32 E2 M32b E2 M32c E2 M32d E2)
BOTH(m32_smx, f32x16, 16, M32 SMX_A M32b SMX_B M32c SMX_C M32d SMX_A)
BOTH(m16_f0, f32x4, 4, M16 M16b M16c M16d)
BOTH(m16_f2, f32x4, 4, M16 F2 M16b F2 M16c F2 M16d F2)
BOTH(m16_f4, f32x4, 4, M16 F4 M16b F4 M16c F4 M16d F4)
BOTH(m16_e2, f32x4, 4, M16 E2 M16b E2 M16c E2 M16d E2)
BOTH(m0_f8, f32x4, 4, F8 F8 F8 F8)
BOTH(m0_e2, f32x4, 4, E2 E2 E2 E2 E2 E2 E2 E2)

// ---- LDS read / LDS-DMA as fillers beside 32x32x16 MFMAs (one wave per SIMD, 256-thread blocks)
template <int NDS, int NDMA, int NVALU>
__global__ __launch_bounds__(256) void lds_fill(float* sink, const __bf16* src, int iters, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[64 * 1024 / 2];  // 64 KiB
  bf16x8 a, b;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
  }
  for (int i = threadIdx.x; i < 32 * 1024; i += 256) smem[i] = (__bf16)(0.001f * (i & 255));
  __syncthreads();
  f32x16 c0, c1, c2, c3;
  for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = -0.001f * (threadIdx.x + e);
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const __bf16* lbase = smem + w * 8192 + l * 8;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(src), 0, (int)0xFFFFFFFFu, 0x00020000);
  bf16x8 fr[4];
  for (int e = 0; e < 4; ++e) fr[e] = a;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x16& c = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : c3;
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j], b, c, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < NDS; ++d) fr[(j + 2) & 3] = *reinterpret_cast<const bf16x8*>(lbase + ((i + j * NDS + d) & 7) * 512);
#pragma unroll
      for (int d = 0; d < NDMA; ++d)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDSP void*)(smem + 16384 + w * 2048 + ((j * NDMA + d) & 3) * 512), 16,
                                                 (unsigned)(l * 16), (unsigned)((((i * 4 + j) * NDMA + d) & 1023) * 1024 + blockIdx.x * (1 << 20)), 0, 0);
#pragma unroll
      for (int d = 0; d < NVALU; ++d) v[d] = __builtin_fmaf(v[d], 0.999f, -1e-4f);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NDMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = c0[0] + c1[0] + c2[0] + c3[0];
  for (int e = 0; e < 8; ++e) acc += v[e];
  acc += (float)fr[0][0] + (float)fr[1][0] + (float)fr[2][0] + (float)fr[3][0];
  if (acc == 12345.f) sink[0] = acc + (float)smem[threadIdx.x + 16384];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int threads, double mfma_per_iter) {
  unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
  unsigned long long h[2] = {0, 0};
  const int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(d, 0, 16);
    hipEventRecord(e0);
    kern<<<256, threads>>>(sink, iters, d);
    hipEventRecord(e1); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-14s threads=%d  wave0 %7.1f cyc/iter  wave4 %7.1f cyc/iter  (%5.1f cyc per mfma, wave 0)  %.2f ms  clock %.2f GHz\n", name, threads,
         (double)h[0] / iters, (double)h[1] / iters, mfma_per_iter ? (double)h[0] / iters / mfma_per_iter : 0.0, ms, (double)h[0] / (ms * 1e6));
}
template <int NDS, int NDMA, int NVALU>
void run_lds(const __bf16* src) {
  unsigned long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
  unsigned long long h = 0;
  const int iters = 5000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    lds_fill<NDS, NDMA, NVALU><<<256, 256>>>(sink, src, iters, d);
    hipEventRecord(e1); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("lds_fill ds_read_b128=%d dma=%d valu=%d per mfma: %6.1f cyc per mfma  %.2f ms\n", NDS, NDMA, NVALU, (double)h / iters / 4.0, ms);
}
#define RUNBOTH(N, M) run(#N, N##_1, 256, M); run(#N, N##_2, 512, M);
int main() {
  RUNBOTH(m32_f0, 4) RUNBOTH(m32_f4, 4) RUNBOTH(m32_f8, 4) RUNBOTH(m32_e2, 4) RUNBOTH(m32_smx, 4)
  RUNBOTH(m16_f0, 4) RUNBOTH(m16_f2, 4) RUNBOTH(m16_f4, 4) RUNBOTH(m16_e2, 4) RUNBOTH(m0_f8, 0) RUNBOTH(m0_e2, 0)
  __bf16* src; hipMalloc(&src, (size_t)300 << 20); hipMemset(src, 0, (size_t)300 << 20);
  run_lds<0, 0, 0>(src); run_lds<1, 0, 0>(src); run_lds<2, 0, 0>(src); run_lds<1, 0, 4>(src); run_lds<1, 0, 6>(src);
  run_lds<0, 1, 0>(src); run_lds<0, 2, 0>(src); run_lds<1, 1, 4>(src); run_lds<0, 0, 4>(src); run_lds<0, 0, 6>(src);
  return 0;
}
