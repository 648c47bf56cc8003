// does s_store_dwordx4 (+ s_dcache_wb) work on gfx950?  each wave writes 4 ballot masks (8 dwords) to out[wave_global][8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, unsigned* out, int n) {
  const int gw = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int l = threadIdx.x & 63;
  unsigned long long m[4];
  for (int j = 0; j < 4; ++j) m[j] = __builtin_amdgcn_ballot_w64(x[((size_t)gw * 4 + j) * 64 + l] > 0.f);
  unsigned* dst = out + (size_t)gw * 8;
  u32x4 a = {(unsigned)m[0], (unsigned)(m[0] >> 32), (unsigned)m[1], (unsigned)(m[1] >> 32)};
  u32x4 b = {(unsigned)m[2], (unsigned)(m[2] >> 32), (unsigned)m[3], (unsigned)(m[3] >> 32)};
  asm volatile("s_store_dwordx4 %0, %1, 0x0\n\ts_store_dwordx4 %2, %1, 0x10" ::"s"(a), "s"(dst), "s"(b) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
int main() {
  const int waves = 4096, n = waves * 4 * 64;
  std::vector<float> hx(n);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 8) % 2001 - 1000); }
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, waves * 8 * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0xff, waves * 8 * 4);
  k<<<waves / 4, 256>>>(dx, dout, n);
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned> ho(waves * 8);
  hipMemcpy(ho.data(), dout, waves * 8 * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int w = 0; w < waves; ++w)
    for (int j = 0; j < 4; ++j) {
      unsigned long long ref = 0;
      for (int l = 0; l < 64; ++l) if (hx[((size_t)w * 4 + j) * 64 + l] > 0.f) ref |= 1ull << l;
      unsigned long long got = ho[w * 8 + 2 * j] | ((unsigned long long)ho[w * 8 + 2 * j + 1] << 32);
      bad += got != ref;
    }
  printf("err=%d bad=%ld of %d\n", (int)e, bad, waves * 4);
  return bad != 0;
}
