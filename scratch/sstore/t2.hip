// are the data SGPRs of s_store read at issue?  overwrite them right behind the store and look at what arrives
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  const int gw = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  unsigned* dst = out + (size_t)gw * 64;
  // 16 stores from the SAME four SGPRs, rewritten between stores with no wait
  asm volatile(
      "s_mov_b32 s20, %1\n\t"
      "s_mov_b32 s21, 0\n\t"
      ".rept 16\n\t"
      "s_add_u32 s21, s21, 1\n\t"
      "s_mov_b32 s16, s21\n\ts_mov_b32 s17, s21\n\ts_mov_b32 s18, s21\n\ts_mov_b32 s19, s21\n\t"
      "s_store_dwordx4 s[16:19], %0, s20\n\t"
      "s_add_u32 s20, s20, 16\n\t"
      ".endr\n\t"
      "s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(dst), "s"(0)
      : "s16", "s17", "s18", "s19", "s20", "s21", "memory");
}
int main() {
  const int waves = 8192;
  unsigned* dout;
  hipMalloc(&dout, waves * 64 * 4);
  hipMemset(dout, 0xff, waves * 64 * 4);
  k<<<waves / 4, 256>>>(dout);
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned> ho(waves * 64);
  hipMemcpy(ho.data(), dout, waves * 64 * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int w = 0; w < waves; ++w)
    for (int i = 0; i < 64; ++i) bad += ho[w * 64 + i] != (unsigned)(i / 4 + 1);
  printf("err=%d bad=%ld of %d\n", (int)e, bad, waves * 64);
  return 0;
}
