// Does buffer_load_dwordx4 ... lds place lane l's 16 bytes at base + 16 l?  Dump the LDS image of the MX8 X-tile fill.
#include "../../chadavit_amd/csrc/common.h"
#include <cstdio>
#include <vector>
using namespace chada;
__global__ void k(const uint8_t* X, int K, uint8_t* dump) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[16384];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const BufRsrc xr = make_rsrc(X);
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * i + 8 * w + (l >> 3);
    const int slot = (l & 7) ^ ((row >> 1) & 7);
    lds_dma16(xr, reinterpret_cast<bf16_t*>(smem + i * 4096 + w * 1024), (unsigned)row * K + slot * 16, 0u);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = tid; i < 16384; i += 256) dump[i] = smem[i];
}
int main() {
  const int K = 128;
  std::vector<uint8_t> X(128 * K), D(16384);
  for (int r = 0; r < 128; ++r) for (int c = 0; c < K; ++c) X[r * K + c] = (uint8_t)((r * 8 + c / 16) & 0xff);  // id of the 16-byte piece
  uint8_t *dX, *dD; hipMalloc(&dX, X.size()); hipMalloc(&dD, 16384);
  hipMemcpy(dX, X.data(), X.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dX, K, dD);
  hipMemcpy(D.data(), dD, 16384, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int row = 0; row < 128; ++row) for (int sl = 0; sl < 8; ++sl) {
    const int expect_piece = (row * 8 + (sl ^ ((row >> 1) & 7))) & 0xff;
    for (int b = 0; b < 16; ++b) if (D[row * 128 + sl * 16 + b] != expect_piece) { if (bad < 10) printf("row %d slot %d byte %d: got %d expect %d\n", row, sl, b, D[row * 128 + sl * 16 + b], expect_piece); ++bad; }
  }
  printf("mismatching bytes: %d\n", bad);
  printf("first 64 bytes by 16: "); for (int i = 0; i < 16; ++i) printf("%d ", D[i * 16]); printf("\n");
  return 0;
}
