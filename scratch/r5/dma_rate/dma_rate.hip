// What does a CU sustain in LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) from an L2-resident source, as a function of how the
// 64 lanes' 16-byte pieces are spread over 128-byte lines?  Every CU runs WPB waves that issue PIECES instructions per round into their own LDS
// slice and wait for them (vmcnt(0)); no MFMA, no LDS reads.  Patterns (row stride RS bytes):
//   0: contiguous 1 KiB (8 lines per instruction)            -- pre-packed records (block kernels' weight stream)
//   1: 8 rows x 128 B  (lane l: row l >> 3, chunk l & 7)     -- row-major tile, full lines
//   2: 16 rows x 64 B  (row l & 15, chunk l >> 4)            -- 16x16x32 K fragment record
//   3: 32 rows x 32 B  (row l & 31, chunk l >> 5)            -- 32x32x16 K fragment / 16x16x32 V record
//   4: 32 rows x 16 B x 2 columns 384 B apart                -- worst case: 64 distinct lines
//   5: 8 rows x 128 B with the chunk XOR-swizzled by the row -- row-major tile as the swizzled stages fetch it
//   6 / 7 / 8: 64 consecutive 16-byte chunks of a row-major tile with 192- / 384- / 768-byte rows (the backward kernels' stages at dh 96 / 192 / 384)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t window, int rounds, unsigned RS, long long* __restrict__ cycles, int PATTERN, int PIECES) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(src);
  // a round = one TILE of 32 rows x 768 bytes (one head of Base: 384 bf16) = 24 pieces of 1 KiB, piece p = w * PIECES + i (nw * PIECES = 24);
  // how a piece's 64 x 16 bytes lie on the tile is the pattern
  constexpr int RB = 768;
  unsigned piece_off[12];
  unsigned lane_off;
  if (PATTERN == 0) lane_off = l * 16;
  else if (PATTERN == 1) lane_off = (l >> 3) * RS + (l & 7) * 16;
  else if (PATTERN == 2) lane_off = (l & 15) * RS + (l >> 4) * 16;
  else if (PATTERN == 3) lane_off = (l & 31) * RS + (l >> 5) * 16;
  else if (PATTERN == 4) lane_off = (l & 31) * RS + (l >> 5) * 384;   // 32 rows x 16 B, two columns 384 B apart: 64 distinct lines
  else if (PATTERN == 5) lane_off = (l >> 3) * RS + (((l & 7) ^ (l >> 3)) & 7) * 16;
  else lane_off = 0;   // (6 / 7 / 8: per piece, below)
  const int CH = PATTERN == 6 ? 12 : PATTERN == 7 ? 24 : 48;   // 16-byte chunks per row: dh 96 / 192 / 384
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int p = w * PIECES + i;
    if (PATTERN >= 6) { const int id = p * 64 + l; piece_off[i] = (id / CH) * RS + (id % CH) * 16; }   // row-major image, 64 consecutive chunks per piece
    else if (PATTERN == 0) piece_off[i] = p * 1024;                                            // (contiguous: the tile is a packed 24 KiB record)
    else if (PATTERN == 1 || PATTERN == 5) piece_off[i] = (p / 6) * 8 * RS + (p % 6) * 128;   // 4 row groups x 6 column blocks
    else if (PATTERN == 2) piece_off[i] = (p / 12) * 16 * RS + (p % 12) * 64;              // 2 x 12
    else if (PATTERN == 3) piece_off[i] = p * 32;                                          // 1 x 24
    else piece_off[i] = p * 16;                                                            // 1 x 24 (+ the 384-byte partner column)
  }
  const size_t tile_step = PATTERN == 0 ? 24 * 1024 : (PATTERN >= 6 ? (size_t)(24 * 64 / CH) * RS : 32 * (size_t)RS);
  // the blocks of one XCD (blockIdx % 8) walk the SAME region together, as the query tiles of one image share its K / V: L2 hits after the warm-up
  const size_t region0 = (size_t)(blockIdx.x & 7) * window;
  size_t base = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < PIECES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (w * PIECES + i) * 1024), 16, lane_off + piece_off[i], (unsigned)(region0 + base), 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    base += tile_step;
    if (base + 2 * tile_step > window) base = 0;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int PATTERN, int PIECES>
void run(const char* name, const char* src, size_t window, int wpb, unsigned RS, long long* dcyc) {
  const int rounds = 400, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)wpb * PIECES * 1024;
  hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dma_kernel<<<blocks, wpb * 64, lds>>>(src, window, 50, RS, dcyc, PATTERN, PIECES);   // warm-up (L2 / MALL)
  hipEventRecord(e0);
  dma_kernel<<<blocks, wpb * 64, lds>>>(src, window, rounds, RS, dcyc, PATTERN, PIECES);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> cyc(blocks); hipMemcpy(cyc.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto c : cyc) avg += c; avg /= blocks;
  const double bytes_cu = (double)rounds * wpb * PIECES * 1024;
  printf("%-28s waves/CU %2d pieces/wave/round %2d: %7.1f us  %6.1f GB/s per CU  %5.1f TB/s chip  (s_memtime ticks per piece per CU %.1f)\n", name, wpb, PIECES, ms * 1e3,
         bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * blocks / (ms * 1e-3) / 1e12, avg / ((double)rounds * wpb * PIECES));
}
int main() {
  const size_t window = 3u << 20;   // per XCD: 3 MiB of its 4 MiB L2 (a Base image's K + V of one head)
  char* src; hipMalloc(&src, 8 * window + (16u << 20)); hipMemset(src, 1, 8 * window + (16u << 20));
  long long* dcyc; hipMalloc(&dcyc, 256 * sizeof(long long));
  const unsigned RS = 4608;   // bytes between rows: 3 * 768 bf16 (Base qkv)
  for (int wpb : {4, 8}) {
    if (wpb == 4) {
      run<0, 12>("contiguous 1 KiB", src, window, 4, RS, dcyc); run<1, 12>("8 rows x 128 B", src, window, 4, RS, dcyc); run<5, 12>("8 rows x 128 B swizzled", src, window, 4, RS, dcyc);
      run<2, 12>("16 rows x 64 B", src, window, 4, RS, dcyc); run<3, 12>("32 rows x 32 B", src, window, 4, RS, dcyc); run<4, 12>("64 lines x 16 B", src, window, 4, RS, dcyc);
    } else {
      run<0, 6>("contiguous 1 KiB", src, window, 8, RS, dcyc); run<1, 6>("8 rows x 128 B", src, window, 8, RS, dcyc); run<5, 6>("8 rows x 128 B swizzled", src, window, 8, RS, dcyc);
      run<2, 6>("16 rows x 64 B", src, window, 8, RS, dcyc); run<3, 6>("32 rows x 32 B", src, window, 8, RS, dcyc); run<4, 6>("64 lines x 16 B", src, window, 8, RS, dcyc);
    }
  }
  // row stride of the Tiny qkv (1152 B) and a small window (L2-resident per XCD)
  run<2, 6>("16 rows x 64 B, RS 1152", src, window, 8, 1152, dcyc); run<3, 6>("32 rows x 32 B, RS 1152", src, window, 8, 1152, dcyc);
  run<1, 6>("8 rows x 128 B, RS 1152", src, window, 8, 1152, dcyc); run<0, 6>("contiguous", src, window, 8, 1152, dcyc);
  run<6, 6>("row-major 192-B rows, RS 1152", src, window, 8, 1152, dcyc); run<7, 6>("row-major 384-B rows, RS 2304", src, window, 8, 2304, dcyc);
  run<8, 6>("row-major 768-B rows, RS 4608", src, window, 8, 4608, dcyc); run<6, 12>("row-major 192-B rows, RS 1152", src, window, 4, 1152, dcyc);
  return 0;
}
