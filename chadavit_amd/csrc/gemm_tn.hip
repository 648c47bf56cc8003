// C[I,J] (+)= A[T,I]^T * B[T,J]  -- weight-gradient GEMM, reduction over the (ragged-packed) token rows.
//
// Both operands are row-major with the reduction index as the ROW, i.e. "k-strided" for the MFMA.
// Stages of 32 token rows go global -> LDS by DMA (row-major, XOR-swizzled) and the fragments are fetched with the
// gfx950 hardware transpose read (ds_read_b64_tr_b16): one instruction hands each lane 4 consecutive t for its own
// column.  A and B use the same k-slot permutation (rows {4g..4g+3} U {16+4g..}) so the product is exact.
//
// T is split into `splits` chunks -> fp32 partial slabs, combined by a second (deterministic) kernel;
// the bias gradient colsum(A) rides along as one extra MFMA against an all-ones fragment.
#include "common.h"

#ifndef CHADA_AB_SWITCHES
#define CHADA_AB_SWITCHES 0   // 1 (side builds only): environment switches for same-box A/B runs
#endif

using namespace chada;

namespace {

// 1-D grid, XCD-aware: the hardware deals consecutive blocks round-robin over the 8 XCDs, so block L runs on XCD L % 8.
// All tiles of one T-split are given to the SAME XCD (and to consecutive dispatch slots there): the operand the tiles
// share (x for dW1/dWqkv, dz for dW2) is then fetched into one L2 instead of eight, and the column segments of the other
// operand that the tiles read side by side complete whole rows inside that L2.  Grid = tiles * round_up(splits, 8).
__device__ __forceinline__ bool tn_decode(int tiles, int nsplits, int& tile, int& split) {
  const int L = blockIdx.x, slot = L >> 3;
  tile = slot % tiles;
  split = (slot / tiles) * 8 + (L & 7);
  return split < nsplits;
}

// 32-row stages go global -> LDS with buffer_load_dwordx4 ... lds (lds_dma16) into a ring of NST stages, so NST-1
// stages (not one register set) are in flight per block and no VGPRs are spent on staging.  A DMA instruction fills
// 1 KiB of LDS linearly (lane l -> +16 l), which rules out row padding; the bank spread the padded layout gave the
// transpose reads comes from the SOURCE side instead: LDS row r, 32-byte pair p holds source pair p ^ swz(r), and the
// reader applies the same XOR.  swz makes the 8 rows a half-wave reads in one ds_read_b64_tr_b16 hit 8 disjoint
// 32-byte bank groups for row strides of 128, 256 and 384 bytes.
constexpr int BTD = 32;
template <int RB>
__device__ __forceinline__ int tn_swz(int row) {
  return RB == 256 ? (row & 7) : ((row >> 1) & 3);
}

// Issue the DMA of one stage into `dst` (when `issue`) and fetch the MFMA fragments of the stage held in `rd`.
// Both live in ONE function with __restrict__ pointers on purpose: after inlining, the LDS reads carry scoped-noalias
// metadata against the DMA, which is what lets the compiler's waitcnt insertion NOT drain the whole DMA queue
// (s_waitcnt vmcnt(0)) in front of every LDS read that follows an LDS-DMA load -- it cannot tell ring slots apart by
// itself.  Completion of the stage being read is established by the caller's explicit counted wait + barrier.
template <int BI, int BJ>
__device__ __forceinline__ void tn_dma_and_read(BufRsrc ag, unsigned abytes, BufRsrc bg, unsigned bbytes,
                                                bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd, bool issue,
                                                int left, const int (&aoffg)[BTD * BI / 2048], const int (&arow)[BTD * BI / 2048],
                                                const int (&boffg)[BTD * BJ / 2048], const int (&brow)[BTD * BJ / 2048],
                                                int w, int l, int wi, int wj, bf16x8 (&af)[BI / 32], bf16x8 (&bfr)[BJ / 32]) {
  constexpr int ARW = BTD * BI / 2048, BRW = BTD * BJ / 2048;  // 1 KiB records per wave per stage
  constexpr int IB = BI / 32, JB = BJ / 32;
  if (issue) {
    bf16_t* sa = dst + w * 512;
    bf16_t* sb = sa + BTD * BI;
    if (left >= BTD) {
#pragma unroll
      for (int i = 0; i < ARW; ++i)
        lds_dma16(ag, sa + i * 2048, aoffg[i] * 2, abytes);
#pragma unroll
      for (int i = 0; i < BRW; ++i)
        lds_dma16(bg, sb + i * 2048, boffg[i] * 2, bbytes);
    } else {
      // ragged last stage: rows past the end are zero-filled by their lanes (the wait before it is read is vmcnt(0))
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < ARW; ++i) {
        if (arow[i] < left)
          lds_dma16(ag, sa + i * 2048, aoffg[i] * 2, abytes);
        else
          *reinterpret_cast<u32x4*>(sa + i * 2048 + l * 8) = zero4;
      }
#pragma unroll
      for (int i = 0; i < BRW; ++i) {
        if (brow[i] < left)
          lds_dma16(bg, sb + i * 2048, boffg[i] * 2, bbytes);
        else
          *reinterpret_cast<u32x4*>(sb + i * 2048 + l * 8) = zero4;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST: free of the alias edge, the scheduler may sink it below the math
  // transpose-read addressing: this lane reads row 4g + (ii >> 2) (and +16) of the stage, 8 bytes at (ii & 3) * 8 inside
  // the swizzled 32-byte pair of its 16-column block
  const int ii = l & 15, g = l >> 4;
  const int rrow = 4 * g + (ii >> 2);
  const int swa = tn_swz<BI * 2>(rrow), swb = tn_swz<BJ * 2>(rrow);  // identical for row + 16
  const bf16_t* sA = rd + rrow * BI + (ii & 3) * 4;
  const bf16_t* sB = rd + BTD * BI + rrow * BJ + (ii & 3) * 4;
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const bf16_t* p = sA + (((wi * IB + i) ^ swa) << 4);
    af[i] = __builtin_shufflevector(lds_read_tr4(p), lds_read_tr4(p + 16 * BI), 0, 1, 2, 3, 4, 5, 6, 7);
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const bf16_t* p = sB + (((wj * JB + j) ^ swb) << 4);
    bfr[j] = __builtin_shufflevector(lds_read_tr4(p), lds_read_tr4(p + 16 * BJ), 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

template <int BI, int BJ, int NST>
__global__ __launch_bounds__(256, (NST == 2 ? 3 : 2)) void gemm_tn_dma_kernel(const bf16_t* __restrict__ A, int lda,
                                                          const bf16_t* __restrict__ B, int ldb,
                                                          float* __restrict__ part, float* __restrict__ part_cs, int T,
                                                          int I, int J, int tchunk, int nsplits) {
  constexpr int TI = BI / 2, TJ = BJ / 2;
  constexpr int IB = TI / 16, JB = TJ / 16;
  constexpr int ACPR = BI / 8, BCPR = BJ / 8;                      // 16-byte chunks per row
  constexpr int ARW = BTD * BI / 2048, BRW = BTD * BJ / 2048;      // 1 KiB records per wave per stage
  constexpr int STAGE = BTD * (BI + BJ);
  __shared__ __attribute__((aligned(16))) bf16_t smem[NST * STAGE];

  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = w >> 1, wj = w & 1;
  const int tiles_j = J / BJ;
  int tile, split;
  if (!tn_decode(tiles_j * (I / BI), nsplits, tile, split)) return;
  const int i0 = (tile / tiles_j) * BI;
  const int j0 = (tile % tiles_j) * BJ;
  const int t_begin = split * tchunk;
  const int t_end = min(T, t_begin + tchunk);
  const int rows = t_end - t_begin;
  const bool do_cs = (part_cs != nullptr) && (j0 == 0) && (wj == 0);
  const int nst = (rows + BTD - 1) / BTD;  // stages; only the last may be ragged
  const int nfull = rows / BTD;

  // record (w + 4 i) of a stage: this lane's row inside the stage and its (un-swizzled) source column
  int arow[ARW], brow[BRW], aoffg[ARW], boffg[BRW];
#pragma unroll
  for (int i = 0; i < ARW; ++i) {
    const int id = (w + 4 * i) * 64 + l, row = id / ACPR, ch = id % ACPR;
    const int sc = (((ch >> 1) ^ tn_swz<BI * 2>(row)) << 1) | (ch & 1);
    arow[i] = row;
    aoffg[i] = row * lda + i0 + sc * 8;
  }
#pragma unroll
  for (int i = 0; i < BRW; ++i) {
    const int id = (w + 4 * i) * 64 + l, row = id / BCPR, ch = id % BCPR;
    const int sc = (((ch >> 1) ^ tn_swz<BJ * 2>(row)) << 1) | (ch & 1);
    brow[i] = row;
    boffg[i] = row * ldb + j0 + sc * 8;
  }

  f32x4 acc[JB][IB];
  f32x4 accs[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int k = 0; k < 8; ++k) ones[k] = (bf16_t)1.0f;
  bf16x8 af[IB], bfr[JB];

  // LDS-DMA through buffer resources based at the split's first row (see lds_dma16): per-stage offsets fit 32 bits
  const BufRsrc abase = make_rsrc(A + (size_t)t_begin * lda), bbase = make_rsrc(B + (size_t)t_begin * ldb);
  if (nst > 0) {
    // prologue: stages 0 .. NST-2 (the fragment reads of these calls are dead and dropped by the compiler)
#pragma unroll
    for (int k = 0; k < NST - 1; ++k)
      if (k < nst) {
        bf16_t* dst = smem + k * STAGE;
        const int other = (k + 1) % NST;
        tn_dma_and_read<BI, BJ>(abase, (unsigned)k * BTD * lda * 2, bbase, (unsigned)k * BTD * ldb * 2, dst, smem + other * STAGE, true,
                                rows - k * BTD, aoffg, arow, boffg, brow, w, l, wi, wj, af, bfr);
      }
  }
  for (int k = 0; k < nst; ++k) {
    // stage k has landed; the NST-2 younger stages may stay in flight when they are full ones (counted, in-order
    // retirement of loads); everyone is done reading the buffer stage k+NST-1 goes into (it held stage k-1)
    if (k + NST - 2 < nfull) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NST - 2) * (ARW + BRW)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int kn = k + NST - 1;
    tn_dma_and_read<BI, BJ>(abase, (unsigned)kn * BTD * lda * 2, bbase, (unsigned)kn * BTD * ldb * 2, smem + (kn % NST) * STAGE,
                            smem + (k % NST) * STAGE, kn < nst, rows - kn * BTD, aoffg, arow, boffg, brow, w, l, wi, wj, af,
                            bfr);
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int i = 0; i < IB; ++i) acc[j][i] = mfma16(bfr[j], af[i], acc[j][i]);
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < IB; ++i) accs[i] = mfma16(ones, af[i], accs[i]);
    }
  }

  const int g = l >> 4;
  float* slab = part + (size_t)split * I * J;
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int ci = i0 + wi * TI + i * 16 + (l & 15);
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int jj = j0 + wj * TJ + j * 16 + 4 * g;
      *reinterpret_cast<f32x4*>(slab + (size_t)ci * J + jj) = acc[j][i];
    }
    if (do_cs && g == 0) part_cs[(size_t)split * I + ci] = accs[i][0];
  }
}

// slabs -> C (and colsum partials -> cs) in one launch, fixed summation order (deterministic).
// The reduction is one memory round trip deep: a block owns 64 float4 elements, wave q of it sums the q-th quarter of the splits
// with up to 8 INDEPENDENT loads in flight per lane, the four quarter sums meet in LDS and wave 0 adds them in a fixed order.
// (First version: one thread per element walking all splits four at a time -- 8 dependent round trips at 6 waves per CU, 27 us per
// launch for the 2048 x 192 gradients, 52 launches a step.)
template <typename V>
__device__ __forceinline__ V tn_quarter_sum(const float* __restrict__ part, size_t stride, size_t e, int k0, int k1, bool ok) {
  V s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = V{};
  for (int k = k0; k < k1; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (ok && k + u < k1) s[u] += *reinterpret_cast<const V*>(part + (size_t)(k + u) * stride + e);
  }
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int ldc,
                                                        int I, int J, int splits, int accumulate,
                                                        const float* __restrict__ part_cs, float* __restrict__ cs) {
  __shared__ f32x4 red[3][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t n4 = (size_t)I * J / 4;
  const size_t slab = (size_t)I * J;
  const int per = (splits + 3) / 4, k0 = min(q * per, splits), k1 = min(k0 + per, splits);
  const int cblocks = (int)((n4 + 63) / 64);  // blocks [0, cblocks): C; the rest: the column sums, same scheme on scalars
  if ((int)blockIdx.x < cblocks) {
    const size_t idx = (size_t)blockIdx.x * 64 + lane;
    const bool ok = idx < n4;
    const size_t e = idx * 4;
    f32x4 t = tn_quarter_sum<f32x4>(part, slab, e, k0, k1, ok);
    if (q) red[q - 1][lane] = t;
    __syncthreads();
    if (q == 0 && ok) {
      const int i = (int)(e / J), j = (int)(e % J);
      t = (t + red[0][lane]) + (red[1][lane] + red[2][lane]);
      if (accumulate) t += *reinterpret_cast<const f32x4*>(C + (size_t)i * ldc + j);
      *reinterpret_cast<f32x4*>(C + (size_t)i * ldc + j) = t;
    }
  } else if (cs != nullptr) {
    float* redf = reinterpret_cast<float*>(&red[0][0]);
    const int i = ((int)blockIdx.x - cblocks) * 64 + lane;
    const bool ok = i < I;
    float t = tn_quarter_sum<float>(part_cs, (size_t)I, (size_t)i, k0, k1, ok);
    if (q) redf[(q - 1) * 64 + lane] = t;
    __syncthreads();
    if (q == 0 && ok) {
      t = (t + redf[lane]) + (redf[64 + lane] + redf[128 + lane]);
      cs[i] = (accumulate ? cs[i] : 0.f) + t;
    }
  }
}

template <int BI, int BJ>
void launch_tn(const bf16_t* A, int lda, const bf16_t* B, int ldb, float* part, float* part_cs, int T, int I, int J,
               int tchunk, int splits, hipStream_t s, bool occ3 = false) {
  if constexpr (BI == 128 && BJ == 192) {
    // 128 x 192 tiles: TWO stages (40 KB) and 167 registers let three blocks live on a CU instead of two with three stages -- a third wave
    // per SIMD to issue into the gaps of the other two is worth more than the deeper ring: 1 155 -> 1 116 us at 1 206 272 x 2048 x 192
    // (4.67 -> 4.84 TB/s), -5 % on the Small shapes (profiles/r05g_tn_three_blocks_per_cu.log).  The mirrored 192 x 128 tile spills 15
    // registers under the same bound and runs 2x slower: it keeps three stages and two blocks.
    if (occ3) {
      hipLaunchKernelGGL((gemm_tn_dma_kernel<BI, BJ, 2>), dim3((I / BI) * (J / BJ) * ((splits + 7) / 8 * 8)), dim3(256), 0, s, A, lda,
                         B, ldb, part, part_cs, T, I, J, tchunk, splits);
      return;
    }
  }
  hipLaunchKernelGGL((gemm_tn_dma_kernel<BI, BJ, 3>), dim3((I / BI) * (J / BJ) * ((splits + 7) / 8 * 8)), dim3(256), 0, s, A, lda,
                     B, ldb, part, part_cs, T, I, J, tchunk, splits);
}

}  // namespace

extern "C" int chadavit_gemm_tn(const chada_bf16* A_, int lda, const chada_bf16* B_, int ldb, float* C, int ldc,
                                float* colsumA, int T, int I, int J, int accumulate, float* workspace,
                                long long workspace_floats, void* stream) {
  CHADA_ENTRY();
  if (!A_ || !B_ || !C || !workspace || T <= 0 || I <= 0 || J <= 0) return 1;
  if (I % 64 != 0 || J % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0) return 2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(A_);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(B_);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static const int cfgs[8][2] = {{128, 192}, {192, 128}, {128, 128}, {64, 192}, {192, 64}, {128, 64}, {64, 128}, {64, 64}};
  int bi = 0, bj = 0;
  for (auto& c : cfgs)
    if (I % c[0] == 0 && J % c[1] == 0) { bi = c[0]; bj = c[1]; break; }
  // dW_qkv at D = 192 (576 x 192): three 192 x 192 tiles instead of nine 64 x 192 ones -- the shared operand h is pulled through LDS
  // three times instead of nine
  // (A/B switches -- CHADA_TN_NO192, CHADA_TN_WIDE=0, CHADA_TN_OCC3=0, CHADA_TN_OLD_SPLITS -- exist in side builds only: -DCHADA_AB_SWITCHES=1)
#if CHADA_AB_SWITCHES
  static const bool big = getenv("CHADA_TN_NO192") == nullptr;
  static const int wide = getenv("CHADA_TN_WIDE") ? atoi(getenv("CHADA_TN_WIDE")) : 1;
  static const bool occ3 = !(getenv("CHADA_TN_OCC3") && atoi(getenv("CHADA_TN_OCC3")) == 0);   // (0: the three-stage 128 x 192 kernel)
  static const bool old_rule = getenv("CHADA_TN_OLD_SPLITS") != nullptr;   // (the round-2 split rule)
#else
  constexpr bool big = true, occ3 = true, old_rule = false;
  constexpr int wide = 1;
#endif
  if (big && bi == 64 && bj == 192 && I % 192 == 0 && I >= 384) bi = 192;
  // wide outputs (Base: 2304 x 768, 768 x 768): there the kernel is bound by the CU's vector-memory path (LDS-DMA, ~58 B/clk), not by
  // HBM, and a 192 x 192 tile stages 25 % fewer bytes per FLOP than 128 x 192
  if (wide && I % 192 == 0 && J % 192 == 0 && (long long)I * J >= 768ll * 768) { bi = 192; bj = 192; }
  const int tiles = (I / bi) * (J / bj);
  // T-splits.  tn_decode gives split s to XCD s % 8 (all tiles of a split share one L2), each XCD has 32 CUs x 2 resident blocks = 64
  // slots, and a block's time goes with its rows: with n splits per XCD the launch takes ceil(tiles n / 64) rounds of T / (8 n) rows.
  // The old rule (512 / tiles splits, any number) left XCDs with 66 or 72 blocks for 64 slots -- a second round for a handful of
  // blocks: 2304 x 768 ran at 620 TFLOP/s beside 2048 x 768 at 890 on the same tile.  Now: the n with the lowest modelled time,
  // the partial slabs (8 n I J floats written and read once) priced in.
  const long long per = (long long)I * J + I;
  const int max_by_t = (T + 255) / 256;  // at least 256 rows per split
  const long long max_by_ws = workspace_floats / per;
  if (max_by_ws < 1) return 1;
  int splits;
  {
    const double t_full = fmax(2.0 * T * I * J / 0.95e15, 2.0 * T * (double)(I + J) / 4.6e12);   // seconds at full occupancy
    double best = 1e30;
    int best_n = 0;
    const int slots = (occ3 && bi == 128 && bj == 192) ? 96 : 64;   // (128 x 192 tiles: three blocks per CU)
    for (int n = 1; n <= 64; ++n) {
      if (8 * n > max_by_t || 8 * n > max_by_ws) break;
      const int rounds = (tiles * n + slots - 1) / slots;
      const double t = t_full * ((double)slots * rounds) / ((double)tiles * n) + 8.0 * n * (double)I * J * 8.0 / 4.0e12;
      if (t < best) { best = t; best_n = n; }
    }
    splits = best_n ? 8 * best_n : (int)((max_by_ws < max_by_t ? max_by_ws : max_by_t) < 1 ? 1 : (max_by_ws < max_by_t ? max_by_ws : max_by_t));
  }
  if (old_rule) {
    splits = (512 + tiles - 1) / tiles;
    if (splits > max_by_t) splits = max_by_t;
    if (splits > max_by_ws) splits = (int)max_by_ws;
  }
  int tchunk = (T + splits - 1) / splits;
  tchunk = (tchunk + 63) / 64 * 64;
  splits = (T + tchunk - 1) / tchunk;
  float* part = workspace;
  float* part_cs = colsumA ? workspace + (size_t)splits * I * J : nullptr;
#define TN_CASE(a, b) \
  if (bi == a && bj == b) launch_tn<a, b>(A, lda, B, ldb, part, part_cs, T, I, J, tchunk, splits, s, occ3);
  TN_CASE(128, 192) TN_CASE(192, 128) TN_CASE(128, 128) TN_CASE(64, 192) TN_CASE(192, 64) TN_CASE(128, 64)
  TN_CASE(64, 128) TN_CASE(64, 64) TN_CASE(192, 192)
#undef TN_CASE
  CHADA_CHECK_LAUNCH();
  const size_t n4 = (size_t)I * J / 4;
  const int rb = (int)((n4 + 63) / 64) + (colsumA ? (I + 63) / 64 : 0);
  hipLaunchKernelGGL(tn_reduce_kernel, dim3(rb), dim3(256), 0, s, part, C, ldc, I, J, splits, accumulate, part_cs, colsumA);
  CHADA_CHECK_LAUNCH();
  return 0;
}
