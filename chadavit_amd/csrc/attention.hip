// Variable-length (ragged-packed) multi-head self-attention, flash style, for gfx950.
//
// Everything is "column-centric": the MFMAs are issued so that the lane's MFMA column (l & 15) is the
// softmax ROW owner (a query in fwd / dQ, a key in dK/dV).  Then
//   * S^T = K Q^T puts 16 scores of ONE query in each lane -> row max / row sum need only two
//     permlane swaps (lanes l, l^16, l^32, l^48 share a query), no LDS round trip;
//   * the probabilities are already in B-operand layout for the second MFMA (O^T = V^T P^T), with the
//     k-slot order {4g..4g+3} U {16+4g..16+4g+3} that the hardware transpose read
//     (ds_read_b64_tr_b16, lds_read_tr8) produces for the V^T / K^T / Q^T / dO^T operand;
//   * the accumulators hold 4 consecutive head-dim elements of one row -> 8-byte bf16 stores.
// Work items are (image, 128-row tile) pairs from the host-built list (sequences are 1 + C_i * p
// tokens: 109 ... 1961), grid = (n_work, heads).
//
// Backward is two kernels (dQ; dK+dV) that recompute the probabilities from the saved LSE: no atomics,
// deterministic.  replaces chada_vit.py:105-111 (nn.MultiheadAttention + key padding mask) fwd/bwd.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace chada;

namespace {

#ifndef CHADA_AB_SWITCHES
#define CHADA_AB_SWITCHES 0   // 1 (side builds only): the superseded kernels and the environment switches that select them
#endif
constexpr int TILE = 128;  // rows per work item
constexpr int KV = 64;     // rows staged per inner step
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// stage ROWS x DH bf16 from global (row stride ld_g, rows clamped to [0, nrows)) into LDS (stride LD)
template <int DH, int LD, int ROWS>
struct Stager {
  static constexpr int CPR = DH / 8;               // 16-byte chunks per row
  static constexpr int NCH = ROWS * CPR / 256;     // chunks per thread
  static_assert((ROWS * CPR) % 256 == 0, "tile must split evenly over 256 threads");
  u32x4 r[NCH];
  __device__ __forceinline__ void load(const bf16_t* __restrict__ base, size_t ld_g, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + 256 * i, row = id / CPR, ch = id % CPR;
      const int rr = min(row0 + row, nrows - 1);
      r[i] = *reinterpret_cast<const u32x4*>(base + (size_t)rr * ld_g + ch * 8);
    }
  }
  __device__ __forceinline__ void store(bf16_t* lds, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + 256 * i, row = id / CPR, ch = id % CPR;
      *reinterpret_cast<u32x4*>(lds + row * LD + ch * 8) = r[i];
    }
  }
};

// Work decode.  The grid is 1-D; hardware places block i on XCD i % 8 and the host lays the work list out so that entry
// j belongs to XCD j % 8, with all tiles of one sequence in the SAME XCD's sub-list (RaggedBatch): the tiles of a
// sequence then share one L2 for their K/V (fwd, dQ) or Q/dO (dK/dV) sweeps instead of each pulling its own copy over
// the fabric.  Entries with image < 0 are padding.
struct WorkItem { int b, t, h, part; };
template <int SPLIT>
__device__ __forceinline__ WorkItem decode_work(const int* __restrict__ work, int H) {
  const int lin = blockIdx.x, xcd = lin & 7;
  int rest = lin >> 3;
  WorkItem it;
  it.part = rest % SPLIT; rest /= SPLIT;
  it.h = rest % H;
  const int wi = (rest / H) * 8 + xcd;
  it.b = work[2 * wi];
  it.t = work[2 * wi + 1];
  return it;
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
  r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
  return r;
}

// Epilogue store widening (round 5).  An accumulator block holds, in lane (li, g), 4 consecutive head-dim elements
// d = blk * 16 + 4 g + r of row li: stored as it is, one instruction writes 16 rows x 32 bytes.  What an attention block pays at its
// end is the ISSUE of these stores through the CU's vector-memory path, not their completion: a persistent dK/dV kernel that never
// waits for a store (and prefetches the next work item's first tile and fragments across the item boundary) ties the one-item
// kernel, its timeline shows 4 us of a 32 us item in the epilogue's 24 loads + stores per wave, and without the stores it is 8 %
// faster (profiles/r05a_attention_persistent.md; source scratch/r5/attention_with_persistent_dkv.hip.txt).  Two blocks
// (blk, blk + 1) packed to bf16 and exchanged between the 16-lane rows (v_permlane16_swap) leave 8 consecutive elements = 16 bytes per
// lane: lane group g owns elements [blk * 16 + (g & 1) * 16 + (g >> 1) * 8, + 8) -- one instruction writes 16 rows x 64 contiguous
// bytes, half as many instructions: dK/dV -5.5 %, dQ -3.6 %, forward -3.5 % at 589 tokens, forward -7 % at 109 (same box, A/B).
__device__ __forceinline__ u32x4 widen_pair(const f32x4& a, const f32x4& b) {
  const u32x2 pa = __builtin_bit_cast(u32x2, pack4(a[0], a[1], a[2], a[3]));
  const u32x2 pb = __builtin_bit_cast(u32x2, pack4(b[0], b[1], b[2], b[3]));
  unsigned a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
  swap16x(a0, b0);
  swap16x(a1, b1);
  return u32x4{a0, a1, b0, b1};
}
__device__ __forceinline__ int widen_col(int g) { return (g & 1) * 16 + (g >> 1) * 8; }  // element offset of the lane's 16 bytes inside the pair
// One row of DB accumulator blocks times `mul` -> bf16 at `row` (the row's first element of this head; a dereferenceable address also
// when !valid).  Every lane of the wave must get here (the exchanges run unpredicated); only the stores are predicated.
template <int DB>
__device__ __forceinline__ void store_row_blocks(bf16_t* row, bool valid, const f32x4 (&acc)[DB], float mul, int g) {
  if constexpr (DB % 2 == 0) {
#pragma unroll
    for (int db = 0; db < DB; db += 2) {
      const u32x4 v = widen_pair(acc[db] * mul, acc[db + 1] * mul);
      if (valid) *reinterpret_cast<u32x4*>(row + db * 16 + widen_col(g)) = v;
    }
  } else {
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const f32x4 v = acc[db] * mul;
      if (valid) *reinterpret_cast<bf16x4*>(row + db * 16 + 4 * g) = pack4(v[0], v[1], v[2], v[3]);
    }
  }
}

// =====================================================================================
// forward
// =====================================================================================
template <int DH, int CB>
__global__ __launch_bounds__(256, (DH <= 96 ? 2 : 1)) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, const int* __restrict__ cu,
                                                       const int* __restrict__ work, int T, int D, int H, float scale) {
  // CB = 16-query column blocks per wave: a block covers 64*CB query rows; a 128-row work item is split over 2/CB blocks
  constexpr int KS = DH / 32;      // k-steps over the head dim
  constexpr int DB = DH / 16;      // 16-wide output blocks over the head dim
  constexpr int LDK = DH + 16;     // K tile stride = 32 B x odd: conflict-free for the lane groups ds_read_b128 serves (see dkv_swz)
  constexpr int LDV = DH + 16;     // V tile stride: bytes = 32 mod 64 -> conflict-free transpose reads
  __shared__ __attribute__((aligned(16))) bf16_t smem[KV * (LDK + LDV)];
  bf16_t* sK = smem;
  bf16_t* sV = smem + KV * LDK;

  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  constexpr int SPLIT = 2 / CB;
  const WorkItem it = decode_work<SPLIT>(work, H);
  const int b = it.b, qt = it.t, h = it.h, part = it.part;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE + part * 64 * CB >= len) return;  // this part of the tile is beyond the sequence (block-uniform)
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* kbase = qbase + D;
  const bf16_t* vbase = qbase + 2 * D;
  const float c = scale * LOG2E;

  // Q fragments (B operand: column = query, k = head dim), resident for the whole KV sweep
  bf16x8 qf[CB][KS];
  int qrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qrow[cb] = qt * TILE + part * 64 * CB + w * 16 * CB + cb * 16 + li;
    const int qr = min(qrow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[cb][ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 32 + g * 8);
  }
  f32x4 o[CB][DB];
  float m[CB], ls[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    m[cb] = -INFINITY;
    ls[cb] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db) o[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  Stager<DH, LDK, KV> stK;
  Stager<DH, LDV, KV> stV;
  const int nkt = (len + KV - 1) / KV;
  stK.load(kbase, ld, 0, len, tid);
  stV.load(vbase, ld, 0, len, tid);
  // one KV tile; MASKED only for the sequence's last tile (keys >= len get -inf)
  auto tile = [&](int kt, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    stK.store(sK, tid);
    stV.store(sV, tid);
    __syncthreads();
    if (!MASKED) {
      stK.load(kbase, ld, (kt + 1) * KV, len, tid);
      stV.load(vbase, ld, (kt + 1) * KV, len, tid);
    }
    // ---- S^T = K Q^T : lane holds scores of query column li for keys kt*64 + kb*16 + 4g + r
    f32x4 s[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[cb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = lds_read8(sK + (kb * 16 + li) * LDK + ks * 32 + g * 8);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) s[cb][kb] = mfma16(kf, qf[cb][ks], s[cb][kb]);
      }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (MASKED && (kt * KV + kb * 16 + 4 * g + r >= len)) s[cb][kb][r] = -INFINITY;
          mx = fmaxf(mx, s[cb][kb][r]);
        }
      mx = rows_max(mx);
      const float mn = fmaxf(m[cb], mx * c);  // running max in the scaled log2 domain (c > 0)
      const float alpha = __builtin_amdgcn_exp2f(m[cb] - mn);
      m[cb] = mn;
      float ps = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[cb][kb][r], c, -mn));
          s[cb][kb][r] = p;
          ps += p;
        }
      ls[cb] = ls[cb] * alpha + ps;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // wave-uniform: once the running max has settled, no rescale
#pragma unroll
        for (int db = 0; db < DB; ++db) o[cb][db] *= alpha;
      }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      bf16x8 pf[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) pf[cb] = pack8(s[cb][2 * k2], s[cb][2 * k2 + 1]);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x8 vf = lds_read_tr8(sV + (k2 * 32) * LDV + db * 16, LDV);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) o[cb][db] = mfma16(vf, pf[cb], o[cb][db]);
      }
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nkt - 1; ++kt) tile(kt, std::false_type{});
  tile(nkt - 1, std::true_type{});
  // ---- finish: row sums across the 4 lane groups, normalise, store
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    float lt = ls[cb];
    lt = rows_sum(lt);
    const float inv = 1.0f / lt;
    store_row_blocks<DB>(out + (size_t)(seq0 + min(qrow[cb], len - 1)) * D + h * DH, qrow[cb] < len, o[cb], inv, g);
    if (g == 0 && qrow[cb] < len) lse[(size_t)h * T + seq0 + qrow[cb]] = (m[cb] + log2f(lt)) * LN2;
  }
}

// One key tile of the LDS-DMA forward: issue the DMA of tile kt+1 into `dst` (when `issue`), then S^T = K Q^T, the online
// softmax and O^T += V^T P^T out of the stage `sK`.  DMA issue and LDS reads share ONE function with __restrict__ pointers on
// purpose: after inlining the reads carry scoped-noalias metadata against the DMA, which keeps the compiler's waitcnt
// insertion from draining the DMA queue (s_waitcnt vmcnt(0)) in front of the first LDS read -- by itself it cannot tell the
// two stages apart and would put the whole load latency of tile kt+1 in front of the math of tile kt.  (The kernel must
// also keep a single __shared__ object: with several, the LDS lowering replaces these scopes by per-variable ones.)
template <int DH>
__device__ __forceinline__ int dkv_swz(int row);   // (defined with the backward kernels' row-major stages)

// -DCHADA_FWD_TIMELINE (side builds only, scratch/r5/fwd_timeline.py): wave 0 of every block sums s_memtime differences per phase of the tile
#ifdef CHADA_FWD_TIMELINE
__device__ unsigned long long g_fwd_tl[4096 * 8 * 8];   // [block][wave][slot]
#define TL_NOW() __builtin_amdgcn_s_memtime()
#define TL_ADD(slot, a, b) tl[slot] += (b) - (a)
#else
#define TL_NOW() 0ull
#define TL_ADD(slot, a, b) ((void)0)
#endif

template <int DH>
struct FwdDmaCfg {
  static constexpr int KS = (DH + 31) / 32, DB = DH / 16;
  // keys per tile: 64 up to dh = 96, 32 above -- the same 24 records (24 KiB) per stage and 48 MFMAs per wave and tile, so a
  // dh = 192 block keeps 2 stages in 48 KiB and fits 256 VGPRs: 2 blocks per CU instead of one with 96 KiB / 396 VGPRs
  static constexpr int KVT = (DH > 96) ? 32 : 64, KB = KVT / 16, K2 = KVT / 32;
  static constexpr int NKR = KB * KS, NVR = K2 * DB, NR = NKR + NVR;  // 1 KiB records per stage
  // waves per block: 4 x 32 (16 above dh = 96... see CB) query rows; dh = 384 (Base) runs EIGHT waves x 16 rows as the CU's only block:
  // its 48 KiB stages (2 x 48 = 96 KiB) are then shared by 128 query rows instead of 64, two waves per SIMD
  static constexpr int NW = (DH > 192) ? 8 : 4;
  static constexpr int NRW = (NR + NW - 1) / NW;                     // LDS-DMA instructions per wave and tile
  static constexpr int STAGE = NR * 512;
};

template <int DH, int CB, bool MASKED, bool RM = false>
__device__ __forceinline__ void attn_fwd_tile(BufRsrc qb, bf16_t* __restrict__ dst,
                                              const bf16_t* __restrict__ sK, bool issue, int kt, int len, int qrow0, unsigned ldu, float c,
                                              int w, int l, const int (&rec_row)[FwdDmaCfg<DH>::NRW],
                                              const unsigned (&rec_col)[FwdDmaCfg<DH>::NRW],
                                              const bf16x8 (&qf)[CB][FwdDmaCfg<DH>::KS], f32x4 (&o)[CB][DH / 16], float (&m)[CB],
                                              float (&ls)[CB], unsigned long long (&tl)[8]) {
  using C = FwdDmaCfg<DH>;
  [[maybe_unused]] const unsigned long long tl0 = TL_NOW();
  constexpr int KS = C::KS, DB = C::DB, KVT = C::KVT, KB = C::KB, K2 = C::K2, NKR = C::NKR, NR = C::NR, NRW = C::NRW;
  const int g = l >> 4;
  if (issue) {
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      if (NR % C::NW != 0 && w + C::NW * i >= NR) continue;  // wave-uniform: NR is not a multiple of the wave count for dh = 16
      const unsigned off = (unsigned)min((kt + 1) * KVT + rec_row[i], len - 1) * ldu + rec_col[i];
      lds_dma16(qb, dst + (w + C::NW * i) * 512, off * 2, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST: free of the alias edge, the scheduler would sink it below the math
  if (qrow0 >= len) return;  // (wave-uniform) none of this wave's query rows exists: it only feeds the DMA and the barriers
  [[maybe_unused]] const unsigned long long tl1 = TL_NOW();
  TL_ADD(1, tl0, tl1);   // DMA issue
  const bf16_t* sV = sK + NKR * 512;
  // The fragment reads run TWO steps ahead of the MFMAs that consume them (ring of three, pinned with sched_barrier):
  // left alone hipcc issues each read right in front of its MFMAs and waits for it.  The first two V fragments are
  // requested before the softmax and land under it.
  f32x4 s[CB][KB];
  constexpr int PD = 2;   // look-ahead of the fragment ring in steps (dh 384: 3, 4, 6 and 10 measured the same as 2)
  bf16x8 fr[PD + 1];
  constexpr int NS = KB * KS, NP = K2 * DB;
  // last tile of the sequence: only the 16-key blocks that hold a valid key are multiplied / exponentiated (len = 589:
  // 13 keys = one block of four) -- wave-uniform branches, identical results (the skipped scores are -inf, their P is 0)
  const int nvb = MASKED ? min(KB, (len - kt * KVT + 15) >> 4) : KB;
  // RM: the stage is the ROW-MAJOR image of the K and V tiles (whole 128-byte lines per LDS-DMA instruction, chunks XOR-swizzled on the source
  // side as in the backward kernels' stages: dkv_swz) -- K fragments by row reads, V^T fragments by the transpose read of the row-major tile
  const int li_ = l & 15;
  auto k_read = [&](int st) {
    if constexpr (RM) {
      const int row = (st / KS) * 16 + li_, ch = ((st % KS) * 4 + g) ^ dkv_swz<DH>(row);
      return lds_read8(sK + row * DH + ch * 8);
    } else {
      return lds_read8(sK + st * 512 + l * 8);
    }
  };
  auto v_read = [&](int st) {
    if constexpr (RM) {
      const int k2 = st / DB, db = st % DB;
      const int trow = k2 * 32 + 4 * g + (li_ >> 2);
      const int ch = (2 * db + ((li_ & 3) >> 1)) ^ dkv_swz<DH>(trow);
      const int off = trow * DH + ch * 8 + (li_ & 1) * 4;
      return __builtin_shufflevector(lds_read_tr4(sV + off), lds_read_tr4(sV + off + 16 * DH), 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      return lds_read_tr8(sV + st * 512, 16);
    }
  };
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < NS) fr[i] = k_read(i);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int kb = st / KS, ks = st % KS;
    if (st + PD < NS) fr[(st + PD) % (PD + 1)] = k_read(st + PD);
    __builtin_amdgcn_sched_barrier(0);
    if (!MASKED || kb < nvb) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
        s[cb][kb] = (ks == 0) ? mfma16(fr[st % (PD + 1)], qf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(fr[st % (PD + 1)], qf[cb][ks], s[cb][kb]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  [[maybe_unused]] const unsigned long long tl2 = TL_NOW();
  TL_ADD(2, tl1, tl2);   // S^T MFMAs (+ their fragment reads)
  bf16x8 vr[PD + 1];
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < NP) vr[i] = v_read(i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (MASKED && kb >= nvb) continue;
      if (MASKED) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * KVT + kb * 16 + 4 * g + r >= len) s[cb][kb][r] = -INFINITY;
      }
      // plain fmaxf (folds to v_max3_f32): an inline-asm max here reads MFMA results the hazard recogniser cannot see --
      // with a single k-step (dh = 16) the asm followed the last MFMA too closely and read garbage
      mx = fmaxf(fmaxf(mx, s[cb][kb][0]), s[cb][kb][1]);
      mx = fmaxf(fmaxf(mx, s[cb][kb][2]), s[cb][kb][3]);
    }
    mx = rows_max(mx);
    const float mn = fmaxf(m[cb], mx * c);
    const float alpha = __builtin_amdgcn_exp2f(m[cb] - mn);
    m[cb] = mn;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (MASKED && kb >= nvb) {
        s[cb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[cb][kb][r], c, -mn));
        s[cb][kb][r] = p;
        ps += p;
      }
    }
    ls[cb] = ls[cb] * alpha + ps;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int db = 0; db < DB; ++db) o[cb][db] *= alpha;
    }
  }
  bf16x8 pf[K2][CB];
#pragma unroll
  for (int k2 = 0; k2 < K2; ++k2)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) pf[k2][cb] = pack8(s[cb][2 * k2], s[cb][2 * k2 + 1]);
  __builtin_amdgcn_sched_barrier(0);
  [[maybe_unused]] const unsigned long long tl3 = TL_NOW();
  TL_ADD(3, tl2, tl3);   // softmax
#pragma unroll
  for (int st = 0; st < NP; ++st) {
    const int k2 = st / DB, db = st % DB;
    if (st + PD < NP) vr[(st + PD) % (PD + 1)] = v_read(st + PD);
    __builtin_amdgcn_sched_barrier(0);
    if (!MASKED || 2 * k2 < nvb) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) o[cb][db] = mfma16(vr[st % (PD + 1)], pf[k2][cb], o[cb][db]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  [[maybe_unused]] const unsigned long long tl4 = TL_NOW();
  TL_ADD(4, tl3, tl4);   // O^T MFMAs (+ reads)
}

// =====================================================================================
// forward, LDS-DMA variant (DH <= 192): K/V tiles go global -> LDS by LDS-DMA into two FRAGMENT-MAJOR stages -- no staging
// registers, no ds_write pass (those writes were the kernel's 28 % LDS bank-conflict cycles), one barrier per 64-key tile.
//   K record (kb, ks): 1 KiB, lane l = (li, g) holds K[key kb*16 + li][d = ks*32 + 8g .. +7]  -> A operand of S^T by one
//                      conflict-free ds_read_b128 at lane * 16;
//   V record (k2, db): 1 KiB = V[keys k2*32 .. +31][d = db*16 .. +15] row-major (32 B rows)     -> A operand of O^T by the
//                      hardware transpose read (two ds_read_b64_tr_b16), 256 contiguous bytes per 32-lane phase.
// Same arithmetic (and bit-identical results) as attn_fwd_kernel.
// =====================================================================================
template <int DH, int CB, bool RM = false>
__global__ __launch_bounds__(64 * FwdDmaCfg<DH>::NW, (DH <= 192 ? 2 : 1)) void attn_fwd_dma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse, const int* __restrict__ cu,
                                                           const int* __restrict__ work, int T, int D, int H, float scale) {
  // dh = 16 (12 heads at D = 192: the reference's default constructor, HOW_TO_USE.ipynb cell 13) runs as ONE 32-wide k-step
  // whose upper 16 slots are zero in the Q fragments; the K records then carry 16 columns of the neighbouring head (or of
  // the V section) in those slots -- finite values times zero.
  using C = FwdDmaCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, KVT = C::KVT, NKR = C::NKR, NR = C::NR, NRW = C::NRW, STAGE = C::STAGE, NW = C::NW;
  constexpr int QPB = NW * 16 * CB;  // query rows per block
  // (dh 384, the CU's only block: THREE stages with a counted wait -- tile t+2 requested while tile t is consumed -- and fragment rings four to ten
  // steps deep were built and measured: 880 against 885-893 us and no change; scratch/r5/attention_fwd_rowmajor_3stage_ring.patch, profiles/r05o_*)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, li = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SPLIT = TILE / QPB;
  const WorkItem it = decode_work<SPLIT>(work, H);
  const int b = it.b, qt = it.t, h = it.h, part = it.part;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE + part * QPB >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const float c = scale * LOG2E;

  bf16x8 qf[CB][KS];
  int qrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qrow[cb] = qt * TILE + part * QPB + w * 16 * CB + cb * 16 + li;
    const int qr = min(qrow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks * 32 + g * 8 < DH) {
        qf[cb][ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 32 + g * 8);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[cb][ks][e] = (bf16_t)0.f;
      }
    }
  }
  f32x4 o[CB][DB];
  float m[CB], ls[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    m[cb] = -INFINITY;
    ls[cb] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db) o[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // record r of a tile is fetched by wave r & 3 (instruction r >> 2 of that wave).  Per lane and record: the key row inside
  // the tile and the element offset of its 16-byte piece; per tile only `min(row, len-1) * ld` is recomputed (32-bit).
  int rec_row[NRW];
  unsigned rec_col[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int r = w + NW * i;
    if constexpr (RM) {   // piece r = 64 consecutive 16-byte chunks of the row-major K (r < NKR) / V image; source chunk un-swizzled
      static_assert(!RM || (NR == 2 * NKR && DH % 32 == 0), "row-major stages: K and V tiles of equal size");
      const int id = (r % NKR) * 64 + l, row = id / (DH / 8), ch = id % (DH / 8);
      rec_row[i] = row;
      rec_col[i] = (r < NKR ? D : 2 * D) + (ch ^ dkv_swz<DH>(row)) * 8;
    } else if (r < NKR) {
      rec_row[i] = (r / KS) * 16 + li;
      rec_col[i] = D + (r % KS) * 32 + g * 8;
    } else {
      const int rv = r - NKR;
      rec_row[i] = (rv / DB) * 32 + (l >> 1);
      rec_col[i] = 2 * D + (rv % DB) * 16 + (l & 1) * 8;
    }
  }
  const unsigned ldu = 3u * (unsigned)D;
  const int nkt = (len + KVT - 1) / KVT;
  const int qrow0 = qt * TILE + part * QPB + w * 16 * CB;  // first query row of this wave
  const BufRsrc qrs = make_rsrc(qbase);  // LDS-DMA through a buffer resource: see lds_dma16
  unsigned long long tl[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  [[maybe_unused]] const unsigned long long tl_start = TL_NOW();
  // tile 0: no LDS read follows before the first barrier, issued bare
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    if (NR % NW != 0 && w + NW * i >= NR) continue;  // wave-uniform: NR is not a multiple of the wave count for dh = 16
    const unsigned off = (unsigned)min(rec_row[i], len - 1) * ldu + rec_col[i];
    lds_dma16(qrs, smem + (w + NW * i) * 512, off * 2, 0);
  }
  for (int kt = 0; kt < nkt - 1; ++kt) {
    // tile kt has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt) and everybody is done
    // reading the other stage
    [[maybe_unused]] const unsigned long long tb0 = TL_NOW();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    [[maybe_unused]] const unsigned long long tb1 = TL_NOW();
    if (kt == 0) TL_ADD(5, tl_start, tb1); else TL_ADD(0, tb0, tb1);   // slot 5: block start -> first tile ready; slot 0: wait + barrier
    attn_fwd_tile<DH, CB, false, RM>(qrs, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE, true, kt, len, qrow0, ldu, c, w, l,
                                     rec_row, rec_col, qf, o, m, ls, tl);
  }
  [[maybe_unused]] const unsigned long long tb0 = TL_NOW();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  [[maybe_unused]] const unsigned long long tb1 = TL_NOW();
  TL_ADD(0, tb0, tb1);
  attn_fwd_tile<DH, CB, true, RM>(qrs, smem + (nkt & 1) * STAGE, smem + ((nkt - 1) & 1) * STAGE, false, nkt - 1, len, qrow0, ldu, c, w, l,
                                  rec_row, rec_col, qf, o, m, ls, tl);
  [[maybe_unused]] const unsigned long long tl_loop_end = TL_NOW();
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const float lt = rows_sum(ls[cb]);
    const float inv = 1.0f / lt;
    store_row_blocks<DB>(out + (size_t)(seq0 + min(qrow[cb], len - 1)) * D + h * DH, qrow[cb] < len, o[cb], inv, g);
    if (g == 0 && qrow[cb] < len) lse[(size_t)h * T + seq0 + qrow[cb]] = (m[cb] + log2f(lt)) * LN2;
  }
#ifdef CHADA_FWD_TIMELINE
  if (l == 0 && blockIdx.x < 4096 && w < 8) {
    const unsigned long long tl_end = TL_NOW();
    tl[6] = tl_end - tl_loop_end;   // epilogue issue
    tl[7] = tl_end - tl_start;      // block life
    for (int i = 0; i < 8; ++i) g_fwd_tl[(blockIdx.x * 8 + w) * 8 + i] = tl[i];
  }
#endif
}

// =====================================================================================
// forward at dh = 384 with the two waves of every SIMD in COMPLEMENTARY phases (round 5; profiles/r05o_*, section 4: in attn_fwd_dma_kernel<384>
// both waves of a SIMD leave the tile's barrier together, multiply together and exponentiate together -- 768 cycles of MFMAs in ~3 600 per wave
// and tile, nobody covering the other's non-matrix phases).  Here the block's eight waves are two halves (A = waves 0-3, B = 4-7; wave w and w + 4
// share a SIMD) that run the SAME program one segment apart, two segments and two barriers per key tile:
//   X_t (matrix):     O += P(t-1) V(t-1)   then   S(t) = K(t) Q^T          -- 48 MFMAs, their fragment reads, nothing else
//   Y_t (everything else):  softmax of S(t) -> P(t), rescale of O, and the LDS-DMA issue of a later tile
// A runs X at even segments and Y at odd ones, B the other way round: beside every matrix segment sits the partner's softmax / DMA segment.
// Stages (fragment-major records as in attn_fwd_dma_kernel): K tiles in a ring of TWO, V tiles in a ring of THREE (120 KiB) -- K(t) is read at
// segments 2t (A) and 2t+1 (B), V(t) at 2t+2 and 2t+3 -- so that every refill is issued by the half that is in its Y segment when the slot falls
// free: B fetches K(t+2) in its Y_t, A fetches V(t+1) in its Y_t; each wave waits for its own pieces (vmcnt(0)) at the end of its next X segment,
// the barrier behind it publishes them.  Same arithmetic and the same order of operations as attn_fwd_dma_kernel<384>: bit-identical results.
// =====================================================================================
template <int DH>
__global__ __launch_bounds__(512, 1) void attn_fwd_pair_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                               const int* __restrict__ cu, const int* __restrict__ work, int T, int D, int H,
                                                               float scale) {
  constexpr int KS = DH / 32, DB = DH / 16, KVT = 32, NKR = 2 * KS, NVR = DB;   // 1 KiB records of a K tile (kb, ks) / of a V tile (db)
  constexpr int KSLOT = NKR * 512, VSLOT = NVR * 512, NPW = NKR / 4;            // bf16 elements per slot; pieces per wave and tile
  static_assert(NKR == NVR && NKR % 4 == 0, "K and V tiles of equal size, split over the four waves of a half");
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * KSLOT + 3 * VSLOT];
  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, li = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool halfB = w >= 4;
  const int wq = w & 3;
  const WorkItem it = decode_work<1>(work, H);
  const int b = it.b, qt = it.t, h = it.h;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const float c = scale * LOG2E;
  const int qrow = qt * TILE + w * 16 + li, qrow0 = qt * TILE + w * 16;
  const bool idle = qrow0 >= len;   // (wave-uniform) none of this wave's query rows exists: it only feeds the DMA and the barriers
  bf16x8 qf[KS];
  {
    const int qr = min(qrow, len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 32 + g * 8);
  }
  f32x4 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, ls = 0.f;
  // this wave's pieces of a tile: half B fetches K records, half A V records (record r = wq + 4 i)
  int rec_row[NPW];
  unsigned rec_col[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int r = wq + 4 * i;
    if (halfB) {
      rec_row[i] = (r / KS) * 16 + li;
      rec_col[i] = D + (r % KS) * 32 + g * 8;
    } else {
      rec_row[i] = l >> 1;
      rec_col[i] = 2 * D + r * 16 + (l & 1) * 8;
    }
  }
  const unsigned ldu = 3u * (unsigned)D;
  const int nkt = (len + KVT - 1) / KVT;
  const BufRsrc qrs = make_rsrc(qbase);
  int opq = 0;
  asm volatile("" : "+s"(opq));   // (the LDS bases go through an opaque zero: see ffn_fused.hip)
  bf16_t* const sKb = smem + opq;
  bf16_t* const sVb = smem + 2 * KSLOT + opq;
  auto fetch = [&](int kt, bf16_t* __restrict__ dst) {   // this wave's NPW pieces of tile kt (K for half B, V for half A)
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const unsigned off = (unsigned)min(kt * KVT + rec_row[i], len - 1) * ldu + rec_col[i];
      lds_dma16(qrs, dst + (wq + 4 * i) * 512, off * 2, 0);
    }
  };
  // ---- the two segment bodies.  Everything that reads a stage lives in these (restrict-scoped pointers: no vmcnt(0) drain in front of the reads)
  f32x4 sacc[2];
  bf16x8 pf;
  auto seg_x = [&](const bf16_t* __restrict__ sV, const bf16_t* __restrict__ sK, auto do_pv_tag, auto do_s_tag) {
    constexpr bool DO_PV = decltype(do_pv_tag)::value, DO_S = decltype(do_s_tag)::value;
    if (idle) return;
    // one ring of three over the 24 V^T fragments (transpose reads) and the 24 K fragments (row reads).  The S^T steps alternate between the two
    // 16-key blocks: each block's 12 MFMAs are a dependent chain, and with the partner wave in its Y segment nothing else fills the gaps of ONE chain
    constexpr int NPV = DB, NS = 2 * KS;
    constexpr int FIRST = DO_PV ? 0 : NPV, LAST = DO_S ? NPV + NS : NPV;
#ifndef CHADA_PAIR_ABL
#define CHADA_PAIR_ABL 0   // ablations for timing only (wrong results): 1 = no refills, 2 = no fragment reads, 4 = no MFMAs, 8 = no softmax, 16 = V by plain reads
#endif
    auto rd = [&](int st) {   // step st >= NPV: k-step (st - NPV) / 2 of key block (st - NPV) % 2
      if constexpr ((CHADA_PAIR_ABL & 2) != 0) return qf[st % KS];
      if constexpr ((CHADA_PAIR_ABL & 16) != 0) { if (st < NPV) return lds_read8(sV + st * 512 + l * 8); }   // (plain 16-byte reads instead of the transpose reads)
      if (st < NPV) return lds_read_tr8(sV + st * 512, 16);
      const int kb = (st - NPV) % 2, ks = (st - NPV) / 2;
      return lds_read8(sK + (kb * KS + ks) * 512 + l * 8);
    };
#ifndef CHADA_PAIR_PD
#define CHADA_PAIR_PD 2
#endif
    constexpr int PD = CHADA_PAIR_PD, NF = PD + 1;   // fragments requested PD steps ahead of their MFMA
    bf16x8 fr[NF];
#pragma unroll
    for (int i = 0; i < PD; ++i) fr[(FIRST + i) % NF] = rd(FIRST + i);
#pragma unroll
    for (int st = FIRST; st < LAST; ++st) {
      if (st + PD < LAST) fr[(st + PD) % NF] = rd(st + PD);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((CHADA_PAIR_ABL & 4) != 0) {
        if (st < NPV) o[st][0] += (float)fr[st % NF][0]; else sacc[(st - NPV) % 2][0] += (float)fr[st % NF][1];
      } else if (st < NPV) {
        o[st] = mfma16(fr[st % NF], pf, o[st]);
      } else {
        const int kb = (st - NPV) % 2, ks = (st - NPV) / 2;
        sacc[kb] = (ks == 0) ? mfma16(fr[st % NF], qf[0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(fr[st % NF], qf[ks], sacc[kb]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto seg_y = [&](int kt, auto masked_tag) {   // softmax of tile kt (attn_fwd_tile's arithmetic, CB = 1).  The last tile's S^T is computed for both
    // 16-key blocks whatever the sequence length (clamped key rows, finite): a block beyond the end exponentiates -inf to exactly the 0 that
    // skipping it produced in attn_fwd_tile -- one MFMA stream without a branch per step
    constexpr bool masked = decltype(masked_tag)::value;
    if (idle) return;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (masked) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * KVT + kb * 16 + 4 * g + r >= len) sacc[kb][r] = -INFINITY;
      }
      mx = fmaxf(fmaxf(mx, sacc[kb][0]), sacc[kb][1]);
      mx = fmaxf(fmaxf(mx, sacc[kb][2]), sacc[kb][3]);
    }
    mx = rows_max(mx);
    const float mn = fmaxf(m, mx * c);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c, -mn));
        sacc[kb][r] = p;
        ps += p;
      }
    }
    ls = ls * alpha + ps;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] *= alpha;
    }
    pf = pack8(sacc[0], sacc[1]);
  };
  // ---- prologue: K(0), K(1) by half B, V(0) by half A; first barrier publishes them
  if (halfB) {
    fetch(0, sKb);
    if (nkt > 1) fetch(1, sKb + KSLOT);
  } else {
    fetch(0, sVb);
  }
  // (the builtin, not asm: hipcc's wait insertion must KNOW that the Q fragment loads above have landed -- it does not read asm waits, is
  // path-insensitive, and would otherwise count them down with vmcnt(11) .. vmcnt(0) in front of the S^T MFMAs of EVERY tile, draining the
  // LDS-DMA pieces issued one segment earlier with them)
  __builtin_amdgcn_s_waitcnt(0x0070);
  asm volatile("s_barrier" ::: "memory");
  if (halfB) asm volatile("s_barrier" ::: "memory");   // half B runs one segment behind
  // X_0: S(0) only
  seg_x(sVb, sKb, std::false_type{}, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int kslot = 0, vslot = 0;   // element offsets of K(kt) / V(kt)
  auto refill = [&](int kt) {   // in Y_kt: half B fetches K(kt + 2) over K(kt) (last read one segment ago: this half's X_kt), half A fetches V(kt + 1)
    if constexpr ((CHADA_PAIR_ABL & 1) == 0) {   // into the slot of V(kt - 2) (last read by half B two segments ago)
      if (halfB) {
        if (kt + 2 < nkt) fetch(kt + 2, sKb + kslot);
      } else {
        const int vnext = (vslot == 2 * VSLOT) ? 0 : vslot + VSLOT;
        if (kt + 1 < nkt) fetch(kt + 1, sVb + vnext);
      }
    }
  };
  for (int kt = 0; kt < nkt - 1; ++kt) {
    // Y_kt: the refill first (it goes out while the exponentials run), then the softmax
    refill(kt);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((CHADA_PAIR_ABL & 8) == 0) seg_y(kt, std::false_type{}); else pf = pack8(sacc[0], sacc[1]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // X_{kt+1}: O += P(kt) V(kt), then S(kt + 1)
    const int knext = KSLOT - kslot;
    seg_x(sVb + vslot, sKb + knext, std::true_type{}, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    kslot = knext;
    vslot = (vslot == 2 * VSLOT) ? 0 : vslot + VSLOT;
  }
  // the last tile: Y (masked softmax), then O += P V alone
  if constexpr ((CHADA_PAIR_ABL & 8) == 0) seg_y(nkt - 1, std::true_type{}); else pf = pack8(sacc[0], sacc[1]);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  seg_x(sVb + vslot, sKb, std::true_type{}, std::false_type{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (!halfB) asm volatile("s_barrier" ::: "memory");   // half A's trailing segment
  if (idle) return;
  const float lt = rows_sum(ls);
  const float inv = 1.0f / lt;
  store_row_blocks<DB>(out + (size_t)(seq0 + min(qrow, len - 1)) * D + h * DH, qrow < len, o, inv, g);
  if (g == 0 && qrow < len) lse[(size_t)h * T + seq0 + qrow] = (m + log2f(lt)) * LN2;
}

// =====================================================================================
// backward: delta[h][t] = sum_d dO[t,h,d] * O[t,h,d]
// =====================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int T, int D, int H) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int DH = D / H;
  for (int row = blockIdx.x * 4 + w; row < T; row += gridDim.x * 4) {
    for (int c0 = 0; c0 < D; c0 += 256) {
      const int c = c0 + 4 * l;
      float part = 0.f;
      int hh = -1;
      if (c < D) {
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(o + (size_t)row * D + c);
        const bf16x4 d = *reinterpret_cast<const bf16x4*>(dout + (size_t)row * D + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) part += (float)a[k] * (float)d[k];
        hh = c / DH;
      }
      const int h_lo = c0 / DH, h_hi = min(H - 1, (min(D, c0 + 256) - 1) / DH);
      for (int hq = h_lo; hq <= h_hi; ++hq) {
        const float sum = wave_sum(hh == hq ? part : 0.f);
        if (l == 0) {
          // a head may straddle two 256-column sweeps (DH > 256): accumulate
          float* dst = delta + (size_t)hq * T + row;
          const bool first = (hq * DH >= c0);
          *dst = (first ? 0.f : *dst) + sum;
        }
      }
    }
  }
}

// =====================================================================================
// backward dQ: block = (128-query tile, head); sweep over KV tiles of 64
// =====================================================================================
// FUSE_DELTA: the block computes delta = rowsum(dO * O) of its own query rows from the dO fragments it holds anyway (plus one
// read of the O rows) and WRITES it for the dK/dV kernel that follows on the stream -- the separate delta pass (a full
// read of O and dO, 68 us at 300k tokens) disappears.
template <int DH, int CB, bool FUSE_DELTA>
__global__ __launch_bounds__(256, (DH <= 192 ? 2 : 1)) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                          const int* __restrict__ work, int T, int D, int H, float scale,
                                                          const bf16_t* __restrict__ out) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int LDK = DH + 16;  // K read both row-wise (b128) and transposed -> transpose-friendly stride
  constexpr int LDV = DH + 16;  // (row stride = 32 B x odd: conflict-free b128 row reads, see dkv_swz; + 8 was 2-way)
  constexpr int KVT = (DH > 96) ? 32 : 64, K2 = KVT / 32;  // keys staged per step (32 above dh = 96: half the LDS and registers)
  __shared__ __attribute__((aligned(16))) bf16_t smem[KVT * (LDK + LDV)];
  bf16_t* sK = smem;
  bf16_t* sV = smem + KVT * LDK;

  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  constexpr int SPLIT = 2 / CB;
  const WorkItem it = decode_work<SPLIT>(work, H);
  const int b = it.b, qt = it.t, h = it.h, part = it.part;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE + part * 64 * CB >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* kbase = qbase + D;
  const bf16_t* vbase = qbase + 2 * D;
  const float c = scale * LOG2E;

  bf16x8 qf[CB][KS], dof[CB][KS];
  float L2[CB], dl[CB];
  int qrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qrow[cb] = qt * TILE + part * 64 * CB + w * 16 * CB + cb * 16 + li;
    const int qr = min(qrow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[cb][ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 32 + g * 8);
      dof[cb][ks] = *reinterpret_cast<const bf16x8*>(dout + (size_t)(seq0 + qr) * D + h * DH + ks * 32 + g * 8);
    }
    L2[cb] = lse[(size_t)h * T + seq0 + qr] * LOG2E;
    if constexpr (FUSE_DELTA) {
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(out + (size_t)(seq0 + qr) * D + h * DH + ks * 32 + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)dof[cb][ks][e], (float)of[e], part);
      }
      dl[cb] = rows_sum(part);
      if (g == 0 && qrow[cb] < len) delta[(size_t)h * T + seq0 + qrow[cb]] = dl[cb];
    } else {
      dl[cb] = delta[(size_t)h * T + seq0 + qr];
    }
  }
  f32x4 dq[CB][DB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stager<DH, LDK, KVT> stK;
  Stager<DH, LDV, KVT> stV;
  const int nkt = (len + KVT - 1) / KVT;
  stK.load(kbase, ld, 0, len, tid);
  stV.load(vbase, ld, 0, len, tid);
  auto tile = [&](int kt, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    stK.store(sK, tid);
    stV.store(sV, tid);
    __syncthreads();
    if (!MASKED) {
      stK.load(kbase, ld, (kt + 1) * KVT, len, tid);
      stV.load(vbase, ld, (kt + 1) * KVT, len, tid);
    }
    if constexpr (DH <= 96) {  // (above, the look-ahead fragments do not fit 256 VGPRs: dh = 192 spills 140 registers)
      // two halves of 32 keys (k2): S, dP of the half -> dS -> the half's contribution to dQ; only half of the score registers
      // are live at a time.  The LDS reads run one group ahead of the MFMAs that consume them (pinned with sched_barrier):
      // left alone hipcc issues each group right in front of its MFMAs and waits for it.
      auto row_read = [&](int step, bf16x8& kfr, bf16x8& vfr) {  // step = (k2 * 2 + k1) * KS + ks
        const int kb = step / KS, ks = step % KS;
        kfr = lds_read8(sK + (kb * 16 + li) * LDK + ks * 32 + g * 8);
        vfr = lds_read8(sV + (kb * 16 + li) * LDV + ks * 32 + g * 8);
      };
      bf16x8 kfr[2], vfr[2], ktf[3];
      row_read(0, kfr[0], vfr[0]);
  #pragma unroll
      for (int k2 = 0; k2 < K2; ++k2) {
        f32x4 s[CB][2], dp[CB][2];
  #pragma unroll
        for (int st = 0; st < 2 * KS; ++st) {
          const int k1 = st / KS, ks = st % KS, cur = st & 1;
          if (st + 1 < 2 * KS) {
            row_read(k2 * 2 * KS + st + 1, kfr[cur ^ 1], vfr[cur ^ 1]);
          } else {  // first transposed fragments: land under the dS arithmetic
            ktf[0] = lds_read_tr8(sK + (k2 * 32) * LDK, LDK);
            ktf[1] = lds_read_tr8(sK + (k2 * 32) * LDK + 16, LDK);
          }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            s[cb][k1] = (ks == 0) ? mfma16(kfr[cur], qf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(kfr[cur], qf[cb][ks], s[cb][k1]);
            dp[cb][k1] = (ks == 0) ? mfma16(vfr[cur], dof[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(vfr[cur], dof[cb][ks], dp[cb][k1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        bf16x8 dsf[CB];
  #pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
  #pragma unroll
          for (int k1 = 0; k1 < 2; ++k1)
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = __builtin_amdgcn_exp2f(fmaf(s[cb][k1][r], c, -L2[cb]));
              if (MASKED && (kt * KVT + (2 * k2 + k1) * 16 + 4 * g + r >= len)) p = 0.f;
              s[cb][k1][r] = p * (dp[cb][k1][r] - dl[cb]);  // dS (unscaled)
            }
          dsf[cb] = pack8(s[cb][0], s[cb][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int db = 0; db < DB; ++db) {
          if (db + 2 < DB) {
            ktf[(db + 2) % 3] = lds_read_tr8(sK + (k2 * 32) * LDK + (db + 2) * 16, LDK);
          } else if (db + 2 == DB && k2 + 1 < K2) {
            row_read((k2 + 1) * 2 * KS, kfr[0], vfr[0]);  // first row pair of the next half
          }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int cb = 0; cb < CB; ++cb) dq[cb][db] = mfma16(ktf[db % 3], dsf[cb], dq[cb][db]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // two halves of 32 keys (k2): S, dP of the half -> dS -> the half's contribution to dQ; only half of the score registers
      // are live at a time
  #pragma unroll
      for (int k2 = 0; k2 < K2; ++k2) {
        f32x4 s[CB][2], dp[CB][2];
  #pragma unroll
        for (int k1 = 0; k1 < 2; ++k1) {
          const int kb = 2 * k2 + k1;
  #pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = lds_read8(sK + (kb * 16 + li) * LDK + ks * 32 + g * 8);
            const bf16x8 vf = lds_read8(sV + (kb * 16 + li) * LDV + ks * 32 + g * 8);
  #pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
              s[cb][k1] = (ks == 0) ? mfma16(kf, qf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(kf, qf[cb][ks], s[cb][k1]);
              dp[cb][k1] = (ks == 0) ? mfma16(vf, dof[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(vf, dof[cb][ks], dp[cb][k1]);
            }
          }
        }
        bf16x8 dsf[CB];
  #pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
  #pragma unroll
          for (int k1 = 0; k1 < 2; ++k1)
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = __builtin_amdgcn_exp2f(fmaf(s[cb][k1][r], c, -L2[cb]));
              if (MASKED && (kt * KVT + (2 * k2 + k1) * 16 + 4 * g + r >= len)) p = 0.f;
              s[cb][k1][r] = p * (dp[cb][k1][r] - dl[cb]);  // dS (unscaled)
            }
          dsf[cb] = pack8(s[cb][0], s[cb][1]);
        }
  #pragma unroll
        for (int db = 0; db < DB; ++db) {
          const bf16x8 ktf = lds_read_tr8(sK + (k2 * 32) * LDK + db * 16, LDK);
  #pragma unroll
          for (int cb = 0; cb < CB; ++cb) dq[cb][db] = mfma16(ktf, dsf[cb], dq[cb][db]);
        }
      }
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nkt - 1; ++kt) tile(kt, std::false_type{});
  tile(nkt - 1, std::true_type{});
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    store_row_blocks<DB>(dqkv + (size_t)(seq0 + min(qrow[cb], len - 1)) * ld + h * DH, qrow[cb] < len, dq[cb], scale, g);
  }
}

// =====================================================================================
// backward dK, dV: block = (64*CBK keys of a 128-key tile, head); sweep over query tiles of 64.
// CBK = 16-key column blocks per wave.  Every Q / dO fragment read from LDS (row-wise for S and dP, transposed for dK and dV)
// feeds CBK MFMAs: with one block per wave the kernel moved 48 KB of LDS per wave and tile for 48 MFMAs -- LDS-bound 2:1;
// CBK = 2 halves that and halves the number of blocks streaming the sequence's Q / dO.  It fits 256 VGPRs at dh = 96 only
// because a tile is processed in two 32-query halves (S, dP -> P, dS -> dV, dK per half): 872 -> 764 us for the backward.
// =====================================================================================
template <int DH, int CBK>
__global__ __launch_bounds__(256, (DH <= 192 ? 2 : 1)) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                           const int* __restrict__ work, int T, int D, int H, float scale) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int LDQ = DH + 16;  // Q and dO tiles are read row-wise (S, dP) and transposed (dK, dV)
  constexpr int KVT = (DH > 96) ? 32 : 64, K2 = KVT / 32;  // query rows staged per step
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * KVT * LDQ];
  __shared__ __attribute__((aligned(16))) float sL[KVT], sD[KVT];
  bf16_t* sQ = smem;
  bf16_t* sO = smem + KVT * LDQ;

  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  constexpr int SPLIT = 2 / CBK;
  const WorkItem it = decode_work<SPLIT>(work, H);
  const int b = it.b, kt = it.t, h = it.h, part = it.part;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (kt * TILE + part * KV * CBK >= len) return;  // this part of the key tile is beyond the sequence (uniform per block)
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* kbase = qbase + D;
  const bf16_t* vbase = qbase + 2 * D;
  const bf16_t* dobase = dout + (size_t)seq0 * D + h * DH;
  const float c = scale * LOG2E;

  int krow[CBK];
  bf16x8 kf[CBK][KS], vf[CBK][KS];
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb) {
    krow[cb] = kt * TILE + part * KV * CBK + w * 16 * CBK + cb * 16 + li;
    const int kr = min(krow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[cb][ks] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)kr * ld + ks * 32 + g * 8);
      vf[cb][ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)kr * ld + ks * 32 + g * 8);
    }
  }
  f32x4 dk[CBK][DB], dv[CBK][DB];
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      dk[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  Stager<DH, LDQ, KVT> stQ, stO;
  const int nqt = (len + KVT - 1) / KVT;
  stQ.load(qbase, ld, 0, len, tid);
  stO.load(dobase, (size_t)D, 0, len, tid);
  auto tile = [&](int q0, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    stQ.store(sQ, tid);
    stO.store(sO, tid);
    if (tid < KVT) {
      const int qr = min(q0 * KVT + tid, len - 1);
      sL[tid] = lse[(size_t)h * T + seq0 + qr] * LOG2E;
      sD[tid] = delta[(size_t)h * T + seq0 + qr];
    }
    __syncthreads();
    if (!MASKED) {
      stQ.load(qbase, ld, (q0 + 1) * KVT, len, tid);
      stO.load(dobase, (size_t)D, (q0 + 1) * KVT, len, tid);
    }
    // Two halves of 32 queries (k2): S, dP for the half, then P / dS, then the half's contribution to dV, dK -- only half of
    // the score registers are live at a time.  S[q][key], dP[q][key]: lane column = key, rows q = qb*16 + 4g + r.
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) {
      f32x4 s[CBK][2], dp[CBK][2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int qb = 2 * k2 + q2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 qfr = lds_read8(sQ + (qb * 16 + li) * LDQ + ks * 32 + g * 8);
          const bf16x8 dofr = lds_read8(sO + (qb * 16 + li) * LDQ + ks * 32 + g * 8);
#pragma unroll
          for (int cb = 0; cb < CBK; ++cb) {
            s[cb][q2] = (ks == 0) ? mfma16(qfr, kf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(qfr, kf[cb][ks], s[cb][q2]);
            dp[cb][q2] = (ks == 0) ? mfma16(dofr, vf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(dofr, vf[cb][ks], dp[cb][q2]);
          }
        }
      }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int qb = 2 * k2 + q2;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + qb * 16 + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + qb * 16 + 4 * g);
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(s[cb][q2][r], c, -l4[r]));
            if (MASKED && (q0 * KVT + qb * 16 + 4 * g + r >= len)) p = 0.f;
            s[cb][q2][r] = p;
            dp[cb][q2][r] = p * (dp[cb][q2][r] - d4[r]);
          }
      }
      bf16x8 pf[CBK], dsf[CBK];
#pragma unroll
      for (int cb = 0; cb < CBK; ++cb) {
        pf[cb] = pack8(s[cb][0], s[cb][1]);
        dsf[cb] = pack8(dp[cb][0], dp[cb][1]);
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x8 dot = lds_read_tr8(sO + (k2 * 32) * LDQ + db * 16, LDQ);
        const bf16x8 qtf = lds_read_tr8(sQ + (k2 * 32) * LDQ + db * 16, LDQ);
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          dv[cb][db] = mfma16(dot, pf[cb], dv[cb][db]);
          dk[cb][db] = mfma16(qtf, dsf[cb], dk[cb][db]);
        }
      }
    }
    __syncthreads();
  };
  for (int q0 = 0; q0 < nqt - 1; ++q0) tile(q0, std::false_type{});
  tile(nqt - 1, std::true_type{});
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb) {
    bf16_t* drow = dqkv + (size_t)(seq0 + min(krow[cb], len - 1)) * ld + h * DH;
    store_row_blocks<DB>(drow + D, krow[cb] < len, dk[cb], scale, g);
    store_row_blocks<DB>(drow + 2 * D, krow[cb] < len, dv[cb], 1.0f, g);
  }
}


// =====================================================================================
// backward dK, dV, LDS-DMA variant (dh = 96 and 192): the Q / dO tiles (and the tile's lse / delta) go global -> LDS by LDS-DMA
// into two stages -- no staging registers, no ds_write pass, ONE barrier per query tile, and the load of tile t+1 flies under
// the whole math of tile t (restrict-scoped tile function: no vmcnt(0) drain, see attn_fwd_tile).  Same arithmetic as
// attn_bwd_dkv_kernel, bit-identical results.
// A tile is read both row-wise (b128: S and dP operands) and transposed (b64_tr: dK and dV operands), so it stays ROW-MAJOR
// and unpadded (the DMA writes lane-linear); the bank spread comes from the source side: LDS row r, 16-byte chunk c' holds
// source chunk c' ^ dkv_swz(r), and the readers apply the same XOR.
// =====================================================================================
// Which lanes a ds_read_b128 serves together is NOT 16 consecutive ones: measured (scratch/ldsbank) the groups are
// {0-3, 12-15, 20-23, 24-27} and {4-7, 8-11, 16-19, 28-31} (+32) -- half of a group reads chunk c, the other half chunk c+1 of
// its rows.  The XOR terms below are conflict-free for THAT grouping and for the 32-lane phases of the transpose read (brute
// force over all (k-step, block) positions); a first version derived for 16 consecutive lanes ran with 40 % conflict cycles.
//   dh =  96 (192-byte rows, 12 chunks): low two chunk bits ^= (4 - (row >> 2)) & 3
//   dh = 192 (384-byte rows, 24 chunks): chunk bits 1..2 ^= row bits 1..2
//   dh = 384 (768-byte rows, 48 chunks: every row starts on bank 0): low four chunk bits ^= 2 (row & 7) ^ 9 (row >> 3 & 1) -- one of
//            the 5376 GF(2)-linear maps of the row's low four bits that are conflict-free for both read kinds (brute force)
template <int DH>
__device__ __forceinline__ int dkv_swz(int row) {
  if constexpr (DH == 384) return (((row & 7) << 1) ^ (((row >> 3) & 1) * 9));
  else if constexpr (DH == 192) return row & 6;
  else return (4 - ((row >> 2) & 3)) & 3;
}

#ifndef CHADA_BWD_ABL
#define CHADA_BWD_ABL 0   // timing-only ablations of the LDS-DMA backward kernels (wrong results): 1 = no refills, 2 = no fragment reads, 8 = no exp / dS arithmetic
#endif
template <int DH, int CBK, bool MASKED>
__device__ __forceinline__ void attn_dkv_tile(BufRsrc qg, BufRsrc dog, BufRsrc lg, BufRsrc dg,
                                              bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd, bool issue, bool idle, int q0, int len,
                                              unsigned ldq, unsigned ldo, float c, int w, int l,
                                              const int (&rec_row)[(DH > 96 ? 32 : 64) * (DH / 8) / 256],
                                              const int (&rec_col)[(DH > 96 ? 32 : 64) * (DH / 8) / 256],
                                              const bf16x8 (&kf)[CBK][DH / 32], const bf16x8 (&vf)[CBK][DH / 32],
                                              f32x4 (&dk)[CBK][DH / 16], f32x4 (&dv)[CBK][DH / 16]) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int KVT = (DH > 96) ? 32 : 64, K2 = KVT / 32;
  constexpr int NRW = KVT * (DH / 8) / 256;  // 1 KiB records per wave and tensor
  constexpr int TILE_E = KVT * DH;           // bf16 elements of one tensor's tile
  const int g = l >> 4, li = l & 15;
  // dh 384 (one block per CU, 13 LDS-DMA instructions per wave and tile): the next tile's pieces are dealt out over the first steps of the tile
  // instead of going out as one burst behind the barrier -- 1 865 -> 1 801 us on cfg5's pass; at dh 96 / 192 (7 instructions, two blocks per CU) the
  // same change is neutral (1 203 vs 1 205 us, 1 278 vs 1 278): burst kept there (profiles/r05k_ffn_spread_dma.log)
  constexpr bool SPREAD = DH > 192;
  const int r0 = (q0 + 1) * KVT;
  auto issue_piece = [&](int i) {   // i < NRW: the Q and dO records; NRW: lse / delta (wave 0)
    if (i < NRW) {
      const unsigned row = (unsigned)min(r0 + rec_row[i], len - 1);
      lds_dma16(qg, dst + (w + 4 * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
      lds_dma16(dog, dst + TILE_E + (w + 4 * i) * 512, (row * ldo + rec_col[i]) * 2, 0);
    } else if (w == 0 && l < KVT) {  // lse / delta of the tile: 4 bytes per lane
      const int qr = min(r0 + l, len - 1);
      lds_dma4(lg, dst + 2 * TILE_E, qr * 4, 0);
      lds_dma4(dg, dst + 2 * TILE_E + 2 * KVT, qr * 4, 0);
    }
  };
  if (issue && (!SPREAD || idle) && (CHADA_BWD_ABL & 1) == 0) {
#pragma unroll
    for (int i = 0; i <= NRW; ++i) issue_piece(i);
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST: free of the alias edge, the scheduler would sink it below the math
  if (idle) return;  // (wave-uniform) none of this wave's keys exists: it only feeds the DMA and the barriers
  const bf16_t* sQ = rd;
  const bf16_t* sO = rd + TILE_E;
  const float* sL = reinterpret_cast<const float*>(rd + 2 * TILE_E);
  const float* sD = sL + KVT;
  // Two halves of 32 queries (k2): S, dP for the half, then P / dS, then the half's contribution to dV, dK -- only half of
  // the score registers are live at a time.  S[q][key], dP[q][key]: lane column = key, rows q = qb*16 + 4g + r.
  // The LDS reads are software-pipelined by hand, one group ahead of the MFMAs that consume them (the fragments of step
  // i+1 are requested BEFORE the MFMAs of step i, pinned with sched_barrier): left alone, hipcc places every group of reads
  // right in front of its MFMAs behind an lgkmcnt(0) -- 25 exposed LDS round trips per tile.
  auto row_read = [&](int step, bf16x8& qfr, bf16x8& dofr) {  // step = (k2 * 2 + q2) * KS + ks
    const int row = (step / KS) * 16 + li, ks = step % KS;
    const int ch = (ks * 4 + g) ^ dkv_swz<DH>(row);  // chunk ks*4 + g of the row, swizzled on its low two bits
    if constexpr ((CHADA_BWD_ABL & 2) != 0) { qfr = kf[0][ks]; dofr = vf[0][ks]; return; }
    qfr = lds_read8(sQ + row * DH + ch * 8);
    dofr = lds_read8(sO + row * DH + ch * 8);
  };
  // transposed operands: this lane supplies row k2*32 + 4g + (li >> 2) (and + 16), 8 bytes at (li & 3) * 8 inside the
  // 32-byte pair of the 16-column block -- chunk 2 db + ((li & 3) >> 1), swizzled like the row reads
  auto tr_read = [&](int k2, int db, bf16x8& dot, bf16x8& qtf) {
    const int trow = k2 * 32 + 4 * g + (li >> 2);
    const int ch = (2 * db + ((li & 3) >> 1)) ^ dkv_swz<DH>(trow);  // the swizzle term is identical for trow + 16
    const int off = trow * DH + ch * 8 + (li & 1) * 4;
    if constexpr ((CHADA_BWD_ABL & 2) != 0) { dot = kf[0][db % KS]; qtf = vf[0][db % KS]; return; }
    dot = __builtin_shufflevector(lds_read_tr4(sO + off), lds_read_tr4(sO + off + 16 * DH), 0, 1, 2, 3, 4, 5, 6, 7);
    qtf = __builtin_shufflevector(lds_read_tr4(sQ + off), lds_read_tr4(sQ + off + 16 * DH), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // last query tile of the sequence: only the 16-row blocks that hold a valid query are multiplied -- wave-uniform
  // branches, identical results (the skipped blocks' P and dS are 0)
  const int nvq = MASKED ? min(KVT / 16, (len - q0 * KVT + 15) >> 4) : KVT / 16;
  // fragments are requested RD steps ahead of their MFMAs through rings of RD + 1 (RD = 1: the schedule of rounds 2-4, kept at dh 96 / 192).  dh 384
  // runs one wave per SIMD -- nobody else covers an LDS round trip -- and has the registers: four steps ahead, 1 802 -> 1 755 us on cfg5's global pass,
  // bit-identical (2 and 3: 1 778 / 1 773)
#ifndef CHADA_DKV_RD384
#define CHADA_DKV_RD384 4
#endif
#ifndef CHADA_DKV_RD
#define CHADA_DKV_RD 1
#endif
  constexpr int RD = (DH > 192) ? CHADA_DKV_RD384 : CHADA_DKV_RD, R = RD + 1;
  static_assert(RD >= 1 && RD <= DB && RD <= 2 * KS, "look-ahead within a phase");
  bf16x8 qfr[R], dofr[R], dot[R], qtf[R];
#pragma unroll
  for (int i = 0; i < RD; ++i) row_read(i, qfr[i], dofr[i]);
#ifndef CHADA_DKV_PIPE
#define CHADA_DKV_PIPE 0   // (built, bit-identical, 12 spilled registers, 1 360 against 1 203 us -- off; profiles/r06a_*)
#endif
  if constexpr (K2 == 2 && CBK == 2 && RD == 1 && CHADA_DKV_PIPE != 0 && 2 * KS >= 4 && DB >= 4) {
    // dh 96 (round 6): the tile's two 32-query halves software-pipelined inside the wave, as in attn_dq_tile:
    //     A0 | A1 + B0 | C0 + B1 | C1        (A = S, dP MFMAs; B = P = exp, dS, pack; C = dV += P^T dO, dK += dS^T Q MFMAs)
    // Same arithmetic, same accumulation order (bit-identical dK / dV); the exponentials, compiled out, are 11 % of this kernel, its fragment reads 22 %.
    f32x4 s[2][CBK][2], dp[2][CBK][2];
    bf16x8 pf[2][CBK], dsf[2][CBK];
    auto mma_a = [&](int k2, int st, int cur) {
      const int q2 = st / KS, ks = st % KS;
      if (!MASKED || 2 * k2 + q2 < nvq) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          s[k2][cb][q2] = (ks == 0) ? mfma16(qfr[cur], kf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(qfr[cur], kf[cb][ks], s[k2][cb][q2]);
          dp[k2][cb][q2] = (ks == 0) ? mfma16(dofr[cur], vf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(dofr[cur], vf[cb][ks], dp[k2][cb][q2]);
        }
      }
    };
    auto valu_slice = [&](int k2, int step) {   // steps 0 .. 3: (q2, cb) = (step >> 1, step & 1); the step after packs
      if (step < 2 * CBK) {
        const int q2 = step >> 1, cb = step & 1, qb = 2 * k2 + q2;
        if (MASKED && qb >= nvq) {
          s[k2][cb][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[k2][cb][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if constexpr ((CHADA_BWD_ABL & 8) == 0) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + qb * 16 + 4 * g) * LOG2E;
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + qb * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(s[k2][cb][q2][r], c, -l4[r]));
            if (MASKED && (q0 * KVT + qb * 16 + 4 * g + r >= len)) p = 0.f;
            s[k2][cb][q2][r] = p;
            dp[k2][cb][q2][r] = p * (dp[k2][cb][q2][r] - d4[r]);
          }
        }
      } else if (step == 2 * CBK) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          pf[k2][cb] = pack8(s[k2][cb][0], s[k2][cb][1]);
          dsf[k2][cb] = pack8(dp[k2][cb][0], dp[k2][cb][1]);
        }
      }
    };
    // ---- A0
#pragma unroll
    for (int st = 0; st < 2 * KS; ++st) {
      const int cur = st & 1;
      row_read(st + 1, qfr[cur ^ 1], dofr[cur ^ 1]);   // (st + 1 == 2 KS: the first row pair of half 1)
      __builtin_amdgcn_sched_barrier(0);
      mma_a(0, st, cur);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- A1 beside B0
#pragma unroll
    for (int st = 0; st < 2 * KS; ++st) {
      const int cur = st & 1;
      if (st + 1 < 2 * KS) row_read(2 * KS + st + 1, qfr[cur ^ 1], dofr[cur ^ 1]);
      else tr_read(0, 0, dot[0], qtf[0]);
      __builtin_amdgcn_sched_barrier(0);
      mma_a(1, st, cur);
      valu_slice(0, st);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int st = 2 * KS; st <= 2 * CBK; ++st) valu_slice(0, st);
    __builtin_amdgcn_sched_barrier(0);
    // ---- C0 beside B1, then C1: one ring of two transposed pairs across both halves
#pragma unroll
    for (int dbg = 0; dbg < 2 * DB; ++dbg) {
      const int k2 = dbg / DB, db = dbg % DB, cur = dbg & 1;
      if (dbg + 1 < 2 * DB) tr_read((dbg + 1) / DB, (dbg + 1) % DB, dot[cur ^ 1], qtf[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (!MASKED || 2 * k2 < nvq) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          dv[cb][db] = mfma16(dot[cur], pf[k2][cb], dv[cb][db]);
          dk[cb][db] = mfma16(qtf[cur], dsf[k2][cb], dk[cb][db]);
        }
      }
      if (k2 == 0) valu_slice(1, db);
      __builtin_amdgcn_sched_barrier(0);
      if (k2 == 0 && db == DB - 1) {
#pragma unroll
        for (int st = DB; st <= 2 * CBK; ++st) valu_slice(1, st);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
#pragma unroll
  for (int k2 = 0; k2 < K2; ++k2) {
    f32x4 s[CBK][2], dp[CBK][2];
#pragma unroll
    for (int st = 0; st < 2 * KS; ++st) {
      const int q2 = st / KS, ks = st % KS, cur = st % R, nxt = st + RD;
      if constexpr (SPREAD && (CHADA_BWD_ABL & 1) == 0) {   // one piece per step of the first half instead of a burst behind the barrier
        if (issue && k2 == 0 && st <= NRW) issue_piece(st);
      }
      if (nxt < 2 * KS) row_read(k2 * 2 * KS + nxt, qfr[nxt % R], dofr[nxt % R]);
      else tr_read(k2, nxt - 2 * KS, dot[(nxt - 2 * KS) % R], qtf[(nxt - 2 * KS) % R]);  // first transposed pairs: land under the softmax
      __builtin_amdgcn_sched_barrier(0);
      if (!MASKED || 2 * k2 + q2 < nvq) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          s[cb][q2] = (ks == 0) ? mfma16(qfr[cur], kf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(qfr[cur], kf[cb][ks], s[cb][q2]);
          dp[cb][q2] = (ks == 0) ? mfma16(dofr[cur], vf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(dofr[cur], vf[cb][ks], dp[cb][q2]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int qb = 2 * k2 + q2;
      if (MASKED && qb >= nvq) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          s[cb][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[cb][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        continue;
      }
      if constexpr ((CHADA_BWD_ABL & 8) != 0) continue;
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + qb * 16 + 4 * g) * LOG2E;
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + qb * 16 + 4 * g);
#pragma unroll
      for (int cb = 0; cb < CBK; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[cb][q2][r], c, -l4[r]));
          if (MASKED && (q0 * KVT + qb * 16 + 4 * g + r >= len)) p = 0.f;
          s[cb][q2][r] = p;
          dp[cb][q2][r] = p * (dp[cb][q2][r] - d4[r]);
        }
    }
    bf16x8 pf[CBK], dsf[CBK];
#pragma unroll
    for (int cb = 0; cb < CBK; ++cb) {
      pf[cb] = pack8(s[cb][0], s[cb][1]);
      dsf[cb] = pack8(dp[cb][0], dp[cb][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const int cur = db % R, nxt = db + RD;
      if (nxt < DB) tr_read(k2, nxt, dot[nxt % R], qtf[nxt % R]);
      else if (k2 + 1 < K2) row_read((k2 + 1) * 2 * KS + (nxt - DB), qfr[(nxt - DB) % R], dofr[(nxt - DB) % R]);  // first row pairs of the next half
      __builtin_amdgcn_sched_barrier(0);
      if (!MASKED || 2 * k2 < nvq) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          dv[cb][db] = mfma16(dot[cur], pf[cb], dv[cb][db]);
          dk[cb][db] = mfma16(qtf[cur], dsf[cb], dk[cb][db]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int DH, int CBK>
__global__ __launch_bounds__(256, (DH > 192 ? 1 : 2)) void attn_bwd_dkv_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                              const int* __restrict__ work, int T, int D, int H, float scale) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int KVT = (DH > 96) ? 32 : 64;
  constexpr int NRW = KVT * (DH / 8) / 256;
  constexpr int STAGE = 2 * KVT * DH + 4 * KVT;  // Q tile | dO tile | lse[KVT] | delta[KVT] (floats = 2 elements each)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, li = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SPLIT = 2 / CBK;
  const WorkItem it = decode_work<SPLIT>(work, H);
  const int b = it.b, kt = it.t, h = it.h, part = it.part;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (kt * TILE + part * KV * CBK >= len) return;  // this part of the key tile is beyond the sequence (uniform per block)
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* kbase = qbase + D;
  const bf16_t* vbase = qbase + 2 * D;
  const bf16_t* dobase = dout + (size_t)seq0 * D + h * DH;
  const float* lbase = lse + (size_t)h * T + seq0;
  const float* dbase = delta + (size_t)h * T + seq0;
  const float c = scale * LOG2E;

  int krow[CBK];
  bf16x8 kf[CBK][KS], vf[CBK][KS];
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb) {
    krow[cb] = kt * TILE + part * KV * CBK + w * 16 * CBK + cb * 16 + li;
    const int kr = min(krow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[cb][ks] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)kr * ld + ks * 32 + g * 8);
      vf[cb][ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)kr * ld + ks * 32 + g * 8);
    }
  }
  f32x4 dk[CBK][DB], dv[CBK][DB];
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      dk[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // record (w + 4 i) of a tile: 64 consecutive 16-byte chunks of the row-major image; per lane the row inside the tile and
  // the element offset of the SOURCE chunk (un-swizzled)
  int rec_row[NRW], rec_col[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int id = (w + 4 * i) * 64 + l, row = id / (DH / 8), ch = id % (DH / 8);
    rec_row[i] = row;
    rec_col[i] = (ch ^ dkv_swz<DH>(row)) * 8;
  }
  const unsigned ldq = 3u * (unsigned)D, ldo = (unsigned)D;
  const int nqt = (len + KVT - 1) / KVT;
  const BufRsrc qrs = make_rsrc(qbase), dors = make_rsrc(dobase), lrs = make_rsrc(lbase), drs = make_rsrc(dbase);
  const bool idle = kt * TILE + part * KV * CBK + w * 16 * CBK >= len;  // no valid key in this wave
  // tile 0 (no LDS read follows before the first barrier: issued bare)
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const unsigned row = (unsigned)min(rec_row[i], len - 1);
    lds_dma16(qrs, smem + (w + 4 * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
    lds_dma16(dors, smem + KVT * DH + (w + 4 * i) * 512, (row * ldo + rec_col[i]) * 2, 0);
  }
  if (w == 0 && l < KVT) {
    const int qr = min(l, len - 1);
    lds_dma4(lrs, smem + 2 * KVT * DH, qr * 4, 0);
    lds_dma4(drs, smem + 2 * KVT * DH + 2 * KVT, qr * 4, 0);
  }
  for (int q0 = 0; q0 < nqt - 1; ++q0) {
    // tile q0 has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt) and everybody is done
    // reading the other stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    attn_dkv_tile<DH, CBK, false>(qrs, dors, lrs, drs, smem + ((q0 + 1) & 1) * STAGE, smem + (q0 & 1) * STAGE, true, idle, q0, len,
                                  ldq, ldo, c, w, l, rec_row, rec_col, kf, vf, dk, dv);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  attn_dkv_tile<DH, CBK, true>(qrs, dors, lrs, drs, smem + (nqt & 1) * STAGE, smem + ((nqt - 1) & 1) * STAGE, false, idle, nqt - 1,
                               len, ldq, ldo, c, w, l, rec_row, rec_col, kf, vf, dk, dv);
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb) {
    bf16_t* drow = dqkv + (size_t)(seq0 + min(krow[cb], len - 1)) * ld + h * DH;
    store_row_blocks<DB>(drow + D, krow[cb] < len, dk[cb], scale, g);
    store_row_blocks<DB>(drow + 2 * D, krow[cb] < len, dv[cb], 1.0f, g);
  }
}

#if CHADA_AB_SWITCHES
// =====================================================================================
// dK / dV at dh = 96 on the PAIRED schedule (round 6; side builds only: CHADAVIT_ATTN_DKV_PAIR=1).  One 512-thread block per (image, 128-key tile): half A
// (waves 0-3) runs head 0, half B (waves 4-7) head 1 -- wave w and w + 4 share a SIMD --, both the arithmetic of attn_bwd_dkv_dma_kernel<96, 2> (same
// fragments, same order of accumulation: bit-identical dK / dV), cut into two segments per 32-query HALF-tile and held one segment apart by the block's barriers:
//   X_h (matrix):          dV += P(h-1)^T dO, dK += dS(h-1)^T Q   then   S(h) = Q K^T, dP(h) = dO V^T      -- 48 MFMAs and their fragment reads
//   Y_h (everything else): the LDS-DMA requests of a later query tile, then P(h) = exp2(..), dS(h) = P (dP - delta), packed to bf16
// Per half: a ring of THREE row-major swizzled stages (Q tile | dO tile | lse | delta of 64 queries; the transposed reads of X_h still need tile (h-1)/2
// while the row reads are already in tile h/2): query tile t + 2 is requested in Y_2t into the stage of tile t - 1, every wave waits for its own pieces at
// the end of each matrix segment.  Four barriers per 64-query tile instead of one.
// =====================================================================================
template <int DH>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkv_pair_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                                   const int* __restrict__ work, int T, int D, float scale) {
  constexpr int CBK = 2, KS = DH / 32, DB = DH / 16, KVT = 64, NRW = KVT * (DH / 8) / 256;
  constexpr int TILE_E = KVT * DH, STAGE = 2 * TILE_E + 4 * KVT, HALF = 3 * STAGE;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * HALF];
  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, li = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = w >> 2, wk = w & 3;
  const int b = work[2 * blockIdx.x], kt = work[2 * blockIdx.x + 1];
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (kt * TILE >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* kbase = qbase + D;
  const bf16_t* vbase = qbase + 2 * D;
  const bf16_t* dobase = dout + (size_t)seq0 * D + h * DH;
  const float c = scale * LOG2E;
  int krow[CBK];
  bf16x8 kf[CBK][KS], vf[CBK][KS];
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb) {
    krow[cb] = kt * TILE + wk * 16 * CBK + cb * 16 + li;
    const int kr = min(krow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[cb][ks] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)kr * ld + ks * 32 + g * 8);
      vf[cb][ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)kr * ld + ks * 32 + g * 8);
    }
  }
  f32x4 dk[CBK][DB], dv[CBK][DB];
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb)
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      dk[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const unsigned ldq = 3u * (unsigned)D, ldo = (unsigned)D;
  const int nqt = (len + KVT - 1) / KVT, NH = 2 * nqt;   // 32-query half-tiles
  const BufRsrc qrs = make_rsrc(qbase), dors = make_rsrc(dobase), lrs = make_rsrc(lse + (size_t)h * T + seq0), drs = make_rsrc(delta + (size_t)h * T + seq0);
  const bool idle = kt * TILE + wk * 16 * CBK >= len;
  const int nvq_last = min(KVT / 16, (len - (nqt - 1) * KVT + 15) >> 4);   // valid 16-query blocks of the last tile
  int opq = 0;
  asm volatile("" : "+s"(opq));
  bf16_t* const base = smem + h * HALF + opq;
  auto fetch = [&](int t, bf16_t* __restrict__ dst) {   // this wave's pieces of query tile t
    const int r0 = t * KVT;
    int lx = l;
    asm volatile("" : "+v"(lx));   // (the pieces' row / chunk are recomputed per tile instead of living in 14 registers across the loop: 244 of 256 are taken)
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      const int id = (wk + 4 * i) * 64 + lx, rrow = id / (DH / 8), ch = id % (DH / 8);
      const unsigned col = (unsigned)((ch ^ dkv_swz<DH>(rrow)) * 8);
      const unsigned row = (unsigned)min(r0 + rrow, len - 1);
      lds_dma16(qrs, dst + (wk + 4 * i) * 512, (row * ldq + col) * 2, 0);
      lds_dma16(dors, dst + TILE_E + (wk + 4 * i) * 512, (row * ldo + col) * 2, 0);
    }
    if (wk == 0 && l < KVT) {
      const int qr = min(r0 + l, len - 1);
      lds_dma4(lrs, dst + 2 * TILE_E, qr * 4, 0);
      lds_dma4(drs, dst + 2 * TILE_E + 2 * KVT, qr * 4, 0);
    }
  };
  f32x4 s[CBK][2], dp[CBK][2];
  bf16x8 pf[CBK], dsf[CBK];
  // ---- X: [C of half hc (transposed reads out of stage sC), A of half ha (row reads out of stage sA)]; nvq_* = that half's tile's valid query blocks
  auto seg_x = [&](const bf16_t* __restrict__ sC, int k2c, int nvq_c, const bf16_t* __restrict__ sA, int k2a, int nvq_a, auto do_c_tag, auto do_a_tag) {
    constexpr bool DO_C = decltype(do_c_tag)::value, DO_A = decltype(do_a_tag)::value;
    // (the last query tile's masks are run-time conditions here; five compile-time instantiations of the two segment bodies instead -- unconditional
    // MFMAs everywhere but in the last tile -- spill 213 registers, this form 51: the unpaired kernel's state alone is 244 of the 256 registers)
    if (idle) return;
    int li = l & 15, g = l >> 4;
    asm volatile("" : "+v"(li), "+v"(g));   // (keeps the segment's swizzled fragment addresses from being hoisted out of the half-tile loop: 64 spilled registers otherwise)
    if constexpr (DO_C) {
      const bf16_t* sQ = sC;
      const bf16_t* sO = sC + TILE_E;
      bf16x8 dot[2], qtf[2];
      auto tr_read = [&](int db, bf16x8& d_, bf16x8& q_) {
        const int trow = k2c * 32 + 4 * g + (li >> 2);
        const int ch = (2 * db + ((li & 3) >> 1)) ^ dkv_swz<DH>(trow);
        const int off = trow * DH + ch * 8 + (li & 1) * 4;
        d_ = __builtin_shufflevector(lds_read_tr4(sO + off), lds_read_tr4(sO + off + 16 * DH), 0, 1, 2, 3, 4, 5, 6, 7);
        q_ = __builtin_shufflevector(lds_read_tr4(sQ + off), lds_read_tr4(sQ + off + 16 * DH), 0, 1, 2, 3, 4, 5, 6, 7);
      };
      tr_read(0, dot[0], qtf[0]);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const int cur = db & 1;
        if (db + 1 < DB) tr_read(db + 1, dot[cur ^ 1], qtf[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * k2c < nvq_c) {
#pragma unroll
          for (int cb = 0; cb < CBK; ++cb) {
            dv[cb][db] = mfma16(dot[cur], pf[cb], dv[cb][db]);
            dk[cb][db] = mfma16(qtf[cur], dsf[cb], dk[cb][db]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (DO_A) {
      const bf16_t* sQ = sA;
      const bf16_t* sO = sA + TILE_E;
      bf16x8 qfr[2], dofr[2];
      auto row_read = [&](int st, bf16x8& q_, bf16x8& d_) {
        const int row = k2a * 32 + (st / KS) * 16 + li, ks = st % KS;
        const int ch = (ks * 4 + g) ^ dkv_swz<DH>(row);
        q_ = lds_read8(sQ + row * DH + ch * 8);
        d_ = lds_read8(sO + row * DH + ch * 8);
      };
      row_read(0, qfr[0], dofr[0]);
#pragma unroll
      for (int st = 0; st < 2 * KS; ++st) {
        const int q2 = st / KS, ks = st % KS, cur = st & 1;
        if (st + 1 < 2 * KS) row_read(st + 1, qfr[cur ^ 1], dofr[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * k2a + q2 < nvq_a) {
#pragma unroll
          for (int cb = 0; cb < CBK; ++cb) {
            s[cb][q2] = (ks == 0) ? mfma16(qfr[cur], kf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(qfr[cur], kf[cb][ks], s[cb][q2]);
            dp[cb][q2] = (ks == 0) ? mfma16(dofr[cur], vf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(dofr[cur], vf[cb][ks], dp[cb][q2]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- Y: P, dS of half-tile (q0, k2) out of its stage's lse / delta
  auto seg_y = [&](const bf16_t* __restrict__ st_, int q0, int k2, int nvq, bool masked) {
    if (idle) return;
    const float* sL = reinterpret_cast<const float*>(st_ + 2 * TILE_E);
    const float* sD = sL + KVT;
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int qb = 2 * k2 + q2;
      if (qb >= nvq) {
#pragma unroll
        for (int cb = 0; cb < CBK; ++cb) {
          s[cb][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[cb][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        continue;
      }
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + qb * 16 + 4 * g) * LOG2E;
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + qb * 16 + 4 * g);
#pragma unroll
      for (int cb = 0; cb < CBK; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[cb][q2][r], c, -l4[r]));
          if (masked && (q0 * KVT + qb * 16 + 4 * g + r >= len)) p = 0.f;
          s[cb][q2][r] = p;
          dp[cb][q2][r] = p * (dp[cb][q2][r] - d4[r]);
        }
    }
#pragma unroll
    for (int cb = 0; cb < CBK; ++cb) {
      pf[cb] = pack8(s[cb][0], s[cb][1]);
      dsf[cb] = pack8(dp[cb][0], dp[cb][1]);
    }
  };
  auto stage_of = [&](int t) { return base + (t % 3) * STAGE; };
  auto nvq_of = [&](int t) { return t == nqt - 1 ? nvq_last : KVT / 16; };
  // ---- prologue: query tiles 0 and 1
  fetch(0, stage_of(0));
  if (nqt > 1) fetch(1, stage_of(1));
  __builtin_amdgcn_s_waitcnt(0x0070);   // (the builtin: hipcc must know that the K / V fragment loads have landed -- attn_fwd_pair_kernel)
  asm volatile("s_barrier" ::: "memory");
  if (h == 1) asm volatile("s_barrier" ::: "memory");   // half B runs one segment behind
  seg_x(base, 0, 0, stage_of(0), 0, nvq_of(0), std::false_type{}, std::true_type{});   // X_0: A(0)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // (two half-tiles per trip, k2 a compile-time constant in every segment body)
  auto step = [&](int t, auto k2_tag) {
    constexpr int k2 = decltype(k2_tag)::value;
    const int hh = 2 * t + k2;
    // Y_hh: the refill first (query tile t + 2 over tile t - 1, whose last reader was X_2t), then the exponentials
    if (k2 == 0 && t + 2 < nqt) fetch(t + 2, stage_of(t + 2));
    __builtin_amdgcn_sched_barrier(0);
    seg_y(stage_of(t), t, k2, nvq_of(t), t == nqt - 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // X_{hh+1}: C(hh), then A(hh + 1)
    if (hh + 1 < NH) {
      const int tn = (hh + 1) >> 1;
      seg_x(stage_of(t), k2, nvq_of(t), stage_of(tn), 1 - k2, nvq_of(tn), std::true_type{}, std::true_type{});
    } else {
      seg_x(stage_of(t), k2, nvq_of(t), base, 0, 0, std::true_type{}, std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
#pragma clang loop unroll(disable)
  for (int t = 0; t < nqt; ++t) {
    step(t, std::integral_constant<int, 0>{});
    step(t, std::integral_constant<int, 1>{});
  }
  if (h == 0) asm volatile("s_barrier" ::: "memory");   // half A's trailing segment
#pragma unroll
  for (int cb = 0; cb < CBK; ++cb) {
    bf16_t* drow = dqkv + (size_t)(seq0 + min(krow[cb], len - 1)) * ld + h * DH;
    store_row_blocks<DB>(drow + D, krow[cb] < len, dk[cb], scale, g);
    store_row_blocks<DB>(drow + 2 * D, krow[cb] < len, dv[cb], 1.0f, g);
  }
}
#endif  // CHADA_AB_SWITCHES

// =====================================================================================
// backward dQ, LDS-DMA variant (dh = 96; dh = 192 with one query block per wave): the K / V tiles go global -> LDS by LDS-DMA into two row-major stages swizzled on
// the source side (dkv_swz: K is read both row-wise and transposed), one barrier per key tile, no staging registers; LDS
// reads one / two groups ahead of the MFMAs.  Same arithmetic as attn_bwd_dq_kernel, bit-identical results.
// =====================================================================================
template <int DH, int CB, bool MASKED, int NW = 4>
__device__ __forceinline__ void attn_dq_tile(BufRsrc kg, BufRsrc vg, bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd,
                                             bool issue, bool idle, int kt, int len, unsigned ldq, float c, int w, int l,
                                             const int (&rec_row)[(DH > 96 ? 32 : 64) * (DH / 8) / (64 * NW)],
                                             const int (&rec_col)[(DH > 96 ? 32 : 64) * (DH / 8) / (64 * NW)],
                                             const bf16x8 (&qf)[CB][DH / 32], const bf16x8 (&dof)[CB][DH / 32],
                                             const float (&L2)[CB], const float (&dl)[CB], f32x4 (&dq)[CB][DH / 16]) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int KVT = (DH > 96) ? 32 : 64, K2 = KVT / 32;
  constexpr int NRW = KVT * (DH / 8) / (64 * NW);
  constexpr int TILE_E = KVT * DH;
#ifndef CHADA_DQ_FOLD_DELTA
#define CHADA_DQ_FOLD_DELTA 1   // (round 6: dQ 977 -> 953 us at cfg2's global pass, 880 -> 874 at dh 192; one VALU op of ~4.5 per score less.  dh 384 keeps the
                                // subtraction: its dQ is held bit-identical to the fragment-major kernel it replaced)
#endif
  constexpr bool FOLD = CHADA_DQ_FOLD_DELTA != 0 && DH <= 192;
  const int g = l >> 4;
  int li = l & 15;
  if constexpr (NW == 8) asm volatile("" : "+v"(li));   // (dh 384, 192 registers of Q / dO / dQ state: keeps the tile's swizzled fragment addresses from being hoisted out of the tile loop)
  if (issue && (CHADA_BWD_ABL & 1) == 0) {
    const int r0 = (kt + 1) * KVT;
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      const unsigned row = (unsigned)min(r0 + rec_row[i], len - 1);
      lds_dma16(kg, dst + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
      lds_dma16(vg, dst + TILE_E + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST: free of the alias edge, the scheduler would sink it below the math
  if (idle) return;  // (wave-uniform) none of this wave's query rows exists: it only feeds the DMA and the barriers
  const bf16_t* sK = rd;
  const bf16_t* sV = rd + TILE_E;
  auto row_read = [&](int step, bf16x8& kfr, bf16x8& vfr) {  // step = (k2 * 2 + k1) * KS + ks
    const int row = (step / KS) * 16 + li, ks = step % KS;
    const int ch = (ks * 4 + g) ^ dkv_swz<DH>(row);
    if constexpr ((CHADA_BWD_ABL & 2) != 0) { kfr = qf[0][ks]; vfr = dof[0][ks]; return; }
    kfr = lds_read8(sK + row * DH + ch * 8);
    vfr = lds_read8(sV + row * DH + ch * 8);
  };
  auto tr_read = [&](int k2, int db) {
    const int trow = k2 * 32 + 4 * g + (li >> 2);
    const int ch = (2 * db + ((li & 3) >> 1)) ^ dkv_swz<DH>(trow);
    const int off = trow * DH + ch * 8 + (li & 1) * 4;
    if constexpr ((CHADA_BWD_ABL & 2) != 0) return qf[0][db % KS];
    return (bf16x8)__builtin_shufflevector(lds_read_tr4(sK + off), lds_read_tr4(sK + off + 16 * DH), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // (two / three groups of look-ahead instead of one / two measured the same: 739 vs 736 us for the whole backward)
  // last key tile of the sequence: only the 16-key blocks that hold a valid key are multiplied (len = 589: 13 keys = one
  // block of four) -- wave-uniform branches, identical results (the skipped blocks' dS is 0)
  const int nvb = MASKED ? min(KVT / 16, (len - kt * KVT + 15) >> 4) : KVT / 16;
  bf16x8 kfr[2], vfr[2], ktf[3];
  row_read(0, kfr[0], vfr[0]);
#ifndef CHADA_DQ_PIPE
#define CHADA_DQ_PIPE 0   // (built, bit-identical, measured: 988 against 989 us on cfg2's global pass -- off; profiles/r06a_*)
#endif
  if constexpr (K2 == 2 && CB == 2 && CHADA_DQ_PIPE != 0 && 2 * KS >= 4 && DB >= 4 && DB % 3 == 0) {
    // dh 96 (round 6, side builds with -DCHADA_DQ_PIPE=1 only): the two 32-key halves of the tile software-pipelined INSIDE the wave.  In the order
    // [S, dP of half 0][exp / dS of half 0][dQ += of half 0][S, dP of half 1] ... every phase waits for the one before it; compiled out, the
    // exponentials are 29 % of the kernel (990 -> 706 us).  Dealing them out behind the other half's MFMAs changes NOTHING (988 us): what they cost is
    // their issue slots, not a wait -- the kernel is the sum of its instruction classes' issue costs (profiles/r06a_*).  The schedule:
    //     A0 | A1 + B0 | C0 + B1 | C1        (A = S, dP MFMAs; B = exp, dS, pack; C = dQ += K^T dS MFMAs)
    // the VALU work of a half is dealt out, one (query block, 16-key block) pair per step, behind the MFMAs of the other half's neighbouring phase.
    // Same arithmetic and the same accumulation order as the unpipelined body (bit-identical dQ); both halves' scores live at once (+ 32 registers).
    f32x4 s[2][CB][2], dp[2][CB][2];
    bf16x8 dsf[2][CB];
    auto mma_a = [&](int k2, int st, int cur) {
      const int k1 = st / KS, ks = st % KS;
      if (!MASKED || 2 * k2 + k1 < nvb) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          s[k2][cb][k1] = (ks == 0) ? mfma16(kfr[cur], qf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(kfr[cur], qf[cb][ks], s[k2][cb][k1]);
          dp[k2][cb][k1] = (ks == 0) ? mfma16(vfr[cur], dof[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(vfr[cur], dof[cb][ks], dp[k2][cb][k1]);
        }
      }
    };
    auto valu_slice = [&](int k2, int step) {   // steps 0 .. 3: pair (cb, k1) = (step >> 1, step & 1); the step after the last pair packs
      if (step < 2 * CB) {
        const int cb = step >> 1, k1 = step & 1;
        if (MASKED && 2 * k2 + k1 >= nvb) {
          s[k2][cb][k1] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if constexpr ((CHADA_BWD_ABL & 8) == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(s[k2][cb][k1][r], c, -L2[cb]));
            if (MASKED && (kt * KVT + (2 * k2 + k1) * 16 + 4 * g + r >= len)) p = 0.f;
            s[k2][cb][k1][r] = p * (dp[k2][cb][k1][r] - dl[cb]);  // dS (unscaled)
          }
        }
      } else if (step == 2 * CB) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) dsf[k2][cb] = pack8(s[k2][cb][0], s[k2][cb][1]);
      }
    };
    // ---- A0
#pragma unroll
    for (int st = 0; st < 2 * KS; ++st) {
      const int cur = st & 1;
      row_read(st + 1, kfr[cur ^ 1], vfr[cur ^ 1]);   // (st + 1 == 2 KS: the first row pair of half 1)
      __builtin_amdgcn_sched_barrier(0);
      mma_a(0, st, cur);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- A1 beside B0
#pragma unroll
    for (int st = 0; st < 2 * KS; ++st) {
      const int cur = st & 1;
      if (st + 1 < 2 * KS) {
        row_read(2 * KS + st + 1, kfr[cur ^ 1], vfr[cur ^ 1]);
      } else {
        ktf[0] = tr_read(0, 0);
        ktf[1] = tr_read(0, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_a(1, st, cur);
      valu_slice(0, st);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int st = 2 * KS; st <= 2 * CB; ++st) valu_slice(0, st);   // (whatever of B0 the steps above did not reach)
    __builtin_amdgcn_sched_barrier(0);
    // ---- C0 beside B1, then C1; the transposed fragments run through one ring of three across both halves
#pragma unroll
    for (int dbg = 0; dbg < 2 * DB; ++dbg) {
      const int k2 = dbg / DB, db = dbg % DB;
      if (dbg + 2 < 2 * DB) ktf[(dbg + 2) % 3] = tr_read((dbg + 2) / DB, (dbg + 2) % DB);
      __builtin_amdgcn_sched_barrier(0);
      if (!MASKED || 2 * k2 < nvb) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) dq[cb][db] = mfma16(ktf[dbg % 3], dsf[k2][cb], dq[cb][db]);
      }
      if (k2 == 0) valu_slice(1, db);
      __builtin_amdgcn_sched_barrier(0);
      if (k2 == 0 && db == DB - 1) {
#pragma unroll
        for (int st = DB; st <= 2 * CB; ++st) valu_slice(1, st);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
#pragma unroll
  for (int k2 = 0; k2 < K2; ++k2) {
    f32x4 s[CB][2], dp[CB][2];
#pragma unroll
    for (int st = 0; st < 2 * KS; ++st) {
      const int k1 = st / KS, ks = st % KS, cur = st & 1;
      if (st + 1 < 2 * KS) {
        row_read(k2 * 2 * KS + st + 1, kfr[cur ^ 1], vfr[cur ^ 1]);
      } else {  // first transposed fragments: land under the dS arithmetic
        ktf[0] = tr_read(k2, 0);
        ktf[1] = tr_read(k2, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!MASKED || 2 * k2 + k1 < nvb) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          s[cb][k1] = (ks == 0) ? mfma16(kfr[cur], qf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(kfr[cur], qf[cb][ks], s[cb][k1]);
          // (CHADA_DQ_FOLD_DELTA: delta enters as the accumulator's initial value -- dP - delta comes out of the MFMAs, one VALU op per score less)
          dp[cb][k1] = (ks == 0) ? mfma16(vfr[cur], dof[cb][0], FOLD ? f32x4{-dl[cb], -dl[cb], -dl[cb], -dl[cb]} : f32x4{0.f, 0.f, 0.f, 0.f})
                                 : mfma16(vfr[cur], dof[cb][ks], dp[cb][k1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 dsf[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int k1 = 0; k1 < 2; ++k1) {
        if (MASKED && 2 * k2 + k1 >= nvb) {
          s[cb][k1] = f32x4{0.f, 0.f, 0.f, 0.f};
          continue;
        }
        if constexpr ((CHADA_BWD_ABL & 8) != 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(fmaf(s[cb][k1][r], c, -L2[cb]));
          if (MASKED && (kt * KVT + (2 * k2 + k1) * 16 + 4 * g + r >= len)) p = 0.f;
          s[cb][k1][r] = FOLD ? p * dp[cb][k1][r] : p * (dp[cb][k1][r] - dl[cb]);  // dS (unscaled)
        }
      }
      dsf[cb] = pack8(s[cb][0], s[cb][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      if (db + 2 < DB) {
        ktf[(db + 2) % 3] = tr_read(k2, db + 2);
      } else if (db + 2 == DB && k2 + 1 < K2) {
        row_read((k2 + 1) * 2 * KS, kfr[0], vfr[0]);  // first row pair of the next half
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!MASKED || 2 * k2 < nvb) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) dq[cb][db] = mfma16(ktf[db % 3], dsf[cb], dq[cb][db]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// One 16-row query block per wave (CB = 1: dh 192, where two spill) fits three blocks per CU: 175 registers at two waves per SIMD, 168 with 4
// spilled outside the loop at three -- and the third wave is worth 6 %: 1 017 -> 957 us on cfg3's global pass, 257 -> 241 us on its local
// one (same box, interleaved; profiles/r05g_dq192_three_blocks_per_cu.log).
template <int DH, int CB, bool FUSE_DELTA, int NW = 4>
__global__ __launch_bounds__(64 * NW, (CB == 1 && DH <= 192 ? 3 : (NW == 8 ? 1 : 2))) void attn_bwd_dq_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                          const int* __restrict__ work, int T, int D, int H, float scale,
                                                          const bf16_t* __restrict__ out) {
  constexpr int KS = DH / 32, DB = DH / 16;
  constexpr int KVT = (DH > 96) ? 32 : 64;
  constexpr int NRW = KVT * (DH / 8) / (64 * NW);  // 1 KiB records per wave and tensor
  constexpr int STAGE = 2 * KVT * DH;        // K tile | V tile, row-major, swizzled on the DMA source side (see dkv_swz)
  constexpr int QPB = NW * 16 * CB;          // query rows per block
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, li = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int SPLIT = TILE / QPB;
  const WorkItem it = decode_work<SPLIT>(work, H);
  const int b = it.b, qt = it.t, h = it.h, part = it.part;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE + part * QPB >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* kbase = qbase + D;
  const bf16_t* vbase = qbase + 2 * D;
  const float c = scale * LOG2E;

  bf16x8 qf[CB][KS], dof[CB][KS];
  float L2[CB], dl[CB];
  int qrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qrow[cb] = qt * TILE + part * QPB + w * 16 * CB + cb * 16 + li;
    const int qr = min(qrow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[cb][ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 32 + g * 8);
      dof[cb][ks] = *reinterpret_cast<const bf16x8*>(dout + (size_t)(seq0 + qr) * D + h * DH + ks * 32 + g * 8);
    }
    L2[cb] = lse[(size_t)h * T + seq0 + qr] * LOG2E;
    if constexpr (FUSE_DELTA) {
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(out + (size_t)(seq0 + qr) * D + h * DH + ks * 32 + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)dof[cb][ks][e], (float)of[e], part);
      }
      dl[cb] = rows_sum(part);
      if (g == 0 && qrow[cb] < len) delta[(size_t)h * T + seq0 + qrow[cb]] = dl[cb];
    } else {
      dl[cb] = delta[(size_t)h * T + seq0 + qr];
    }
  }
  f32x4 dq[CB][DB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  int rec_row[NRW], rec_col[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int id = (w + NW * i) * 64 + l, row = id / (DH / 8), ch = id % (DH / 8);
    rec_row[i] = row;
    rec_col[i] = (ch ^ dkv_swz<DH>(row)) * 8;
  }
  const unsigned ldq = 3u * (unsigned)D;
  const int nkt = (len + KVT - 1) / KVT;
  const BufRsrc krs = make_rsrc(kbase), vrs = make_rsrc(vbase);
  const bool idle = qt * TILE + part * QPB + w * 16 * CB >= len;  // no valid query row in this wave
  // tile 0 (no LDS read follows before the first barrier: issued bare)
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const unsigned row = (unsigned)min(rec_row[i], len - 1);
    lds_dma16(krs, smem + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
    lds_dma16(vrs, smem + KVT * DH + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
  }
  for (int kt = 0; kt < nkt - 1; ++kt) {
    // tile kt has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt) and everybody is done
    // reading the other stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    attn_dq_tile<DH, CB, false, NW>(krs, vrs, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE, true, idle, kt, len, ldq, c, w, l, rec_row,
                                rec_col, qf, dof, L2, dl, dq);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  attn_dq_tile<DH, CB, true, NW>(krs, vrs, smem + (nkt & 1) * STAGE, smem + ((nkt - 1) & 1) * STAGE, false, idle, nkt - 1, len, ldq, c, w, l,
                             rec_row, rec_col, qf, dof, L2, dl, dq);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    store_row_blocks<DB>(dqkv + (size_t)(seq0 + min(qrow[cb], len - 1)) * ld + h * DH, qrow[cb] < len, dq[cb], scale, g);
  }
}

// =====================================================================================
// backward dQ (+ delta) for dh = 384 (Base) on the forward's LDS-DMA structure: eight waves x 16 query rows share one K/V stage.
// dQ needs K twice -- row-wise as the A operand of S^T = K Q^T, and key-major as the A operand of dQ^T = K^T dS -- and V once
// (dP^T = V dO^T).  Instead of one row-major image that serves both reads through a swizzle (the dh <= 192 kernels: 768-byte rows
// would need a new one), the tile is fetched in the forward's two FRAGMENT-MAJOR record layouts, K in both:
//   K / V records (kb, ks): lane (li, g) = row kb*16 + li, d = ks*32 + 8g .. +7      (conflict-free ds_read_b128)
//   Kt records (db):        keys 0..31 x d = db*16 .. +15, 32-byte rows              (hardware transpose read)
// 72 KiB per 32-key stage, two stages (one block per CU, two waves per SIMD).  Same arithmetic as attn_bwd_dq_kernel.
// =====================================================================================
template <int DH, int CB, bool MASKED>
__device__ __forceinline__ void attn_dq_fm_tile(BufRsrc qb, bf16_t* __restrict__ dst, const bf16_t* __restrict__ sK, bool issue, bool idle, int kt,
                                                int len, unsigned ldu, float c, int w, int l, const int (&rec_row)[9], const unsigned (&rec_col)[9],
                                                const bf16x8 (&qf)[CB][DH / 32], const bf16x8 (&dof)[CB][DH / 32], const float (&L2)[CB],
                                                const float (&dl)[CB], f32x4 (&dq)[CB][DH / 16]) {
  constexpr int KS = DH / 32, DB = DH / 16, KVT = 32, NKR = 2 * KS, NW = DH / 48;  // NKR K records, NKR V records, DB (= NKR) Kt records
  static_assert(6 * KS == 9 * NW && DB == NKR, "nine LDS-DMA instructions per wave and tile");
  const int g = l >> 4;
  if (issue) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const unsigned off = (unsigned)min((kt + 1) * KVT + rec_row[i], len - 1) * ldu + rec_col[i];
      lds_dma16(qb, dst + (w + NW * i) * 512, off * 2, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST
  if (idle) return;
  const bf16_t* sV = sK + NKR * 512;
  const bf16_t* sT = sK + 2 * NKR * 512;
  const int nvb = MASKED ? min(2, (len - kt * KVT + 15) >> 4) : 2;
  f32x4 s[CB][2], dp[CB][2];
  bf16x8 kf[2], vf[2];
  kf[0] = lds_read8(sK + l * 8);
  vf[0] = lds_read8(sV + l * 8);
#pragma unroll
  for (int st = 0; st < NKR; ++st) {
    const int kb = st / KS, ks = st % KS, cur = st & 1;
    if (st + 1 < NKR) {
      kf[cur ^ 1] = lds_read8(sK + (st + 1) * 512 + l * 8);
      vf[cur ^ 1] = lds_read8(sV + (st + 1) * 512 + l * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!MASKED || kb < nvb) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        s[cb][kb] = (ks == 0) ? mfma16(kf[cur], qf[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(kf[cur], qf[cb][ks], s[cb][kb]);
        dp[cb][kb] = (ks == 0) ? mfma16(vf[cur], dof[cb][0], f32x4{0.f, 0.f, 0.f, 0.f}) : mfma16(vf[cur], dof[cb][ks], dp[cb][kb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16x8 tr[2];
  tr[0] = lds_read_tr8(sT, 16);
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 dsf[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (MASKED && kb >= nvb) {
        s[cb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(s[cb][kb][r], c, -L2[cb]));
        if (MASKED && (kt * KVT + kb * 16 + 4 * g + r >= len)) p = 0.f;
        s[cb][kb][r] = p * (dp[cb][kb][r] - dl[cb]);  // dS (unscaled)
      }
    }
    dsf[cb] = pack8(s[cb][0], s[cb][1]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    if (db + 1 < DB) tr[(db + 1) & 1] = lds_read_tr8(sT + (db + 1) * 512, 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) dq[cb][db] = mfma16(tr[db & 1], dsf[cb], dq[cb][db]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// CB query blocks of 16 rows per wave, NW = DH / 48 waves: 8 x 16 rows at dh = 384 (the instance in use).  The dh = 192 instance
// <192, 2> (4 waves x 32 rows, the same 192 registers of Q / dO / dQ state) was measured and is NOT dispatched: at two waves per SIMD
// it spills 77 registers (2089 us on cfg3's global pass), at one wave per SIMD (168 VGPRs + 122 AGPRs, no spill) it takes 1320 us --
// against 1180 us for the row-major <192, 1> kernel.
template <int DH, int CB>
__global__ __launch_bounds__(64 * (DH / 48), 1) void attn_bwd_dq_fm_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ lse, float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                                const int* __restrict__ work, int T, int D, int H, float scale,
                                                                const bf16_t* __restrict__ out) {
  constexpr int KS = DH / 32, DB = DH / 16, KVT = 32, NKR = 2 * KS, STAGE = 3 * NKR * 512, NW = DH / 48;
  static_assert(NW * 16 * CB == TILE, "one block per 128-row work item");
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];
  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, li = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WorkItem it = decode_work<1>(work, H);
  const int b = it.b, qt = it.t, h = it.h;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const float c = scale * LOG2E;
  bf16x8 qf[CB][KS], dof[CB][KS];
  float L2[CB], dl[CB];
  int qrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qrow[cb] = qt * TILE + w * 16 * CB + cb * 16 + li;
    const int qr = min(qrow[cb], len - 1);
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[cb][ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 32 + g * 8);
      dof[cb][ks] = *reinterpret_cast<const bf16x8*>(dout + (size_t)(seq0 + qr) * D + h * DH + ks * 32 + g * 8);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(out + (size_t)(seq0 + qr) * D + h * DH + ks * 32 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf((float)dof[cb][ks][e], (float)of[e], part);
    }
    L2[cb] = lse[(size_t)h * T + seq0 + qr] * LOG2E;
    dl[cb] = rows_sum(part);  // delta = rowsum(dO * O), produced here for the dK/dV kernel as well
    if (g == 0 && qrow[cb] < len) delta[(size_t)h * T + seq0 + qrow[cb]] = dl[cb];
  }
  f32x4 dq[CB][DB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[cb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

  // record r of a tile is fetched by wave r % NW (instruction r / NW): NKR K fragments, NKR V fragments, DB K key-major
  int rec_row[9];
  unsigned rec_col[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int r = w + NW * i;
    if (r < 2 * NKR) {
      const int rr = r % NKR;
      rec_row[i] = (rr / KS) * 16 + li;
      rec_col[i] = (r < NKR ? D : 2 * D) + (rr % KS) * 32 + g * 8;
    } else {
      const int rv = r - 2 * NKR;
      rec_row[i] = (l >> 1);
      rec_col[i] = D + rv * 16 + (l & 1) * 8;
    }
  }
  const unsigned ldu = 3u * (unsigned)D;
  const int nkt = (len + KVT - 1) / KVT;
  const bool idle = qt * TILE + w * 16 * CB >= len;
  const BufRsrc qrs = make_rsrc(qbase);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const unsigned off = (unsigned)min(rec_row[i], len - 1) * ldu + rec_col[i];
    lds_dma16(qrs, smem + (w + NW * i) * 512, off * 2, 0);
  }
  for (int kt = 0; kt < nkt - 1; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    attn_dq_fm_tile<DH, CB, false>(qrs, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE, true, idle, kt, len, ldu, c, w, l, rec_row, rec_col,
                                   qf, dof, L2, dl, dq);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  attn_dq_fm_tile<DH, CB, true>(qrs, smem + (nkt & 1) * STAGE, smem + ((nkt - 1) & 1) * STAGE, false, idle, nkt - 1, len, ldu, c, w, l, rec_row,
                                rec_col, qf, dof, L2, dl, dq);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    store_row_blocks<DB>(dqkv + (size_t)(seq0 + min(qrow[cb], len - 1)) * ld + h * DH, qrow[cb] < len, dq[cb], scale, g);
  }
}

// =====================================================================================
// attention-map export: the softmax probabilities themselves, P[b][h] = softmax(Q K^T / sqrt(dh)) (len_b x len_b, fp32).
// replaces the need_weights=True / average_attn_weights=False path of nn.MultiheadAttention used by
// get_last_selfattention (chada_vit.py:313-320, :105-110).  Not a training-path kernel: one wave per query row, plain FMAs.
// =====================================================================================
constexpr int PROB_MAXK = 32;  // keys per lane: sequences up to 2048 tokens (1 + 10 * 196 = 1961)
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ probs,
                                                         const int* __restrict__ cu, const long long* __restrict__ offs, int B,
                                                         int T, int D, int H, float scale) {
  __shared__ float sq[4][384];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + w, h = blockIdx.y;
  if (t >= T) return;
  const int dh = D / H;
  int lo = 0, hi = B;  // sequence b with cu[b] <= t < cu[b+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] <= t) lo = mid; else hi = mid;
  }
  const int seq0 = cu[lo], len = cu[lo + 1] - seq0;
  const size_t ld = 3 * (size_t)D;
  for (int d = l; d < dh; d += 64) sq[w][d] = (float)qkv[(size_t)t * ld + h * dh + d] * scale;
  __builtin_amdgcn_wave_barrier();
  float sc[PROB_MAXK];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PROB_MAXK; ++i) {
    const int j = l + 64 * i;
    float acc = -INFINITY;
    if (j < len) {
      const bf16_t* kr = qkv + (size_t)(seq0 + j) * ld + D + h * dh;
      acc = 0.f;
      for (int d = 0; d < dh; d += 8) {
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kr + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf((float)kv[e], sq[w][d + e], acc);
      }
    }
    sc[i] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PROB_MAXK; ++i) {
    sc[i] = (l + 64 * i < len) ? __expf(sc[i] - mx) : 0.f;
    sum += sc[i];
  }
  const float inv = 1.0f / wave_sum(sum);
  float* orow = probs + offs[lo] + ((size_t)h * len + (t - seq0)) * len;
#pragma unroll
  for (int i = 0; i < PROB_MAXK; ++i) {
    const int j = l + 64 * i;
    if (j < len) orow[j] = sc[i] * inv;
  }
}

}  // namespace

extern "C" int chadavit_attn_tile_rows(void) { return TILE; }
#ifdef CHADA_FWD_TIMELINE
extern "C" int chadavit_debug_read_fwd_timeline(void* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fwd_tl), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

#define ATTN_DISPATCH(DHV, CALL) \
  case DHV: { constexpr int DH_ = DHV; CALL; break; }

extern "C" int chadavit_attn_fwd(const chada_bf16* qkv_, chada_bf16* out_, float* lse, const int* cu_seqlens,
                                 const int* work, int n_work, int T, int D, int H, void* stream) {
  CHADA_ENTRY();
  if (!qkv_ || !out_ || !lse || !cu_seqlens || !work || n_work <= 0 || n_work % 8 != 0 || T <= 0 || H <= 0 || D % H != 0) return 1;
  const int dh = D / H;
  // head widths 96 / 192: the 32x32x16-MFMA forward (attention_m32.hip); dh 384: the paired schedule (attn_fwd_pair_kernel: bit-identical to
  // attn_fwd_dma_kernel<384>, 908 -> 815-821 us on cfg5's global pass).  The kernels these replaced -- the 16x16x32 LDS-DMA forward at dh 96 / 192 /
  // 384, its ROW-MAJOR-stage form -- and the environment switches that select them (CHADAVIT_ATTN_FWD_M32 = -1 / 1 / 2, CHADAVIT_ATTN_FWD_RM = 1,
  // CHADAVIT_ATTN_FWD_PAIR = 0) exist only in side builds (-DCHADA_AB_SWITCHES=1: chadavit_amd.build.build(side="ab"), same-box A/B runs and the
  // bit-identity tests); the product library has neither the switches nor those instances.
#if CHADA_AB_SWITCHES
  static const int m32_variant = getenv("CHADAVIT_ATTN_FWD_M32") ? atoi(getenv("CHADAVIT_ATTN_FWD_M32")) : 0;
  static const int fwd_rm = getenv("CHADAVIT_ATTN_FWD_RM") ? atoi(getenv("CHADAVIT_ATTN_FWD_RM")) : 0;
  static const int fwd_pair = getenv("CHADAVIT_ATTN_FWD_PAIR") ? atoi(getenv("CHADAVIT_ATTN_FWD_PAIR")) : 1;
#else
  constexpr int m32_variant = 0, fwd_rm = 0, fwd_pair = 1;
#endif
  if (fwd_pair > 0 && fwd_rm <= 0 && dh == 384) {
    hipLaunchKernelGGL((attn_fwd_pair_kernel<384>), dim3(n_work * H), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(qkv_),
                       reinterpret_cast<bf16_t*>(out_), lse, cu_seqlens, work, T, D, H, 1.0f / sqrtf((float)dh));
    CHADA_CHECK_LAUNCH();
    return 0;
  }
#if CHADA_AB_SWITCHES
  if (fwd_rm > 0 && (dh == 96 || dh == 192 || dh == 384)) {
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
    bf16_t* out = reinterpret_cast<bf16_t*>(out_);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)dh);
    if (dh == 96) hipLaunchKernelGGL((attn_fwd_dma_kernel<96, 2, true>), dim3(n_work * H), dim3(256), 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale);
    else if (dh == 192) hipLaunchKernelGGL((attn_fwd_dma_kernel<192, 2, true>), dim3(n_work * H), dim3(256), 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale);
    else hipLaunchKernelGGL((attn_fwd_dma_kernel<384, 1, true>), dim3(n_work * H), dim3(512), 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale);
    CHADA_CHECK_LAUNCH();
    return 0;
  }
#endif
  if (m32_variant >= 0 && (dh == 96 || dh == 192))
    return chadavit_attn_fwd_m32(qkv_, out_, lse, cu_seqlens, work, n_work, T, D, H, m32_variant, stream);
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
  bf16_t* out = reinterpret_cast<bf16_t*>(out_);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float scale = 1.0f / sqrtf((float)dh);
  const dim3 blk(256);
  // (round 4: the CHADAVIT_ATTN_NO_DMA fallbacks -- register-staged instances of every LDS-DMA kernel -- are gone; the register-staged
  // kernels remain for the head widths that have no DMA instance: 128 / 256 forward, 32 / 64 / 128 / 256 backward)
#define FWD_DMA_CASE(DHV, CBV)                                                                                     \
  case DHV:                                                                                                       \
    hipLaunchKernelGGL((attn_fwd_dma_kernel<DHV, CBV>), dim3(n_work * (2 / CBV) * H), blk, 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale); \
    break;
#define FWD_CASE(DHV, CBV)                                                                                         \
  case DHV:                                                                                                       \
    hipLaunchKernelGGL((attn_fwd_kernel<DHV, CBV>), dim3(n_work * (2 / CBV) * H), blk, 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale); \
    break;
  switch (dh) {
    case 16:  // forward only (feature extraction with the 12-head default constructor); training uses the 2-head factory
      hipLaunchKernelGGL((attn_fwd_dma_kernel<16, 2>), dim3(n_work * H), blk, 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale);
      break;
    FWD_DMA_CASE(32, 2) FWD_DMA_CASE(64, 2)
    FWD_CASE(128, 2) FWD_CASE(256, 1)   // embed_dim 256 / 512 with the factory's two heads: the register-staged kernels (cold path)
#if CHADA_AB_SWITCHES   // (the product dispatches dh 96 / 192 / 384 above)
    FWD_DMA_CASE(96, 2) FWD_DMA_CASE(192, 2)
    case 384:  // eight waves x 16 query rows per 128-row tile (FwdDmaCfg<384>::NW)
      hipLaunchKernelGGL((attn_fwd_dma_kernel<384, 1>), dim3(n_work * H), dim3(512), 0, s, qkv, out, lse, cu_seqlens, work, T, D, H, scale);
      break;
#endif
    default: return 2;
  }
#undef FWD_CASE
#undef FWD_DMA_CASE
  CHADA_CHECK_LAUNCH();
  return 0;
}

// Backward pieces.  `parts` bit 1 = delta, 2 = dQ kernel, 4 = dK/dV kernel.  With bits 1 and 2 together delta is produced by
// the dQ kernel itself (no separate pass); dK/dV needs delta and must follow on the same stream.  parts = 7 runs everything.
// `scale` = the softmax scale 1/sqrt(head width of the MODEL); it differs from 1/sqrt(D/H) only for the zero-padded dh = 16 path.
static int attn_bwd_launch(const chada_bf16* qkv_, const chada_bf16* out_, const chada_bf16* dout_, const float* lse,
                           chada_bf16* dqkv_, float* delta, const int* cu_seqlens, const int* work, int n_work,
                           int T, int D, int H, int parts, float scale, void* stream) {
  if (!qkv_ || !out_ || !dout_ || !lse || !dqkv_ || !delta || !cu_seqlens || !work || n_work <= 0 || n_work % 8 != 0 || T <= 0 || H <= 0 ||
      D % H != 0 || D % 4 != 0)
    return 1;
  const int dh = D / H;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
  const bf16_t* out = reinterpret_cast<const bf16_t*>(out_);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(dout_);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(dqkv_);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dh != 32 && dh != 64 && dh != 96 && dh != 128 && dh != 192 && dh != 256 && dh != 384) return 2;
  const bool fuse_delta = (parts & 3) == 3;  // delta comes out of the dQ kernel; the stand-alone pass only if dQ is not run here
  if ((parts & 1) && !fuse_delta) {
    int dgrid = (T + 3) / 4;
    if (dgrid > 4096) dgrid = 4096;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, s, out, dout, delta, T, D, H);
    CHADA_CHECK_LAUNCH();
  }
  // (the 32x32x16-MFMA backward pair of round 4 -- correct, 3 % / 14 % slower than the kernels below -- left the library in round 5:
  // scratch/r4/attention_bwd_m32.hip.txt, profiles/r04b_attention_bwd_m32.md)
  const dim3 blk(256);
#define BWD_REG_CASE(DHV, CBV) /* register-staged kernels: head widths without an LDS-DMA instance */                      \
  case DHV:                                                                                                       \
    if ((parts & 2) && fuse_delta)                                                                                \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV, CBV, true>), dim3(n_work * (2 / CBV) * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out); \
    else if (parts & 2)                                                                                           \
      hipLaunchKernelGGL((attn_bwd_dq_kernel<DHV, CBV, false>), dim3(n_work * (2 / CBV) * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out); \
    if (parts & 4)                                                                                                \
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHV, (DHV <= 96 ? 2 : 1)>), dim3((DHV <= 96 ? 1 : 2) * n_work * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale); \
    break;
  // LDS-DMA kernels (dh 96 / 192 / 384).  With the delta bit the dQ kernel produces delta itself; a dQ request WITHOUT it (the host's
  // two-stream form: delta first, then dQ and dK/dV side by side) reads the caller's.
  switch (dh) {
    BWD_REG_CASE(32, 2) BWD_REG_CASE(64, 2) BWD_REG_CASE(128, 2) BWD_REG_CASE(256, 1)
    case 96:
      // (dh 96 keeps two query blocks per wave at two blocks per CU: one block per wave at three blocks per CU -- 123 registers -- measured
      // 1 123 against 991 us: half the MFMAs per K / V fragment read outweighs the third wave; profiles/r05g_dq192_three_blocks_per_cu.log)
      if ((parts & 2) && fuse_delta)
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<96, 2, true>), dim3(n_work * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
      else if (parts & 2)
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<96, 2, false>), dim3(n_work * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
      if (parts & 4) {
#if CHADA_AB_SWITCHES
        static const int dkv_pair = getenv("CHADAVIT_ATTN_DKV_PAIR") ? atoi(getenv("CHADAVIT_ATTN_DKV_PAIR")) : 0;
        if (dkv_pair > 0 && H == 2)
          hipLaunchKernelGGL((attn_bwd_dkv_pair_kernel<96>), dim3(n_work), dim3(512), 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, scale);
        else
#endif
        hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<96, 2>), dim3(n_work * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale);
      }
      break;
    case 192:  // one query block per wave in dQ: with two, Q + dO + dQ spill (65-124 VGPRs)
      if ((parts & 2) && fuse_delta)
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<192, 1, true>), dim3(n_work * 2 * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
      else if (parts & 2)
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<192, 1, false>), dim3(n_work * 2 * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
      if (parts & 4)
        hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<192, 1>), dim3(2 * n_work * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale);
      break;
    case 384:  // dQ: eight waves x 16 query rows on row-major K / V stages; dK/dV: one wave per SIMD (288 registers of state)
      if ((parts & 2) && fuse_delta) {
        // the row-major-stage kernel as eight waves x 16 rows (round 5: K fetched ONCE per tile and in whole 128-byte lines -- 48 KiB per tile instead of
        // the fragment-major kernel's 72, 96 KiB of LDS instead of 144; bit-identical; 1 261 -> 1 188 us on cfg5's global pass, ragged 939 -> 866).
        // Side builds (-DCHADA_AB_SWITCHES=1) keep attn_bwd_dq_fm_kernel behind CHADAVIT_ATTN_DQ_RM=0 (same-box A/B, the bit-identity test)
#if CHADA_AB_SWITCHES
        static const int dq_rm = getenv("CHADAVIT_ATTN_DQ_RM") ? atoi(getenv("CHADAVIT_ATTN_DQ_RM")) : 1;
        if (dq_rm <= 0)
          hipLaunchKernelGGL((attn_bwd_dq_fm_kernel<384, 1>), dim3(n_work * H), dim3(512), 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
        else
#endif
        hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<384, 1, true, 8>), dim3(n_work * H), dim3(512), 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
      }
      else if (parts & 2)   // (the fragment-major kernel always derives delta: the register-staged one serves the two-stream form)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<384, 1, false>), dim3(n_work * 2 * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, out);
      if (parts & 4)
        hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<384, 1>), dim3(2 * n_work * H), blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale);
      break;
    default: return 2;
  }
#undef BWD_REG_CASE
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_attn_bwd_parts(const chada_bf16* qkv_, const chada_bf16* out_, const chada_bf16* dout_, const float* lse,
                                       chada_bf16* dqkv_, float* delta, const int* cu_seqlens, const int* work, int n_work,
                                       int T, int D, int H, int parts, void* stream) {
  CHADA_ENTRY();
  if (H <= 0 || D % H != 0) return 1;
  return attn_bwd_launch(qkv_, out_, dout_, lse, dqkv_, delta, cu_seqlens, work, n_work, T, D, H, parts,
                         1.0f / sqrtf((float)(D / H)), stream);
}

// ---- dh = 16 backward (the reference's DEFAULT constructor: 12 heads at D = 192, chada_vit.py:138-139 -- the notebook's
// model; fine-tuning it needs a backward).  MFMA k-steps are 32 wide, so every head is widened to 32 lanes with zeros in a
// caller-provided workspace and the dh = 32 kernels run on the widened tensors with the MODEL's softmax scale 1/sqrt(16):
// zero lanes add nothing to QK^T, dP = dO V^T or delta, and receive exactly zero gradient, so the result is the dh = 16
// backward bit for bit in the arithmetic that matters.  Cold path: 3 small copy kernels around the two attention kernels.
namespace {
// src [T, n_sec * H * 16] -> dst [T, n_sec * H * 32] (WIDEN) or back (!WIDEN); one 16-byte piece (8 bf16) per thread
template <bool WIDEN>
__global__ __launch_bounds__(256) void head_pad_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long long n_pieces_wide) {
  const uint4 zero = {0u, 0u, 0u, 0u};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n_pieces_wide; i += (long long)gridDim.x * 256ll) {
    const long long head = i >> 2;        // 4 pieces per widened head
    const int piece = (int)(i & 3);
    if (WIDEN) {
      uint4 v = zero;
      if (piece < 2) v = *reinterpret_cast<const uint4*>(src + head * 16 + piece * 8);
      *reinterpret_cast<uint4*>(dst + head * 32 + piece * 8) = v;
    } else if (piece < 2) {
      *reinterpret_cast<uint4*>(dst + head * 16 + piece * 8) = *reinterpret_cast<const uint4*>(src + head * 32 + piece * 8);
    }
  }
}
}  // namespace

extern "C" long long chadavit_attn_bwd_dh16_workspace_bytes(int T, int H) {
  if (T <= 0 || H <= 0) return -1;
  return (long long)T * H * 32 * 2 * 8;  // widened qkv (3), out (1), dout (1), dqkv (3)
}

extern "C" int chadavit_attn_bwd_dh16(const chada_bf16* qkv_, const chada_bf16* out_, const chada_bf16* dout_, const float* lse,
                                      chada_bf16* dqkv_, float* delta, const int* cu_seqlens, const int* work, int n_work,
                                      int T, int D, int H, void* workspace, long long workspace_bytes, void* stream) {
  CHADA_ENTRY();
  if (!qkv_ || !out_ || !dout_ || !dqkv_ || !workspace || T <= 0 || H <= 0 || D != 16 * H) return 1;
  if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < chadavit_attn_bwd_dh16_workspace_bytes(T, H)) return 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long unit = (long long)T * H * 32;  // elements of one widened [T, H*32] section
  bf16_t* w_qkv = reinterpret_cast<bf16_t*>(workspace);
  bf16_t* w_out = w_qkv + 3 * unit;
  bf16_t* w_dout = w_out + unit;
  bf16_t* w_dqkv = w_dout + unit;
  auto grid = [](long long pieces) { long long g = (pieces + 255) / 256; return dim3((unsigned)(g > 8192 ? 8192 : g)); };
  const long long p1 = (long long)T * H * 4, p3 = 3 * p1;
  hipLaunchKernelGGL((head_pad_kernel<true>), grid(p3), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(qkv_), w_qkv, p3);
  hipLaunchKernelGGL((head_pad_kernel<true>), grid(p1), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(out_), w_out, p1);
  hipLaunchKernelGGL((head_pad_kernel<true>), grid(p1), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(dout_), w_dout, p1);
  CHADA_CHECK_LAUNCH();
  // NB the widened qkv is [T, 3 * (H*32)] with the q | k | v sections contiguous per row, exactly the layout the kernels index
  const int rc = attn_bwd_launch(reinterpret_cast<const chada_bf16*>(w_qkv), reinterpret_cast<const chada_bf16*>(w_out),
                                 reinterpret_cast<const chada_bf16*>(w_dout), lse, reinterpret_cast<chada_bf16*>(w_dqkv), delta,
                                 cu_seqlens, work, n_work, T, 32 * H, H, 7, 0.25f, stream);
  if (rc != 0) return rc;
  hipLaunchKernelGGL((head_pad_kernel<false>), grid(p3), dim3(256), 0, s, w_dqkv, reinterpret_cast<bf16_t*>(dqkv_), p3);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_attn_bwd(const chada_bf16* qkv_, const chada_bf16* out_, const chada_bf16* dout_, const float* lse,
                                 chada_bf16* dqkv_, float* delta, const int* cu_seqlens, const int* work, int n_work,
                                 int T, int D, int H, void* stream) {
  return chadavit_attn_bwd_parts(qkv_, out_, dout_, lse, dqkv_, delta, cu_seqlens, work, n_work, T, D, H, 7, stream);
}

extern "C" int chadavit_attn_probs(const chada_bf16* qkv_, float* probs, const int* cu_seqlens, const long long* prob_offsets,
                                   int B, int T, int D, int H, int max_len, void* stream) {
  CHADA_ENTRY();
  if (!qkv_ || !probs || !cu_seqlens || !prob_offsets || B <= 0 || T <= 0 || H <= 0 || D % H != 0) return 1;
  const int dh = D / H;
  if (dh % 8 != 0 || dh > 384 || max_len > 64 * PROB_MAXK) return 2;
  hipLaunchKernelGGL(attn_probs_kernel, dim3((T + 3) / 4, H), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(qkv_), probs, cu_seqlens, prob_offsets, B, T, D, H, 1.0f / sqrtf((float)dh));
  CHADA_CHECK_LAUNCH();
  return 0;
}
