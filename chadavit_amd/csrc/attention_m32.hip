// Variable-length flash attention forward on v_mfma_f32_32x32x16_bf16 (gfx950), head widths 96 and 192.
// replaces chada_vit.py:105-111 (nn.MultiheadAttention + key padding mask), forward.
//
// What differs from attention.hip's 16x16x32 forward, and why (measurements: profiles/r03a_coissue*.txt, r03b_attention_fwd.md):
//   * a wave of the 16x16x32 kernel is bound by its own in-order ISSUE, not by the matrix pipe, LDS or HBM: per (32 queries x 64 keys)
//     wave-tile it issues 48 MFMAs (16 issue cycles each inside one wave), ~215 VALU (4 cycles, v_exp_f32 8), 36 LDS reads and 6
//     LDS-DMA pieces (~50-60 cycles each while the CU's other waves feed the same 64 B/clk address path): ~1700 issue cycles against
//     768 cycles of matrix pipe.  32x32x16 does the same FLOPs in HALF the MFMA instructions;
//   * S^T = K Q^T with 32x32 tiles leaves 16 keys of ONE query per lane, already in the k-slot order of the B operand of
//     O^T = V^T P^T: no cross-lane traffic between the two GEMMs; row max / sum need one v_permlane32_swap;
//   * a leaner softmax (LEAN = 2): the row maximum is taken ONCE, from the first key tile, and kept as the exponent reference of the
//     whole row -- softmax is invariant to the reference, and fp32 (and bf16: same exponent range) hold exp2 of anything within
//     +-126 of it.  Per (query, key) that removes the max, the rescale test and the running-max bookkeeping; the scores themselves
//     are the exact fp32 ones (P = exp2(fma(s, scale * log2 e, -m))).  A row whose later scores exceed the first tile's maximum by more
//     than 64 (44 nats) is detected at the end (row sum not below 2^64, which also catches inf / NaN) and the whole block re-runs
//     that work item with the textbook online recurrence (LEAN = 0): same result either way.
//   * LEAN = 1 additionally folds scale * log2 e into the Q fragments and the reference into the MFMA's C operand (no FMA left): same
//     speed as LEAN = 2 on the hardware (392 vs 393 us at 1024 x 589 tokens) but a second bf16 rounding of Q -- not used.
//
// LDS image of a 64-key (dh 96) / 32-key (dh 192) tile, written by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction).  Since round 6
// the ROW-MAJOR image of the tile (CHADA_M32_RM below: bit-identical results, -1 ... -5 % by shape); rounds 3-5 and CHADA_M32_RM=0:
//   K record (kb, ks):  lane l = K[key kb*32 + (l & 31)][d = ks*16 + (l >> 5)*8 .. +7]  -> A operand of S^T, one conflict-free
//                       ds_read_b128 at lane * 16;
//   V record (kp, db):  V[keys kp*16 .. +15][d = db*32 .. +31] row-major (64-byte rows)   -> A operand of O^T by two
//                       ds_read_b64_tr_b16; each 32-lane half reads 256 contiguous bytes (conflict-free).
//
// Built and measured, not kept (sources: scratch/r3/attention_m32_all_variants.hip.txt; numbers: profiles/r03b_attention_fwd.md):
// 64 query rows per wave at one wave per SIMD, and a software-pipelined body (S'(t+1) MFMAs interleaved with the exponentials of
// tile t, P V MFMAs with those of tile t+1, fragments two steps ahead, K / V in separate rings) at two waves per SIMD.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace chada;

#ifndef CHADA_AB_SWITCHES
#define CHADA_AB_SWITCHES 0   // 1 (side builds only): kernels that were built, measured and not adopted, behind their switches
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TILE = 128;  // query rows per work item (the host's work list, chadavit_attn_tile_rows)
#ifndef CHADA_M32_KV32
#define CHADA_M32_KV32 0
#endif
// (dh 192 at three waves per SIMD: 41 spilled registers, 1 103 against 632 us on cfg3's global pass -- stays at two; round 5)
constexpr bool KV32_AT_DH96 = CHADA_M32_KV32 != 0;   // 32-key tiles at dh 96 too: 24 KiB of LDS per block, five blocks per CU instead of three
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float OVERFLOW_GUARD = 18446744073709551616.0f;  // 2^64: see SAFE

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 r;
#pragma unroll
  for (int e = 0; e < 16; ++e) r[e] = v;
  return r;
}
__device__ __forceinline__ bf16x8 pack8f(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  bf16x8 r;
  r[0] = (bf16_t)a0; r[1] = (bf16_t)a1; r[2] = (bf16_t)a2; r[3] = (bf16_t)a3;
  r[4] = (bf16_t)a4; r[5] = (bf16_t)a5; r[6] = (bf16_t)a6; r[7] = (bf16_t)a7;
  return r;
}
// {v[l], v[l ^ 32]} combined
__device__ __forceinline__ float half_max(float v) {
  float a, b;
  swap32(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float half_sum(float v) {
  float a, b;
  swap32(v, a, b);
  return a + b;
}

// Row-major stages (round 6).  CHADA_M32_RM = 1: a K / V tile is staged as its ROW-MAJOR image (rows of DH bf16, 64 consecutive 16-byte chunks per
// LDS-DMA instruction: 11-12 lines of 128 bytes touched instead of the fragment records' 32 -- 24-27 cycles of the CU's address path instead of a
// flat 64, profiles/r05o_*), the 16-byte chunks of a row XOR-swizzled on the SOURCE side (the DMA writes lane-linear) so that the 32x32x16
// fragment reads stay conflict-free: K fragments by ds_read_b128 of row (l & 31), chunk 2 ks + (l >> 5); V^T fragments by the transpose read of
// the row-major tile.  The swizzles (brute force over the measured ds_read_b128 lane groups {0-3, 12-15, 20-23, 24-27} / {4-7, 8-11, 16-19,
// 28-31} (+32) and the 32-lane phases of the transpose read; scratch/r6/swizzle_search.py): dh 96 -- low two chunk bits ^= (4 - (row >> 2)) & 3
// (attention.hip's dkv_swz<96>); dh 192 -- low three chunk bits ^= row bits (2, 3, 1).
#ifndef CHADA_M32_RM
#define CHADA_M32_RM 1   // (0: the fragment-major records of rounds 3-5 -- bit-identical results; same-box A/B in profiles/r06a_*)
#endif
template <int DH>
__device__ __forceinline__ int rm_swz(int row) {
  if constexpr (DH == 192) return ((row >> 2) & 3) | (((row >> 1) & 1) << 2);
  else return (4 - ((row >> 2) & 3)) & 3;
}

template <int DH, int CB, int NW>
struct Cfg {
  static constexpr int KVT = (DH > 96 || KV32_AT_DH96) ? 32 : 64;   // keys per tile
  static constexpr int KS = DH / 16;                // 16-wide k-steps of S^T
  static constexpr int DB = DH / 32;                // 32-wide head-dim blocks of O^T
  static constexpr int KB = KVT / 32;               // 32-key blocks of S^T
  static constexpr int KP = KVT / 16;               // 16-key k-steps of O^T
  static constexpr int NKR = KB * KS, NVR = KP * DB, NR = NKR + NVR;  // 1 KiB records per stage
  static constexpr int NRW = NR / NW;               // LDS-DMA instructions per wave and tile
  static constexpr int STAGE = NR * 512;            // bf16 elements
  static constexpr int QPB = NW * 32 * CB;          // query rows per block
  static_assert(NR % NW == 0, "records must split evenly over the waves");
  static_assert(TILE % QPB == 0, "a work item is a whole number of blocks");
};

// Record r of a K / V tile as lane l fetches it: the key row inside the tile and the element offset of the lane's 16 bytes inside a token row of qkv
// (K section at D, V section at 2 D; this head's columns start at the resource's base).
template <int DH, int CB, int NW>
__device__ __forceinline__ void m32_rec_of(int r, int l, int D, int& row, unsigned& col) {
  using C = Cfg<DH, CB, NW>;
  constexpr int KS = C::KS, DB = C::DB, NKR = C::NKR;
  const int li = l & 31, hi = l >> 5;
  if (CHADA_M32_RM) {   // piece r = 64 consecutive 16-byte chunks of the row-major K (r < NKR) / V image; source chunk un-swizzled
    static_assert(!CHADA_M32_RM || C::NVR == NKR, "row-major stages: K and V tiles of equal size");
    const int id = (r % NKR) * 64 + l, ch = id % (DH / 8);
    row = id / (DH / 8);
    col = (r < NKR ? D : 2 * D) + (ch ^ rm_swz<DH>(row)) * 8;
  } else if (r < NKR) {
    row = (r / KS) * 32 + li;
    col = D + (r % KS) * 16 + hi * 8;
  } else {
    const int rv = r - NKR;
    row = (rv / DB) * 16 + (l >> 2);
    col = 2 * D + (rv % DB) * 32 + (l & 3) * 8;
  }
}
// The wave's LDS-DMA instructions for the key tile that starts at row `first`.  A tile that lies wholly inside the sequence -- every tile but the last --
// needs no per-lane arithmetic at all: the lane's byte offset inside the tile is a constant (`lane_off`, one register per piece) and the tile's start rides
// in the instruction's scalar offset.  (Round 6: computed per piece -- add, clamp, 32-bit multiply, add -- it was 46 of the loop's ~160 vector instructions.)
// Only the sequence's last tile clamps its rows, and recomputes the records for that.
template <int DH, int CB, int NW>
__device__ __forceinline__ void m32_issue_tile(BufRsrc qb, bf16_t* __restrict__ dst, int first, int len, unsigned ldu, int D, int w, int l,
                                               const unsigned (&lane_off)[Cfg<DH, CB, NW>::NRW]) {
  using C = Cfg<DH, CB, NW>;
  constexpr int NRW = C::NRW, KVT = C::KVT;
  if (first + KVT <= len) {
    const unsigned tile_bytes = (unsigned)first * ldu * 2u;
#pragma unroll
    for (int i = 0; i < NRW; ++i) lds_dma16(qb, dst + (w + NW * i) * 512, lane_off[i], tile_bytes);
  } else {
    int lx = l;
    asm volatile("" : "+v"(lx));   // (recomputed here, once per work item, instead of living in registers across the tile loop)
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      int row;
      unsigned col;
      m32_rec_of<DH, CB, NW>(w + NW * i, lx, D, row, col);
      const unsigned off = (unsigned)min(first + row, len - 1) * ldu + col;
      lds_dma16(qb, dst + (w + NW * i) * 512, off * 2, 0);
    }
  }
}

struct WorkItem { int b, t, h, part; };
template <int SPLIT>
__device__ __forceinline__ WorkItem decode_work(const int* __restrict__ work, int H) {
  // same convention as attention.hip: block i runs on XCD i % 8, work-list entry j belongs to XCD j % 8
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

// One key tile.  MODE 0: textbook online softmax (running max m, rescale);  MODE 1: the first tile of the diet path (takes the
// row maximum, which becomes the fixed reference held in `minit` = -m);  MODE 2: diet steady state (no max, no FMA, no rescale).
// In MODE 1 / 2 the Q fragments carry the scale (c == 1 is passed).
#ifndef CHADA_M32_ABL
#define CHADA_M32_ABL 0   // timing-only ablations of the unpaired kernel (wrong results): 1 = no refills, 2 = no fragment reads, 8 = no softmax,
                          // 16 = refills issued but never waited for, 32 = every refill fetches key tile 0 (the same L2-hot lines)
#endif
template <int DH, int CB, int NW, int MODE, bool MASKED>
__device__ __forceinline__ void fwd_tile(BufRsrc qb, bf16_t* __restrict__ dst, const bf16_t* __restrict__ sK, bool issue, int kt, int len,
                                         int qrow0, unsigned ldu, float c, int w, int l, int D,
                                         const unsigned (&lane_off)[Cfg<DH, CB, NW>::NRW], const bf16x8 (&qf)[CB][Cfg<DH, CB, NW>::KS],
                                         f32x16 (&o)[CB][Cfg<DH, CB, NW>::DB], float (&m)[CB], float (&ls)[CB], f32x16 (&minit)[CB]) {
  using C = Cfg<DH, CB, NW>;
  constexpr int KS = C::KS, DB = C::DB, KB = C::KB, KP = C::KP, KVT = C::KVT, NKR = C::NKR;
  const int hi = l >> 5;
  if (issue && (CHADA_M32_ABL & 1) == 0) m32_issue_tile<DH, CB, NW>(qb, dst, (CHADA_M32_ABL & 32) ? 0 : (kt + 1) * KVT, len, ldu, D, w, l, lane_off);
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out first (see attention.hip / DESIGN 3a)
  if (qrow0 >= len) return;           // (wave-uniform) no query row of this wave exists: it only feeds the DMA and the barriers
  const bf16_t* sV = sK + NKR * 512;
  const int valid = MASKED ? (len - kt * KVT) : KVT;       // valid keys in this tile (>= 1)
  const int nkb = MASKED ? min(KB, (valid + 31) >> 5) : KB;
  const int nkp = MASKED ? min(KP, (valid + 15) >> 4) : KP;

  // ---- S^T = K Q^T (+ the C operand: zero, or -m in the diet steady state).  (Requesting the fragments two steps ahead through a
  // register ring, as attention.hip does, costs this kernel its third wave per SIMD at dh = 96 -- 168 registers -- and measured
  // slower: 472 against ~380 us; hipcc's own placement stays.)
  f32x16 s[CB][KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (MASKED && kb >= nkb) continue;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 kf;
      if (CHADA_M32_ABL & 2) kf = qf[0][(ks + kb) % KS];
      else if (CHADA_M32_RM) kf = lds_read8(sK + (kb * 32 + (l & 31)) * DH + ((2 * ks + hi) ^ rm_swz<DH>(l & 31)) * 8);   // (the swizzle sees row bits 1-3 only)
      else kf = lds_read8(sK + (kb * KS + ks) * 512 + l * 8);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        if (ks == 0)
          s[cb][kb] = mfma32(kf, qf[cb][0], MODE == 2 ? minit[cb] : splat16(0.f));
        else
          s[cb][kb] = mfma32(kf, qf[cb][ks], s[cb][kb]);
      }
    }
  }
  // ---- softmax
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    if (MASKED) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3) >= valid) s[cb][kb][r] = -INFINITY;
      }
    }
    float ps = 0.f;
    if (CHADA_M32_ABL & 8) {
      ls[cb] += s[cb][0][0];
    } else if (MODE == 4) {  // fixed reference m (taken from the first tile), exact scores: one FMA + exp per score, no maximum
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[cb][kb][r], c, -m[cb]));
          s[cb][kb][r] = p;
          ps += p;
        }
      }
      ls[cb] += ps;
    } else if (MODE == 2) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[cb][kb][r]);
          s[cb][kb][r] = p;
          ps += p;
        }
      }
      ls[cb] += ps;
    } else {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[cb][kb][r]), s[cb][kb][r + 1]);
      }
      mx = half_max(mx);
      const float mn = fmaxf(m[cb], mx * c);
      const float alpha = __builtin_amdgcn_exp2f(m[cb] - mn);
      m[cb] = mn;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[cb][kb][r], c, -mn));
          s[cb][kb][r] = p;
          ps += p;
        }
      }
      ls[cb] = ls[cb] * alpha + ps;
      if (MODE == 0) {
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[cb][db][r] *= alpha;
        }
      } else {
        minit[cb] = splat16(-mn);  // (first tile: o and ls are still zero, nothing to rescale)
      }
    }
  }
  // ---- O^T += V^T P^T: P of k-step kp = registers 8 (kp & 1) .. +7 of the 32-key block kp >> 1, already in B-operand order
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    if (MASKED && kp >= nkp) continue;
    bf16x8 pf[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const f32x16& sv = s[cb][kp >> 1];
      const int b0 = 8 * (kp & 1);
      pf[cb] = pack8f(sv[b0], sv[b0 + 1], sv[b0 + 2], sv[b0 + 3], sv[b0 + 4], sv[b0 + 5], sv[b0 + 6], sv[b0 + 7]);
    }
    const int g = l >> 4, ii = l & 15;
    const bf16_t* vrow = sV + kp * DB * 512 + (4 * (g >> 1) + (ii >> 2)) * 32 + (g & 1) * 16 + (ii & 3) * 4;
    // row-major stage: the lane's 8-byte piece of key row kp * 16 + 4 (g >> 1) + (ii >> 2) (and that + 8), chunk 4 db + 2 (g & 1) + ((ii & 3) >> 1)
    const int trow = 4 * (g >> 1) + (ii >> 2), low2 = 2 * (g & 1) + ((ii & 3) >> 1);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      bf16x8 vf;
      if (CHADA_M32_ABL & 2) {
        vf = qf[0][(db + kp) % KS];
      } else if (CHADA_M32_RM) {
        const bf16x4 lo = lds_read_tr4(sV + (kp * 16 + trow) * DH + ((4 * db + low2) ^ rm_swz<DH>(trow)) * 8 + (ii & 1) * 4);
        const bf16x4 hi4 = lds_read_tr4(sV + (kp * 16 + trow + 8) * DH + ((4 * db + low2) ^ rm_swz<DH>(trow + 8)) * 8 + (ii & 1) * 4);
        vf = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
        const bf16x4 lo = lds_read_tr4(vrow + db * 512);
        const bf16x4 hi4 = lds_read_tr4(vrow + db * 512 + 8 * 32);
        vf = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) o[cb][db] = mfma32(vf, pf[cb], o[cb][db]);
    }
  }
}

// All key tiles of one work item for this block.  Returns with o / m / ls final.
#define M32_TILE_BARRIER() do { if (CHADA_M32_ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
                                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
template <int DH, int CB, int NW, int LEAN>
__device__ __forceinline__ void fwd_item(BufRsrc qrs, bf16_t* smem, int len, int qrow0, unsigned ldu, float c, int w, int l,
                                         int D, const unsigned (&lane_off)[Cfg<DH, CB, NW>::NRW],
                                         const bf16x8 (&qf)[CB][Cfg<DH, CB, NW>::KS], f32x16 (&o)[CB][Cfg<DH, CB, NW>::DB], float (&m)[CB],
                                         float (&ls)[CB]) {
  using C = Cfg<DH, CB, NW>;
  constexpr int KVT = C::KVT, STAGE = C::STAGE, DB = C::DB;
  f32x16 minit[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    m[cb] = -INFINITY;
    ls[cb] = 0.f;
    minit[cb] = splat16(0.f);
#pragma unroll
    for (int db = 0; db < DB; ++db) o[cb][db] = splat16(0.f);
  }
  const int nkt = (len + KVT - 1) / KVT;
  // tile 0: no LDS read follows before the first barrier, issued bare
  m32_issue_tile<DH, CB, NW>(qrs, smem, 0, len, ldu, D, w, l, lane_off);
  constexpr int M_FIRST = LEAN ? 1 : 0, M_REST = LEAN == 1 ? 2 : (LEAN == 2 ? 4 : 0);
  if (nkt == 1) {
    M32_TILE_BARRIER();
    fwd_tile<DH, CB, NW, M_FIRST, true>(qrs, smem + STAGE, smem, false, 0, len, qrow0, ldu, c, w, l, D, lane_off, qf, o, m, ls, minit);
    return;
  }
  M32_TILE_BARRIER();
  fwd_tile<DH, CB, NW, M_FIRST, false>(qrs, smem + STAGE, smem, true, 0, len, qrow0, ldu, c, w, l, D, lane_off, qf, o, m, ls, minit);
  for (int kt = 1; kt < nkt - 1; ++kt) {
    // tile kt has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt) and everybody is done reading
    // the other stage
    M32_TILE_BARRIER();
    fwd_tile<DH, CB, NW, M_REST, false>(qrs, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE, true, kt, len, qrow0, ldu, c, w, l, D,
                                        lane_off, qf, o, m, ls, minit);
  }
  M32_TILE_BARRIER();
  fwd_tile<DH, CB, NW, M_REST, true>(qrs, smem + (nkt & 1) * STAGE, smem + ((nkt - 1) & 1) * STAGE, false, nkt - 1, len, qrow0, ldu, c, w, l,
                                     D, lane_off, qf, o, m, ls, minit);
}

template <int DH, int CB, int NW, int LEAN>
__global__ __launch_bounds__(64 * NW, (CB > 1 ? 1 : (DH <= 96 ? (KV32_AT_DH96 ? 4 : 3) : 2))) void attn_fwd_m32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                                    float* __restrict__ lse, const int* __restrict__ cu,
                                                                                    const int* __restrict__ work, int T, int D, int H,
                                                                                    float scale) {
  using C = Cfg<DH, CB, NW>;
  constexpr int KS = C::KS, DB = C::DB, NRW = C::NRW, STAGE = C::STAGE, QPB = C::QPB;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, li = l & 31;
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

  // Q as the B operand of S^T: lane l = Q[row q0 + (l & 31)][d = ks*16 + (l >> 5)*8 .. +7]; LEAN: times scale*log2(e), rounded once
  bf16x8 qf[CB][KS];
  int qrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    qrow[cb] = qt * TILE + part * QPB + w * 32 * CB + cb * 32 + li;
    const int qr = min(qrow[cb], len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 16 + hi * 8);
      if (LEAN == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * c);
      }
      qf[cb][ks] = v;
    }
  }
  // record r of a tile is fetched by wave r % NW (instruction r / NW of that wave): the lane's byte offset inside a tile
  unsigned lane_off[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    int row;
    unsigned col;
    m32_rec_of<DH, CB, NW>(w + NW * i, l, D, row, col);
    lane_off[i] = ((unsigned)row * 3u * (unsigned)D + col) * 2u;
  }
  const unsigned ldu = 3u * (unsigned)D;
  const int qrow0 = qt * TILE + part * QPB + w * 32 * CB;  // first query row of this wave
  const BufRsrc qrs = make_rsrc(qbase);

  f32x16 o[CB][DB];
  float m[CB], ls[CB], lt[CB];
  fwd_item<DH, CB, NW, LEAN>(qrs, smem, len, qrow0, ldu, LEAN == 1 ? 1.0f : c, w, l, D, lane_off, qf, o, m, ls);
  bool bad = false;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    lt[cb] = half_sum(ls[cb]);
    if (LEAN && qrow0 < len) bad |= !(lt[cb] < OVERFLOW_GUARD);
  }
  if (LEAN) {
    // Fixed-reference softmax went out of range for some row (a score more than 64 above the first tile's maximum, or inf / NaN
    // inputs): the BLOCK re-runs the item with the running-max recurrence (block-uniform: the tile loop has barriers).  The
    // per-wave flags live in the stage the last tile did not read (no second __shared__ object: DESIGN 3a).
    const int nkt = (len + C::KVT - 1) / C::KVT;
    int* flags = reinterpret_cast<int*>(smem + (nkt & 1) * STAGE);
    const bool wbad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (l == 0) flags[w] = wbad ? 1 : 0;
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) any |= flags[i];
    if (any) {
      __syncthreads();  // everybody has read the flags before the re-run's first DMA may land on them
      fwd_item<DH, CB, NW, 0>(qrs, smem, len, qrow0, ldu, LEAN == 1 ? 1.0f : c, w, l, D, lane_off, qf, o, m, ls);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) lt[cb] = half_sum(ls[cb]);
    }
  }
  if (qrow0 >= len) return;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const float inv = 1.0f / lt[cb];
    // accumulator register r of block db = O[query][d = db*32 + 8 (r >> 2) + 4 hi + (r & 3)]: four consecutive d per quad, the other four
    // of the same eight in the lane 32 away.  Stored 8 bytes per lane an instruction writes 32 rows x 16 bytes, and the store ISSUE is
    // what a block's end costs (attention.hip, widen_pair): two quads are packed to bf16 and the halves exchanged, so that the lower
    // half holds all eight elements of quad q4 and the upper half those of quad q4 + 1 -- 16 bytes per lane, 32 rows x 32 bytes per
    // instruction, half as many instructions.  (The exchange runs in every lane; only the stores are predicated.)
    bf16_t* orow = out + (size_t)(seq0 + min(qrow[cb], len - 1)) * D + h * DH + 8 * hi;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int q4 = 0; q4 < 4; q4 += 2) {
        const u32x2 pa = __builtin_bit_cast(u32x2, pack4(o[cb][db][4 * q4] * inv, o[cb][db][4 * q4 + 1] * inv, o[cb][db][4 * q4 + 2] * inv, o[cb][db][4 * q4 + 3] * inv));
        const u32x2 pb = __builtin_bit_cast(u32x2, pack4(o[cb][db][4 * q4 + 4] * inv, o[cb][db][4 * q4 + 5] * inv, o[cb][db][4 * q4 + 6] * inv, o[cb][db][4 * q4 + 7] * inv));
        unsigned a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
        swap32x(a0, b0);
        swap32x(a1, b1);
        if (qrow[cb] < len) *reinterpret_cast<u32x4*>(orow + db * 32 + 8 * q4) = u32x4{a0, a1, b0, b1};
      }
    if (hi == 0 && qrow[cb] < len) lse[(size_t)h * T + seq0 + qrow[cb]] = (m[cb] + log2f(lt[cb])) * LN2;
  }
}

#if CHADA_AB_SWITCHES
// =====================================================================================
// The paired ("ping-pong") schedule on 32x32x16 bodies, head widths 96 and 192 with the factory's TWO heads (round 6).
//
// One 512-thread block per (image, 128-row query tile): half A (waves 0-3) runs head 0, half B (waves 4-7) head 1 of the same 128 query rows
// -- wave w and w + 4 share a SIMD.  Both halves run the program of attn_fwd_m32_kernel<DH, 1, 4, 2> (same fragments, same order of operations:
// results bit-identical to it), cut into two segments per key tile and held ONE SEGMENT APART by the block's barriers:
//   X_t (matrix):          O += P(t-1) V(t-1)   then   S(t) = K(t) Q^T      -- 24 MFMAs of 32 cycles and their fragment reads, nothing else
//   Y_t (everything else): the LDS-DMA requests for K(t+2) and V(t+1), then the softmax of S(t) -> P(t)
// so that beside every matrix segment on a SIMD sits the partner wave's softmax / DMA segment (the three independent blocks per CU of the
// unpaired kernel leave that to chance: MFMA busy 0.42-0.45).  The halves share no data: each owns a ring of two K tiles and two V tiles (48 KiB;
// K(t+2) replaces K(t), last read in this half's X_t; V(t+1) replaces V(t-1), last read in X_t), a wave waits for its own pieces at the end of
// its next X segment, the barrier behind it publishes them.  The last key tile multiplies only the 32-key blocks that hold a valid key (block-
// uniform choice between two straight-line segment bodies, no branch inside an MFMA stream); its masked scores are -inf -> P = 0 exactly.
// A row that leaves the fixed-reference softmax's range makes the block run the loop a second time, that half with the running-max recurrence.
//
// MEASURED (round 6, profiles/r06a_*), bit-identical to the unpaired kernel on every shape incl. the re-run, and NOT ADOPTED -- a side-build kernel
// (-DCHADA_AB_SWITCHES=1, variant 6 / CHADAVIT_ATTN_FWD_PAIR32=1): cfg2's global pass 862 against 745 us (+15 %), local crops +26 %, dh 192 global
// 588-609 against 607 (-3 ... 0 %, with s_setprio 1 for the second half), dh 192 local +9 %.  Why: one 512-thread block per CU leaves nothing beside a
// block's prologue and epilogue (5.4 us of a 14.7 us block at 589 tokens; MFMAs + barriers alone: 590 us), which three independent 4-wave blocks
// per CU cover for one another; at 10 key tiles per item the schedule inside the loop cannot pay that back.
// =====================================================================================
#ifndef CHADA_P32_PD
#define CHADA_P32_PD 2      // fragments requested this many steps ahead of their MFMA
#endif
#ifndef CHADA_P32_PRIO
#define CHADA_P32_PRIO 0    // 1: s_setprio 1 for the second-dispatched half (the guide's static form)
#endif
#ifndef CHADA_P32_RERUN
#define CHADA_P32_RERUN 1   // 0 (timing only): no range check of the lean softmax
#endif
#ifndef CHADA_P32_ABL
#define CHADA_P32_ABL 0     // timing-only ablations (wrong results): 1 = no refills, 2 = no fragment reads, 4 = no MFMAs, 8 = no softmax
#endif
template <int DH>
__global__ __launch_bounds__(512, 1) void attn_fwd_pair32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                                 const int* __restrict__ cu, const int* __restrict__ work, int T, int D,
                                                                 float scale) {
  using C = Cfg<DH, 1, 4>;
  constexpr int KS = C::KS, DB = C::DB, KB = C::KB, KP = C::KP, KVT = C::KVT, NKR = C::NKR, NVR = C::NVR;
  constexpr int KSLOT = NKR * 512, VSLOT = NVR * 512, HALF = 2 * KSLOT + 2 * VSLOT;   // bf16 elements
  static_assert(HALF == 2 * C::STAGE, "a half's rings are also the two stages of the unpaired loop (the re-run path)");
  static_assert(NKR % 4 == 0 && NVR % 4 == 0, "K and V records split evenly over the four waves of a half");
  constexpr int NKW = NKR / 4, NVW = NVR / 4;   // pieces per wave and tile
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * HALF];

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, li = l & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = w >> 2, wq = w & 3;
  const int b = work[2 * blockIdx.x], qt = work[2 * blockIdx.x + 1];   // block i runs on XCD i % 8, entry i belongs to XCD i % 8
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const float c = scale * LOG2E;
  const int qrow0 = qt * TILE + wq * 32, qrow = qrow0 + li;
  const bool idle = qrow0 >= len;   // (wave-uniform) none of this wave's query rows exists: it only feeds the DMA and the barriers

  bf16x8 qfr[1][KS];
  {
    const int qr = min(qrow, len - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qfr[0][ks] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qr * ld + ks * 16 + hi * 8);
  }
  // this wave's pieces: K records wq + 4 i (i < NKW), V records wq + 4 i (i < NVW); row inside the tile and element offset of the lane's 16 bytes
  int krow[NKW], vrow_[NVW];
  unsigned kcol[NKW], vcol[NVW];
#pragma unroll
  for (int i = 0; i < NKW; ++i) {
    const int r = wq + 4 * i;
    krow[i] = (r / KS) * 32 + li;
    kcol[i] = D + (r % KS) * 16 + hi * 8;
  }
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    const int r = wq + 4 * i;
    vrow_[i] = (r / DB) * 16 + (l >> 2);
    vcol[i] = 2 * D + (r % DB) * 32 + (l & 3) * 8;
  }
  const unsigned ldu = 3u * (unsigned)D;
  const int nkt = (len + KVT - 1) / KVT;
  const int valid_last = len - (nkt - 1) * KVT;                 // valid keys of the last tile (>= 1)
  const int nkb_last = min(KB, (valid_last + 31) >> 5);         // its 32-key blocks that hold one
  const BufRsrc qrs = make_rsrc(qbase);
  int opq = 0;
  asm volatile("" : "+s"(opq));   // (the LDS bases go through an opaque zero: see ffn_fused.hip)
  bf16_t* const sKb = smem + h * HALF + opq;
  bf16_t* const sVb = sKb + 2 * KSLOT;
  auto fetch_k = [&](int kt, bf16_t* __restrict__ dst) {
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
      const unsigned off = (unsigned)min(kt * KVT + krow[i], len - 1) * ldu + kcol[i];
      lds_dma16(qrs, dst + (wq + 4 * i) * 512, off * 2, 0);
    }
  };
  auto fetch_v = [&](int kt, bf16_t* __restrict__ dst) {
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
      const unsigned off = (unsigned)min(kt * KVT + vrow_[i], len - 1) * ldu + vcol[i];
      lds_dma16(qrs, dst + (wq + 4 * i) * 512, off * 2, 0);
    }
  };

  f32x16 o[1][DB], s[KB];
  bf16x8 pf[KP];
  float m[1], ls[1];

  int lx = l, vlast = valid_last;   // laundered at the top of every pass (run): what is derived from them must not be hoisted out of the pass loop
  // ---- X: the matrix segment.  NB_PV / NB_S = the 32-key blocks of the P V tile / the S tile that are multiplied (KB except for the last tile)
  auto seg_x = [&](const bf16_t* __restrict__ sV, const bf16_t* __restrict__ sK, auto pv_tag, auto s_tag) {
    constexpr int NB_PV = decltype(pv_tag)::value, NB_S = decltype(s_tag)::value;
    if (idle) return;
    constexpr int NPV = 2 * NB_PV * DB, NS = NB_S * KS, N = NPV + NS, NBS = NB_S > 0 ? NB_S : 1;
    // step order.  Two key blocks (dh 96): the P V steps (DB independent accumulators), then the S steps alternating between the two blocks' chains.
    // One key block (dh 192): S is ONE dependent chain of KS MFMAs -- its steps alternate with the P V steps while both last.
    constexpr bool MIX = (KB == 1) && NPV > 0 && NS > 0;
    auto kind = [](int st, int& idx) {   // true = P V step idx, false = S step idx
      if (MIX) {
        constexpr int NM = (NPV < NS ? NPV : NS);
        if (st < 2 * NM) { idx = st >> 1; return (st & 1) != 0; }
        idx = st - NM;
        return NPV > NS;
      }
      if (st < NPV) { idx = st; return true; }
      idx = st - NPV;
      return false;
    };
    const int g = lx >> 4, ii = lx & 15;
    const bf16_t* vlane = sV + (4 * (g >> 1) + (ii >> 2)) * 32 + (g & 1) * 16 + (ii & 3) * 4;
    auto rd = [&](int st) {
      int idx = 0;
      const bool is_pv = kind(st, idx);
      if constexpr ((CHADA_P32_ABL & 2) != 0) return qfr[0][idx % KS];
      if (is_pv) {   // idx = kp * DB + db
        const bf16x4 lo = lds_read_tr4(vlane + idx * 512);
        const bf16x4 hi4 = lds_read_tr4(vlane + idx * 512 + 8 * 32);
        return (bf16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      const int kb = idx % NBS, ks = idx / NBS;
      return lds_read8(sK + (kb * KS + ks) * 512 + lx * 8);
    };
    constexpr int PD = CHADA_P32_PD, NF = PD + 1;
    bf16x8 fr[NF];
#pragma unroll
    for (int i = 0; i < PD; ++i)
      if (i < N) fr[i % NF] = rd(i);
#pragma unroll
    for (int st = 0; st < N; ++st) {
      if (st + PD < N) fr[(st + PD) % NF] = rd(st + PD);
      __builtin_amdgcn_sched_barrier(0);
      int idx = 0;
      const bool is_pv = kind(st, idx);
      if constexpr ((CHADA_P32_ABL & 4) != 0) {
        if (is_pv) o[0][idx % DB][0] += (float)fr[st % NF][0]; else s[idx % NBS][0] += (float)fr[st % NF][1];
      } else if (is_pv) {
        o[0][idx % DB] = mfma32(fr[st % NF], pf[idx / DB], o[0][idx % DB]);
      } else {
        const int kb = idx % NBS, ks = idx / NBS;
        s[kb] = mfma32(fr[st % NF], qfr[0][ks], ks == 0 ? splat16(0.f) : s[kb]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- Y: softmax of tile kt (fwd_tile's arithmetic: MODE 1 = the first tile, takes the row maximum as the fixed reference; MODE 4 = the rest;
  // REDO pass, half flagged `textbook`: fwd_tile's MODE 0 on every tile -- running maximum, O rescaled)
  bool textbook = false;
  auto seg_y = [&](int kt, auto mode_tag, auto masked_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool MASKED = decltype(masked_tag)::value;
    if (idle) return;
    const int valid = MASKED ? vlast : KVT;
    const int nkb = MASKED ? nkb_last : KB;
    if (MASKED) {
      const int hx = lx >> 5;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + 8 * (r >> 2) + 4 * hx + (r & 3) >= valid) s[kb][r] = -INFINITY;
      }
    }
    float ps = 0.f;
    if (MODE == 4 && !textbook) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -m[0]));
          s[kb][r] = p;
          ps += p;
        }
      }
      ls[0] += ps;
    } else {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[kb][r]), s[kb][r + 1]);
      }
      mx = half_max(mx);
      const float mn = fmaxf(m[0], mx * c);
      const float alpha = __builtin_amdgcn_exp2f(m[0] - mn);
      m[0] = mn;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (MASKED && kb >= nkb) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -mn));
          s[kb][r] = p;
          ps += p;
        }
      }
      ls[0] = ls[0] * alpha + ps;
      if (textbook) {
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][db][r] *= alpha;
        }
      }
    }
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      if (MASKED && (kp >> 1) >= nkb) continue;   // (never multiplied)
      const f32x16& sv = s[kp >> 1];
      const int b0 = 8 * (kp & 1);
      pf[kp] = pack8f(sv[b0], sv[b0 + 1], sv[b0 + 2], sv[b0 + 3], sv[b0 + 4], sv[b0 + 5], sv[b0 + 6], sv[b0 + 7]);
    }
  };
  using One = std::integral_constant<int, 1>;
  using Full = std::integral_constant<int, KB>;
  using None = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M4 = std::integral_constant<int, 4>;

  // ---- all key tiles of the item, both halves
  auto run = [&]() {
    asm volatile("" : "+v"(lx), "+s"(vlast));
    m[0] = -INFINITY;
    ls[0] = 0.f;
#pragma unroll
    for (int db = 0; db < DB; ++db) o[0][db] = splat16(0.f);
    // prologue: K(0), K(1), V(0); the first barrier publishes them
    fetch_k(0, sKb);
    if (nkt > 1) fetch_k(1, sKb + KSLOT);
    fetch_v(0, sVb);
    // (the builtin, not asm: hipcc's wait insertion must KNOW that the Q fragment loads have landed -- attn_fwd_pair_kernel in attention.hip)
    __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("s_barrier" ::: "memory");
    if (h == 1) asm volatile("s_barrier" ::: "memory");   // half B runs one segment behind
    // X_0: S(0) only
    if (nkt > 1 || KB == 1 || nkb_last == KB) seg_x(sVb, sKb, None{}, Full{}); else seg_x(sVb, sKb, None{}, One{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nkt - 1; ++kt) {
      // Y_kt: the refills first (they go out while the exponentials run): K(kt + 2) over K(kt), V(kt + 1) over V(kt - 1)
      if constexpr ((CHADA_P32_ABL & 1) == 0) {
        if (kt + 2 < nkt) fetch_k(kt + 2, sKb + (kt & 1) * KSLOT);
        fetch_v(kt + 1, sVb + ((kt + 1) & 1) * VSLOT);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((CHADA_P32_ABL & 8) == 0) {
        if (kt == 0) seg_y(kt, M1{}, std::false_type{}); else seg_y(kt, M4{}, std::false_type{});
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // X_{kt+1}: O += P(kt) V(kt), then S(kt + 1)
      const bf16_t* sVc = sVb + (kt & 1) * VSLOT;
      const bf16_t* sKn = sKb + ((kt + 1) & 1) * KSLOT;
      if (KB == 1 || kt + 1 < nkt - 1 || nkb_last == KB) seg_x(sVc, sKn, Full{}, Full{}); else seg_x(sVc, sKn, Full{}, One{});
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // the last tile: Y (masked softmax), then O += P V alone
    if constexpr ((CHADA_P32_ABL & 8) == 0) {
      if (nkt == 1) seg_y(0, M1{}, std::true_type{}); else seg_y(nkt - 1, M4{}, std::true_type{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
      const bf16_t* sVc = sVb + ((nkt - 1) & 1) * VSLOT;
      if (KB == 1 || nkb_last == KB) seg_x(sVc, sKb, Full{}, None{}); else seg_x(sVc, sKb, One{}, None{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (h == 0) asm volatile("s_barrier" ::: "memory");   // half A's trailing segment
  };
  if (CHADA_P32_PRIO && h == 1) __builtin_amdgcn_s_setprio(1);
  // ---- pass 0: the lean softmax.  Its range check (attn_fwd_m32_kernel's): a half with a bad row runs the item again with the running-max
  // recurrence -- ONE instance of the loop, run a second time with `textbook` set for that half (the block repeats the loop: its barriers are
  // the block's; the other half repeats its lean pass and gets what it had).  (A second inlined instance of the loop, or the unpaired kernel's
  // fwd_item, costs dh 192 44-300 spilled registers.)
  float lt;
  for (int pass = 0;; ++pass) {
    run();
    lt = half_sum(ls[0]);
    if (pass == 1 || CHADA_P32_RERUN == 0) break;
    const bool bad = !idle && !(lt < OVERFLOW_GUARD);
    const bool wbad = __builtin_amdgcn_ballot_w64(bad) != 0;
    int* flags = reinterpret_cast<int*>(smem);   // (every DMA has landed and every fragment read is done: the barriers above)
    if (l == 0) flags[w] = wbad ? 1 : 0;
    __syncthreads();
    int any[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) any[i >> 2] |= flags[i];
    if ((any[0] | any[1]) == 0) break;
    __syncthreads();   // everybody has read the flags before the re-run's first DMA may land on them
    textbook = any[h] != 0;
  }
  if (idle) return;
  const float inv = 1.0f / lt;
  bf16_t* orow = out + (size_t)(seq0 + min(qrow, len - 1)) * D + h * DH + 8 * hi;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; q4 += 2) {
      const f32x16& ov = o[0][db];
      const u32x2 pa = __builtin_bit_cast(u32x2, pack4(ov[4 * q4] * inv, ov[4 * q4 + 1] * inv, ov[4 * q4 + 2] * inv, ov[4 * q4 + 3] * inv));
      const u32x2 pb = __builtin_bit_cast(u32x2, pack4(ov[4 * q4 + 4] * inv, ov[4 * q4 + 5] * inv, ov[4 * q4 + 6] * inv, ov[4 * q4 + 7] * inv));
      unsigned a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
      swap32x(a0, b0);
      swap32x(a1, b1);
      if (qrow < len) *reinterpret_cast<u32x4*>(orow + db * 32 + 8 * q4) = u32x4{a0, a1, b0, b1};
    }
  if (hi == 0 && qrow < len) lse[(size_t)h * T + seq0 + qrow] = (m[0] + log2f(lt)) * LN2;
}

#endif  // CHADA_AB_SWITCHES

}  // namespace

// variant: 0 / 5 = lean softmax on exact scores (what chadavit_attn_fwd dispatches to), 1 = textbook online softmax, 2 = lean softmax with
// the scale folded into Q (see the file header)
extern "C" int chadavit_attn_fwd_m32(const chada_bf16* qkv_, chada_bf16* out_, float* lse, const int* cu_seqlens, const int* work,
                                     int n_work, int T, int D, int H, int variant, void* stream) {
  CHADA_ENTRY();
  if (!qkv_ || !out_ || !lse || !cu_seqlens || !work || n_work <= 0 || n_work % 8 != 0 || T <= 0 || H <= 0 || D % H != 0) return 1;
  const int dh = D / H;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
  bf16_t* out = reinterpret_cast<bf16_t*>(out_);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float scale = 1.0f / sqrtf((float)dh);
#define M32_LAUNCH(DHV, CBV, NWV, LEANV)                                                                                              \
  hipLaunchKernelGGL((attn_fwd_m32_kernel<DHV, CBV, NWV, LEANV>), dim3(n_work * (TILE / (NWV * 32 * CBV)) * H), dim3(64 * NWV), 0, s, qkv, out, \
                     lse, cu_seqlens, work, T, D, H, scale)
#if CHADA_AB_SWITCHES
  // the paired schedule (two heads of one 128-row tile per 512-thread block): CHADAVIT_ATTN_FWD_PAIR32=1 or variant 6
  static const int pair32 = getenv("CHADAVIT_ATTN_FWD_PAIR32") ? atoi(getenv("CHADAVIT_ATTN_FWD_PAIR32")) : 0;
  if (H == 2 && (dh == 96 || dh == 192) && (variant == 6 || (variant == 0 && pair32 > 0))) {
    if (dh == 96) hipLaunchKernelGGL((attn_fwd_pair32_kernel<96>), dim3(n_work), dim3(512), 0, s, qkv, out, lse, cu_seqlens, work, T, D, scale);
    else hipLaunchKernelGGL((attn_fwd_pair32_kernel<192>), dim3(n_work), dim3(512), 0, s, qkv, out, lse, cu_seqlens, work, T, D, scale);
    CHADA_CHECK_LAUNCH();
    return 0;
  }
#endif
  const int lean = variant == 1 ? 0 : (variant == 2 ? 1 : 2);
  if (dh == 96) {
    if (lean == 0) M32_LAUNCH(96, 1, 4, 0);
    else if (lean == 1) M32_LAUNCH(96, 1, 4, 1);
    else M32_LAUNCH(96, 1, 4, 2);
  } else if (dh == 192) {
    if (lean == 0) M32_LAUNCH(192, 1, 4, 0);
    else if (lean == 1) M32_LAUNCH(192, 1, 4, 1);
    else M32_LAUNCH(192, 1, 4, 2);
  } else {
    return 2;
  }
#undef M32_LAUNCH
  CHADA_CHECK_LAUNCH();
  return 0;
}
