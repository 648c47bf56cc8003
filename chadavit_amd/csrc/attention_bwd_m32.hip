// Variable-length flash attention BACKWARD on v_mfma_f32_32x32x16_bf16 (gfx950), head widths 96 and 192.
// replaces autograd of chada_vit.py:105-111 (nn.MultiheadAttention + key padding mask).
//
// Two kernels, no atomics, deterministic -- as the 16x16x32 pair in attention.hip (which stays for the other head widths):
//   dQ    (attn_bwd_dq_m32_kernel):  a wave owns 32 QUERY rows, walks the key tiles;   S^T, dP^T, then dQ^T += K^T dS^T
//   dK/dV (attn_bwd_dkv_m32_kernel): a wave owns 32 KEY rows,   walks the query tiles; S, dP, then dV^T += dO^T P, dK^T += Q^T dS
//
// Why 32x32x16 (DESIGN 7-1 of round 3, built in round 4): with 16 keys / rows per wave every 1 KiB operand fragment read from LDS fed
// ONE 16x16x32 MFMA (16 KFLOP) -- at dh 192 eight waves read 250 B/clk, the LDS peak, and at dh 96 the wave's own issue stream was twice
// as long as its matrix-pipe time.  A 32x32x16 MFMA does 32 KFLOP per 1 KiB fragment: half the LDS bytes, half the MFMA issues and half
// the fragment reads per FLOP.  The layouts chain without any cross-lane traffic:
//   * dK/dV: S[q][key] = Q K^T and dP[q][key] = dO V^T are issued with the tile's Q / dO rows as the A operand and the wave's K / V rows
//     (resident in registers for the whole kernel) as B.  A lane then holds 16 queries of ONE key (its column), in exactly the k-slot
//     order of the B operand of dV^T[d][key] += dO^T[d][q] P[q][key] and dK^T[d][key] += Q^T[d][q] dS[q][key]; the A operands of those
//     are the tile's dO / Q read TRANSPOSED (ds_read_b64_tr_b16).
//   * dQ: S^T[key][q] = K Q^T, dP^T[key][q] = V dO^T with the wave's Q / dO rows resident as B; a lane holds 16 keys of ONE query, the
//     B operand of dQ^T[d][q] += K^T[d][key] dS^T[key][q]; lse and delta of the query are per-lane scalars.
//   * the saved LSE is the softmax's exponent reference (no running maximum, no rescale: P = exp2(s * scale*log2e - lse*log2e), the
//     same expression the forward's last pass evaluates), and -delta rides in the C operand of the dP MFMAs (dS = P * dP', one multiply).
//
// LDS image of a tile, written by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction), the forward's two record types:
//   row record (rb, ks):  lane l = X[row rb*32 + (l & 31)][d = ks*16 + (l >> 5)*8 .. +7]   -> A operand, one ds_read_b128 at lane * 16
//   tr  record (kp, db):  X[rows kp*16 .. +15][d = db*32 .. +31] row-major, 64-byte rows     -> A operand of the transposed product by two
//                         ds_read_b64_tr_b16 (each 32-lane half reads 256 contiguous bytes)
// Both are conflict-free (PMC: 0 bank-conflict cycles in the forward, which reads the same images).  A tensor needed both ways (Q and dO
// in dK/dV, K in dQ) is staged TWICE -- the second copy comes out of L2 -- instead of searching for a swizzle that serves both reads.
#include <cstdlib>

#include "common.h"

using namespace chada;

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TILE = 128;  // rows per work item (the host's work list, chadavit_attn_tile_rows)
constexpr int NW = 4;      // waves per block: 4 x 32 rows = one work item
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 r;
#pragma unroll
  for (int e = 0; e < 16; ++e) r[e] = v;
  return r;
}
// registers 8 * half .. +7 of a 32x32 accumulator as a bf16 B-operand fragment (k-slot j of lane group hi = row 8 (j >> 2) + 4 hi + (j & 3)
// of the 16-row k-step `half`: the order the transposed A operands below are read in)
__device__ __forceinline__ bf16x8 pack_half(const f32x16& v, int half) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[8 * half + e];
  return r;
}
__device__ __forceinline__ float half_sum(float v) {  // v[l] + v[l ^ 32]
  float a, b;
  swap32(v, a, b);
  return a + b;
}
// transposed A operand from a tr record: lane l <- X[row 8 (j >> 2) + 4 (l >> 5) + (j & 3)][d = l & 31], j < 8
__device__ __forceinline__ bf16x8 read_tr_rec(const bf16_t* rec, int l) {
  const int g = l >> 4, ii = l & 15;
  const bf16_t* p = rec + (4 * (g >> 1) + (ii >> 2)) * 32 + (g & 1) * 16 + (ii & 3) * 4;
  return __builtin_shufflevector(lds_read_tr4(p), lds_read_tr4(p + 8 * 32), 0, 1, 2, 3, 4, 5, 6, 7);
}

struct WorkItem { int b, t, h; };
__device__ __forceinline__ WorkItem decode_work(const int* __restrict__ work, int H) {
  // same convention as attention.hip: block i runs on XCD i % 8, work-list entry j belongs to XCD j % 8
  const int lin = blockIdx.x, xcd = lin & 7;
  const int rest = lin >> 3;
  WorkItem it;
  it.h = rest % H;
  const int wi = (rest / H) * 8 + xcd;
  it.b = work[2 * wi];
  it.t = work[2 * wi + 1];
  return it;
}

// =====================================================================================================================================
// dK / dV
// =====================================================================================================================================
template <int DH>
struct DkvCfg {
  static constexpr int QT = 32;                  // query rows per tile
  static constexpr int KS = DH / 16;             // 16-wide k-steps over the head dim (S, dP)
  static constexpr int DB = DH / 32;             // 32-wide head-dim blocks of dK^T / dV^T
  static constexpr int KP = QT / 16;             // 16-query k-steps of dK^T / dV^T
  static constexpr int NROW = KS, NTR = KP * DB; // records of one tensor: row image, transposed image
  static constexpr int NT = NROW + NTR;          // records per tensor
  static constexpr int NR = 2 * NT;              // Q then dO
  static constexpr int NRW = NR / NW;            // LDS-DMA pieces per wave and tile (the first half Q, the second half dO)
  static constexpr int STAGE = NR * 512 + 128;   // bf16 elements: records | lse[32] | delta[32] (floats)
  static_assert(NT % NW == 0, "a tensor's records must split evenly over the waves");
};

template <int DH, bool MASKED>
__device__ __forceinline__ void dkv_tile(BufRsrc qg, BufRsrc dog, BufRsrc lg, BufRsrc dg, bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd,
                                         bool issue, bool idle, int qt_next, int q0, int len, unsigned ldq, unsigned ldo, float c, int w, int l,
                                         const int (&rec_row)[DkvCfg<DH>::NRW / 2], const unsigned (&rec_col)[DkvCfg<DH>::NRW / 2],
                                         const bf16x8 (&kf)[DkvCfg<DH>::KS], const bf16x8 (&vf)[DkvCfg<DH>::KS],
                                         f32x16 (&dk)[DkvCfg<DH>::DB], f32x16 (&dv)[DkvCfg<DH>::DB]) {
  using C = DkvCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, KP = C::KP, QT = C::QT, NT = C::NT, NROW = C::NROW, NH = C::NRW / 2;
  const int hi = l >> 5;
  if (issue) {
    const int r0 = qt_next * QT;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const unsigned row = (unsigned)min(r0 + rec_row[i], len - 1);
      lds_dma16(qg, dst + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
      lds_dma16(dog, dst + (NT + w + NW * i) * 512, (row * ldo + rec_col[i]) * 2, 0);
    }
    if (w == 0 && l < QT) {  // lse / delta of the tile's queries: 4 bytes per lane
      const int qr = min(r0 + l, len - 1);
      lds_dma4(lg, dst + C::NR * 512, qr * 4, 0);
      lds_dma4(dg, dst + C::NR * 512 + 2 * QT, qr * 4, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out first
  if (idle) return;                   // (wave-uniform) none of this wave's keys exists: it only feeds the DMA and the barriers
  const bf16_t* sQr = rd;
  const bf16_t* sQt = rd + NROW * 512;
  const bf16_t* sOr = rd + NT * 512;
  const bf16_t* sOt = rd + (NT + NROW) * 512;
  const float* sL = reinterpret_cast<const float*>(rd + C::NR * 512);
  const float* sD = sL + QT;

  // accumulator register r = query 8 (r >> 2) + 4 hi + (r & 3) of the tile, column = this lane's key
  f32x16 s = splat16(0.f), dp;
  f32x4 l4[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    l4[q4] = *reinterpret_cast<const f32x4*>(sL + 8 * q4 + 4 * hi) * LOG2E;
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 8 * q4 + 4 * hi);
#pragma unroll
    for (int e = 0; e < 4; ++e) dp[4 * q4 + e] = -d4[e];
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) s = mfma32(lds_read8(sQr + ks * 512 + l * 8), kf[ks], s);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) dp = mfma32(lds_read8(sOr + ks * 512 + l * 8), vf[ks], dp);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float p = __builtin_amdgcn_exp2f(fmaf(s[r], c, -l4[r >> 2][r & 3]));
    if (MASKED && (q0 * QT + 8 * (r >> 2) + 4 * hi + (r & 3) >= len)) p = 0.f;
    s[r] = p;
    dp[r] = p * dp[r];
  }
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    if (MASKED && q0 * QT + kp * 16 >= len) continue;  // (wave-uniform) no valid query in this k-step: P and dS are zero
    const bf16x8 pf = pack_half(s, kp), dsf = pack_half(dp, kp);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      dv[db] = mfma32(read_tr_rec(sOt + (kp * DB + db) * 512, l), pf, dv[db]);
      dk[db] = mfma32(read_tr_rec(sQt + (kp * DB + db) * 512, l), dsf, dk[db]);
    }
  }
}

template <int DH>
__global__ __launch_bounds__(64 * NW, (DH <= 96 ? 2 : 1)) void attn_bwd_dkv_m32_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                                                       bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                                                       const int* __restrict__ work, int T, int D, int H, float scale) {
  using C = DkvCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, QT = C::QT, NH = C::NRW / 2, NROW = C::NROW, STAGE = C::STAGE;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, li = l & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WorkItem it = decode_work(work, H);
  const int b = it.b, kt = it.t, h = it.h;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (kt * TILE >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const bf16_t* dobase = dout + (size_t)seq0 * D + h * DH;
  const float c = scale * LOG2E;

  // this wave's 32 keys as B operands: lane l = K / V[key k0 + (l & 31)][d = ks*16 + (l >> 5)*8 .. +7]
  const int krow = kt * TILE + w * 32 + li;
  const bool idle = kt * TILE + w * 32 >= len;
  bf16x8 kf[KS], vf[KS];
  {
    const bf16_t* kr = qbase + (size_t)min(krow, len - 1) * ld + D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kr + ks * 16);
      vf[ks] = *reinterpret_cast<const bf16x8*>(kr + D + ks * 16);
    }
  }
  f32x16 dk[DB], dv[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    dk[db] = splat16(0.f);
    dv[db] = splat16(0.f);
  }
  // record r of a tensor is fetched by wave r % NW (instruction r / NW of that wave): row inside the tile, column inside the head
  int rec_row[NH];
  unsigned rec_col[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int r = w + NW * i;
    if (r < NROW) {
      rec_row[i] = li;
      rec_col[i] = r * 16 + hi * 8;
    } else {
      const int rv = r - NROW;
      rec_row[i] = (rv / DB) * 16 + (l >> 2);
      rec_col[i] = (rv % DB) * 32 + (l & 3) * 8;
    }
  }
  const unsigned ldq = 3u * (unsigned)D, ldo = (unsigned)D;
  const int nqt = (len + QT - 1) / QT;
  const BufRsrc qrs = make_rsrc(qbase), dors = make_rsrc(dobase), lrs = make_rsrc(lse + (size_t)h * T + seq0),
                drs = make_rsrc(delta + (size_t)h * T + seq0);
  // tile 0 (no LDS read follows before the first barrier: issued bare)
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const unsigned row = (unsigned)min(rec_row[i], len - 1);
    lds_dma16(qrs, smem + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
    lds_dma16(dors, smem + (C::NT + w + NW * i) * 512, (row * ldo + rec_col[i]) * 2, 0);
  }
  if (w == 0 && l < QT) {
    const int qr = min(l, len - 1);
    lds_dma4(lrs, smem + C::NR * 512, qr * 4, 0);
    lds_dma4(drs, smem + C::NR * 512 + 2 * QT, qr * 4, 0);
  }
  for (int q0 = 0; q0 < nqt - 1; ++q0) {
    // tile q0 has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt) and everybody is done reading the
    // other stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dkv_tile<DH, false>(qrs, dors, lrs, drs, smem + ((q0 + 1) & 1) * STAGE, smem + (q0 & 1) * STAGE, true, idle, q0 + 1, q0, len, ldq, ldo, c, w, l,
                        rec_row, rec_col, kf, vf, dk, dv);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  dkv_tile<DH, true>(qrs, dors, lrs, drs, smem + (nqt & 1) * STAGE, smem + ((nqt - 1) & 1) * STAGE, false, idle, 0, nqt - 1, len, ldq, ldo, c, w, l,
                     rec_row, rec_col, kf, vf, dk, dv);
  if (krow < len) {
    // accumulator register r of block db = dK^T / dV^T[d = db*32 + 8 (r >> 2) + 4 hi + (r & 3)][key]: four consecutive d per quad
    bf16_t* drow = dqkv + (size_t)(seq0 + krow) * ld + h * DH + 4 * hi;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        *reinterpret_cast<bf16x4*>(drow + D + db * 32 + 8 * q4) =
            pack4(dk[db][4 * q4] * scale, dk[db][4 * q4 + 1] * scale, dk[db][4 * q4 + 2] * scale, dk[db][4 * q4 + 3] * scale);
        *reinterpret_cast<bf16x4*>(drow + 2 * D + db * 32 + 8 * q4) = pack4(dv[db][4 * q4], dv[db][4 * q4 + 1], dv[db][4 * q4 + 2], dv[db][4 * q4 + 3]);
      }
  }
}

// =====================================================================================================================================
// dQ (+ delta)
// =====================================================================================================================================
template <int DH>
struct DqCfg {
  static constexpr int KVT = (DH > 96) ? 32 : 64;  // keys per tile
  static constexpr int KS = DH / 16, DB = DH / 32, KB = KVT / 32, KP = KVT / 16;
  static constexpr int NKR = KB * KS, NKT = KP * DB, NVR = KB * KS;  // K row image, K transposed image, V row image
  static constexpr int NR = NKR + NKT + NVR;
  static constexpr int NRW = NR / NW;
  static constexpr int STAGE = NR * 512;
  static_assert(NR % NW == 0, "records must split evenly over the waves");
};

template <int DH, bool MASKED>
__device__ __forceinline__ void dq_tile(BufRsrc qb, bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd, bool issue, bool idle, int kt, int len,
                                        unsigned ldu, float c, float l2, float ndelta, int w, int l, const int (&rec_row)[DqCfg<DH>::NRW],
                                        const unsigned (&rec_col)[DqCfg<DH>::NRW], const bf16x8 (&qf)[DqCfg<DH>::KS],
                                        const bf16x8 (&dof)[DqCfg<DH>::KS], f32x16 (&dq)[DqCfg<DH>::DB]) {
  using C = DqCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, KB = C::KB, KP = C::KP, KVT = C::KVT, NKR = C::NKR, NKT = C::NKT, NRW = C::NRW;
  const int hi = l >> 5;
  if (issue) {
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      const unsigned off = (unsigned)min((kt + 1) * KVT + rec_row[i], len - 1) * ldu + rec_col[i];
      lds_dma16(qb, dst + (w + NW * i) * 512, off * 2, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (idle) return;
  const bf16_t* sKr = rd;
  const bf16_t* sKt = rd + NKR * 512;
  const bf16_t* sVr = rd + (NKR + NKT) * 512;
  const int valid = MASKED ? (len - kt * KVT) : KVT;  // valid keys in this tile (>= 1)
  const int nkb = MASKED ? min(KB, (valid + 31) >> 5) : KB;
  const int nkp = MASKED ? min(KP, (valid + 15) >> 4) : KP;

  // accumulator register r of block kb = key kb*32 + 8 (r >> 2) + 4 hi + (r & 3) of the tile, column = this lane's query
  f32x16 s[KB], dp[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (MASKED && kb >= nkb) continue;
    s[kb] = splat16(0.f);
    dp[kb] = splat16(ndelta);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s[kb] = mfma32(lds_read8(sKr + (kb * KS + ks) * 512 + l * 8), qf[ks], s[kb]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dp[kb] = mfma32(lds_read8(sVr + (kb * KS + ks) * 512 + l * 8), dof[ks], dp[kb]);
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (MASKED && kb >= nkb) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -l2));
      if (MASKED && (kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3) >= valid)) p = 0.f;
      dp[kb][r] = p * dp[kb][r];
    }
  }
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) {
    if (MASKED && kp >= nkp) continue;
    const bf16x8 dsf = pack_half(dp[kp >> 1], kp & 1);
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = mfma32(read_tr_rec(sKt + (kp * DB + db) * 512, l), dsf, dq[db]);
  }
}

template <int DH>
__global__ __launch_bounds__(64 * NW, (DH <= 96 ? 2 : 1)) void attn_bwd_dq_m32_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                                      const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                                      float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                                                      const int* __restrict__ cu, const int* __restrict__ work, int T,
                                                                                      int D, int H, float scale, int write_delta) {
  using C = DqCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, KVT = C::KVT, NKR = C::NKR, NKT = C::NKT, NRW = C::NRW, STAGE = C::STAGE;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x, l = tid & 63, hi = l >> 5, li = l & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WorkItem it = decode_work(work, H);
  const int b = it.b, qt = it.t, h = it.h;
  if (b < 0) return;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  if (qt * TILE >= len) return;
  const size_t ld = 3 * (size_t)D;
  const bf16_t* qbase = qkv + (size_t)seq0 * ld + h * DH;
  const float c = scale * LOG2E;

  // this wave's 32 queries as B operands: lane l = Q / dO[query q0 + (l & 31)][d = ks*16 + (l >> 5)*8 .. +7]; delta = rowsum(dO * O)
  const int qrow = qt * TILE + w * 32 + li;
  const bool idle = qt * TILE + w * 32 >= len;
  const int qr = min(qrow, len - 1);
  bf16x8 qf[KS], dof[KS];
  float dsum = 0.f;
  {
    const bf16_t* qp = qbase + (size_t)qr * ld + hi * 8;
    const bf16_t* dop = dout + (size_t)(seq0 + qr) * D + h * DH + hi * 8;
    const bf16_t* op = out + (size_t)(seq0 + qr) * D + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dop + ks * 16);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + ks * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = fmaf((float)dof[ks][e], (float)of[e], dsum);
    }
  }
  float dl;
  if (write_delta) {
    dl = half_sum(dsum);
    if (hi == 0 && qrow < len) delta[(size_t)h * T + seq0 + qrow] = dl;
  } else {
    dl = delta[(size_t)h * T + seq0 + qr];
  }
  const float l2 = lse[(size_t)h * T + seq0 + qr] * LOG2E;

  f32x16 dq[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) dq[db] = splat16(0.f);
  // record r of a tile is fetched by wave r % NW (instruction r / NW of that wave): K rows, K transposed image, V rows
  int rec_row[NRW];
  unsigned rec_col[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int r = w + NW * i;
    if (r < NKR) {
      rec_row[i] = (r / KS) * 32 + li;
      rec_col[i] = D + (r % KS) * 16 + hi * 8;
    } else if (r < NKR + NKT) {
      const int rv = r - NKR;
      rec_row[i] = (rv / DB) * 16 + (l >> 2);
      rec_col[i] = D + (rv % DB) * 32 + (l & 3) * 8;
    } else {
      const int rv = r - NKR - NKT;
      rec_row[i] = (rv / KS) * 32 + li;
      rec_col[i] = 2 * D + (rv % KS) * 16 + hi * 8;
    }
  }
  const unsigned ldu = 3u * (unsigned)D;
  const BufRsrc qrs = make_rsrc(qbase);
  const int nkt = (len + KVT - 1) / KVT;
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const unsigned off = (unsigned)min(rec_row[i], len - 1) * ldu + rec_col[i];
    lds_dma16(qrs, smem + (w + NW * i) * 512, off * 2, 0);
  }
  for (int kt = 0; kt < nkt - 1; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dq_tile<DH, false>(qrs, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE, true, idle, kt, len, ldu, c, l2, -dl, w, l, rec_row, rec_col, qf, dof,
                       dq);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  dq_tile<DH, true>(qrs, smem + (nkt & 1) * STAGE, smem + ((nkt - 1) & 1) * STAGE, false, idle, nkt - 1, len, ldu, c, l2, -dl, w, l, rec_row, rec_col, qf,
                    dof, dq);
  if (qrow < len) {
    bf16_t* drow = dqkv + (size_t)(seq0 + qrow) * ld + h * DH + 4 * hi;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<bf16x4*>(drow + db * 32 + 8 * q4) =
            pack4(dq[db][4 * q4] * scale, dq[db][4 * q4 + 1] * scale, dq[db][4 * q4 + 2] * scale, dq[db][4 * q4 + 3] * scale);
  }
}

}  // namespace

// parts: bit 1 = delta, 2 = dQ kernel, 4 = dK/dV kernel (as chadavit_attn_bwd_parts).  delta is produced by the dQ kernel when bits 1 and 2
// are both set; with bit 2 alone the dQ kernel READS the caller's delta; bit 1 alone is not served here (returns 3: the caller runs the
// stand-alone delta pass of attention.hip).
extern "C" int chadavit_attn_bwd_m32(const chada_bf16* qkv_, const chada_bf16* out_, const chada_bf16* dout_, const float* lse, chada_bf16* dqkv_,
                                     float* delta, const int* cu_seqlens, const int* work, int n_work, int T, int D, int H, int parts, float scale,
                                     void* stream) {
  CHADA_ENTRY();
  if (!qkv_ || !out_ || !dout_ || !lse || !dqkv_ || !delta || !cu_seqlens || !work || n_work <= 0 || n_work % 8 != 0 || T <= 0 || H <= 0 ||
      D % H != 0)
    return 1;
  const int dh = D / H;
  if (dh != 96 && dh != 192) return 2;
  if ((parts & 3) == 1) return 3;
  const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
  const bf16_t* out = reinterpret_cast<const bf16_t*>(out_);
  const bf16_t* dout = reinterpret_cast<const bf16_t*>(dout_);
  bf16_t* dqkv = reinterpret_cast<bf16_t*>(dqkv_);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(n_work * H), blk(64 * NW);
  const int write_delta = (parts & 1) ? 1 : 0;
  if (parts & 2) {
    if (dh == 96)
      hipLaunchKernelGGL((attn_bwd_dq_m32_kernel<96>), grid, blk, 0, s, qkv, out, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, write_delta);
    else
      hipLaunchKernelGGL((attn_bwd_dq_m32_kernel<192>), grid, blk, 0, s, qkv, out, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale, write_delta);
    CHADA_CHECK_LAUNCH();
  }
  if (parts & 4) {
    if (dh == 96)
      hipLaunchKernelGGL((attn_bwd_dkv_m32_kernel<96>), grid, blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale);
    else
      hipLaunchKernelGGL((attn_bwd_dkv_m32_kernel<192>), grid, blk, 0, s, qkv, dout, lse, delta, dqkv, cu_seqlens, work, T, D, H, scale);
    CHADA_CHECK_LAUNCH();
  }
  return 0;
}
