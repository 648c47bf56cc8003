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
// LDS image of a tile, written by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction).  ONE image per tensor serves both ways a
// tile is read (round 4, second version: two images -- the forward's two record types side by side -- doubled the LDS-DMA pieces and
// halved the tile that fits, and ran 15 % SLOWER than the 16x16x32 pair):
//   record (kp, db) = X[rows kp*16 .. +15][d = db*32 .. +31], row-major, 64-byte rows = four 16-byte units per row, and unit u of row r
//   is stored at position u ^ ((r >> 2) & 3)  (the swizzle is applied on the DMA's SOURCE address: lane l of a piece lands at byte 16 l);
//   * transposed A operand (ds_read_b64_tr_b16): a 32-lane half reads four whole rows = 256 contiguous bytes whatever the unit order;
//   * row-wise A operand (ds_read_b128, lane l <- X[row l & 31][d = ks*16 + (l >> 5)*8 .. +7]): the hardware serves 16 lanes at a time
//     ({0-3,12-15,20-27}, {4-11,16-19,28-31} and their upper-half twins); a row's four units cover bank slots 4 (r & 3) .. +3, each
//     16-lane group holds four rows of every r & 3 class, and (r >> 2) & 3 is different for those four: 16 distinct slots.
// Tiles are 64 rows; LDS reads are software-pipelined by hand across the MFMA groups (sched_barrier fences): a lone wave's exposed LDS
// round trips, not issue slots or bytes, were what the first version spent its time on (5 200 cycles per 768 cycles of matrix pipe).
#include <cstdlib>

#include "common.h"

using namespace chada;

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TILE = 128;  // rows per work item (the host's work list, chadavit_attn_tile_rows)
constexpr int NW = 4;      // waves per block: 4 x 32 rows = one work item
constexpr int RT = 64;     // rows of the streamed tensor per LDS stage
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
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// Per-lane LDS offsets (bf16 elements) into a tile image, see the file header.
struct LaneOffs {
  int row0, row1;  // row-wise read of 32-row block 0: k-step parity 0 / 1 (+ (ks >> 1) * 512 + rb * 2 * DB * 512)
  int tr0, tr1;    // transposed read, rows 0-7 / 8-15 of a record (+ record * 512)
};
template <int DB>
__device__ __forceinline__ LaneOffs lane_offs(int l) {
  LaneOffs o;
  const int rr = l & 31, hi = l >> 5, swz = (rr >> 2) & 3;
  const int rowbase = (rr >> 4) * DB * 512 + (rr & 15) * 32;
  o.row0 = rowbase + ((hi ^ swz) << 3);
  o.row1 = rowbase + (((2 + hi) ^ swz) << 3);
  const int g = l >> 4, ii = l & 15;
  const int r = 4 * (g >> 1) + (ii >> 2), u = (g & 1) * 2 + ((ii & 3) >> 1);
  o.tr0 = r * 32 + ((u ^ (g >> 1)) << 3) + (ii & 1) * 4;
  o.tr1 = (r + 8) * 32 + ((u ^ (2 + (g >> 1))) << 3) + (ii & 1) * 4;
  return o;
}
// A operand, row-wise: lane l <- X[row rb*32 + (l & 31)][d = ks*16 + (l >> 5)*8 .. +7]
template <int DB>
__device__ __forceinline__ bf16x8 read_row(const bf16_t* img, const LaneOffs& o, int rb, int ks) {
  return lds_read8(img + rb * 2 * DB * 512 + (ks >> 1) * 512 + ((ks & 1) ? o.row1 : o.row0));
}
// A operand of the transposed product: lane l <- X[row kp*16 + 8 (j >> 2) + 4 (l >> 5) + (j & 3)][d = db*32 + (l & 31)], j < 8
template <int DB>
__device__ __forceinline__ bf16x8 read_tr(const bf16_t* img, const LaneOffs& o, int kp, int db) {
  const bf16_t* rec = img + (kp * DB + db) * 512;
  return __builtin_shufflevector(lds_read_tr4(rec + o.tr0), lds_read_tr4(rec + o.tr1), 0, 1, 2, 3, 4, 5, 6, 7);
}
// LDS-DMA source of record r of an image (row inside the tile, column inside the head), for lane l
template <int DB>
__device__ __forceinline__ void rec_src(int r, int l, int& row, unsigned& col) {
  row = (r / DB) * 16 + (l >> 2);
  col = (unsigned)((r % DB) * 32 + (((l & 3) ^ ((l >> 4) & 3)) << 3));
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
  static constexpr int KS = DH / 16;             // 16-wide k-steps over the head dim (S, dP)
  static constexpr int DB = DH / 32;             // 32-wide head-dim blocks of dK^T / dV^T
  static constexpr int NT = (RT / 16) * DB;      // records per tensor and tile
  static constexpr int NR = 2 * NT;              // Q then dO
  static constexpr int NH = NT / NW;             // LDS-DMA pieces per wave, tile and tensor
  static constexpr int STAGE = NR * 512 + 256;   // bf16 elements: records | lse[64] | delta[64] (floats)
  static_assert(NT % NW == 0, "a tensor's records must split evenly over the waves");
};

template <int DH>
__device__ __forceinline__ void dkv_tile(BufRsrc qg, BufRsrc dog, BufRsrc lg, BufRsrc dg, bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd,
                                         bool issue, bool idle, int qt_next, int q0, int len, unsigned ldq, unsigned ldo, float c, int w, int l,
                                         const LaneOffs& lo, const int (&rec_row)[DkvCfg<DH>::NH], const unsigned (&rec_col)[DkvCfg<DH>::NH],
                                         const bf16x8 (&kf)[DkvCfg<DH>::KS], const bf16x8 (&vf)[DkvCfg<DH>::KS],
                                         f32x16 (&dk)[DkvCfg<DH>::DB], f32x16 (&dv)[DkvCfg<DH>::DB]) {
  using C = DkvCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, NT = C::NT, NH = C::NH;
  const int hi = l >> 5;
  if (issue) {
    const int r0 = qt_next * RT;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const unsigned row = (unsigned)min(r0 + rec_row[i], len - 1);
      lds_dma16(qg, dst + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
      lds_dma16(dog, dst + (NT + w + NW * i) * 512, (row * ldo + rec_col[i]) * 2, 0);
    }
    if (w == 0) {  // lse / delta of the tile's queries: 4 bytes per lane
      const int qr = min(r0 + l, len - 1);
      lds_dma4(lg, dst + C::NR * 512, qr * 4, 0);
      lds_dma4(dg, dst + C::NR * 512 + 2 * RT, qr * 4, 0);
    }
  }
  FENCE();               // the DMA goes out first
  if (idle) return;      // (wave-uniform) none of this wave's keys exists: it only feeds the DMA and the barriers
  const bf16_t* sQ = rd;
  const bf16_t* sO = rd + NT * 512;
  const float* sL = reinterpret_cast<const float*>(rd + C::NR * 512);
  const float* sD = sL + RT;
  // queries of this tile that exist (all 64 but in a sequence's last tile): rows past them are clamped copies whose P is forced to
  // zero, and 16-query halves without any valid row are skipped (wave-uniform)
  const int nvalid = min(RT, len - q0 * RT);
  const int nu = (nvalid + 31) >> 5;                         // 32-query sub-tiles with a valid query (1 or 2)

  // accumulator register r = query 8 (r >> 2) + 4 hi + (r & 3) of the sub-tile, column = this lane's key.
  // Schedule of one 32-query sub-tile (fences between the groups; the LDS reads named first in a group are requested BEFORE its MFMAs
  // and consumed by a later group -- at most ~80 registers of fragments in flight beside the 144 of K / V / dK / dV at dh 96):
  //   A  [dO rows]                       S   = Q K^T - lse          (C operand: -lse * log2 e; the K fragments carry scale * log2 e)
  //   B  [dO^T, Q^T of queries 0-15]     dP' = dO V^T - delta       (C operand: -delta)
  //   C                                  P = exp2(S), dS = P dP'
  //   D  [dO^T, Q^T of queries 16-31]    dV^T += dO^T P, dK^T += Q^T dS   (queries 0-15)
  //   E  [next sub-tile's Q rows, lse]   the same for queries 16-31
  // (A shallower read-ahead -- one tensor's transposed fragments per group, no spills -- measured 11 % SLOWER than this one with its
  //  20 spilled registers: the kernel lives on how much it keeps in flight, and 256 registers at two waves per SIMD are the limit.)
  bf16x8 qa[KS], da[KS];
  f32x16 s, dp;
  auto load_q = [&](int u, f32x16& acc) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qa[ks] = read_row<DB>(sQ, lo, u, ks);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {   // -lse * log2 e of the lane's 16 query rows: the C operand of the S MFMAs
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(sL + u * 32 + 8 * q4 + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = -LOG2E * v4[e];
    }
  };
  auto load_do = [&](int u, f32x16& acc) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) da[ks] = read_row<DB>(sO, lo, u, ks);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {   // -delta: the C operand of the dP MFMAs
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(sD + u * 32 + 8 * q4 + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * q4 + e] = -v4[e];
    }
  };
  load_q(0, s);
  FENCE();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u >= nu) break;
    bf16x8 tq[2][DB], to[2][DB];
    // ---- A
    load_do(u, dp);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(qa[ks], kf[ks], s);
    FENCE();
    // ---- B
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      to[0][db] = read_tr<DB>(sO, lo, 2 * u, db);
      tq[0][db] = read_tr<DB>(sQ, lo, 2 * u, db);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dp = mfma32(da[ks], vf[ks], dp);
    FENCE();
    // ---- C
    const int nrem = nvalid - 4 * hi;   // row 8 (r >> 2) + 4 hi + (r & 3) of sub-tile u exists iff u*32 + 8 (r >> 2) + (r & 3) < nrem
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = __builtin_amdgcn_exp2f(s[r]);
      if (u * 32 + 8 * (r >> 2) + (r & 3) >= nrem) p = 0.f;
      s[r] = p;
      dp[r] = p * dp[r];
    }
    const bf16x8 pf0 = pack_half(s, 0), pf1 = pack_half(s, 1), ds0 = pack_half(dp, 0), ds1 = pack_half(dp, 1);
    FENCE();
    // ---- D
    const bool second = u * 32 + 16 < nvalid;   // (wave-uniform) queries 16-31 of the sub-tile exist
    if (second) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        to[1][db] = read_tr<DB>(sO, lo, 2 * u + 1, db);
        tq[1][db] = read_tr<DB>(sQ, lo, 2 * u + 1, db);
      }
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      dv[db] = mfma32(to[0][db], pf0, dv[db]);
      dk[db] = mfma32(tq[0][db], ds0, dk[db]);
    }
    FENCE();
    // ---- E
    if (u == 0 && nu > 1) load_q(1, s);
    if (second) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        dv[db] = mfma32(to[1][db], pf1, dv[db]);
        dk[db] = mfma32(tq[1][db], ds1, dk[db]);
      }
    }
    FENCE();
  }
}

template <int DH>
__global__ __launch_bounds__(64 * NW, (DH <= 96 ? 2 : 1)) void attn_bwd_dkv_m32_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                                                       bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                                                       const int* __restrict__ work, int T, int D, int H, float scale) {
  using C = DkvCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, NH = C::NH, STAGE = C::STAGE;
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
      // the S MFMAs produce the exponent itself: scale * log2 e rides in the K fragments (one more bf16 rounding of K, of the size of the
      // rounding K already carries), -lse * log2 e in the C operand
#pragma unroll
      for (int e = 0; e < 8; ++e) kf[ks][e] = (bf16_t)((float)kf[ks][e] * c);
    }
  }
  f32x16 dk[DB], dv[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    dk[db] = splat16(0.f);
    dv[db] = splat16(0.f);
  }
  // record r of a tensor is fetched by wave r % NW (instruction r / NW of that wave)
  int rec_row[NH];
  unsigned rec_col[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) rec_src<DB>(w + NW * i, l, rec_row[i], rec_col[i]);
  const LaneOffs lo = lane_offs<DB>(l);
  const unsigned ldq = 3u * (unsigned)D, ldo = (unsigned)D;
  const int nqt = (len + RT - 1) / RT;
  const BufRsrc qrs = make_rsrc(qbase), dors = make_rsrc(dobase), lrs = make_rsrc(lse + (size_t)h * T + seq0),
                drs = make_rsrc(delta + (size_t)h * T + seq0);
  // tile 0 (no LDS read follows before the first barrier: issued bare)
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const unsigned row = (unsigned)min(rec_row[i], len - 1);
    lds_dma16(qrs, smem + (w + NW * i) * 512, (row * ldq + rec_col[i]) * 2, 0);
    lds_dma16(dors, smem + (C::NT + w + NW * i) * 512, (row * ldo + rec_col[i]) * 2, 0);
  }
  if (w == 0) {
    const int qr = min(l, len - 1);
    lds_dma4(lrs, smem + C::NR * 512, qr * 4, 0);
    lds_dma4(drs, smem + C::NR * 512 + 2 * RT, qr * 4, 0);
  }
  for (int q0 = 0; q0 < nqt; ++q0) {
    // tile q0 has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt) and everybody is done reading the
    // other stage.  ONE body for all tiles (the last one's masking is a handful of compares): a second, masked instantiation raised
    // the kernel's register peak past 256 and the spills it caused sat in front of the main loop.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dkv_tile<DH>(qrs, dors, lrs, drs, smem + ((q0 + 1) & 1) * STAGE, smem + (q0 & 1) * STAGE, q0 + 1 < nqt, idle, q0 + 1, q0, len, ldq, ldo, c, w, l, lo,
                 rec_row, rec_col, kf, vf, dk, dv);
  }
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
  static constexpr int KS = DH / 16, DB = DH / 32;
  static constexpr int NT = (RT / 16) * DB;   // records per tensor and tile: K (read row-wise and transposed), V (row-wise)
  static constexpr int NR = 2 * NT;
  static constexpr int NH = NT / NW;
  static constexpr int STAGE = NR * 512;
  static_assert(NT % NW == 0, "a tensor's records must split evenly over the waves");
};

template <int DH, bool MASKED>
__device__ __forceinline__ void dq_tile(BufRsrc qb, bf16_t* __restrict__ dst, const bf16_t* __restrict__ rd, bool issue, bool idle, int kt, int len,
                                        unsigned ldu, unsigned dcol, float c, float l2, float ndelta, int w, int l, const LaneOffs& lo,
                                        const int (&rec_row)[DqCfg<DH>::NH], const unsigned (&rec_col)[DqCfg<DH>::NH],
                                        const bf16x8 (&qf)[DqCfg<DH>::KS], const bf16x8 (&dof)[DqCfg<DH>::KS], f32x16 (&dq)[DqCfg<DH>::DB]) {
  using C = DqCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, NT = C::NT, NH = C::NH;
  const int hi = l >> 5;
  if (issue) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const unsigned off = (unsigned)min((kt + 1) * RT + rec_row[i], len - 1) * ldu + dcol + rec_col[i];
      lds_dma16(qb, dst + (w + NW * i) * 512, off * 2, 0);
      lds_dma16(qb, dst + (NT + w + NW * i) * 512, (off + dcol) * 2, 0);
    }
  }
  FENCE();
  if (idle) return;
  const bf16_t* sK = rd;
  const bf16_t* sV = rd + NT * 512;
  const int valid = MASKED ? (len - kt * RT) : RT;   // valid keys in this tile (>= 1)
  const int nkb = MASKED ? ((valid + 31) >> 5) : 2;

  // accumulator register r = key 8 (r >> 2) + 4 hi + (r & 3) of the 32-key block, column = this lane's query
  bf16x8 ka[KS], va[KS];
  auto load_kv = [&](int kb) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = read_row<DB>(sK, lo, kb, ks);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) va[ks] = read_row<DB>(sV, lo, kb, ks);
  };
  load_kv(0);
  FENCE();
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    if (MASKED && kb >= nkb) break;
    f32x16 s = splat16(-l2), dp = splat16(ndelta);
    bf16x8 tk[2][DB];
#pragma unroll
    for (int kp = 0; kp < 2; ++kp)
#pragma unroll
      for (int db = 0; db < DB; ++db) tk[kp][db] = read_tr<DB>(sK, lo, 2 * kb + kp, db);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(ka[ks], qf[ks], s);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dp = mfma32(va[ks], dof[ks], dp);
    FENCE();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = __builtin_amdgcn_exp2f(s[r]);
      if (MASKED && (kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3) >= valid)) p = 0.f;
      dp[r] = p * dp[r];
    }
    const bf16x8 ds0 = pack_half(dp, 0), ds1 = pack_half(dp, 1);
    FENCE();
    if (kb == 0 && (!MASKED || nkb > 1)) load_kv(1);   // the next 32 keys' rows, requested under this block's last MFMAs
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = mfma32(tk[0][db], ds0, dq[db]);
    if (!MASKED || kb * 32 + 16 < valid) {
#pragma unroll
      for (int db = 0; db < DB; ++db) dq[db] = mfma32(tk[1][db], ds1, dq[db]);
    }
    FENCE();
  }
}

template <int DH>
__global__ __launch_bounds__(64 * NW, (DH <= 96 ? 2 : 1)) void attn_bwd_dq_m32_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                                      const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                                      float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                                                      const int* __restrict__ cu, const int* __restrict__ work, int T,
                                                                                      int D, int H, float scale, int write_delta) {
  using C = DqCfg<DH>;
  constexpr int KS = C::KS, DB = C::DB, NH = C::NH, STAGE = C::STAGE;
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
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16_t)((float)qf[ks][e] * c);   // scale * log2 e rides in the Q fragments (see dK/dV)
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
  // record r of a tensor is fetched by wave r % NW (instruction r / NW of that wave); K sits D columns into the row, V another D
  int rec_row[NH];
  unsigned rec_col[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) rec_src<DB>(w + NW * i, l, rec_row[i], rec_col[i]);
  const LaneOffs lo = lane_offs<DB>(l);
  const unsigned ldu = 3u * (unsigned)D, dcol = (unsigned)D;
  const BufRsrc qrs = make_rsrc(qbase);
  const int nkt = (len + RT - 1) / RT;
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const unsigned off = (unsigned)min(rec_row[i], len - 1) * ldu + dcol + rec_col[i];
    lds_dma16(qrs, smem + (w + NW * i) * 512, off * 2, 0);
    lds_dma16(qrs, smem + (C::NT + w + NW * i) * 512, (off + dcol) * 2, 0);
  }
  for (int kt = 0; kt < nkt - 1; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    dq_tile<DH, false>(qrs, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE, true, idle, kt, len, ldu, dcol, c, l2, -dl, w, l, lo, rec_row, rec_col,
                       qf, dof, dq);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  dq_tile<DH, true>(qrs, smem + (nkt & 1) * STAGE, smem + ((nkt - 1) & 1) * STAGE, false, idle, nkt - 1, len, ldu, dcol, c, l2, -dl, w, l, lo, rec_row,
                    rec_col, qf, dof, dq);
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
