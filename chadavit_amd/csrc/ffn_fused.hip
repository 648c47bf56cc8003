// Fused transformer FFN for D = 192 (ChAda-ViT Tiny):  Out = resid + b2 + relu(X W1^T + b1) W2^T  in ONE kernel.
// reference: TransformerEncoderLayer feed-forward, linear2(dropout(relu(linear1(x)))) + residual
// (torch.nn.TransformerEncoderLayer._ff_block as instantiated at src/backbones/vit/chada_vit.py:256-264).
//
// Why: as two GEMMs the 2048-wide hidden activation H makes a round trip through HBM (M x 2048 bf16 written by FFN1,
// read back by FFN2) and both GEMMs are HBM-bound at D = 192.  Here a block owns 32*RT*4 token rows for the whole
// hidden range; H only ever exists 32 hidden units at a time, in registers:
//   * X fragments of the wave's 16*RT rows stay in registers for the block's lifetime (as MFMA B operands);
//   * the weights stream through LDS by LDS-DMA in a FRAGMENT-MAJOR packing built once per optimiser step by
//     ffn_pack_kernel: every 16x32 MFMA operand is one contiguous 1 KiB record in lane order, so the DMA is a 1 KiB
//     burst and the ds_read_b128 of a fragment is conflict-free by construction;
//   * GEMM1 is issued "transposed" (A = W1 fragment, B = X fragment) so a lane ends up holding, for its token row,
//     8 consecutive hidden units -- exactly the B-operand k-slots of GEMM2: no LDS round trip for H;
//   * hidden row -> MFMA row and output column -> MFMA row permutations are folded into the packing so that the 8
//     values a lane owns are CONTIGUOUS in memory: H (optional, for the backward) and Out are written with 16-byte
//     stores straight from registers;
//   * software pipelining across hidden chunks: iteration k runs GEMM1 of chunk k and GEMM2 of chunk k-1, which are
//     independent, so the MFMA pipe never waits for the bias/ReLU/convert step in between.
#include "common.h"

// This file is compiled twice: as is (D = 192, 4 waves x 32 rows, with the whole-block prologue / postlogue) and from
// ffn_fused_d384.hip with FFN_FD = 384 (Small): 8 waves x 16 rows per block -- the X fragments (12 k-steps) and the 16 x 384 output
// accumulators of ONE 16-row tile are the same 144 registers as two tiles at D = 192, the 48 KiB weight blocks of a chunk are
// shared by eight waves (three stages = 144 KiB: one block per CU).  The secondary build exports its launchers under internal
// names (chada_int_*_d384); the public entry points below dispatch on D.
#ifndef FFN_FD
#define FFN_FD 192
// (the tiling can be overridden from the command line for A/B builds.  Measured at D = 192, 603 136 rows, same box, against 4 x 32 rows at two waves
// per SIMD -- training 1 398 / no-grad 1 018 / backward dX 1 064 us: eight waves x 16 rows at FOUR waves per SIMD (-DFFN_NW=8 -DFFN_RT=1
// -DFFN_MIN_WAVES=4; 16-80 spilled registers in the whole-block instances) 1 523 / 1 202 / 1 244, the same without forcing the registers
// (one block per CU for the whole-block instances) 1 650 / 1 270 / 1 238, six waves x 16 rows at three per SIMD 1 823 / 1 458 / 1 385: one
// row tile per wave doubles the LDS fragment reads per MFMA, and that costs more than the extra waves hide -- profiles/r05m_block_tilings.log)
#ifndef FFN_NW
#define FFN_NW 4
#define FFN_RT 2
#endif
#define FFN_PRIMARY 1
#define FFN_NAME(x) chadavit_##x
#else
#define FFN_PRIMARY 0
#define FFN_NAME(x) chada_int_##x##_d384
#endif

namespace {
using namespace chada;

constexpr int FD = FFN_FD;              // model width this build is specialised for
constexpr int NWV = FFN_NW;             // waves per block
constexpr int DRT = FFN_RT;             // 16-row tiles per wave of the default instances
constexpr int HC = 32;                  // hidden units per chunk
constexpr int KS1 = FD / 32;            // k-steps of GEMM1
constexpr int NT2 = FD / 16;            // output column tiles of GEMM2
constexpr int W1_FRAGS = 2 * KS1;       // 12
constexpr int W2_FRAG0 = W1_FRAGS;      // 12
constexpr int BLK_FRAGS = W2_FRAG0 + NT2;  // 24 records of 1 KiB per packed block: 6 LDS-DMA instructions per wave
constexpr int NPB = KS1 / 2;            // stream blocks of a [D x D] projection matrix (two k-steps of all NT2 tiles per block)
constexpr int MAX_FF = 2048;            // b1 is staged in LDS whole
constexpr int FRAG_ELEMS = 512;         // bf16 elements per record

// hidden unit (within a chunk) that sits at MFMA row i of GEMM1 tile nt; output column (within a 32-wide pair of
// tiles) that sits at MFMA row i of GEMM2 tile t:  8*(i>>2) + 4*t + (i&3)
__device__ __forceinline__ int perm_row(int t, int i) { return 8 * (i >> 2) + 4 * t + (i & 3); }

// Packed block k (k = 0..NC):  [ W1 fragments of chunk k | W2 fragments of chunk k-1 ]
__global__ __launch_bounds__(256) void ffn_pack_kernel(const bf16_t* __restrict__ W1, const bf16_t* __restrict__ W2,
                                                       bf16_t* __restrict__ packed, int FF) {
  const int NC = FF / HC;
  const int k = blockIdx.x;  // 0..NC
  const int tid = threadIdx.x;
  bf16_t* blk = packed + (size_t)k * BLK_FRAGS * FRAG_ELEMS;
  for (int id = tid; id < BLK_FRAGS * 64; id += 256) {
    const int f = id >> 6, l = id & 63, li = l & 15, g = l >> 4;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
    if (f < W1_FRAGS) {
      if (k < NC) {
        const int ks = f >> 1, nt = f & 1;
        v = *reinterpret_cast<const bf16x8*>(W1 + (size_t)(k * HC + perm_row(nt, li)) * FD + ks * 32 + g * 8);
      }
    } else if (k > 0) {
      const int nt2 = f - W2_FRAG0, p = nt2 >> 1, t = nt2 & 1;
      v = *reinterpret_cast<const bf16x8*>(W2 + (size_t)(32 * p + perm_row(t, li)) * FF + (k - 1) * HC + g * 8);
    }
    *reinterpret_cast<bf16x8*>(blk + f * FRAG_ELEMS + l * 8) = v;
  }
}

// the same for all layers of a model in one launch: desc[t] = {W1 offset, W2 offset (bf16 elements into the slab's bf16
// shadow), packed offset (bf16 elements)}; blockIdx.y = layer
__global__ __launch_bounds__(256) void ffn_pack_batched_kernel(const bf16_t* __restrict__ slab, bf16_t* __restrict__ packed,
                                                               const long long* __restrict__ desc, int FF) {
  const long long* d = desc + 3 * blockIdx.y;
  const bf16_t* W1 = slab + d[0];
  const bf16_t* W2 = slab + d[1];
  const int NC = FF / HC;
  const int k = blockIdx.x;  // 0..NC
  const int tid = threadIdx.x;
  bf16_t* blk = packed + d[2] + (size_t)k * BLK_FRAGS * FRAG_ELEMS;
  for (int id = tid; id < BLK_FRAGS * 64; id += 256) {
    const int f = id >> 6, l = id & 63, li = l & 15, g = l >> 4;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
    if (f < W1_FRAGS) {
      if (k < NC) {
        const int ks = f >> 1, nt = f & 1;
        v = *reinterpret_cast<const bf16x8*>(W1 + (size_t)(k * HC + perm_row(nt, li)) * FD + ks * 32 + g * 8);
      }
    } else if (k > 0) {
      const int nt2 = f - W2_FRAG0, p = nt2 >> 1, t = nt2 & 1;
      v = *reinterpret_cast<const bf16x8*>(W2 + (size_t)(32 * p + perm_row(t, li)) * FF + (k - 1) * HC + g * 8);
    }
    *reinterpret_cast<bf16x8*>(blk + f * FRAG_ELEMS + l * 8) = v;
  }
}

// Optional LayerNorm tail of the block, done on the accumulators before anything leaves the chip (a block owns whole rows):
//   mode 0: Out = z (the pre-norm sum) only;
//   mode 1: X2 = LN_a(z)                      -- norm2 of the last block;
//   mode 2: X2 = LN_a(z), Hn = LN_b(X2)       -- norm2 and the NEXT block's norm1 (chada_vit.py:96,100): replaces the separate
//           two-LayerNorm pass (one read of z, two writes) and, in the no-grad passes, the write of z itself.
// As in the stand-alone kernels the statistics are taken over the bf16-ROUNDED z / X2 (what a separate pass would read).
struct FfnLnTail {
  int mode;
  const float* ga; const float* ba; const float* gb; const float* bb;
  float eps_a, eps_b;
  bf16_t* X2; bf16_t* Hn;
  float* mean_a; float* rstd_a; float* mean_b; float* rstd_b;
};

// Optional PROLOGUE (PRO): the block's attention-output projection, residual and norm1 in front of the FFN, on the same rows:
//   y = x + a Wo^T + bo ;  x1 = LN1(y)   (chada_vit.py:96-99: x = norm1(x + self_attn(...)) in the post-norm layer)
// Wo streams through the same LDS ring as three extra packed blocks IN FRONT of the FFN blocks (block j = k-steps 2j, 2j+1 of
// all 12 output tiles, rows permuted like W2's), the a-fragments of the wave's rows sit in the registers that hold the X
// fragments afterwards, the 32 x 192 accumulators are the FFN's output accumulators.  With W2's row permutation a lane ends
// up with 8 consecutive columns of its row per 32-wide pair -- exactly GEMM1's B-operand k-slots: x1 goes from the
// LayerNorm straight into the X fragments.  y and x1 are also written out (backward / residual); what disappears is the
// out-proj GEMM launch, the norm1 launch, one read of y and one read of x1 per layer and pass.
struct FfnPro {
  const bf16_t* A; int lda;        // attention output (heads concatenated), [M, D]
  const bf16_t* Xres; int ldx;     // the block's input x (residual), [M, D]
  const float* bo; const float* g1; const float* be1; float eps1;
  bf16_t* Y; int ldy;              // x + a Wo^T + bo (bf16), optional (needed by norm1's backward)
  bf16_t* X1; int ldx1;            // norm1(y): FFN input and residual (both taken from registers); optional output
  float* mean1; float* rstd1;      // optional
  // optional POSTLOGUE: the NEXT block's QKV projection of hn = norm1_next(x2), qkv = hn Wqkv^T + bqkv ([M, 3 D]); its weight
  // follows the FFN blocks in the packed stream as 3 NPB more blocks (three [D x D] row slices packed like Wo), at block qkv_at
  bf16_t* QKV; int ldqkv; const float* bqkv; int qkv_at;
};

// eight bf16 values -> bit j = value j != 0 (v_pk_min_u16 by hand: hipcc expands the vector minimum into compare + select pairs)
__device__ __forceinline__ unsigned relu_mask8(u32x4 x) {
  const unsigned one2 = 0x00010001u;
  unsigned m = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned f;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(x[i] & 0x7FFF7FFFu), "v"(one2));  // bit 0 / bit 16 = low / high half non-zero
    m |= f << (2 * i);
  }
  return (m | (m >> 15)) & 0xFFu;
}

// One pipeline step: issue the LDS-DMA of a packed block into `dst` (when `issue`), then GEMM1 of chunk k and GEMM2 of chunk
// k-1 out of the stage `st`, bias/ReLU/convert, and (WRITE_H) the hidden-slab traffic.  Everything that touches LDS between
// two barriers lives in THIS function, with __restrict__ pointers, on purpose: after inlining, the LDS reads carry
// scoped-noalias metadata against the DMA, which is what keeps the compiler's waitcnt insertion from draining the DMA queue
// (s_waitcnt vmcnt(0)) in front of the first LDS read after an LDS-DMA load -- it cannot tell ring slots (or even different
// __shared__ arrays) apart by itself, and would expose one full DMA latency per chunk.  Completion of the stage being read is
// established by the caller's explicit wait + barrier.
// MODE 0: forward.  MODE 1: forward that also reports the ReLU pattern of chunk k in `rbits` (bit rt * 8 + j = hidden value j of
// the lane's row in row tile rt is > 0; j = the lane's B-operand k-slot of GEMM2).  MODE 2: the BACKWARD dX pass on the same
// machinery -- X fragments = dz, "W1" records = W2^T chunk (GEMM1 gives dH = dz W2 for the chunk, no bias), the ReLU is replaced
// by the mask `rbits` recorded by the forward (dpre = dH where the forward's hidden value was > 0), "W2" records = W1^T chunk
// (GEMM2 accumulates dx1 += dpre W1).
__device__ __forceinline__ void ffn_issue(BufRsrc wrs, unsigned blk_bytes, bf16_t* __restrict__ dst, int w, int l) {
#pragma unroll
  for (int i = 0; i < BLK_FRAGS / NWV; ++i) {   // record f of a block goes to wave f % NWV
    const int f = w + NWV * i;
    lds_dma16(wrs, dst + f * FRAG_ELEMS, l * 16, blk_bytes + f * (FRAG_ELEMS * 2));
  }
}
#ifndef CHADA_FFN_BFE_SELECT
#define CHADA_FFN_BFE_SELECT 1
#endif
__device__ __forceinline__ float relu_select(float x, unsigned bits, int idx) {
#if CHADA_FFN_BFE_SELECT
  const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)bits, idx, 1);   // 0 or 0xffffffff
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & m);
#else
  return ((bits >> idx) & 1u) ? x : 0.f;
#endif
}
template <int RT, bool WRITE_H, bool DO_G1, bool DO_G2, bool GATHER, bool DO_P = false, int MODE = 0>
__device__ __forceinline__ void ffn_core(BufRsrc wrs, unsigned blk_bytes, bf16_t* __restrict__ dst,
                                         const bf16_t* __restrict__ st, const float* __restrict__ sb1,
                                         bf16_t* __restrict__ sh, bool issue, int k, int w, int l,
                                         const bf16x8 (&xf)[RT][KS1], f32x4 (&oacc)[RT][NT2], bf16x8 (&hb)[RT],
                                         bf16x8 (&pend)[WRITE_H ? RT * 2 : 1], unsigned& rbits) {
  constexpr int HROW = 72;
  constexpr int NPEND = WRITE_H ? RT * 2 : 1;
  const int li = l & 15, g = l >> 4;
// The instances that write H / dpre (two LDS stages: the slab takes the third's room) deal the next block's six LDS-DMA pieces out over the
// iteration's steps instead of issuing them as one burst behind the barrier: all eight waves of a CU reach their burst together and queue at
// the CU's one vector-memory path, and these instances have their H stores in the same queue.  Training whole-block kernel 2 759 -> 2 712 us,
// backward dX 2 080 -> 2 061 us at 1 206 272 rows; cfg2 +0.2 %, cfg3 +0.5 % (three interleaved rounds: profiles/r05k_ffn_spread_dma.log).  On the
// three-stage no-grad instances the same change gave the step nothing back (the pieces land one iteration later there anyway): burst kept.
#ifndef CHADA_FFN_SPREAD_DMA
#define CHADA_FFN_SPREAD_DMA 1
#endif
  constexpr int PPW = BLK_FRAGS / NWV;   // LDS-DMA pieces per wave and block
  constexpr int NSTEP_ALL = DO_P ? BLK_FRAGS / 2 : ((DO_G1 ? KS1 : 0) + (DO_G2 ? NT2 / 2 : 0));
  constexpr bool SPREAD = CHADA_FFN_SPREAD_DMA && WRITE_H && NSTEP_ALL >= PPW;
  auto issue_piece = [&](int i) {  // record f of a block goes to wave f % NWV
    const int f = w + NWV * i;
    lds_dma16(wrs, dst + f * FRAG_ELEMS, l * 16, blk_bytes + f * (FRAG_ELEMS * 2));
  };
  auto issue_at = [&](int sidx) {  // the block's pieces dealt out over the steps instead of going out as one burst
    if constexpr (SPREAD) {
      if (issue) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
          if ((i * NSTEP_ALL) / PPW == sidx) issue_piece(i);
      }
    }
  };
  if constexpr (!SPREAD) {
    if (issue) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(i);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST: free of the alias edge, the scheduler would sink it below the math
  if constexpr (DO_P) {  // prologue block k: records (ksl, n) = k-step 2k + ksl of output tile n; two records per step
    bf16x8 pr[2][2];
    pr[0][0] = lds_read8(st + l * 8);
    pr[0][1] = lds_read8(st + FRAG_ELEMS + l * 8);
#pragma unroll
    for (int sidx = 0; sidx < BLK_FRAGS / 2; ++sidx) {
      issue_at(sidx);
      if (sidx + 1 < BLK_FRAGS / 2) {
        pr[(sidx + 1) & 1][0] = lds_read8(st + (2 * sidx + 2) * FRAG_ELEMS + l * 8);
        pr[(sidx + 1) & 1][1] = lds_read8(st + (2 * sidx + 3) * FRAG_ELEMS + l * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ksl = sidx / (NT2 / 2), n = 2 * (sidx % (NT2 / 2));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        // (k < NPB at run time: select the fragment with uniform branches so that xf stays in registers)
        bf16x8 a = xf[rt][ksl];
#pragma unroll
        for (int kk = 1; kk < NPB; ++kk)
          if (k == kk) a = xf[rt][2 * kk + ksl];
        oacc[rt][n] = mfma16(pr[sidx & 1][0], a, oacc[rt][n]);
        oacc[rt][n + 1] = mfma16(pr[sidx & 1][1], a, oacc[rt][n + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 hacc[RT][2];
  if constexpr (DO_G1) {
    if constexpr (MODE == 2) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { hacc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    } else {
      const f32x4 bia0 = *reinterpret_cast<const f32x4*>(sb1 + k * HC + 8 * g);
      const f32x4 bia1 = *reinterpret_cast<const f32x4*>(sb1 + k * HC + 8 * g + 4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { hacc[rt][0] = bia0; hacc[rt][1] = bia1; }
    }
  }
  // GEMM1 (KS1 steps: the two W1 fragments of a k-step) then GEMM2 (NT2 / 2 steps: two W2 fragments): every step is two
  // 1 KiB fragment reads and 2 RT MFMAs.  The reads run PD steps ahead of the MFMAs, pinned with sched_barrier -- left
  // alone hipcc issues each pair of reads right in front of its MFMAs and waits for them.
  constexpr int G1S = DO_G1 ? KS1 : 0, NSTEP = G1S + (DO_G2 ? NT2 / 2 : 0);
  auto rd_step = [&](int sidx, bf16x8 (&f)[2]) {
    const int f0 = (sidx < G1S) ? 2 * sidx : W2_FRAG0 + 2 * (sidx - G1S);
    f[0] = lds_read8(st + f0 * FRAG_ELEMS + l * 8);
    f[1] = lds_read8(st + (f0 + 1) * FRAG_ELEMS + l * 8);
  };
  constexpr int PD = 1;  // look-ahead in steps (2 measured the same: 390 vs 391 us)
  bf16x8 fr[PD + 1][2];
#pragma unroll
  for (int i = 0; i < PD; ++i)
    if (i < NSTEP) rd_step(i, fr[i]);
#pragma unroll
  for (int sidx = 0; sidx < NSTEP; ++sidx) {
    issue_at(sidx);
    if (sidx + PD < NSTEP) rd_step(sidx + PD, fr[(sidx + PD) % (PD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
    if (sidx < G1S) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        hacc[rt][0] = mfma16(fr[sidx % (PD + 1)][0], xf[rt][sidx < G1S ? sidx : 0], hacc[rt][0]);
        hacc[rt][1] = mfma16(fr[sidx % (PD + 1)][1], xf[rt][sidx < G1S ? sidx : 0], hacc[rt][1]);
      }
    } else {
      const int n = 2 * (sidx - G1S);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        oacc[rt][n] = mfma16(fr[sidx % (PD + 1)][0], hb[rt], oacc[rt][n]);
        oacc[rt][n + 1] = mfma16(fr[sidx % (PD + 1)][1], hb[rt], oacc[rt][n + 1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (DO_G1) {
    if constexpr (MODE == 1) rbits = 0u;  // 8 RT bits per chunk
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (MODE == 2) {  // dpre = dH where the forward's hidden activation was positive
          // (the pattern bit as a 0 / -1 mask by one signed bit-field extract, then one AND: two vector instructions per value instead of the three --
          // and, compare, select -- hipcc makes of `bit ? x : 0`; same values)
          hb[rt][r] = (bf16_t)relu_select(hacc[rt][0][r], rbits, rt * 8 + r);
          hb[rt][4 + r] = (bf16_t)relu_select(hacc[rt][1][r], rbits, rt * 8 + 4 + r);
        } else {
          hb[rt][r] = (bf16_t)relu_f(hacc[rt][0][r]);
          hb[rt][4 + r] = (bf16_t)relu_f(hacc[rt][1][r]);
        }
      }
      // "> 0" on the bf16 values the backward would otherwise read back from H: they left a ReLU, so "!= 0" (sign masked off for a
      // -0) -- a packed 16-bit minimum against 1 per dword, the four dwords merged two bits apart and folded once: 15 VALU per eight
      // values instead of 4 per value (convert, compare, select, shift-or)
      if constexpr (MODE == 1) rbits |= relu_mask8(__builtin_bit_cast(u32x4, hb[rt])) << (rt * 8);
      if constexpr (WRITE_H) *reinterpret_cast<bf16x8*>(sh + (w * 16 * RT + rt * 16 + li) * HROW + (k & 1) * HC + g * 8) = hb[rt];
    }
  }
  if constexpr (GATHER) {
#pragma unroll
    for (int i = 0; i < NPEND; ++i) {
      const int id = l + 64 * i;
      pend[i] = *reinterpret_cast<const bf16x8*>(sh + (w * 16 * RT + (id >> 3)) * HROW + (id & 7) * 8);
    }
  }
}

// RB (MODE 1 writes, MODE 2 reads): the ReLU pattern of the hidden activation, 1 bit per (row, hidden unit), in the kernel's own
// lane order -- record (32-row tile t = 4 * block + wave, chunk group q = k / 8) is 64 lanes x 16 bytes: dword (k % 8) / 2 of lane
// l holds the 16 bits of chunk k (low half: even k) described at ffn_core.  M/32 tiles x FF/256 groups x 1 KiB = M * FF / 8 bytes,
// written as whole 1 KiB records; the backward dX instance reads it with the same lane mapping, nothing else looks inside.
template <int RT, bool WRITE_H, bool PRO = false, int MODE = 0>
#ifndef FFN_MIN_WAVES
#define FFN_MIN_WAVES (RT * NWV <= 8 ? 2 : 1)
#endif
__global__ __launch_bounds__(64 * NWV, FFN_MIN_WAVES) void ffn_fwd_kernel(const bf16_t* __restrict__ X, int ldx,
                                                                         const bf16_t* __restrict__ packed,
                                                                         const float* __restrict__ b1,
                                                                         const float* __restrict__ b2,
                                                                         const bf16_t* __restrict__ resid, int ldr,
                                                                         bf16_t* __restrict__ Out, int ldo,
                                                                         bf16_t* __restrict__ H, int ldh, int M, int FF, FfnLnTail ln,
                                                                         FfnPro pro, unsigned* __restrict__ RB) {
  static_assert(MODE == 0 || RT <= 2, "the ReLU-bit record layout is defined for 16 or 32 rows per wave");
  static_assert(!(MODE == 2 && PRO), "the backward instance has no prologue");
  static_assert(!PRO || RT == DRT, "the prologue / postlogue run on the build's default row tiling");
  constexpr int STAGE = BLK_FRAGS * FRAG_ELEMS;  // bf16 elements per stage (25 KiB)
  // forward-only instance: THREE stages, the LDS-DMA of block k+2 is issued while block k is consumed and the wait at a
  // barrier is counted (vmcnt(6): only the six DMA instructions of the newest block may still be in flight -- loads retire
  // in order); 3 x 24 KiB + 8 KiB of bias = 80 KiB, two blocks fill the CU's 160 KiB exactly.  With H the slab needs the room.
  constexpr int NST = WRITE_H ? 2 : 3;
  // ONE __shared__ object, carved by hand: with several, the LDS lowering tags every access with per-variable alias scopes
  // that replace the finer ones ffn_core's __restrict__ pointers produce (see there)
  constexpr int SH_ELEMS = WRITE_H ? NWV * 16 * RT * 72 : 0;
  __shared__ __attribute__((aligned(16))) bf16_t smem[NST * STAGE + 2 * MAX_FF + SH_ELEMS];
  float* const sB1 = reinterpret_cast<float*>(smem + NST * STAGE);
  // WRITE_H: per-wave slab where two consecutive hidden chunks (64 units = 128 B per row) are gathered before they are
  // written out as full 128-byte row segments (8 rows per store instruction); row stride 144 B keeps the b128 writes
  // conflict-free
  // (72 bf16 elements per staged row = 64 + 8 pad: HROW in ffn_core, SH_ELEMS above)
  constexpr int NPEND = WRITE_H ? RT * 2 : 1;    // 16-byte pieces per lane per chunk pair
  bf16_t* const sH = smem + NST * STAGE + 2 * MAX_FF;
  const int tid = threadIdx.x, l = tid & 63, li = l & 15, g = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = FF / HC;
  const int m0 = blockIdx.x * (NWV * 16 * RT) + w * (16 * RT);

  bf16x8 hb[RT];
  bf16x8 pend[NPEND];
  // ReLU-bit plumbing (see RB above): four named dwords, selected with uniform branches (a runtime-indexed vector would go to
  // scratch)
  // a chunk has 8 RT bits per lane: CPD chunks share a dword, a record of 8 chunks is NDW dwords per lane (RT = 2: 4 x 16 bytes = the
  // 1 KiB record of the header comment; RT = 1: 2 dwords, 512-byte records -- M * FF / 8 bytes either way)
  constexpr int BPC = 8 * RT, CPD = 32 / BPC, NDW = 8 / CPD;
  static_assert(MODE == 0 || NDW == 4 || NDW == 2, "record = 2 or 4 dwords per lane");
  unsigned rbits = 0u, mw0 = 0u, mw1 = 0u, mw2 = 0u, mw3 = 0u;
  unsigned* const rb_rec = (MODE != 0 && RB != nullptr)
                               ? RB + (((size_t)(blockIdx.x * NWV + w) * (size_t)(FF / (8 * HC))) * 64 + l) * NDW
                               : nullptr;
  auto rec_load = [&](int rec, unsigned (&d)[4]) {
    if constexpr (NDW == 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(rb_rec + (size_t)rec * 64 * NDW);
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    } else {
      const u32x2 v = *reinterpret_cast<const u32x2*>(rb_rec + (size_t)rec * 64 * NDW);
      d[0] = v[0]; d[1] = v[1]; d[2] = 0u; d[3] = 0u;
    }
  };
  unsigned mnext[4] = {0u, 0u, 0u, 0u};
  if constexpr (MODE == 2) rec_load(0, mnext);
  auto mask_pre = [&](int k) {   // MODE 2: the bits of chunk k, in front of its GEMM1; the next record is requested 8 chunks ahead
    if constexpr (MODE == 2) {
      if ((k & 7) == 0) {
        mw0 = mnext[0]; mw1 = mnext[1]; mw2 = mnext[2]; mw3 = mnext[3];
        if (k + 8 < FF / HC) rec_load((k >> 3) + 1, mnext);
      }
      const int q = (k & 7) / CPD;
      const unsigned v = q == 0 ? mw0 : q == 1 ? mw1 : q == 2 ? mw2 : mw3;
      rbits = (v >> ((k % CPD) * BPC)) & ((1u << BPC) - 1u);
    }
  };
  auto mask_post = [&](int k) {  // MODE 1: file the bits GEMM1 of chunk k just produced; a full record leaves every 8 chunks
    if constexpr (MODE == 1) {
      const int q = (k & 7) / CPD, sh = (k % CPD) * BPC;
      const unsigned add = rbits << sh;
      if (q == 0) mw0 = sh ? (mw0 | add) : add;
      else if (q == 1) mw1 = sh ? (mw1 | add) : add;
      else if (q == 2) mw2 = sh ? (mw2 | add) : add;
      else mw3 = sh ? (mw3 | add) : add;
      if ((k & 7) == 7 && rb_rec != nullptr) {
        if constexpr (NDW == 4) __builtin_nontemporal_store(u32x4{mw0, mw1, mw2, mw3}, reinterpret_cast<u32x4*>(rb_rec + (size_t)(k >> 3) * 64 * NDW));
        else __builtin_nontemporal_store(u32x2{mw0, mw1}, reinterpret_cast<u32x2*>(rb_rec + (size_t)(k >> 3) * 64 * NDW));
      }
    }
  };
  // (no LDS read follows the first block's DMA before the barrier: issued bare)
  const BufRsrc wrs = make_rsrc(packed);
  constexpr int J3 = PRO ? NPB : 0;   // stream blocks in front of FFN block 0 (NPB is a multiple of 3: the three-stage ring lines up)
  static_assert(NPB % 3 == 0, "ring phase of the FFN blocks behind the projection blocks");
  constexpr int LA = NST - 1;       // DMA look-ahead in blocks
#pragma unroll
  for (int i = 0; i < BLK_FRAGS / NWV; ++i) lds_dma16(wrs, smem + (w + NWV * i) * FRAG_ELEMS, l * 16, (w + NWV * i) * (FRAG_ELEMS * 2));
  if constexpr (MODE != 2)
    for (int i = tid; i < FF / 4; i += 64 * NWV) reinterpret_cast<f32x4*>(sB1)[i] = reinterpret_cast<const f32x4*>(b1)[i];
  // Small per-column vectors are read from LDS wherever their use sits between STORES: loads and stores share the one vmcnt
  // counter, so "load bias, wait, add, store, load the next bias, wait ..." makes every wait sit out the store before it -- a
  // store round trip (microseconds while the write queues are busy) per 16-byte piece, ~70 of them per 128-row block.  bo / gamma1 /
  // beta1 of the prologue's LayerNorm go into the H slab (idle until the FFN loop), b2 behind the tail's gamma / beta in the bias
  // slab (b1 is dead by then; no DMA is in flight there, so hipcc adds no vmcnt guard to those LDS reads).
  constexpr bool PRO_LDS = PRO && WRITE_H;
  if constexpr (PRO_LDS) {
    float* const sP = reinterpret_cast<float*>(sH);
    for (int i = tid; i < FD; i += 64 * NWV) { sP[i] = pro.bo[i]; sP[FD + i] = pro.g1[i]; sP[2 * FD + i] = pro.be1[i]; }
  }

  bf16x8 xf[RT][KS1];
  int mrow[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    mrow[rt] = min(m0 + rt * 16 + li, M - 1);  // rows past M duplicate row M-1 (identical values, benign duplicate stores)
    const bf16_t* src = PRO ? pro.A + (size_t)mrow[rt] * pro.lda : X + (size_t)mrow[rt] * ldx;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) xf[rt][ks] = *reinterpret_cast<const bf16x8*>(src + ks * 32 + g * 8);
  }

  f32x4 oacc[RT][NT2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int n = 0; n < NT2; ++n) oacc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // H rows leave the chip one chunk PAIR late: gathered from the slab into `pend` after the odd chunk, stored right after
  // the next barrier, so the stores drain under a whole iteration of MFMA work
  auto store_h = [&](int kpair) {  // kpair = first chunk of the pair
#pragma unroll
    for (int i = 0; i < NPEND; ++i) {
      const int id = l + 64 * i;
      const int m = min(m0 + (id >> 3), M - 1);
      __builtin_nontemporal_store(pend[i], reinterpret_cast<bf16x8*>(H + (size_t)m * ldh + kpair * HC + (id & 7) * 8));  // streamed: read again only in the backward
    }
  };

  // block k of the packed stream carries W1 of chunk k and W2 of chunk k-1: iteration k runs GEMM1(k) beside GEMM2(k-1)
  auto blk = [&](int k) { return (unsigned)(k + J3) * (STAGE * 2); };  // byte offset of FFN block k in the packed stream
  // the LDS bases handed to ffn_core go through an opaque zero: were they compile-time constants, interprocedural constant
  // propagation would substitute them INSIDE ffn_core before it is inlined and the accesses would lose their noalias scopes
  int opq = 0;
  asm volatile("" : "+s"(opq));
  bf16_t* const smem_o = smem + opq;
  float* const sB1_o = sB1 + opq;
  bf16_t* const sH_o = sH + opq;
  const float* const sP_o = reinterpret_cast<const float*>(sH_o);
  if constexpr (PRO) {
    // stream blocks 0..2 = Wo; the ring continues into the FFN blocks (stream block 3 + k), so the steps below already issue
    // FFN block 0 (and 1 with three stages)
#pragma unroll
    for (int j = 1; j < LA; ++j)
      ffn_core<RT, WRITE_H, false, false, false, false, MODE>(wrs, (unsigned)j * (STAGE * 2), smem_o + (j % NST) * STAGE, smem_o, sB1_o, sH_o, true, 0,
                                                 w, l, xf, oacc, hb, pend, rbits);
    bf16x8 rv[RT][NT2 / 2];  // the residual rows: requested before the last projection step, they land under its MFMAs
    for (int j = 0; j < NPB; ++j) {
      if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (j == NPB - 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int p = 0; p < NT2 / 2; ++p)
            rv[rt][p] = *reinterpret_cast<const bf16x8*>(pro.Xres + (size_t)mrow[rt] * pro.ldx + 32 * p + 8 * g);
      }
      ffn_core<RT, WRITE_H, false, false, false, true, MODE>(wrs, (unsigned)(j + LA) * (STAGE * 2), smem_o + ((j + LA) % NST) * STAGE,
                                                       smem_o + (j % NST) * STAGE, sB1_o, sH_o, true, j, w, l, xf, oacc, hb, pend, rbits);
    }
    // y = acc + bo + x ;  x1 = LN1(y) -> the X fragments (same arithmetic as the LayerNorm tail below / the stand-alone kernel)
    constexpr float invDp = 1.0f / FD;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int m = m0 + rt * 16 + li;
      const bool live = m < M;
      float sum = 0.f;
#pragma unroll
      for (int p = 0; p < NT2 / 2; ++p) {
        const int col = 32 * p + 8 * g;
        f32x4 v0, v1;
        if constexpr (PRO_LDS) {
          v0 = oacc[rt][2 * p] + *reinterpret_cast<const f32x4*>(sP_o + col);
          v1 = oacc[rt][2 * p + 1] + *reinterpret_cast<const f32x4*>(sP_o + col + 4);
        } else {
          v0 = oacc[rt][2 * p] + *reinterpret_cast<const f32x4*>(pro.bo + col);
          v1 = oacc[rt][2 * p + 1] + *reinterpret_cast<const f32x4*>(pro.bo + col + 4);
        }
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (bf16_t)(v0[r] + (float)rv[rt][p][r]);
          o[4 + r] = (bf16_t)(v1[r] + (float)rv[rt][p][4 + r]);
        }
        if (pro.Y && live) *reinterpret_cast<bf16x8*>(pro.Y + (size_t)m * pro.ldy + col) = o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          oacc[rt][2 * p][r] = (float)o[r];
          oacc[rt][2 * p + 1][r] = (float)o[4 + r];
          sum += (float)o[r] + (float)o[4 + r];
        }
      }
      const float mean1 = rows_sum(sum) * invDp;
      float q = 0.f;
#pragma unroll
      for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = oacc[rt][n][r] - mean1; q = __builtin_fmaf(d, d, q); }
      const float r1 = rsqrtf(rows_sum(q) * invDp + pro.eps1);
#pragma unroll
      for (int p = 0; p < NT2 / 2; ++p) {
        const int col = 32 * p + 8 * g;
        f32x4 g0, g1v, e0, e1;
        if constexpr (PRO_LDS) {
          g0 = *reinterpret_cast<const f32x4*>(sP_o + FD + col); g1v = *reinterpret_cast<const f32x4*>(sP_o + FD + col + 4);
          e0 = *reinterpret_cast<const f32x4*>(sP_o + 2 * FD + col); e1 = *reinterpret_cast<const f32x4*>(sP_o + 2 * FD + col + 4);
        } else {
          g0 = *reinterpret_cast<const f32x4*>(pro.g1 + col); g1v = *reinterpret_cast<const f32x4*>(pro.g1 + col + 4);
          e0 = *reinterpret_cast<const f32x4*>(pro.be1 + col); e1 = *reinterpret_cast<const f32x4*>(pro.be1 + col + 4);
        }
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (bf16_t)__builtin_fmaf((oacc[rt][2 * p][r] - mean1) * r1, g0[r], e0[r]);
          o[4 + r] = (bf16_t)__builtin_fmaf((oacc[rt][2 * p + 1][r] - mean1) * r1, g1v[r], e1[r]);
        }
        if (pro.X1 && live) *reinterpret_cast<bf16x8*>(pro.X1 + (size_t)m * pro.ldx1 + col) = o;
        xf[rt][p] = o;  // pair p = GEMM1's k-step p: 8 consecutive columns of the lane's row
      }
      if (g == 0 && live && pro.mean1) { pro.mean1[m] = mean1; pro.rstd1[m] = r1; }
#pragma unroll
      for (int n = 0; n < NT2; ++n) oacc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if constexpr (!WRITE_H) {
    // block 1 after the X / bias loads above: they are older than it in the in-order VMEM queue (PRO: already issued)
    if constexpr (!PRO)
      ffn_core<RT, false, false, false, false, false, MODE>(wrs, blk(1), smem_o + STAGE, smem_o, sB1_o, sH_o, true, 0, w, l, xf, oacc, hb, pend, rbits);
    {
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      mask_pre(0);
      ffn_core<RT, false, true, false, false, false, MODE>(wrs, blk(2), smem_o + 2 * STAGE, smem_o, sB1_o, sH_o, 2 <= NC, 0, w, l, xf, oacc, hb, pend, rbits);
      mask_post(0);
    }
    for (int k = 1; k < NC; ++k) {
      // block k has landed (LDS-DMA completion is visible only through the issuing wave's vmcnt), block k+1 may still be
      // in flight; everyone is done reading the stage block k+2 goes into (it held block k-1)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      mask_pre(k);
      ffn_core<RT, false, true, true, false, false, MODE>(wrs, blk(k + 2), smem_o + ((k + 2) % 3) * STAGE, smem_o + (k % 3) * STAGE, sB1_o, sH_o,
                                             k + 2 <= NC, k, w, l, xf, oacc, hb, pend, rbits);
      mask_post(k);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ffn_core<RT, false, false, true, false, false, MODE>(wrs, 0u, smem_o, smem_o + (NC % 3) * STAGE, sB1_o, sH_o, false, NC, w, l, xf, oacc, hb, pend, rbits);
  } else {
    {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // block 0 has landed, b1 is staged
      mask_pre(0);
      ffn_core<RT, true, true, false, false, false, MODE>(wrs, blk(1), smem_o + ((1 + J3) & 1) * STAGE, smem_o + (J3 & 1) * STAGE, sB1_o, sH_o, true,
                                             0, w, l, xf, oacc, hb, pend, rbits);
      mask_post(0);
    }
    // NC is even: iterations come in (odd, even) pairs
    for (int k = 1; k < NC; k += 2) {
      {  // odd k
        // (vmcnt(0), not "all but the H stores of the even iteration before": tried -- stores issued behind the DMA, vmcnt(NPEND)
        // here -- and WRONG: a store can retire before an older load, so a count that relies on younger stores still being in
        // flight lets the barrier pass with DMA pieces missing; test_ffn_bwd_dx_from_relu_bits caught it: 0.16 % of dx1 off)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        mask_pre(k);
        ffn_core<RT, true, true, true, true, false, MODE>(wrs, blk(k + 1), smem_o + ((k + 1 + J3) & 1) * STAGE, smem_o + ((k + J3) & 1) * STAGE, sB1_o,
                                             sH_o, true, k, w, l, xf, oacc, hb, pend, rbits);
        mask_post(k);
      }
      if (k + 1 < NC) {  // even k + 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        store_h(k - 1);
        mask_pre(k + 1);
        ffn_core<RT, true, true, true, false, false, MODE>(wrs, blk(k + 2), smem_o + ((k + J3) & 1) * STAGE, smem_o + ((k + 1 + J3) & 1) * STAGE, sB1_o,
                                              sH_o, true, k + 1, w, l, xf, oacc, hb, pend, rbits);
        mask_post(k + 1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    store_h(NC - 2);
    ffn_core<RT, true, false, true, false, false, MODE>(wrs, 0u, smem_o, smem_o + ((NC + J3) & 1) * STAGE, sB1_o, sH_o, false, NC, w, l, xf, oacc, hb, pend, rbits);
  }

  // ---- epilogue: + b2 + residual (+ LayerNorm tail), 16-byte stores straight from the accumulators
  constexpr int NP = NT2 / 2;
  constexpr float invD = 1.0f / FD;
  float* const sLn = sB1;  // the bias slab is dead now: gamma / beta of the tail are staged there ([ga | ba | gb | bb], FD each), then b2
  {
    __syncthreads();
    for (int i = tid; i < FD; i += 64 * NWV) {
      if (ln.mode) {
        sLn[i] = ln.ga[i];
        sLn[FD + i] = ln.ba[i];
        if (ln.mode == 2) { sLn[2 * FD + i] = ln.gb[i]; sLn[3 * FD + i] = ln.bb[i]; }
      }
      if constexpr (MODE != 2) sLn[4 * FD + i] = b2[i];
    }
    __syncthreads();
  }
  const bool resid_is_x = !PRO && resid == X && ldr == ldx;   // (uniform)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int m = m0 + rt * 16 + li;
    const bool live = m < M;
    const int mr = min(m, M - 1);
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = 32 * p + 8 * g;
      f32x4 v0 = oacc[rt][2 * p], v1 = oacc[rt][2 * p + 1];
      if constexpr (MODE != 2) {
        v0 += *reinterpret_cast<const f32x4*>(sLn + 4 * FD + col);
        v1 += *reinterpret_cast<const f32x4*>(sLn + 4 * FD + col + 4);
      }
      if (PRO || resid_is_x) {  // the residual is the X fragment of this pair: same rows, same 8 columns, already in registers
#pragma unroll                   // (PRO: x1; otherwise whenever the caller passes resid == X -- a load here would sit between the stores
        for (int r = 0; r < 4; ++r) { v0[r] += (float)xf[rt][p][r]; v1[r] += (float)xf[rt][p][4 + r]; }   // and drain them one by one)
      } else if (resid) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(resid + (size_t)mr * ldr + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v0[r] += (float)rv[r]; v1[r] += (float)rv[4 + r]; }
      }
      bf16x8 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)v0[r]; o[4 + r] = (bf16_t)v1[r]; }
      if (Out && live) *reinterpret_cast<bf16x8*>(Out + (size_t)m * ldo + col) = o;
      if (ln.mode) {  // keep the rounded z in the accumulator registers for the normalisation
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          oacc[rt][2 * p][r] = (float)o[r];
          oacc[rt][2 * p + 1][r] = (float)o[4 + r];
          sum += (float)o[r] + (float)o[4 + r];
        }
      }
    }
    if (ln.mode) {  // wave-uniform
      const float mean1 = rows_sum(sum) * invD;
      float q = 0.f;
#pragma unroll
      for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = oacc[rt][n][r] - mean1; q = __builtin_fmaf(d, d, q); }
      const float r1 = rsqrtf(rows_sum(q) * invD + ln.eps_a);
      float sum2 = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = 32 * p + 8 * g;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sLn + col), g1 = *reinterpret_cast<const f32x4*>(sLn + col + 4);
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(sLn + FD + col), e1 = *reinterpret_cast<const f32x4*>(sLn + FD + col + 4);
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (bf16_t)__builtin_fmaf((oacc[rt][2 * p][r] - mean1) * r1, g0[r], e0[r]);
          o[4 + r] = (bf16_t)__builtin_fmaf((oacc[rt][2 * p + 1][r] - mean1) * r1, g1[r], e1[r]);
        }
        if (live) *reinterpret_cast<bf16x8*>(ln.X2 + (size_t)m * FD + col) = o;
        if (ln.mode == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            oacc[rt][2 * p][r] = (float)o[r];
            oacc[rt][2 * p + 1][r] = (float)o[4 + r];
            sum2 += (float)o[r] + (float)o[4 + r];
          }
        }
      }
      if (g == 0 && live && ln.mean_a) { ln.mean_a[m] = mean1; ln.rstd_a[m] = r1; }
      if (ln.mode == 2) {
        const float mean2 = rows_sum(sum2) * invD;
        float q2 = 0.f;
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = oacc[rt][n][r] - mean2; q2 = __builtin_fmaf(d, d, q2); }
        const float r2 = rsqrtf(rows_sum(q2) * invD + ln.eps_b);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int col = 32 * p + 8 * g;
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(sLn + 2 * FD + col), g1 = *reinterpret_cast<const f32x4*>(sLn + 2 * FD + col + 4);
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(sLn + 3 * FD + col), e1 = *reinterpret_cast<const f32x4*>(sLn + 3 * FD + col + 4);
          bf16x8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (bf16_t)__builtin_fmaf((oacc[rt][2 * p][r] - mean2) * r2, g0[r], e0[r]);
            o[4 + r] = (bf16_t)__builtin_fmaf((oacc[rt][2 * p + 1][r] - mean2) * r2, g1[r], e1[r]);
          }
          if (ln.Hn && live) *reinterpret_cast<bf16x8*>(ln.Hn + (size_t)m * FD + col) = o;
          if constexpr (PRO) xf[rt][p] = o;  // hn, in B-operand layout, for the postlogue (the X fragments are dead by now)
        }
        if (g == 0 && live && ln.mean_b) { ln.mean_b[m] = mean2; ln.rstd_b[m] = r2; }
      }
    }
  }
  if constexpr (PRO) {
    if (pro.QKV != nullptr && ln.mode == 2) {  // (uniform) the next block's QKV projection of hn, three [M x D] column slices
      // the ring is free (every wave has passed the barrier in front of the LayerNorm tail): 3 NPB stream blocks, same steps as
      // the prologue with hn in the X-fragment registers
      auto qblk = [&](int q) { return (unsigned)(pro.qkv_at + q) * (STAGE * 2); };
#pragma unroll
      for (int q = 0; q < LA; ++q)
        ffn_core<RT, WRITE_H, false, false, false, false, MODE>(wrs, qblk(q), smem_o + (q % NST) * STAGE, smem_o, sB1_o, sH_o, true, 0, w, l, xf, oacc, hb,
                                                   pend, rbits);
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int n = 0; n < NT2; ++n) oacc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < NPB; ++j) {
          const int q = NPB * c + j;
          // block q has landed; block q + 1 (six pieces per wave) may still be in flight -- IF the step before this one issued it.
          // The last step's predecessor issues nothing (q + LA - 1 = 3 NPB): the six pieces the counted wait leaves in flight
          // would then be block q's own.  (Until round 3's last day the wait was vmcnt(6) throughout: the V third of the teacher's
          // and the local-crop pass's QKV was computed from a weight block still landing, in a few % of the rows at bench size --
          // found by tests/test_kernels_gpu.py::test_hot_kernels_are_deterministic.)
          if (NST == 3 && q + LA - 1 < 3 * NPB) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          ffn_core<RT, WRITE_H, false, false, false, true, MODE>(wrs, qblk(q + LA), smem_o + ((q + LA) % NST) * STAGE, smem_o + (q % NST) * STAGE,
                                                           sB1_o, sH_o, q + LA < 3 * NPB, j, w, l, xf, oacc, hb, pend, rbits);
        }
        // the slice's bias: ALL of it requested here, before the slice's first store.  Fetched inside the store loop, every piece's
        // wait sat out the store in front of it (one vmcnt counter for loads and stores: 36 store round trips per block); read
        // from LDS, hipcc guards each read against the ring's DMA in flight with a vmcnt wait that does the same.  (Earlier than
        // here there are no registers for it.)
        f32x4 qb[NT2];
#pragma unroll
        for (int n = 0; n < NT2; ++n) qb[n] = *reinterpret_cast<const f32x4*>(pro.bqkv + FD * c + 32 * (n >> 1) + 8 * g + 4 * (n & 1));
#pragma unroll
        for (int n = 0; n < NT2; ++n) asm volatile("" ::"v"(qb[n]));   // (one wait, here)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int m = m0 + rt * 16 + li;
          if (m < M) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              const int col = FD * c + 32 * p + 8 * g;
              const f32x4 v0 = oacc[rt][2 * p] + qb[2 * p];
              const f32x4 v1 = oacc[rt][2 * p + 1] + qb[2 * p + 1];
              bf16x8 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)v0[r]; o[4 + r] = (bf16_t)v1[r]; }
              *reinterpret_cast<bf16x8*>(pro.QKV + (size_t)m * pro.ldqkv + col) = o;
            }
          }
        }
      }
    }
  }
}

}  // namespace

#if FFN_PRIMARY
// the D = 384 build of this file (ffn_fused_d384.hip)
extern "C" long long chada_int_ffn_packed_bytes_d384(int D, int FF);
extern "C" int chada_int_ffn_pack_d384(const chada_bf16* W1, const chada_bf16* W2, void* packed, int D, int FF, void* stream);
extern "C" int chada_int_ffn_pack_batched_d384(const chada_bf16* slab, void* packed, const long long* desc, int n_layers, int D, int FF,
                                               void* stream);
extern "C" int chada_int_ffn_fwd_d384(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2,
                                      const chada_bf16* resid, int ldr, chada_bf16* Out, int ldo, chada_bf16* H, int ldh, void* relu_bits,
                                      int M, int D, int FF, int rows_per_wave, void* stream);
extern "C" int chada_int_ffn_ln_fwd_d384(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2,
                                         const chada_bf16* resid, int ldr, chada_bf16* Z, int ldz, chada_bf16* H, int ldh,
                                         const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a, float* rstd_a,
                                         const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b, float* rstd_b,
                                         void* relu_bits, int M, int D, int FF, void* stream);
extern "C" int chada_int_ffn_bwd_dx_d384(const chada_bf16* dZ, int lddz, const void* packed_bwd, const void* relu_bits, chada_bf16* dX1,
                                         int lddx, chada_bf16* dPre, int lddp, int M, int D, int FF, void* stream);
#define FFN_DISPATCH_384(call) if (D == 384) return call
#else
#define FFN_DISPATCH_384(call)
#endif

extern "C" long long FFN_NAME(ffn_packed_bytes)(int D, int FF) {
  FFN_DISPATCH_384(chada_int_ffn_packed_bytes_d384(D, FF));
  if (D != FD || FF < 2 * HC || FF > MAX_FF || FF % (2 * HC) != 0) return -1;
  return (long long)(FF / HC + 1) * BLK_FRAGS * FRAG_ELEMS * 2;
}

extern "C" int FFN_NAME(ffn_pack)(const chada_bf16* W1, const chada_bf16* W2, void* packed, int D, int FF, void* stream) {
  FFN_DISPATCH_384(chada_int_ffn_pack_d384(W1, W2, packed, D, FF, stream));
  CHADA_ENTRY();
  if (!W1 || !W2 || !packed) return 1;
  if (D != FD || FF < 2 * HC || FF > MAX_FF || FF % (2 * HC) != 0) return 2;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(FF / HC + 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(W1), reinterpret_cast<const bf16_t*>(W2),
                     reinterpret_cast<bf16_t*>(packed), FF);
  CHADA_CHECK_LAUNCH();
  return 0;
}

// [NPB Wo blocks | FFN blocks | 3 NPB next-QKV blocks] per layer: desc[5 t ..] = {W1, W2, Wo, next in_proj weight (or -1) offsets
// (bf16 elements into the slab's bf16 shadow), packed offset}.  A [D x D] matrix (Wo, or row slice c of the next in_proj weight) is
// NPB blocks; block j, record r = ksl * NT2 + n: k-step 2j + ksl of output tile n, rows permuted like W2's.
namespace {
__global__ __launch_bounds__(256) void ffn_pack_proj_batched_kernel(const bf16_t* __restrict__ slab, bf16_t* __restrict__ packed,
                                                                    const long long* __restrict__ desc, int FF) {
  const long long* d = desc + 5 * blockIdx.y;
  const int NC = FF / HC;
  const int tid = threadIdx.x;
  const int xq = (int)blockIdx.x - (NC + 1 + NPB);  // >= 0: one of the 3 NPB next-QKV blocks
  if ((int)blockIdx.x < NPB || xq >= 0) {
    if (xq >= 0 && d[3] < 0) return;
    const bf16_t* Wo = xq >= 0 ? slab + d[3] + (size_t)(xq / NPB) * FD * FD : slab + d[2];
    const int j = xq >= 0 ? xq % NPB : blockIdx.x;
    bf16_t* blk = packed + d[4] + (size_t)blockIdx.x * BLK_FRAGS * FRAG_ELEMS;
    for (int id = tid; id < BLK_FRAGS * 64; id += 256) {
      const int f = id >> 6, l = id & 63, li = l & 15, g = l >> 4;
      const int ks = 2 * j + f / NT2, n = f % NT2;
      *reinterpret_cast<bf16x8*>(blk + f * FRAG_ELEMS + l * 8) =
          *reinterpret_cast<const bf16x8*>(Wo + (size_t)(32 * (n >> 1) + perm_row(n & 1, li)) * FD + ks * 32 + g * 8);
    }
    return;
  }
  const bf16_t* W1 = slab + d[0];
  const bf16_t* W2 = slab + d[1];
  const int k = blockIdx.x - NPB;  // 0..NC
  bf16_t* blk = packed + d[4] + (size_t)(k + NPB) * BLK_FRAGS * FRAG_ELEMS;
  for (int id = tid; id < BLK_FRAGS * 64; id += 256) {
    const int f = id >> 6, l = id & 63, li = l & 15, g = l >> 4;
    bf16x8 v;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = (bf16_t)0.f;
    if (f < W1_FRAGS) {
      if (k < NC) {
        const int ks = f >> 1, nt = f & 1;
        v = *reinterpret_cast<const bf16x8*>(W1 + (size_t)(k * HC + perm_row(nt, li)) * FD + ks * 32 + g * 8);
      }
    } else if (k > 0) {
      const int nt2 = f - W2_FRAG0, p = nt2 >> 1, t = nt2 & 1;
      v = *reinterpret_cast<const bf16x8*>(W2 + (size_t)(32 * p + perm_row(t, li)) * FF + (k - 1) * HC + g * 8);
    }
    *reinterpret_cast<bf16x8*>(blk + f * FRAG_ELEMS + l * 8) = v;
  }
}
}  // namespace

#if FFN_PRIMARY
extern "C" long long chada_int_ffn_proj_packed_bytes_d384(int D, int FF);
extern "C" int chada_int_ffn_pack_proj_batched_d384(const chada_bf16* slab, void* packed, const long long* desc, int n_layers, int D, int FF,
                                                    void* stream);
#endif

extern "C" long long FFN_NAME(ffn_proj_packed_bytes)(int D, int FF) {  // always with room for the 3 NPB next-QKV blocks
  FFN_DISPATCH_384(chada_int_ffn_proj_packed_bytes_d384(D, FF));
  if (D != FD || FF < 2 * HC || FF > MAX_FF || FF % (2 * HC) != 0) return -1;
  return (long long)(FF / HC + 1 + 4 * NPB) * BLK_FRAGS * FRAG_ELEMS * 2;
}

extern "C" int FFN_NAME(ffn_pack_proj_batched)(const chada_bf16* slab, void* packed, const long long* desc, int n_layers, int D, int FF,
                                               void* stream) {
  FFN_DISPATCH_384(chada_int_ffn_pack_proj_batched_d384(slab, packed, desc, n_layers, D, FF, stream));
  CHADA_ENTRY();
  if (!slab || !packed || !desc || n_layers <= 0) return 1;
  if (D != FD || FF < 2 * HC || FF > MAX_FF || FF % (2 * HC) != 0) return 2;
  hipLaunchKernelGGL(ffn_pack_proj_batched_kernel, dim3(FF / HC + 1 + 4 * NPB, n_layers), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(slab), reinterpret_cast<bf16_t*>(packed), desc, FF);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int FFN_NAME(ffn_pack_batched)(const chada_bf16* slab, void* packed, const long long* desc, int n_layers, int D, int FF,
                                          void* stream) {
  FFN_DISPATCH_384(chada_int_ffn_pack_batched_d384(slab, packed, desc, n_layers, D, FF, stream));
  CHADA_ENTRY();
  if (!slab || !packed || !desc || n_layers <= 0) return 1;
  if (D != FD || FF < 2 * HC || FF > MAX_FF || FF % (2 * HC) != 0) return 2;
  hipLaunchKernelGGL(ffn_pack_batched_kernel, dim3(FF / HC + 1, n_layers), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(slab), reinterpret_cast<bf16_t*>(packed), desc, FF);
  CHADA_CHECK_LAUNCH();
  return 0;
}

namespace {
int launch_ffn(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2, const chada_bf16* resid, int ldr,
               chada_bf16* Out, int ldo, chada_bf16* H, int ldh, int M, int D, int FF, int rows_per_wave, const FfnLnTail& ln,
               void* stream, const FfnPro* pro = nullptr, void* relu_bits = nullptr, bool bwd = false) {
  if ((!X && !pro) || !packed || (!bwd && (!b1 || !b2)) || M <= 0 || (!Out && !ln.mode)) return 1;
  if (bwd && (!relu_bits || pro || ln.mode)) return 1;
  if (D != FD || FF < 2 * HC || FF > MAX_FF || FF % (2 * HC) != 0 || ldx % 8 != 0 || (Out && ldo % 8 != 0) || (resid && ldr % 8 != 0) ||
      (H && ldh % 8 != 0))
    return 2;
  if (relu_bits && (FF % (8 * HC) != 0 || ((uintptr_t)relu_bits & 15) != 0)) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* x = reinterpret_cast<const bf16_t*>(X);
  const bf16_t* pk = reinterpret_cast<const bf16_t*>(packed);
  const bf16_t* rs = reinterpret_cast<const bf16_t*>(resid);
  bf16_t* o = reinterpret_cast<bf16_t*>(Out);
  bf16_t* h = reinterpret_cast<bf16_t*>(H);
  unsigned* rb = reinterpret_cast<unsigned*>(relu_bits);
#define FFN_LAUNCH(RT, WH, PR, MD)                                                                                             \
  hipLaunchKernelGGL((ffn_fwd_kernel<RT, WH, PR, MD>), dim3((M + NWV * 16 * RT - 1) / (NWV * 16 * RT)), dim3(64 * NWV), 0, s, x, ldx, pk, \
                     b1, b2, rs, ldr, o, ldo, h, ldh, M, FF, ln, (PR ? *pro : FfnPro{}), rb)
  if (bwd) {  // dX pass: X = dz, packed = [W2^T | W1^T] stream, H (optional) receives dpre
    if (h) FFN_LAUNCH(DRT, true, false, 2); else FFN_LAUNCH(DRT, false, false, 2);
  } else if (pro) {  // with the out-proj + norm1 prologue (the build's default row tiling only)
    if (rb) { if (h) FFN_LAUNCH(DRT, true, true, 1); else FFN_LAUNCH(DRT, false, true, 1); }
    else { if (h) FFN_LAUNCH(DRT, true, true, 0); else FFN_LAUNCH(DRT, false, true, 0); }
  } else if (rows_per_wave == 64) {
    return 2;   // (a 64-rows-per-wave instance existed until round 4: 173 spilled registers, no caller but its own test)
  } else if (rb) {
    if (h) FFN_LAUNCH(DRT, true, false, 1); else FFN_LAUNCH(DRT, false, false, 1);
  } else {
    if (h) FFN_LAUNCH(DRT, true, false, 0); else FFN_LAUNCH(DRT, false, false, 0);
  }
#undef FFN_LAUNCH
  CHADA_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int FFN_NAME(ffn_fwd)(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2,
                                 const chada_bf16* resid, int ldr, chada_bf16* Out, int ldo, chada_bf16* H, int ldh, void* relu_bits,
                                 int M, int D, int FF, int rows_per_wave, void* stream) {
  FFN_DISPATCH_384(chada_int_ffn_fwd_d384(X, ldx, packed, b1, b2, resid, ldr, Out, ldo, H, ldh, relu_bits, M, D, FF, rows_per_wave, stream));
  CHADA_ENTRY();
  FfnLnTail ln{};
  return launch_ffn(X, ldx, packed, b1, b2, resid, ldr, Out, ldo, H, ldh, M, D, FF, rows_per_wave, ln, stream, nullptr, relu_bits);
}

#if FFN_PRIMARY
extern "C" long long chadavit_relu_bits_bytes(int M, int FF) {  // (the same for both builds: 128-row blocks, 1 bit per element)
  if (M <= 0 || FF <= 0 || FF % (8 * HC) != 0) return -1;
  return (long long)((M + 127) / 128) * 4 * (FF / (8 * HC)) * 1024;
}
#endif

// dX pass of the FFN backward in one launch, nothing 2048-wide in HBM:  dX1 = dZ + ((dZ W2) * [H > 0]) W1.
// `packed_bwd` = chadavit_ffn_pack[_batched] applied to (W2^T as "W1", W1^T as "W2"), i.e. to the [FF x D] and [D x FF] transposed
// bf16 copies; `relu_bits` = what the forward recorded.  dPre (optional, [M x FF]) receives (dZ W2) * [H > 0] for a separate dW1 pass.
extern "C" int FFN_NAME(ffn_bwd_dx)(const chada_bf16* dZ, int lddz, const void* packed_bwd, const void* relu_bits, chada_bf16* dX1,
                                    int lddx, chada_bf16* dPre, int lddp, int M, int D, int FF, void* stream) {
  FFN_DISPATCH_384(chada_int_ffn_bwd_dx_d384(dZ, lddz, packed_bwd, relu_bits, dX1, lddx, dPre, lddp, M, D, FF, stream));
  CHADA_ENTRY();
  if (!dZ || !dX1 || !relu_bits) return 1;
  FfnLnTail ln{};
  return launch_ffn(dZ, lddz, packed_bwd, nullptr, nullptr, dZ, lddz, dX1, lddx, dPre, lddp, M, D, FF, 32, ln, stream, nullptr,
                    const_cast<void*>(relu_bits), true);
}

extern "C" int FFN_NAME(ffn_ln_fwd)(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2,
                                    const chada_bf16* resid, int ldr, chada_bf16* Z, int ldz, chada_bf16* H, int ldh,
                                    const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a, float* rstd_a,
                                    const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b, float* rstd_b,
                                    void* relu_bits, int M, int D, int FF, void* stream) {
  FFN_DISPATCH_384(chada_int_ffn_ln_fwd_d384(X, ldx, packed, b1, b2, resid, ldr, Z, ldz, H, ldh, gamma_a, beta_a, eps_a, X2, mean_a, rstd_a,
                                             gamma_b, beta_b, eps_b, Hn, mean_b, rstd_b, relu_bits, M, D, FF, stream));
  CHADA_ENTRY();
  if (!gamma_a || !beta_a || !X2 || (mean_a == nullptr) != (rstd_a == nullptr) || (mean_b == nullptr) != (rstd_b == nullptr)) return 1;
  if (Hn && (!gamma_b || !beta_b)) return 1;
  FfnLnTail ln{};
  ln.mode = Hn ? 2 : 1;
  ln.ga = gamma_a; ln.ba = beta_a; ln.gb = gamma_b; ln.bb = beta_b;
  ln.eps_a = eps_a; ln.eps_b = eps_b;
  ln.X2 = reinterpret_cast<bf16_t*>(X2); ln.Hn = reinterpret_cast<bf16_t*>(Hn);
  ln.mean_a = mean_a; ln.rstd_a = rstd_a; ln.mean_b = mean_b; ln.rstd_b = rstd_b;
  return launch_ffn(X, ldx, packed, b1, b2, resid, ldr, Z, ldz, H, ldh, M, D, FF, 32, ln, stream, nullptr, relu_bits);
}

#if FFN_PRIMARY
extern "C" int chada_int_block_fwd_d384(const chada_bf16* A, int lda, const chada_bf16* Xres, int ldxr, const void* packed, const float* bo,
                                        const float* gamma1, const float* beta1, float eps1, chada_bf16* Y, int ldy, chada_bf16* X1, int ldx1,
                                        float* mean1, float* rstd1, const float* b1, const float* b2, chada_bf16* Z, int ldz, chada_bf16* H,
                                        int ldh, const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a,
                                        float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b,
                                        float* rstd_b, chada_bf16* QKV, int ldqkv, const float* bqkv, void* relu_bits, int M, int D, int FF,
                                        void* stream);
// Out-proj + residual + norm1 + FFN + norm2 (+ next norm1) of one transformer block in ONE launch (see FfnPro): `packed` is the
// [Wo | FFN] stream of chadavit_ffn_pack_proj_batched.
extern "C" int chadavit_block_fwd(const chada_bf16* A, int lda, const chada_bf16* Xres, int ldxr, const void* packed, const float* bo,
                                  const float* gamma1, const float* beta1, float eps1, chada_bf16* Y, int ldy, chada_bf16* X1, int ldx1,
                                  float* mean1, float* rstd1, const float* b1, const float* b2, chada_bf16* Z, int ldz, chada_bf16* H,
                                  int ldh, const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a,
                                  float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b,
                                  float* rstd_b, chada_bf16* QKV, int ldqkv, const float* bqkv, void* relu_bits, int M, int D, int FF, void* stream);

extern "C" int chadavit_proj_ffn_ln_fwd(const chada_bf16* A, int lda, const chada_bf16* Xres, int ldxr, const void* packed, const float* bo,
                                        const float* gamma1, const float* beta1, float eps1, chada_bf16* Y, int ldy, chada_bf16* X1,
                                        int ldx1, float* mean1, float* rstd1, const float* b1, const float* b2, chada_bf16* Z, int ldz,
                                        chada_bf16* H, int ldh, const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2,
                                        float* mean_a, float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b,
                                        chada_bf16* Hn, float* mean_b, float* rstd_b, int M, int D, int FF, void* stream) {
  return chadavit_block_fwd(A, lda, Xres, ldxr, packed, bo, gamma1, beta1, eps1, Y, ldy, X1, ldx1, mean1, rstd1, b1, b2, Z, ldz, H, ldh,
                            gamma_a, beta_a, eps_a, X2, mean_a, rstd_a, gamma_b, beta_b, eps_b, Hn, mean_b, rstd_b, nullptr, 0, nullptr, nullptr,
                            M, D, FF, stream);
}

#endif  // FFN_PRIMARY

// ... plus, optionally, the NEXT block's QKV projection (QKV != NULL: needs gamma_b / beta_b; Hn itself becomes optional)
extern "C" int FFN_NAME(block_fwd)(const chada_bf16* A, int lda, const chada_bf16* Xres, int ldxr, const void* packed, const float* bo,
                                  const float* gamma1, const float* beta1, float eps1, chada_bf16* Y, int ldy, chada_bf16* X1, int ldx1,
                                  float* mean1, float* rstd1, const float* b1, const float* b2, chada_bf16* Z, int ldz, chada_bf16* H,
                                  int ldh, const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a,
                                  float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b,
                                  float* rstd_b, chada_bf16* QKV, int ldqkv, const float* bqkv, void* relu_bits, int M, int D, int FF, void* stream) {
  FFN_DISPATCH_384(chada_int_block_fwd_d384(A, lda, Xres, ldxr, packed, bo, gamma1, beta1, eps1, Y, ldy, X1, ldx1, mean1, rstd1, b1, b2, Z, ldz, H,
                                            ldh, gamma_a, beta_a, eps_a, X2, mean_a, rstd_a, gamma_b, beta_b, eps_b, Hn, mean_b, rstd_b, QKV, ldqkv,
                                            bqkv, relu_bits, M, D, FF, stream));
  CHADA_ENTRY();
  if (QKV && (!gamma_b || !beta_b || !bqkv || ldqkv % 8 != 0)) return 1;
  if (!A || !Xres || !bo || !gamma1 || !beta1 || (mean1 == nullptr) != (rstd1 == nullptr)) return 1;
  if (!gamma_a || !beta_a || !X2 || (mean_a == nullptr) != (rstd_a == nullptr) || (mean_b == nullptr) != (rstd_b == nullptr)) return 1;
  if (Hn && (!gamma_b || !beta_b)) return 1;
  if (lda % 8 != 0 || ldxr % 8 != 0 || (X1 && ldx1 % 8 != 0) || (Y && ldy % 8 != 0)) return 2;
  FfnLnTail ln{};
  ln.mode = (Hn || QKV) ? 2 : 1;
  ln.ga = gamma_a; ln.ba = beta_a; ln.gb = gamma_b; ln.bb = beta_b;
  ln.eps_a = eps_a; ln.eps_b = eps_b;
  ln.X2 = reinterpret_cast<bf16_t*>(X2); ln.Hn = reinterpret_cast<bf16_t*>(Hn);
  ln.mean_a = mean_a; ln.rstd_a = rstd_a; ln.mean_b = mean_b; ln.rstd_b = rstd_b;
  FfnPro pro{};
  pro.A = reinterpret_cast<const bf16_t*>(A); pro.lda = lda;
  pro.Xres = reinterpret_cast<const bf16_t*>(Xres); pro.ldx = ldxr;
  pro.bo = bo; pro.g1 = gamma1; pro.be1 = beta1; pro.eps1 = eps1;
  pro.Y = reinterpret_cast<bf16_t*>(Y); pro.ldy = ldy;
  pro.X1 = reinterpret_cast<bf16_t*>(X1); pro.ldx1 = ldx1;
  pro.mean1 = mean1; pro.rstd1 = rstd1;
  pro.QKV = reinterpret_cast<bf16_t*>(QKV); pro.ldqkv = ldqkv; pro.bqkv = bqkv; pro.qkv_at = FF / HC + 1 + NPB;
  // the FFN's input rows and its residual are x1: both come from the X fragments the prologue leaves in registers
  return launch_ffn(nullptr, 0, packed, b1, b2, nullptr, 0, Z, ldz, H, ldh, M, D, FF, 32, ln, stream, &pro, relu_bits);
}

