// MX-scaled fp8 GEMM for the fp8 weight path of BASELINE.json configs[4] (ChAda-ViT-Base, D = 768):
//   Out[M,N] = epilogue( dequant(Xq, xs)[M,K] * dequant(Wq, ws)[N,K]^T ),  fp32 accumulate, bf16 out.
// reference: the nn.Linear forwards of the encoder block (in_proj / out_proj / linear1 / linear2, src/backbones/vit/chada_vit.py:
// 95-116) -- the reference computes them in fp32; this is the reduced-precision weight path the north star asks for on Base.
//
// Number format: OCP MX (microscaling) fp8 -- elements e4m3fn, one shared power-of-two scale (E8M0 byte, value 2^(e-127)) per
// 32 consecutive k of a row.  That is the only 8-bit form gfx950 multiplies above the bf16 rate:
// v_mfma_scale_f32_16x16x128_f8f6f4 (2x the FLOP/clk of v_mfma_f32_16x16x32_bf16; the non-scaled fp8 MFMA runs at the bf16 rate).
// BOTH operands of that instruction are 8-bit, so the activations are quantised too (chadavit_mx8_quantize, per row and 32-k
// block, on the fly each forward); weights are quantised once per optimiser step.
//
// Operand layout of the instruction, measured on the hardware (scratch/mx/probe2.hip): lane l = (r = l & 15, g = l >> 4) supplies
// row r (A) / column r (B) and 32 bytes = 8 dwords of it -- NOT one contiguous 32-k block: with the 128 k of a step cut into eight
// 16-byte slots, lane group g holds slots g and g + 4 (its first 16 bytes belong to scale block g >> 1, the second 16 to scale block
// 2 + (g >> 1)).  The scale operand (byte 0 with opsel 0) of lane group s is the E8M0 scale of 32-k block s of that row / column.
// C/D is the usual 16x16 map (row = (l >> 4) * 4 + i, col = l & 15).
//
// Kernel: 256 x 128 x 128 tile, 8 consumer waves (4 x 2, 64 x 64 each = 4 x 4 MFMA tiles) + 4 DMA producer waves, three LDS stages
// filled by LDS-DMA two k-tiles ahead (buffer_load ... lds, 16 bytes per lane; the scale bytes of the tile and its bias floats ride the
// same ring as 4-byte pieces); rows are 128 bytes, the 16-byte slots of a row are XOR-swizzled with (row >> 1) & 7 on the DMA source
// side and on the fragment reads (a ds_read_b128 lane group then covers 16 distinct slots).  The MFMA is issued as D[n][m] (A = W
// fragment, B = X fragment) so a lane ends up with four consecutive n of one output row.
#include "common.h"

namespace {
using namespace chada;

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int MX_BM = 256, MX_BN = 128, MX_BK = 128;  // BK bytes = fp8 elements
constexpr int MX_XT = MX_BM * MX_BK;                   // X tile bytes in LDS (32 KiB)
constexpr int MX_WT = MX_BN * MX_BK;                   // W tile bytes (16 KiB)
constexpr int MX_XS = MX_XT + MX_WT;                   // X scales [4][256] (1 KiB) ...
constexpr int MX_WS = MX_XS + 4 * MX_BM;               // ... W scales [4][128] (512 B, slot padded to 1 KiB)
constexpr int MX_STAGE = MX_WS + 1024;                 // 50 KiB per stage; three stages = 150 KiB: one 8-wave block per CU
constexpr int MX_NSTG = 3;
#ifndef MX_ST_AUX
#define MX_ST_AUX 2    // cache policy of the result stores (buffer aux bits: 2 = nt, 16 = sc1 write-through, 0 = plain)
#endif

enum { MXE_NONE = 0, MXE_RELU = 1, MXE_RESID = 3, MXE_RELUMASK = 4 };    // numbering as the bf16 GEMM's epilogues

struct Mx8Args {
  const uint8_t* Xq; const uint8_t* xs;   // [M, K] fp8, [K/32, lds_x] e8m0 (row stride lds_x >= M, multiple of 4)
  const uint8_t* Wq; const uint8_t* ws;   // [N, K] fp8, [K/32, lds_w] e8m0
  bf16_t* Out; const float* bias; const bf16_t* aux;
  int M, N, K, ldo, ldaux, lds_x, lds_w;
  // optional second output: the result quantised again (e4m3 [M, N] + e8m0 [N/32, lds_o]) as the NEXT GEMM's X operand -- the same values
  // chadavit_mx8_quantize would produce from Out, without the pass over it; Out itself may then be NULL (no-grad passes)
  uint8_t* OutQ; uint8_t* outs; int lds_o;
};

__device__ __forceinline__ int mx_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// fragment / scale reads + 16 MFMAs of one landed k-tile.  The LDS pointer is a __restrict__ parameter on purpose: the
// reads then carry noalias scopes against the DMA and hipcc does not drain it.  STREAM_A: the W fragments are read two ahead of their
// MFMAs into two rotating register sets instead of all four up front (16 registers fewer alive: the residual epilogue's prefetched
// rows are alive across the last k-step of a tile).
__device__ __forceinline__ i32x8 mx8_frag(const uint8_t* __restrict__ st, unsigned a0_addr, unsigned a1_addr) {
  const u32x4 a0 = *reinterpret_cast<const u32x4*>(st + a0_addr);
  const u32x4 a1 = *reinterpret_cast<const u32x4*>(st + a1_addr);
  return i32x8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
}
// MX_PINGPONG (round 6): the two consumer waves of every SIMD (w and w + 4) in COMPLEMENTARY segments, two barriers per k-tile -- one reads the
// k-tile's fragments and scales into registers (Y) while the other multiplies the ones it read a segment earlier (X: 16 MFMAs, nothing else).  As built
// (one barrier per k-tile) all eight waves read, then all multiply: 1 470 cycles per k-tile for 1 024 of matrix work (profiles/r03f_mx8_gemm.md).
#ifndef MX_PINGPONG
#define MX_PINGPONG 0
#endif
struct Mx8Frags {
  i32x8 af[4], bfr[4];
  int sa[4], sb[4];
};
__device__ __forceinline__ void mx8_read(const uint8_t* __restrict__ st, const unsigned (&a_addr)[2], const unsigned (&b_addr)[2], unsigned sa_addr,
                                         unsigned sb_addr, Mx8Frags& f) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f.bfr[t] = mx8_frag(st, b_addr[0] + t * 2048, b_addr[1] + t * 2048);
    f.sa[t] = st[MX_WS + sa_addr + t * 16];
    f.sb[t] = st[MX_XS + sb_addr + t * 16];
    f.af[t] = mx8_frag(st, MX_XT + a_addr[0] + t * 2048, MX_XT + a_addr[1] + t * 2048);
  }
}
__device__ __forceinline__ void mx8_mma(const Mx8Frags& f, f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(f.af[nt], f.bfr[mt], acc[nt][mt], 0, 0, 0, f.sa[nt], 0, f.sb[mt]);
}

template <bool STREAM_A>
__device__ __forceinline__ void mx8_consume(const uint8_t* __restrict__ st, const unsigned (&a_addr)[2], const unsigned (&b_addr)[2],
                                            unsigned sa_addr, unsigned sb_addr, f32x4 (&acc)[4][4]) {
  i32x8 af[4], bfr[4];
  int sa[4], sb[4];
  // tile t = 16 rows further: + 2 KiB, same slot swizzle ((row >> 1) & 7 does not see the 16)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    bfr[t] = mx8_frag(st, b_addr[0] + t * 2048, b_addr[1] + t * 2048);
    sa[t] = st[MX_WS + sa_addr + t * 16];   // W scale of (n row of tile t, k-block g)
    sb[t] = st[MX_XS + sb_addr + t * 16];   // X scale of (m row of tile t, k-block g)
  }
#pragma unroll
  for (int t = 0; t < (STREAM_A ? 2 : 4); ++t) af[t] = mx8_frag(st, MX_XT + a_addr[0] + t * 2048, MX_XT + a_addr[1] + t * 2048);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[STREAM_A ? (nt & 1) : nt], bfr[mt], acc[nt][mt], 0, 0, 0, sa[nt], 0, sb[mt]);
    if (STREAM_A && nt < 2) {
      __builtin_amdgcn_sched_barrier(0);
      af[nt] = mx8_frag(st, MX_XT + a_addr[0] + (nt + 2) * 2048, MX_XT + a_addr[1] + (nt + 2) * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// 256 x 128 x 128 tile, THREE LDS stages with the k-tile fetched two ahead, PERSISTENT blocks (one per CU, 150 KiB of LDS) that walk
// the output tiles, and the waves SPECIALISED: 8 consumer waves (4 x 2, 64 x 64 each: fragment reads, MFMAs, epilogue) + 4 producer
// waves that do nothing but issue the LDS-DMA of the k-tile stream and wait for it.
//
// Why the split (measured, profiles/r03f_mx8_gemm.md): loads and stores share the one vmcnt counter.  A wave that issues both the
// DMA and the result stores cannot wait for a k-tile issued after its epilogue without sitting out the epilogue's stores too, and
// a store retires only when the write pipe has taken it -- several microseconds when 256 CUs write 64 KiB each at about the same
// time.  With every wave doing both, each output tile (~2.5 us of matrix work at K = 768) paid one such drain: DMA stream alone
// 102 us, epilogue alone 132 us, the two together 362 us (125 504 x 2 304 x 768).  (Nor may such a wave count past its stores:
// loads retire in order among themselves, but a store can retire before an older load, so "all but my 8 newest operations, the
// stores" does not mean the DMA has landed -- the previous version of this kernel waited like that.)  Here the consumers never
// wait on vmcnt for anything but the residual rows they fetch themselves, so their stores stay in flight across the following
// k-steps, and the producers issue loads only: their counted wait is exact.  (Three waves per SIMD: 168 registers per wave.)
//
// Per stream position q (one k-tile): barrier B_q = "position q has landed" (each producer waited for its own pieces) and "every
// consumer is done reading position q - 1" (each waited for its LDS reads).  After B_q the producers issue position q + 2 into the
// stage of q - 1; the consumers read and multiply position q, and after the last k-tile of an output tile each consumer wave
// converts and stores its own 64 x 64 piece (in-register transposition: no LDS, no further barrier).
// History at 125 504 x 2 304 x 768: 128 x 128 tile / two stages 509 us; this tile + ring, every wave doing everything, 486 us;
// + the transposing epilogue 427 us; + the stream running across output tiles 390 us; specialised waves: see the profile note.
constexpr int MX_NCONS = 8, MX_NPROD = 4, MX_PIECES = 14;   // DMA instructions per producer and k-tile (the counted wait below)
constexpr int MX_BIAS = MX_WS + 512;                        // the tile's 128 bias floats ride in the W-scale slot's padding

template <int EPI>
__global__ __launch_bounds__((MX_NCONS + MX_NPROD) * 64, 3) void gemm_mx8_kernel(Mx8Args a) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[MX_NSTG * MX_STAGE];
  // (the instances whose epilogue holds 32 registers of residual / ReLU-pattern rows across the last matrix segment do not fit the ping-pong
  // schedule's whole-k-tile fragment set in 168 registers -- 60 spilled --: they keep one barrier per k-tile)
  constexpr bool PP = MX_PINGPONG != 0 && !(EPI == MXE_RESID || EPI == MXE_RELUMASK);
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K;
  const int tiles_n = N / MX_BN;
  const int n_tiles = ((M + MX_BM - 1) / MX_BM) * tiles_n;
  const int KT = K / MX_BK;
  // the block's output tiles: blockIdx.x, + gridDim.x, ... in the XCD-aware order (gridDim.x is a multiple of 8 or == n_tiles)
  const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int Q = my_tiles * KT;  // k-tiles this block streams
  int opq = 0;
  asm volatile("" : "+s"(opq));
  uint8_t* const smem_o = smem + opq;

  if (w >= MX_NCONS) {
    // ================================================= producer p: X records (i, 2p) (i, 2p + 1), i < 4; W records (i, 2p) (i, 2p + 1),
    // i < 2 (a record = 8 rows x 128 B = 1 KiB); X scales of 32-k block p; W scales of blocks 2p, 2p + 1 (p < 2) or half of the bias
    const int p = w - MX_NCONS;
    const BufRsrc xr = make_rsrc(a.Xq), wr = make_rsrc(a.Wq);
    // scale / bias arrays through BOUNDED resources: the 4-byte pieces of a ragged last row tile may reach past the array (reads as
    // 0); no bias = zero records = zeros
    const BufRsrc xsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xs), 0, (K / 32) * a.lds_x, 0x00020000);
    const BufRsrc wsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.ws), 0, (K / 32) * a.lds_w, 0x00020000);
    const BufRsrc brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias ? a.bias : reinterpret_cast<const float*>(a.Wq)), 0,
                                                          a.bias ? N * 4 : 0, 0x00020000);
    unsigned xoff[8], woff[4], xs_off0 = 0, small_off0 = 0;
    auto set_issue_tile = [&](int ti) {
      const int bid = mx_xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, n_tiles);
      const int m0 = (bid / tiles_n) * MX_BM, n0 = (bid % tiles_n) * MX_BN;
      // record (i, ws) = rows [64 i + 8 ws, +8) at LDS i * 8192 + ws * 1024; lane l -> row + (l >> 3), LDS slot l & 7 holds the source
      // slot (l & 7) ^ ((row >> 1) & 7)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 64 * (j >> 1) + 8 * (2 * p + (j & 1)) + (l >> 3);
        xoff[j] = (unsigned)min(m0 + row, M - 1) * (unsigned)K + ((l & 7) ^ ((row >> 1) & 7)) * 16;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 64 * (j >> 1) + 8 * (2 * p + (j & 1)) + (l >> 3);
        woff[j] = (unsigned)(n0 + row) * (unsigned)K + ((l & 7) ^ ((row >> 1) & 7)) * 16;
      }
      xs_off0 = (unsigned)p * a.lds_x + m0 + 4 * l;   // k-tile 0 (4 rows of the scale array further per k-tile); lane = 4 rows
      small_off0 = p < 2 ? (unsigned)(2 * p + (l >> 5)) * a.lds_w + n0 + 4 * (l & 31)   // lanes 0-31 / 32-63: the two k-blocks
                         : (unsigned)(n0 + (p - 2) * 64 + l) * 4u;                        // bias floats [64 (p - 2), +64) of the tile
    };
    int iq_tile = 0, iq_kt = 0;  // (output tile, k-tile) of the NEXT stream position to issue
    auto issue_next = [&](uint8_t* __restrict__ dst) {
      const unsigned kbyte = (unsigned)iq_kt * MX_BK;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        lds_dma16(xr, reinterpret_cast<bf16_t*>(dst + (j >> 1) * 8192 + (2 * p + (j & 1)) * 1024), xoff[j], kbyte);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        lds_dma16(wr, reinterpret_cast<bf16_t*>(dst + MX_XT + (j >> 1) * 8192 + (2 * p + (j & 1)) * 1024), woff[j], kbyte);
      lds_dma4(xsr, dst + MX_XS + p * 256, xs_off0 + (unsigned)(4 * iq_kt) * a.lds_x, 0);
      if (p < 2) lds_dma4(wsr, dst + MX_WS + p * 256, small_off0 + (unsigned)(4 * iq_kt) * a.lds_w, 0);
      else lds_dma4(brs, dst + MX_BIAS + (p - 2) * 256, small_off0, 0);   // (with every k-tile: the count per producer stays fixed)
      if (++iq_kt == KT) { iq_kt = 0; ++iq_tile; if (iq_tile < my_tiles) set_issue_tile(iq_tile); }
    };
    static_assert(MX_PIECES == 8 + 4 + 1 + 1, "the counted wait below");
    set_issue_tile(0);
    issue_next(smem_o);
    if (Q > 1) issue_next(smem_o + MX_STAGE);
    int st_wr = 2;
    for (int q = 0; q < Q; ++q) {
      // position q has landed: only the MX_PIECES instructions of position q + 1 may still be in flight
      if (q + 1 < Q) asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (q + 2 < Q) issue_next(smem_o + st_wr * MX_STAGE);
      st_wr = st_wr == MX_NSTG - 1 ? 0 : st_wr + 1;
      if (PP) asm volatile("s_barrier" ::: "memory");   // the odd slot of position q (second half's fragment reads)
    }
    if (PP) asm volatile("s_barrier" ::: "memory");     // slot 2 Q: the second half's last matrix segment
    return;
  }

  // =================================================== consumer wave (wm, wn): rows [64 wm, +64) x columns [64 wn, +64) of the tile
  const int wm = w >> 1, wn = w & 1;
  const int r = l & 15, g = l >> 4;
  const bool r3 = (l & 8) != 0;
  unsigned a_addr[2], b_addr[2];  // W (A operand) rows of this wave's first n tile, X (B operand) rows of its first m tile
  {
    const int nrow = wn * 64 + r, mrow = wm * 64 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a_addr[h] = nrow * 128 + (((g + 4 * h) ^ ((nrow >> 1) & 7)) * 16);   // 16-byte slots g and g + 4 of the row (see the header)
      b_addr[h] = mrow * 128 + (((g + 4 * h) ^ ((mrow >> 1) & 7)) * 16);
    }
  }
  const unsigned sa_addr = g * MX_BN + wn * 64 + r, sb_addr = g * MX_BM + wm * 64 + r;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool AUXR = EPI == MXE_RESID || EPI == MXE_RELUMASK;   // the epilogue reads an [M, N] bf16 operand (residual / ReLU pattern)
  bf16x8 ep_rr[AUXR ? 4 : 1][2];   // its rows for the tile about to finish (see the prefetch below)
  int st_rd = 0, ckt = 0, ctile = 0;
  // the residual / ReLU-pattern rows of output tile `ctile` are requested behind the barrier of its last k-step and land behind that step's MFMAs.
  // The epilogue then issues no load at all: a load's wait also waits for every OLDER store.
  auto aux_prefetch = [&]() {
    if constexpr (AUXR) {
        // last k-step of output tile `ctile`: its residual rows are requested now and land behind this step's MFMAs.  The epilogue
        // then issues no load at all: a load's wait also waits for every OLDER store.
        const int bid = mx_xcd_remap((int)blockIdx.x + ctile * (int)gridDim.x, n_tiles);
        const int m0 = (bid / tiles_n) * MX_BM, n0 = (bid % tiles_n) * MX_BN;
        const BufRsrc aux_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.aux) + (size_t)m0 * a.ldaux + n0, 0,
                                                                 min(MX_BM, M - m0) * a.ldaux * 2, 0x00020000);
        const unsigned aux_voff = ((wm * 64 + (r & 7)) * (unsigned)a.ldaux + wn * 64 + 32 * r3 + 8 * g) * 2u;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int k = 0; k < 2; ++k)   // the epilogue's own layout: row (r & 7) + 8 k of the slab, columns 32 r3 + 8 g + [0, 8); rows past M read 0
            ep_rr[mt][k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(aux_rs, aux_voff, (mt * 16 + 8 * k) * a.ldaux * 2, 0));
    }
  };
  auto epilogue = [&](const uint8_t* stg) {
      // ---- epilogue of output tile `ctile`, per wave, no LDS and no barrier.  acc[nt][mt][i] = out[m = .. + mt*16 + r][n = .. +
      // nt*16 + 4 g + i]: written straight out that is 8-byte pieces scattered over 16 rows per instruction.  The four lanes
      // (r, g = 0..3) hold 64 consecutive n of row r between them; two rounds of register <-> lane-bit exchanges (v_permlane32_swap:
      // lane bit 5 <-> item bit 0, then v_permlane16_swap: lane bit 4 <-> the same item bit) leave lane (r, g) with
      // n = 32 j + 8 g + [0, 8) for j = 0, 1; a third (DPP, below) trades j for row bit 3, so that each of the two store instructions
      // of a 16-row slab writes eight whole 128-byte row segments.  (The transposition through LDS this replaces cost two barriers and four LDS round trips per tile: 4 000 of the
      // 15 400 cycles a tile took.)
      const int bid = mx_xcd_remap((int)blockIdx.x + ctile * (int)gridDim.x, n_tiles);
      const int m0 = (bid / tiles_n) * MX_BM, n0 = (bid % tiles_n) * MX_BN;
      f32x4 bia[4];   // (MX_PINGPONG: the bias is the accumulators' initial value -- bias_init below -- and nothing is added here)   // bias of the lane's accumulator columns (added before the exchange: the exchange below is inline asm, and its
                      // inputs then come from a VALU instruction hipcc itself has placed after the MFMAs)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bia[nt] = PP ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(stg + MX_BIAS + (wn * 64 + 16 * nt + 4 * g) * 4);
      // ONE wait for the prefetched residual rows, here, before the first store (these empty statements "use" them): left to
      // itself hipcc waits for row k at slab k with a count that ignores the conditional stores in between, and every such wait
      // drains the stores of the slabs before it
      if constexpr (AUXR) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" ::"v"(ep_rr[mt][0]), "v"(ep_rr[mt][1]));
      }
      // the results leave through per-tile buffer resources: one per-lane offset for the whole tile (the slab and the column half
      // are a wave-uniform offset and an immediate), rows past M fall outside the resource and are dropped by the hardware, and the
      // stores can carry a cache policy (nt measured 3-9 % ahead of plain and sc1)
      const int rows = min(MX_BM, M - m0);
      const BufRsrc out_rs = __builtin_amdgcn_make_buffer_rsrc(a.Out ? a.Out + (size_t)m0 * a.ldo + n0 : nullptr, 0,
                                                               a.Out ? rows * a.ldo * 2 : 0, 0x00020000);
      const BufRsrc outq_rs = __builtin_amdgcn_make_buffer_rsrc(a.OutQ ? a.OutQ + (size_t)m0 * a.N + n0 : nullptr, 0,
                                                                a.OutQ ? rows * a.N : 0, 0x00020000);
      const unsigned lane_row = wm * 64 + (r & 7), lane_col = wn * 64 + 32 * r3 + 8 * g;
      const unsigned out_voff = (lane_row * (unsigned)a.ldo + lane_col) * 2u, outq_voff = lane_row * (unsigned)a.N + lane_col;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float x[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            x[nt][i] = acc[nt][mt][i] + bia[nt][i];
            if constexpr (EPI == MXE_RELU) x[nt][i] = relu_f(x[nt][i]);
          }
          acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // the exchanges: on bf16 pairs where the value is final (half the instructions); the residual instance adds in fp32 after them
        constexpr int NX = EPI == MXE_RESID ? 4 : 2;   // dwords per item
        unsigned y[4][NX];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if constexpr (EPI == MXE_RESID) {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[nt][i] = __builtin_bit_cast(unsigned, x[nt][i]);
          } else {
            typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2;
            y[nt][0] = __builtin_bit_cast(unsigned, bf16x2{(bf16_t)x[nt][0], (bf16_t)x[nt][1]});
            y[nt][1] = __builtin_bit_cast(unsigned, bf16x2{(bf16_t)x[nt][2], (bf16_t)x[nt][3]});
          }
        }
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
          for (int i = 0; i < NX; ++i)   // (the s_nops: VALU write -> permlane swap and swap -> swap wait states, invisible to hipcc inside asm)
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1"
                         : "+v"(y[2 * s1][i]), "+v"(y[2 * s1 + 1][i]));
        // third exchange, lane bit 3 (row r + 8 of the slab) <-> column half j, by DPP inside the 16-lane row (row_ror:8 written into
        // one half of the row only): store instruction k then covers rows (r & 7) + 8 k with all 128 bytes of each (eight lanes x
        // 16 bytes; half-line stores measured 3-20 % slower).  P[j] = items 2 j, 2 j + 1;  Q[0] = r3 ? partner's P[1] : P[0],
        // Q[1] = r3 ? P[1] : partner's P[0].
        unsigned q0[2 * NX], q1[2 * NX];
#pragma unroll
        for (int e = 0; e < 2 * NX; ++e) {
          const unsigned p0 = y[e / NX][e % NX], p1 = y[2 + e / NX][e % NX];
          q0[e] = (unsigned)__builtin_amdgcn_update_dpp((int)p0, (int)p1, 0x128, 0xF, 0xC, false);   // lanes 8-15 of a row <- lane - 8's P[1]
          q1[e] = (unsigned)__builtin_amdgcn_update_dpp((int)p1, (int)p0, 0x128, 0xF, 0x3, false);   // lanes 0-7 <- lane + 8's P[0]
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int m = m0 + wm * 64 + mt * 16 + 8 * k + (r & 7);
          const unsigned (&qq)[2 * NX] = k == 0 ? q0 : q1;
          bf16x8 o;
          if constexpr (EPI == MXE_RESID) {
            const bf16x8 rr = ep_rr[mt][k];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(__builtin_bit_cast(float, qq[e]) + (float)rr[e]);
          } else {
            o = __builtin_bit_cast(bf16x8, u32x4{qq[0], qq[1], qq[2], qq[3]});
            if constexpr (EPI == MXE_RELUMASK) {   // threshold_backward: the gradient passes where the forward's ReLU output was > 0
              const bf16x8 rr = ep_rr[mt][k];
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (float)rr[e] > 0.f ? o[e] : (bf16_t)0.f;
            }
          }
          if (a.Out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, out_voff, (mt * 16 + 8 * k) * a.ldo * 2, MX_ST_AUX);
          if (a.OutQ) {  // (uniform) quantise the bf16 values: a 32-column MX block is the four lanes g = 0..3 of one (r & 7, r3)
            float f[8], amax = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { f[i] = (float)o[i]; amax = fmaxf(amax, fabsf(f[i])); }
            amax = rows_max(amax);
            int e8 = 127;
            float inv = 1.f;
            if (amax > 0.f) {  // as mx8_quantize_kernel
              int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xff) - 127 - 8;
              if (amax * __builtin_bit_cast(float, (unsigned)(127 - ex) << 23) > 448.f) ex += 1;
              ex = max(-127, min(127, ex));
              e8 = ex + 127;
              inv = __builtin_bit_cast(float, (unsigned)(127 - ex) << 23);
            }
            int p0 = 0, p1 = 0;
            p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, p0, false);
            p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, p0, true);
            p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, p1, false);
            p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, p1, true);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)p0, (unsigned)p1}, outq_rs, outq_voff, (mt * 16 + 8 * k) * a.N, 0);
            if (g == 0 && m < M) a.outs[(size_t)((n0 + wn * 64 + 32 * r3) >> 5) * a.lds_o + m] = (uint8_t)e8;
          }
        }
      }
  };
  if constexpr (PP) {
    // ---- two barriers per stream position q.  Slot 2q: half A (waves 0-3) reads position q into registers, half B (waves 4-7; wave w and w + 4 share a
    // SIMD) multiplies position q - 1; slot 2q + 1: A multiplies q, B reads q.  A stage is read in slots 2q and 2q + 1 and refilled (position q + 3) behind
    // the barrier of slot 2q + 2 -- so neither half may touch it later than that: the bias of an output tile is taken as the accumulators' INITIAL value
    // when the tile's first position is read (it rides every k-tile), and half A's epilogue, which would otherwise idle the SIMD for a whole slot of its
    // own, runs at the head of slot 2q + 2 beside half B's last matrix segment and epilogue.
    const bool halfB = w >= 4;
    Mx8Frags fr;
    auto read_pos = [&](int q) {
      const uint8_t* const stg = smem_o + (q % MX_NSTG) * MX_STAGE;
      mx8_read(stg, a_addr, b_addr, sa_addr, sb_addr, fr);
      if (q % KT == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(stg + MX_BIAS + (wn * 64 + 16 * nt + 4 * g) * 4);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = bv;
        }
      }
    };
    if (!halfB) {
      for (int q = 0; q < Q; ++q) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slot 2q
        if (q > 0 && q % KT == 0) { epilogue(nullptr); ++ctile; }
        read_pos(q);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slot 2q + 1
        if (q % KT == KT - 1) aux_prefetch();
        mx8_mma(fr, acc);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // slot 2Q
      epilogue(nullptr);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // slot 0
      for (int q = 0; q < Q; ++q) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slot 2q + 1
        read_pos(q);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slot 2q + 2
        if (q % KT == KT - 1) aux_prefetch();
        mx8_mma(fr, acc);
        if (q % KT == KT - 1) { epilogue(nullptr); ++ctile; }
      }
    }
    return;
  }
  for (int q = 0; q < Q; ++q) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B_q
    if (ckt == KT - 1) aux_prefetch();
    const uint8_t* const stg = smem_o + st_rd * MX_STAGE;
    mx8_consume<AUXR>(stg, a_addr, b_addr, sa_addr, sb_addr, acc);
    if (++ckt == KT) {
      epilogue(stg);
      ckt = 0;
      ++ctile;
    }
    st_rd = st_rd == MX_NSTG - 1 ? 0 : st_rd + 1;
  }
}

// ---- quantiser: bf16 [R, K] -> e4m3 [R, K] + e8m0 [K/32, R].  One thread per (row, 32-element block).
// Scale: 2^e with e = floor(log2(amax)) - 8 (OCP MX: the largest element lands in [256, 512)), raised by one when that would
// push the largest element above e4m3's 448 (no saturation, no NaN); an all-zero block gets scale 1.
__global__ __launch_bounds__(256) void mx8_quantize_kernel(const bf16_t* __restrict__ x, int ldx, uint8_t* __restrict__ q,
                                                           uint8_t* __restrict__ s, int lds, long long n_blocks, int R, int K, int relu) {
  const int KB = K / 32;
  for (long long id = blockIdx.x * 256ll + threadIdx.x; id < n_blocks; id += (long long)gridDim.x * 256ll) {
    const int row = (int)(id / KB), kb = (int)(id % KB);
    const bf16_t* src = x + (size_t)row * ldx + kb * 32;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(src + j * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)t[e];
        if (relu) f = fmaxf(f, 0.f);
        v[j * 8 + e] = f;
        amax = fmaxf(amax, fabsf(f));
      }
    }
    int e8 = 127;
    float inv = 1.f;
    if (amax > 0.f) {
      int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xff) - 127 - 8;   // floor(log2(amax)) - 8
      if (amax * __builtin_bit_cast(float, (unsigned)(127 - ex) << 23) > 448.f) ex += 1;
      ex = max(-127, min(127, ex));
      e8 = ex + 127;
      inv = __builtin_bit_cast(float, (unsigned)(127 - ex) << 23);                   // 2^-ex (ex in [-127, 127] -> finite)
    }
    unsigned out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int pk = 0;
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j] * inv, v[4 * j + 1] * inv, pk, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j + 2] * inv, v[4 * j + 3] * inv, pk, true);
      out[j] = (unsigned)pk;
    }
    u32x4* dst = reinterpret_cast<u32x4*>(q + (size_t)row * K + kb * 32);
    dst[0] = u32x4{out[0], out[1], out[2], out[3]};
    dst[1] = u32x4{out[4], out[5], out[6], out[7]};
    s[(size_t)kb * lds + row] = (uint8_t)e8;
  }
}

}  // namespace

extern "C" int chadavit_mx8_quantize(const chada_bf16* x, int ldx, void* q, void* scales, int lds, int R, int K, int relu,
                                     void* stream) {
  CHADA_ENTRY();
  if (!x || !q || !scales || R <= 0 || K <= 0) return 1;
  if (K % 32 != 0 || ldx % 8 != 0 || ((uintptr_t)q & 15) != 0 || lds < R || lds % 4 != 0 || ((uintptr_t)scales & 3) != 0) return 2;
  const long long nb = (long long)R * (K / 32);
  long long grid = (nb + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(mx8_quantize_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(x), ldx, reinterpret_cast<uint8_t*>(q), reinterpret_cast<uint8_t*>(scales), lds, nb, R,
                     K, relu);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_gemm_nt_mx8_q(const void* Xq, const void* xs, int lds_x, const void* Wq, const void* ws, int lds_w, chada_bf16* Out,
                                      int ldo, void* OutQ, void* out_scales, int lds_o, int M, int N, int K, const float* bias, int epilogue,
                                      const chada_bf16* aux, int ldaux, void* stream);

extern "C" int chadavit_gemm_nt_mx8(const void* Xq, const void* xs, int lds_x, const void* Wq, const void* ws, int lds_w, chada_bf16* Out,
                                    int ldo, int M, int N, int K, const float* bias, int epilogue, const chada_bf16* aux, int ldaux,
                                    void* stream) {
  if (!Out) return 1;
  return chadavit_gemm_nt_mx8_q(Xq, xs, lds_x, Wq, ws, lds_w, Out, ldo, nullptr, nullptr, 0, M, N, K, bias, epilogue, aux, ldaux, stream);
}

// ... with the result also (or only: Out == NULL) leaving as the next GEMM's fp8 X operand: OutQ [M, N] e4m3, out_scales [N/32, lds_o] e8m0
extern "C" int chadavit_gemm_nt_mx8_q(const void* Xq, const void* xs, int lds_x, const void* Wq, const void* ws, int lds_w, chada_bf16* Out,
                                      int ldo, void* OutQ, void* out_scales, int lds_o, int M, int N, int K, const float* bias, int epilogue,
                                      const chada_bf16* aux, int ldaux, void* stream) {
  CHADA_ENTRY();
  if (!Xq || !xs || !Wq || !ws || (!Out && !OutQ) || M <= 0 || N <= 0 || K <= 0) return 1;
  if (OutQ && (!out_scales || lds_o < M || ((uintptr_t)OutQ & 7) != 0)) return 1;
  if (Out == nullptr) ldo = 8;
  if (N % MX_BN != 0 || K % MX_BK != 0 || ldo % 8 != 0 || (long long)M * K >= (1ll << 32) || (long long)N * K >= (1ll << 32)) return 2;
  if (lds_x < M || lds_w < N || lds_x % 4 != 0 || lds_w % 4 != 0 || (((uintptr_t)xs | (uintptr_t)ws) & 3) != 0) return 2;
  if ((epilogue == MXE_RESID || epilogue == MXE_RELUMASK) && (!aux || ldaux % 8 != 0)) return 1;
  Mx8Args a;
  a.Xq = reinterpret_cast<const uint8_t*>(Xq); a.xs = reinterpret_cast<const uint8_t*>(xs);
  a.Wq = reinterpret_cast<const uint8_t*>(Wq); a.ws = reinterpret_cast<const uint8_t*>(ws);
  a.Out = reinterpret_cast<bf16_t*>(Out); a.bias = bias; a.aux = reinterpret_cast<const bf16_t*>(aux);
  a.M = M; a.N = N; a.K = K; a.ldo = ldo; a.ldaux = ldaux; a.lds_x = lds_x; a.lds_w = lds_w;
  a.OutQ = reinterpret_cast<uint8_t*>(OutQ); a.outs = reinterpret_cast<uint8_t*>(out_scales); a.lds_o = lds_o;
  const int n_tiles = ((M + MX_BM - 1) / MX_BM) * (N / MX_BN);
  const dim3 grid(n_tiles < 256 ? n_tiles : 256);  // persistent: one 8-wave block per CU (150 KiB of LDS)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (epilogue) {
    case MXE_NONE: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_NONE>), grid, dim3((MX_NCONS + MX_NPROD) * 64), 0, s, a); break;
    case MXE_RELU: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_RELU>), grid, dim3((MX_NCONS + MX_NPROD) * 64), 0, s, a); break;
    case MXE_RESID: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_RESID>), grid, dim3((MX_NCONS + MX_NPROD) * 64), 0, s, a); break;
    case MXE_RELUMASK: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_RELUMASK>), grid, dim3((MX_NCONS + MX_NPROD) * 64), 0, s, a); break;
    default: return 2;
  }
  CHADA_CHECK_LAUNCH();
  return 0;
}
