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
// Kernel: 128 x 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each = 4 x 4 MFMA tiles), two LDS stages filled by LDS-DMA
// (buffer_load ... lds, 16 bytes per lane); rows are 128 bytes, the 16-byte slots of a row are XOR-swizzled with (row >> 1) & 7
// on the DMA source side and on the fragment reads (a ds_read_b128 lane group then covers 16 distinct slots).  The MFMA is
// issued as D[n][m] (A = W fragment, B = X fragment) so a lane ends up with four consecutive n of one output row: 8-byte stores.
#include "common.h"

namespace {
using namespace chada;

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int MX_BM = 128, MX_BN = 128, MX_BK = 128;  // BK bytes = fp8 elements
constexpr int MX_TILE = MX_BM * MX_BK;                // bytes of one operand tile in LDS (16 KiB)

enum { MXE_NONE = 0, MXE_RELU = 1, MXE_RESID = 3 };   // numbering as the bf16 GEMM's epilogues

struct Mx8Args {
  const uint8_t* Xq; const uint8_t* xs;   // [M, K] fp8, [K/32, M] e8m0
  const uint8_t* Wq; const uint8_t* ws;   // [N, K] fp8, [K/32, N] e8m0
  bf16_t* Out; const float* bias; const bf16_t* aux;
  int M, N, K, ldo, ldaux;
};

__device__ __forceinline__ int mx_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// one pipeline step: DMA of the next k-tile (when `issue`), fragment reads + 16 MFMAs of the current one.  The LDS pointers are
// __restrict__ parameters on purpose: the reads then carry noalias scopes against the DMA and hipcc does not drain it.
__device__ __forceinline__ void mx8_step(BufRsrc xr, BufRsrc wr, uint8_t* __restrict__ dst, const uint8_t* __restrict__ st, bool issue,
                                         const unsigned (&xoff)[4], const unsigned (&woff)[4], unsigned kbyte, int l,
                                         const unsigned (&a_addr)[4][2], const unsigned (&b_addr)[4][2],
                                         const int (&sa)[4], const int (&sb)[4], f32x4 (&acc)[4][4]) {
  if (issue) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_dma16(xr, reinterpret_cast<bf16_t*>(dst + i * 4096), xoff[i], kbyte);                // rows 32 i .. of the X tile (per wave: 4 x 8 rows)
      lds_dma16(wr, reinterpret_cast<bf16_t*>(dst + MX_TILE + i * 4096), woff[i], kbyte);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  i32x8 af[4], bfr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(st + MX_TILE + a_addr[t][0]);
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(st + MX_TILE + a_addr[t][1]);
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(st + b_addr[t][0]);
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(st + b_addr[t][1]);
    af[t] = i32x8{(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    bfr[t] = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[nt], bfr[mt], acc[nt][mt], 0, 0, 0, sa[nt], 0, sb[mt]);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mx8_kernel(Mx8Args a) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 2 * MX_TILE];  // 2 stages x (X tile | W tile) = 64 KiB
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int M = a.M, N = a.N, K = a.K;
  const int tiles_n = N / MX_BN;
  const int bid = mx_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * MX_BM, n0 = (bid % tiles_n) * MX_BN;
  const int KT = K / MX_BK;
  const BufRsrc xr = make_rsrc(a.Xq), wr = make_rsrc(a.Wq);
  // DMA sources: wave w fills rows [32 i + 8 w, +8) of both tiles with instruction i; lane l -> row + (l >> 3), LDS slot l & 7 holds
  // the source slot (l & 7) ^ ((row >> 1) & 7)
  unsigned xoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * i + 8 * w + (l >> 3);
    const int slot = (l & 7) ^ ((row >> 1) & 7);
    xoff[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)K + slot * 16;
    woff[i] = (unsigned)(n0 + row) * (unsigned)K + slot * 16;
  }
  // (the LDS image of instruction i of wave w sits at i * 4096 + w * 1024 within a tile: rows 32 i + 8 w ..)
  const int r = l & 15, g = l >> 4;
  unsigned a_addr[4][2], b_addr[4][2];  // W (A operand) rows of this wave's n tiles, X (B operand) rows of its m tiles
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int nrow = wn * 64 + t * 16 + r, mrow = wm * 64 + t * 16 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a_addr[t][h] = nrow * 128 + (((g + 4 * h) ^ ((nrow >> 1) & 7)) * 16);   // 16-byte slots g and g + 4 of the row (see the header)
      b_addr[t][h] = mrow * 128 + (((g + 4 * h) ^ ((mrow >> 1) & 7)) * 16);
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // scale bytes of k-tile kt for this lane: ws[(4 kt + g) * N + n], xs[(4 kt + g) * M + m]
  const uint8_t* wsp[4];
  const uint8_t* xsp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    wsp[t] = a.ws + (size_t)g * N + (n0 + wn * 64 + t * 16 + r);
    xsp[t] = a.xs + (size_t)g * M + min(m0 + wm * 64 + t * 16 + r, M - 1);
  }
  int sa[4], sb[4], san[4], sbn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { sa[t] = wsp[t][0]; sb[t] = xsp[t][0]; }
  int opq = 0;
  asm volatile("" : "+s"(opq));
  uint8_t* const smem_o = smem + opq;
  // stage 0 <- k-tile 0 (bare: nothing reads LDS before the barrier); the DMA image of wave w: tile + i * 4096 + w * 1024
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lds_dma16(xr, reinterpret_cast<bf16_t*>(smem + i * 4096 + w * 1024), xoff[i], 0u);
    lds_dma16(wr, reinterpret_cast<bf16_t*>(smem + MX_TILE + i * 4096 + w * 1024), woff[i], 0u);
  }
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // k-tile kt has landed; the other stage is free
    const bool more = kt + 1 < KT;
    if (more) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        san[t] = wsp[t][(size_t)(4 * (kt + 1)) * N];
        sbn[t] = xsp[t][(size_t)(4 * (kt + 1)) * M];
      }
    }
    mx8_step(xr, wr, smem_o + ((kt + 1) & 1) * 2 * MX_TILE + w * 1024, smem_o + (kt & 1) * 2 * MX_TILE, more, xoff, woff,
             (unsigned)(kt + 1) * MX_BK, l, a_addr, b_addr, sa, sb, acc);
    if (more) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { sa[t] = san[t]; sb[t] = sbn[t]; }
    }
  }
  // epilogue: acc[nt][mt][i] = out[m = m0 + wm*64 + mt*16 + r][n = n0 + wn*64 + nt*16 + 4 g + i]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + r;
    if (m >= M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * g;
      f32x4 v = acc[nt][mt];
      if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
      if constexpr (EPI == MXE_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
      } else if constexpr (EPI == MXE_RESID) {
        const bf16x4 rr = *reinterpret_cast<const bf16x4*>(a.aux + (size_t)m * a.ldaux + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
      }
      *reinterpret_cast<bf16x4*>(a.Out + (size_t)m * a.ldo + n) = pack4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- quantiser: bf16 [R, K] -> e4m3 [R, K] + e8m0 [K/32, R].  One thread per (row, 32-element block).
// Scale: 2^e with e = floor(log2(amax)) - 8 (OCP MX: the largest element lands in [256, 512)), raised by one when that would
// push the largest element above e4m3's 448 (no saturation, no NaN); an all-zero block gets scale 1.
__global__ __launch_bounds__(256) void mx8_quantize_kernel(const bf16_t* __restrict__ x, int ldx, uint8_t* __restrict__ q,
                                                           uint8_t* __restrict__ s, long long n_blocks, int R, int K, int relu) {
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
    s[(size_t)kb * R + row] = (uint8_t)e8;
  }
}

}  // namespace

extern "C" int chadavit_mx8_quantize(const chada_bf16* x, int ldx, void* q, void* scales, int R, int K, int relu, void* stream) {
  CHADA_ENTRY();
  if (!x || !q || !scales || R <= 0 || K <= 0) return 1;
  if (K % 32 != 0 || ldx % 8 != 0 || ((uintptr_t)q & 15) != 0) return 2;
  const long long nb = (long long)R * (K / 32);
  long long grid = (nb + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(mx8_quantize_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(x), ldx, reinterpret_cast<uint8_t*>(q), reinterpret_cast<uint8_t*>(scales), nb, R, K,
                     relu);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_gemm_nt_mx8(const void* Xq, const void* xs, const void* Wq, const void* ws, chada_bf16* Out, int ldo, int M, int N,
                                    int K, const float* bias, int epilogue, const chada_bf16* aux, int ldaux, void* stream) {
  CHADA_ENTRY();
  if (!Xq || !xs || !Wq || !ws || !Out || M <= 0 || N <= 0 || K <= 0) return 1;
  if (N % MX_BN != 0 || K % MX_BK != 0 || ldo % 4 != 0 || (long long)M * K >= (1ll << 32) || (long long)N * K >= (1ll << 32)) return 2;
  if (epilogue == MXE_RESID && (!aux || ldaux % 4 != 0)) return 1;
  Mx8Args a;
  a.Xq = reinterpret_cast<const uint8_t*>(Xq); a.xs = reinterpret_cast<const uint8_t*>(xs);
  a.Wq = reinterpret_cast<const uint8_t*>(Wq); a.ws = reinterpret_cast<const uint8_t*>(ws);
  a.Out = reinterpret_cast<bf16_t*>(Out); a.bias = bias; a.aux = reinterpret_cast<const bf16_t*>(aux);
  a.M = M; a.N = N; a.K = K; a.ldo = ldo; a.ldaux = ldaux;
  const dim3 grid(((M + MX_BM - 1) / MX_BM) * (N / MX_BN));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (epilogue) {
    case MXE_NONE: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_NONE>), grid, dim3(256), 0, s, a); break;
    case MXE_RELU: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_RELU>), grid, dim3(256), 0, s, a); break;
    case MXE_RESID: hipLaunchKernelGGL((gemm_mx8_kernel<MXE_RESID>), grid, dim3(256), 0, s, a); break;
    default: return 2;
  }
  CHADA_CHECK_LAUNCH();
  return 0;
}
