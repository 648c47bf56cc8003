// Out[M,N] = epilogue(X[M,K] * W[N,K]^T): bf16 operands, fp32 MFMA accumulate, fused epilogues.
//
// Tile BM x BN x 64, 4 waves (2x2), single LDS stage + register prefetch of the next k-tile (the
// global loads of tile t+1 are in flight while tile t is multiplied).  LDS rows are padded by 8
// bf16 (144-byte stride) so the 16 rows a ds_read_b128 lane group touches fall on 16 distinct
// 16-byte slots.  The MFMA is issued as D[n][m] = W-frag x X-frag so that each lane ends up with
// FOUR CONSECUTIVE n of one output row m: the epilogue then reads bias / residual and writes the
// result with 8-byte (bf16) or 16-byte (fp32) vector accesses instead of 2-byte scatters.
#include <type_traits>

#include "common.h"

using namespace chada;

namespace {

constexpr int BK = 64;
constexpr int LDK = BK + 16;  // row stride = 32 B x odd: conflict-free for the lane groups a ds_read_b128 actually serves together (see attention.hip, dkv_swz); +8 was 2-way

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_RELUMASK = 4, EPI_GELUBWD = 5, EPI_TOKEN = 6 };

struct NtArgs {
  const bf16_t* X;
  const bf16_t* W;
  void* Out;
  const float* bias;
  const bf16_t* aux;
  bf16_t* aux_out;
  int M, N, K, ldx, ldw, ldo, ldaux;
  // tokenizer epilogue
  const float* pos;
  const float* chan;
  const int* chan_img;
  const int* chan_idx;
  int p;
  // tokenizer with the patch gather folded into the X staging (no im2col buffer): X row m = patch (m % p) of channel image m / p
  // of img [n_chan, S, S] fp32, k = u * 16 + v inside the 16 x 16 patch
  const float* img;
  int S;
};

__device__ __forceinline__ bf16x8 pack8f(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
  r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
  return r;
}

// XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous run of
// logical tiles so the N-tiles that share one X row-panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}


// Epilogue for 8 consecutive output columns n..n+7 of row m (v0 = cols n..n+3, v1 = n+4..n+7).
template <int EPI, bool OUT_F32>
__device__ __forceinline__ void epi_store(const NtArgs& a, f32x4 v0, f32x4 v1, f32x4 b0, f32x4 b1, bf16x8 rr, int m, int n, int N) {
  bf16_t* __restrict__ aux_out = a.aux_out;
  const int ldo = a.ldo, ldaux = a.ldaux;
  void* Out = a.Out;
  v0 += b0;   // (the caller fetched the bias of its columns before its first store: a load between two stores waits for the
  v1 += b1;   //  first one as well -- loads and stores share the one vmcnt counter)
  int orow = m;
  if constexpr (EPI == EPI_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] = relu_f(v0[r]); v1[r] = relu_f(v1[r]); }
  } else if constexpr (EPI == EPI_GELU) {
    bf16x8 pre;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pre[r] = (bf16_t)v0[r]; pre[4 + r] = (bf16_t)v1[r]; }
    *reinterpret_cast<bf16x8*>(aux_out + (size_t)m * ldaux + n) = pre;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] = gelu_erf(v0[r]); v1[r] = gelu_erf(v1[r]); }
  } else if constexpr (EPI == EPI_RESID) {   // (rr: the aux piece, fetched by the caller before its first store)
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] += (float)rr[r]; v1[r] += (float)rr[4 + r]; }
  } else if constexpr (EPI == EPI_RELUMASK) {   // (rr: the aux piece, fetched by the caller before its first store)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v0[r] = ((float)rr[r] > 0.f) ? v0[r] : 0.f;
      v1[r] = ((float)rr[4 + r] > 0.f) ? v1[r] : 0.f;
    }
  } else if constexpr (EPI == EPI_GELUBWD) {   // (rr: the aux piece, fetched by the caller before its first store)
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] *= gelu_erf_grad((float)rr[r]); v1[r] *= gelu_erf_grad((float)rr[4 + r]); }
  } else if constexpr (EPI == EPI_TOKEN) {
    const int ci = m / a.p;
    orow = m + a.chan_img[ci] + 1;
    const float* posrow = a.pos + (size_t)(m - ci * a.p) * N + n;
    v0 += *reinterpret_cast<const f32x4*>(posrow);
    v1 += *reinterpret_cast<const f32x4*>(posrow + 4);
    if (a.chan) {
      const float* chanrow = a.chan + (size_t)a.chan_idx[ci] * N + n;
      v0 += *reinterpret_cast<const f32x4*>(chanrow);
      v1 += *reinterpret_cast<const f32x4*>(chanrow + 4);
    }
  }
  if constexpr (OUT_F32) {
    float* op = reinterpret_cast<float*>(Out) + (size_t)orow * ldo + n;
    *reinterpret_cast<f32x4*>(op) = v0;
    *reinterpret_cast<f32x4*>(op + 4) = v1;
  } else {
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)v0[r]; o[4 + r] = (bf16_t)v1[r]; }
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(Out) + (size_t)orow * ldo + n) = o;
  }
}

template <int BM, int BN, int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NtArgs a) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int MB = TM / 16, NB = TN / 16;
  constexpr int XCH = BM * 8 / 256, WCH = BN * 8 / 256;
  __shared__ __attribute__((aligned(16))) bf16_t smem[(BM + BN) * LDK];
  bf16_t* sX = smem;
  bf16_t* sW = smem + BM * LDK;

  const int tid = threadIdx.x;
  const int l = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  // kernel arguments into registers (taking the struct's address would spill it to scratch)
  const bf16_t* __restrict__ gX = a.X;
  const bf16_t* __restrict__ gW = a.W;
  const int M = a.M, N = a.N, K = a.K, ldx = a.ldx, ldw = a.ldw;
  const int tiles_n = N / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM;
  const int n0 = (bid % tiles_n) * BN;

  // per-thread staging chunks: chunk id = tid + 256*i -> (row = id>>3, 16-byte column chunk = id&7)
  const int srow = tid >> 3, sch = tid & 7;
  const bf16_t* xsrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) xsrc[i] = gX + (size_t)min(m0 + srow + 32 * i, M - 1) * ldx + sch * 8;
  const bf16_t* wsrc = gW + (size_t)(n0 + srow) * ldw + sch * 8;
  u32x4 xr[XCH], wr[WCH];
  // EPI_TOKEN with a.img: the X tile is gathered straight from the fp32 image (conv unfold of chada_vit.py:128-133 folded into
  // the GEMM's staging): chunk (row, sch) of k-tile k0 = 8 consecutive pixels of patch row u = (k0 + 8 sch) / 16
  const float* isrc[XCH];
  const float* const gimg = (EPI == EPI_TOKEN) ? a.img : nullptr;
  if constexpr (EPI == EPI_TOKEN) {
    if (gimg != nullptr) {
      const int S = a.S, gq = S >> 4;
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        const int m = min(m0 + srow + 32 * i, M - 1);
        const int ci = m / a.p, pi = m - ci * a.p;
        const int r = pi / gq, q = pi - r * gq;
        isrc[i] = gimg + ((size_t)ci * S + 16 * r + (sch >> 1)) * S + 16 * q + (sch & 1) * 8;
      }
    }
  }
  const int img_k_stride = (EPI == EPI_TOKEN && gimg != nullptr) ? (a.S * (BK / 16)) : 0;  // floats per k-tile: 4 patch rows
#define LOAD_REGS(k0)                                                                        \
  {                                                                                          \
    if (EPI == EPI_TOKEN && gimg != nullptr) {                                               \
      _Pragma("unroll") for (int i = 0; i < XCH; ++i) {                                      \
        const float* ps_ = isrc[i] + (size_t)((k0) / BK) * img_k_stride;                     \
        const f32x4 a0_ = *reinterpret_cast<const f32x4*>(ps_);                              \
        const f32x4 a1_ = *reinterpret_cast<const f32x4*>(ps_ + 4);                          \
        const bf16x8 o_ = pack8f(a0_, a1_);                                                  \
        xr[i] = __builtin_bit_cast(u32x4, o_);                                               \
      }                                                                                      \
    } else                                                                                   \
    _Pragma("unroll") for (int i = 0; i < XCH; ++i) xr[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + (k0)); \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i)                                          \
        wr[i] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)(32 * i) * ldw + (k0));       \
  }
#define WRITE_LDS()                                                                          \
  {                                                                                          \
    _Pragma("unroll") for (int i = 0; i < XCH; ++i)                                          \
        *reinterpret_cast<u32x4*>(sX + (srow + 32 * i) * LDK + sch * 8) = xr[i];             \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i)                                          \
        *reinterpret_cast<u32x4*>(sW + (srow + 32 * i) * LDK + sch * 8) = wr[i];             \
  }

  f32x4 acc[NB][MB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  LOAD_REGS(0);
  const bf16_t* xbase = sX + (wm * TM + (l & 15)) * LDK + (l >> 4) * 8;
  const bf16_t* wbase = sW + (wn * TN + (l & 15)) * LDK + (l >> 4) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    WRITE_LDS();
    __syncthreads();
    if (kt + 1 < nk) LOAD_REGS((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 32) {
      bf16x8 xf[MB], wf[NB];
#pragma unroll
      for (int j = 0; j < MB; ++j) xf[j] = lds_read8(xbase + j * 16 * LDK + kk);
#pragma unroll
      for (int i = 0; i < NB; ++i) wf[i] = lds_read8(wbase + i * 16 * LDK + kk);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue.  The MFMA result has lane = (row m = l&15, 4 consecutive n); writing that straight out gives 8-byte
  // pieces scattered over 16 rows per instruction.  Instead each wave transposes 16 rows at a time through its own LDS
  // slab (fp32, row stride TN+4 floats -> conflict-free 16-byte writes) and reads them back as 8 consecutive n per lane:
  // bias / residual / mask are then 16-32 byte coalesced reads and the output is written as full 16-byte row segments.
#undef LOAD_REGS
#undef WRITE_LDS
  constexpr int STG = TN + 4;        // floats per staged row
  constexpr int CH = TN / 8;         // 8-float chunks per row
  constexpr int CPL = 16 * CH / 64;  // chunks per lane per 16-row pass
  static_assert(4 * 16 * STG * 4 <= (BM + BN) * LDK * 2, "epilogue staging must fit the tile buffers");
  float* stage = reinterpret_cast<float*>(smem) + w * 16 * STG;
  const int g = l >> 4, li = l & 15;
  f32x4 bb[CPL][2];   // the bias of the lane's column chunks (the same in every 16-row pass)
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int n = n0 + wn * TN + ((l + 64 * cc) % CH) * 8;
    bb[cc][0] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    bb[cc][1] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) asm volatile("" ::"v"(bb[cc][0]), "v"(bb[cc][1]));   // waited for HERE, once, not inside every row guard below
  // ... and the residual / mask / pre-activation pieces of the whole tile, for the same reason (rows past M: row M - 1, not stored)
  constexpr bool HAS_AUX = EPI == EPI_RESID || EPI == EPI_RELUMASK || EPI == EPI_GELUBWD;
  bf16x8 ax[HAS_AUX ? MB : 1][HAS_AUX ? CPL : 1];
  if constexpr (HAS_AUX) {
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int id = l + 64 * cc;
        const int m = min(m0 + wm * TM + j * 16 + id / CH, M - 1);
        ax[j][cc] = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)m * a.ldaux + n0 + wn * TN + (id % CH) * 8);
      }
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) asm volatile("" ::"v"(ax[j][cc]));
  }
#pragma unroll
  for (int j = 0; j < MB; ++j) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(stage + li * STG + i * 16 + 4 * g) = acc[i][j];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
      const int id = l + 64 * cc, row = id / CH, ch = id % CH;
      const int m = m0 + wm * TM + j * 16 + row;
      const int n = n0 + wn * TN + ch * 8;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * STG + ch * 8);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * STG + ch * 8 + 4);
      if (m < M) epi_store<EPI, OUT_F32>(a, v0, v1, bb[cc][0], bb[cc][1], ax[HAS_AUX ? j : 0][HAS_AUX ? cc : 0], m, n, N);
    }
  }
}



// ---------------------------------------------------------------------------------------------------------------
// Long-K variant (K >= 512: FFN2, dx1, dh, head GEMMs): tiles go global -> LDS directly with buffer_load_dwordx4 ... lds
// (LDS-DMA: no staging VGPRs, no ds_write pass), two LDS stages, ONE barrier per k-tile; the loads of tile t+1 are in
// flight while tile t is multiplied.  The DMA writes lane-linear (wave-uniform base + lane*16 B), so the LDS image is
// unpadded [row][64]; bank conflicts are removed by an XOR swizzle applied on the SOURCE address (lane i of row r fetches
// 16-byte chunk c ^ ((r>>1)&7)) and mirrored on the fragment reads -- conflict-free for the ds_read_b128 lane groups.
//
// The DMA issue of tile t+1 and the fragment reads + MFMAs of tile t live in ONE function with __restrict__ pointers:
// after inlining, the LDS reads carry scoped-noalias metadata against the DMA, so the compiler's waitcnt insertion does
// not drain the DMA queue (s_waitcnt vmcnt(0)) in front of the first LDS read -- it cannot tell the two stages apart by
// itself and would otherwise serialise every tile's load with the previous tile's math.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, bool COMPUTE>
__device__ __forceinline__ void glds_step(BufRsrc gx, BufRsrc gw, unsigned kbytes,
                                          bf16_t* __restrict__ dst, const bf16_t* __restrict__ st, bool issue,
                                          const unsigned (&xoff)[BM / 32], const unsigned (&woff)[BN / 32], int w, int xrow,
                                          int wrow, int g, int sw, f32x4 (&acc)[BN / 32][BM / 32]) {
  constexpr int MB = BM / 32, NB = BN / 32, XI = BM / 32, WI = BN / 32;
  if (issue) {
    bf16_t* sx_ = dst + (w * XI) * 512;
    bf16_t* sw_ = dst + BM * BK + (w * WI) * 512;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      lds_dma16(gx, sx_ + i * 512, xoff[i], kbytes);
#pragma unroll
    for (int i = 0; i < WI; ++i)
      lds_dma16(gw, sw_ + i * 512, woff[i], kbytes);
  }
  __builtin_amdgcn_sched_barrier(0);  // the DMA goes out FIRST: free of the alias edge, the scheduler may sink it below the math
  if constexpr (COMPUTE) {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 32) {
      const int ch = (((kk >> 3) + g) ^ sw) * 8;
      bf16x8 xf[MB], wf[NB];
#pragma unroll
      for (int j = 0; j < MB; ++j) xf[j] = lds_read8(st + xrow + j * 16 * BK + ch);
#pragma unroll
      for (int i = 0; i < NB; ++i) wf[i] = lds_read8(st + wrow + i * 16 * BK + ch);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
    }
  }
}

template <int BM, int BN, int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(NtArgs a) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int MB = TM / 16, NB = TN / 16;
  constexpr int STAGE = (BM + BN) * BK;  // bf16 elements per stage
  constexpr int XI = BM / 32, WI = BN / 32;  // LDS-DMA instructions per wave per tile (8 rows x 128 B each)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const bf16_t* __restrict__ gX = a.X;
  const bf16_t* __restrict__ gW = a.W;
  const int M = a.M, N = a.N, K = a.K, ldx = a.ldx, ldw = a.ldw;
  const int tiles_n = N / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM;
  const int n0 = (bid % tiles_n) * BN;

  // per-lane DMA sources: instruction q = w*XI + i fills tile rows 8q..8q+7; lane -> (row 8q + (l>>3), LDS chunk l&7),
  // which holds the GLOBAL chunk (l&7) ^ swz(row)
  // (byte offsets relative to the tile's first row: the buffer resources are based there, see lds_dma16)
  unsigned xoff[XI], woff[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int r = 8 * (w * XI + i) + (l >> 3);
    const int cg = (l & 7) ^ ((r >> 1) & 7);
    xoff[i] = ((unsigned)(min(m0 + r, M - 1) - m0) * ldx + cg * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int r = 8 * (w * WI + i) + (l >> 3);
    const int cg = (l & 7) ^ ((r >> 1) & 7);
    woff[i] = ((unsigned)r * ldw + cg * 8) * 2;
  }
  const BufRsrc xrs = make_rsrc(gX + (size_t)m0 * ldx), wrs = make_rsrc(gW + (size_t)n0 * ldw);

  f32x4 acc[NB][MB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int li = l & 15, g = l >> 4;
  const int sw = (li >> 1) & 7;  // swizzle term of this lane's fragment rows (row bases are multiples of 16)
  const int xrow = (wm * TM + li) * BK, wrow = BM * BK + (wn * TN + li) * BK;
  const int nk = K / BK;
  glds_step<BM, BN, false>(xrs, wrs, 0u, smem, smem + STAGE, true, xoff, woff, w, xrow, wrow, g, sw, acc);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (LDS-DMA completion is only visible through the issuing wave's vmcnt) + everyone is done
    // reading the other stage
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    glds_step<BM, BN, true>(xrs, wrs, (unsigned)(kt + 1) * BK * 2, smem + ((kt + 1) & 1) * STAGE, smem + (kt & 1) * STAGE,
                            kt + 1 < nk, xoff, woff, w, xrow, wrow, g, sw, acc);
  }
  __syncthreads();

  // ---- epilogue (same LDS-transpose scheme as gemm_nt_kernel)
  constexpr int STG = TN + 4, CH = TN / 8, CPL = 16 * CH / 64;
  static_assert(4 * 16 * STG * 4 <= 2 * STAGE * 2, "epilogue staging must fit the stage buffers");
  float* stage = reinterpret_cast<float*>(smem) + w * 16 * STG;
  f32x4 bb[CPL][2];   // the bias of the lane's column chunks (the same in every 16-row pass)
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int n = n0 + wn * TN + ((l + 64 * cc) % CH) * 8;
    bb[cc][0] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    bb[cc][1] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) asm volatile("" ::"v"(bb[cc][0]), "v"(bb[cc][1]));   // waited for HERE, once, not inside every row guard below
  // ... and the residual / mask / pre-activation pieces of the whole tile, for the same reason (rows past M: row M - 1, not stored)
  constexpr bool HAS_AUX = EPI == EPI_RESID || EPI == EPI_RELUMASK || EPI == EPI_GELUBWD;
  bf16x8 ax[HAS_AUX ? MB : 1][HAS_AUX ? CPL : 1];
  if constexpr (HAS_AUX) {
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int id = l + 64 * cc;
        const int m = min(m0 + wm * TM + j * 16 + id / CH, M - 1);
        ax[j][cc] = *reinterpret_cast<const bf16x8*>(a.aux + (size_t)m * a.ldaux + n0 + wn * TN + (id % CH) * 8);
      }
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) asm volatile("" ::"v"(ax[j][cc]));
  }
#pragma unroll
  for (int j = 0; j < MB; ++j) {
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(stage + li * STG + i * 16 + 4 * g) = acc[i][j];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
      const int id = l + 64 * cc, row = id / CH, ch = id % CH;
      const int m = m0 + wm * TM + j * 16 + row;
      const int n = n0 + wn * TN + ch * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * STG + ch * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * STG + ch * 8 + 4);
      if (m < M) epi_store<EPI, OUT_F32>(a, v0, v1, bb[cc][0], bb[cc][1], ax[HAS_AUX ? j : 0][HAS_AUX ? cc : 0], m, n, N);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small-K variant (K = KD <= 256: the D=192 projections QKV, FFN1, dH).  With so few k-tiles the generic kernel is
// latency-bound (one exposed memory round trip per 64-wide k-tile).  Here a block owns a 128-row panel for a whole
// range of N: each wave keeps the X fragments of its 32 rows x KD in REGISTERS for the lifetime of the block; only the
// W tiles (BN x KD, L2-resident weights) stream through LDS, prefetched one tile ahead.
// The output stream is the HBM-bound part, and HBM write latency under load is microseconds, so enough bytes must be
// in flight (Little): a tile's results are converted, parked in registers and STORED AT THE TOP OF THE NEXT ITERATION,
// right after the barrier -- they then have the whole next MFMA phase to drain before the next vmcnt(0) (vmcnt is
// in-order and counts stores on CDNA4, so stores issued just before a load wait would be waited for as well).
// ---------------------------------------------------------------------------------------------------------------
template <int KD, int BN, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_smallk_kernel(NtArgs a, int n_per_item) {
  constexpr int KS = KD / 32, LDW = KD + 16, NB = BN / 16;  // (row stride = 32 B x odd, as LDK)
  constexpr int CPR = KD / 8;                 // 16-byte chunks per W row
  constexpr int WCH = BN * CPR / 256;         // chunks of a W tile per thread
  constexpr int STG = BN + 4, CH = BN / 8, CPL = 16 * CH / 64;
  constexpr bool HAS_AUX = (EPI == EPI_RESID || EPI == EPI_RELUMASK || EPI == EPI_GELUBWD);
  constexpr int MAXN = 2048;                  // widest N range one block sweeps (bias slice kept in LDS)
  static_assert((BN * CPR) % 256 == 0 && (16 * CH) % 64 == 0, "tile must split evenly");
  __shared__ __attribute__((aligned(16))) bf16_t sW[BN * LDW];
  __shared__ __attribute__((aligned(16))) float sStage[4 * 16 * STG];
  __shared__ __attribute__((aligned(16))) float sBias[MAXN];

  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  const int M = a.M, N = a.N;
  const int items_n = N / n_per_item;
  const int panel = blockIdx.x / items_n;
  const int nbeg = (blockIdx.x % items_n) * n_per_item;
  const int m0 = panel * 128 + w * 32;
  const bf16_t* __restrict__ gX = a.X;
  const bf16_t* __restrict__ gW = a.W;
  const bf16_t* __restrict__ aux = a.aux;
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.Out);
  const int ldx = a.ldx, ldw = a.ldw, ldaux = a.ldaux, ldo = a.ldo;

  for (int i = tid; i < n_per_item; i += 256) sBias[i] = a.bias ? a.bias[nbeg + i] : 0.f;

  bf16x8 xf[2][KS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int mr = min(m0 + mb * 16 + li, M - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[mb][ks] = *reinterpret_cast<const bf16x8*>(gX + (size_t)mr * ldx + ks * 32 + g * 8);
  }
  // per-lane epilogue coordinates (fixed for the block): chunk cc of pass mb -> (row, 8-column chunk)
  int erow[CPL], ech[CPL];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int id = l + 64 * cc;
    erow[cc] = id / CH;
    ech[cc] = id % CH;
  }
  u32x4 wr[WCH];
  bf16x8 pend[2][CPL];   // finished outputs of the previous tile, stored one iteration late
  bf16x8 auxr[2][CPL];
  float* stage = sStage + w * 16 * STG;
  const int ntiles = n_per_item / BN;
#define LOAD_W(n0)                                                                          \
  _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                         \
    const int id = tid + 256 * i, row = id / CPR, ch = id % CPR;                            \
    wr[i] = *reinterpret_cast<const u32x4*>(gW + (size_t)((n0) + row) * ldw + ch * 8);      \
  }
  LOAD_W(nbeg);
  // The sweep is branch-free so the compiler can COUNT outstanding VMEM ops (in-order vmcnt) instead of draining to
  // zero.  Rows past M (ragged last panel) are clamped to row M-1 for loads AND stores: their X / aux operands are row
  // M-1's, hence their results are bit-identical to row M-1's and the duplicate stores are benign.
  {
    for (int jt = 0; jt < ntiles; ++jt) {
      const int n0 = nbeg + jt * BN;
#pragma unroll
      for (int i = 0; i < WCH; ++i) {
        const int id = tid + 256 * i, row = id / CPR, ch = id % CPR;
        *reinterpret_cast<u32x4*>(sW + row * LDW + ch * 8) = wr[i];
      }
      __syncthreads();
      if constexpr (HAS_AUX) {  // this tile's epilogue operands first (oldest in the in-order queue) ...
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int cc = 0; cc < CPL; ++cc) {
            const int m = min(m0 + mb * 16 + erow[cc], M - 1);
            auxr[mb][cc] = *reinterpret_cast<const bf16x8*>(aux + (size_t)m * ldaux + n0 + ech[cc] * 8);
          }
      }
      {  // ... then the next W tile (the last iteration re-reads its own tile: keeps the body branch-free) ...
        const int nn = (jt + 1 < ntiles) ? n0 + BN : n0;
        LOAD_W(nn);
      }
      if (jt > 0) {  // ... and only then the parked stores of the previous tile, so no load queues behind them
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int cc = 0; cc < CPL; ++cc) {
            const int m = min(m0 + mb * 16 + erow[cc], M - 1);
            *reinterpret_cast<bf16x8*>(out + (size_t)m * ldo + (n0 - BN) + ech[cc] * 8) = pend[mb][cc];
          }
      }
      f32x4 acc[NB][2];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bf16x8 wf = lds_read8(sW + (nb * 16 + li) * LDW + ks * 32 + g * 8);
          acc[nb][0] = mfma16(wf, xf[0][ks], acc[nb][0]);
          acc[nb][1] = mfma16(wf, xf[1][ks], acc[nb][1]);
        }
      // epilogue math through the wave's private LDS slab; results parked in `pend`
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f32x4*>(stage + li * STG + nb * 16 + 4 * g) = acc[nb][mb];
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          const int row = erow[cc], ch = ech[cc];
          f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * STG + ch * 8);
          f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * STG + ch * 8 + 4);
          v0 += *reinterpret_cast<const f32x4*>(sBias + (n0 - nbeg) + ch * 8);
          v1 += *reinterpret_cast<const f32x4*>(sBias + (n0 - nbeg) + ch * 8 + 4);
          if constexpr (EPI == EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] = relu_f(v0[r]); v1[r] = relu_f(v1[r]); }
          } else if constexpr (EPI == EPI_GELU) {
            const int m = min(m0 + mb * 16 + row, M - 1);
            bf16x8 pre;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pre[r] = (bf16_t)v0[r]; pre[4 + r] = (bf16_t)v1[r]; }
            *reinterpret_cast<bf16x8*>(a.aux_out + (size_t)m * ldaux + n0 + ch * 8) = pre;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] = gelu_erf(v0[r]); v1[r] = gelu_erf(v1[r]); }
          } else if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] += (float)auxr[mb][cc][r]; v1[r] += (float)auxr[mb][cc][4 + r]; }
          } else if constexpr (EPI == EPI_RELUMASK) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v0[r] = ((float)auxr[mb][cc][r] > 0.f) ? v0[r] : 0.f;
              v1[r] = ((float)auxr[mb][cc][4 + r] > 0.f) ? v1[r] : 0.f;
            }
          } else if constexpr (EPI == EPI_GELUBWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v0[r] *= gelu_erf_grad((float)auxr[mb][cc][r]);
              v1[r] *= gelu_erf_grad((float)auxr[mb][cc][4 + r]);
            }
          }
          bf16x8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)v0[r]; o[4 + r] = (bf16_t)v1[r]; }
          pend[mb][cc] = o;
        }
      }
      __syncthreads();
    }
    const int nl = nbeg + (ntiles - 1) * BN;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int m = min(m0 + mb * 16 + erow[cc], M - 1);
        *reinterpret_cast<bf16x8*>(out + (size_t)m * ldo + nl + ech[cc] * 8) = pend[mb][cc];
      }
  }
#undef LOAD_W
}

template <int EPI, bool OUT_F32>
int launch_nt(const NtArgs& a, hipStream_t s) {
  constexpr int BM = 128;
  const int tm = (a.M + BM - 1) / BM;
  if constexpr (EPI != EPI_TOKEN && !OUT_F32) {
    // Small (D = 384): FFN1 with the X fragments (2 x 12 k-steps) in registers: 650 -> 577 us.  Only the ReLU epilogue: the
    // ones that carry an aux operand spill at this K (dH 858 -> 1071 us) and QKV does not gain (400 -> 420 us).
    if (EPI == EPI_RELU && a.K == 384 && a.N >= 1024 && a.N % 64 == 0) {
      int n_per_item = a.N;
      while (n_per_item > 2048 && n_per_item % 128 == 0) n_per_item /= 2;
      while (n_per_item % 128 == 0 && n_per_item > 256 && (long long)tm * (a.N / n_per_item) < 3072) n_per_item /= 2;
      hipLaunchKernelGGL((gemm_nt_smallk_kernel<384, 64, EPI>), dim3(tm * (a.N / n_per_item)), dim3(256), 0, s, a, n_per_item);
      CHADA_CHECK_LAUNCH();
      return 0;
    }
    if (a.K == 192 && a.N >= 192 && a.N % 64 == 0) {
      // how much of N one block sweeps: enough work items to balance 256 CUs x 2 resident blocks, each <= 2048 wide
      int n_per_item = a.N;
      while (n_per_item > 2048 && n_per_item % 128 == 0) n_per_item /= 2;
      while (n_per_item % 128 == 0 && n_per_item > 256 && (long long)tm * (a.N / n_per_item) < 3072) n_per_item /= 2;
      hipLaunchKernelGGL((gemm_nt_smallk_kernel<192, 64, EPI>), dim3(tm * (a.N / n_per_item)), dim3(256), 0, s, a, n_per_item);
      CHADA_CHECK_LAUNCH();
      return 0;
    }
  }
  const bool longk = (a.K >= 512) && (EPI != EPI_TOKEN);
#define NT_LAUNCH(BNV)                                                                                               \
  if (longk) hipLaunchKernelGGL((gemm_nt_glds_kernel<BM, BNV, EPI, OUT_F32>), dim3(tm * (a.N / BNV)), dim3(256), 0, s, a); \
  else hipLaunchKernelGGL((gemm_nt_kernel<BM, BNV, EPI, OUT_F32>), dim3(tm * (a.N / BNV)), dim3(256), 0, s, a);
  static const int wide192 = getenv("CHADA_NT_BN192") ? atoi(getenv("CHADA_NT_BN192")) : 1;
  if ((a.N == 384 || (wide192 && a.N % 192 == 0 && a.N >= 768)) && longk) {
    // N = 384: two 192-wide tiles per row panel instead of three 128-wide ones.  N = 768, 2304 (Base): a 128 x 128 tile stages
    // 64 B/clk of operands at the matrix pipe's pace -- more than the CU's vector-memory path delivers (~58 B/clk through LDS-DMA);
    // 128 x 192 needs 53
    NT_LAUNCH(192)
  } else if (a.N % 128 == 0) {
    NT_LAUNCH(128)
  } else if (a.N % 192 == 0) {
    NT_LAUNCH(192)
  } else {
    NT_LAUNCH(64)
  }
#undef NT_LAUNCH
  CHADA_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int chadavit_gemm_nt(const chada_bf16* X, int ldx, const chada_bf16* W, int ldw, void* Out, int ldo, int M,
                                int N, int K, const float* bias, int epilogue, const chada_bf16* aux, int ldaux,
                                chada_bf16* aux_out, int out_fp32, void* stream) {
  CHADA_ENTRY();
  if (!X || !W || !Out || M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % BK != 0 || N % 64 != 0 || ldx % 8 != 0 || ldw % 8 != 0 || ldo % 8 != 0) return 2;
  if ((epilogue == EPI_RESID || epilogue == EPI_RELUMASK || epilogue == EPI_GELUBWD) && (!aux || ldaux % 8 != 0)) return 1;
  if (epilogue == EPI_GELU && (!aux_out || ldaux % 8 != 0)) return 1;
  NtArgs a{};
  a.X = reinterpret_cast<const bf16_t*>(X);
  a.W = reinterpret_cast<const bf16_t*>(W);
  a.Out = Out;
  a.bias = bias;
  a.aux = reinterpret_cast<const bf16_t*>(aux);
  a.aux_out = reinterpret_cast<bf16_t*>(aux_out);
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldo = ldo; a.ldaux = ldaux;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (out_fp32) {
    if (epilogue != EPI_NONE) return 2;
    return launch_nt<EPI_NONE, true>(a, s);
  }
  switch (epilogue) {
    case EPI_NONE: return launch_nt<EPI_NONE, false>(a, s);
    case EPI_RELU: return launch_nt<EPI_RELU, false>(a, s);
    case EPI_GELU: return launch_nt<EPI_GELU, false>(a, s);
    case EPI_RESID: return launch_nt<EPI_RESID, false>(a, s);
    case EPI_RELUMASK: return launch_nt<EPI_RELUMASK, false>(a, s);
    case EPI_GELUBWD: return launch_nt<EPI_GELUBWD, false>(a, s);
    default: return 1;
  }
}

extern "C" int chadavit_tokenizer_gemm(const chada_bf16* patches, const chada_bf16* Wp, const float* bias,
                                       const float* pos, const float* chan, const int* chan_img, const int* chan_idx,
                                       chada_bf16* tokens, int Mp, int D, int K, int p, void* stream) {
  CHADA_ENTRY();
  if (!patches || !Wp || !pos || !chan_img || !chan_idx || !tokens || Mp <= 0 || p <= 0) return 1;
  if (K % BK != 0 || D % 64 != 0) return 2;
  NtArgs a{};
  a.X = reinterpret_cast<const bf16_t*>(patches);
  a.W = reinterpret_cast<const bf16_t*>(Wp);
  a.Out = tokens;
  a.bias = bias;
  a.M = Mp; a.N = D; a.K = K; a.ldx = K; a.ldw = K; a.ldo = D;
  a.pos = pos; a.chan = chan; a.chan_img = chan_img; a.chan_idx = chan_idx; a.p = p;
  return launch_nt<EPI_TOKEN, false>(a, reinterpret_cast<hipStream_t>(stream));
}

// The same with the conv unfold folded into the GEMM (no im2col buffer): x [n_chan, S, S] fp32, 16 x 16 patches (K = 256).
extern "C" int chadavit_tokenizer_fused(const float* x, const chada_bf16* Wp, const float* bias, const float* pos, const float* chan,
                                        const int* chan_img, const int* chan_idx, chada_bf16* tokens, int n_chan, int S, int D, int p,
                                        void* stream) {
  CHADA_ENTRY();
  if (!x || !Wp || !pos || !chan_img || !chan_idx || !tokens || n_chan <= 0 || S <= 0 || p <= 0) return 1;
  if (S % 16 != 0 || p != (S / 16) * (S / 16) || D % 64 != 0 || ((uintptr_t)x & 15) != 0) return 2;
  NtArgs a{};
  a.X = reinterpret_cast<const bf16_t*>(x);  // unused: the staging reads a.img
  a.W = reinterpret_cast<const bf16_t*>(Wp);
  a.Out = tokens;
  a.bias = bias;
  a.M = n_chan * p; a.N = D; a.K = 256; a.ldx = 256; a.ldw = 256; a.ldo = D;
  a.pos = pos; a.chan = chan; a.chan_img = chan_img; a.chan_idx = chan_idx; a.p = p;
  a.img = x; a.S = S;
  return launch_nt<EPI_TOKEN, false>(a, reinterpret_cast<hipStream_t>(stream));
}
