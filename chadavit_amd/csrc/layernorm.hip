// LayerNorm forward / backward over packed token rows (HBM-bound side kernels).
// One wave per row; a lane owns the same 4-column chunks (c = 4*lane + 256*it) for every row it
// visits, so gamma/beta live in registers and the dgamma/dbeta partial sums of the backward are
// accumulated in registers across rows, then combined across the block's 4 waves through LDS and
// finished by a small deterministic reduce kernel.  Statistics in fp32, biased variance.
#include "common.h"

using namespace chada;

namespace {

constexpr int LN_BWD_PARTIALS = 1024;

template <int NIT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int T,
                                                     int D, float eps) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 gm[NIT], bt[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * l + 256 * it;
    if (c < D) {
      gm[it] = *reinterpret_cast<const f32x4*>(gamma + c);
      bt[it] = *reinterpret_cast<const f32x4*>(beta + c);
    } else {
      gm[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      bt[it] = gm[it];
    }
  }
  const float invD = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + w; row < T; row += gridDim.x * 4) {
    f32x4 v[NIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        const bf16x4 xv = *reinterpret_cast<const bf16x4*>(x + (size_t)row * D + c);
        v[it] = f32x4{(float)xv[0], (float)xv[1], (float)xv[2], (float)xv[3]};
        s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
      } else {
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = v[it][k] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        f32x4 o = (v[it] - mean) * rstd * gm[it] + bt[it];
        *reinterpret_cast<bf16x4*>(y + (size_t)row * D + c) = pack4(o[0], o[1], o[2], o[3]);
      }
    }
    if (l == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}


// Two chained LayerNorms in one pass: x2 = LN_a(z) (the block's norm2) and h = LN_b(x2) (the NEXT block's norm1, whose
// input is exactly x2).  z is read once; x2 is rounded to bf16 before the second normalisation so the result is bit-identical
// to two separate launches.  Saves one full read of x2 and one launch per block.
template <int NIT>
__global__ __launch_bounds__(256) void ln_fwd2_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ga,
                                                      const float* __restrict__ ba, const float* __restrict__ gb,
                                                      const float* __restrict__ bb, bf16_t* __restrict__ y1,
                                                      bf16_t* __restrict__ y2, float* __restrict__ mean1,
                                                      float* __restrict__ rstd1, float* __restrict__ mean2,
                                                      float* __restrict__ rstd2, int T, int D, float eps_a, float eps_b) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 g1[NIT], b1[NIT], g2[NIT], b2[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * l + 256 * it;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    g1[it] = c < D ? *reinterpret_cast<const f32x4*>(ga + c) : z4;
    b1[it] = c < D ? *reinterpret_cast<const f32x4*>(ba + c) : z4;
    g2[it] = c < D ? *reinterpret_cast<const f32x4*>(gb + c) : z4;
    b2[it] = c < D ? *reinterpret_cast<const f32x4*>(bb + c) : z4;
  }
  const float invD = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + w; row < T; row += gridDim.x * 4) {
    f32x4 v[NIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        const bf16x4 xv = *reinterpret_cast<const bf16x4*>(x + (size_t)row * D + c);
        v[it] = f32x4{(float)xv[0], (float)xv[1], (float)xv[2], (float)xv[3]};
        s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
      } else {
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float m1 = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (4 * l + 256 * it < D)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = v[it][k] - m1; q += d * d; }
    const float r1 = rsqrtf(wave_sum(q) * invD + eps_a);
    float s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        const f32x4 o = (v[it] - m1) * r1 * g1[it] + b1[it];
        const bf16x4 ob = pack4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<bf16x4*>(y1 + (size_t)row * D + c) = ob;
        v[it] = f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};  // second LN sees the bf16-rounded x2
        s2 += v[it][0] + v[it][1] + v[it][2] + v[it][3];
      }
    }
    const float m2 = wave_sum(s2) * invD;
    float q2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (4 * l + 256 * it < D)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = v[it][k] - m2; q2 += d * d; }
    const float r2 = rsqrtf(wave_sum(q2) * invD + eps_b);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        const f32x4 o = (v[it] - m2) * r2 * g2[it] + b2[it];
        *reinterpret_cast<bf16x4*>(y2 + (size_t)row * D + c) = pack4(o[0], o[1], o[2], o[3]);
      }
    }
    if (l == 0) {
      if (mean1) { mean1[row] = m1; rstd1[row] = r1; }
      if (mean2) { mean2[row] = m2; rstd2[row] = r2; }
    }
  }
}

template <int NIT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const bf16_t* __restrict__ dres,
                                                     bf16_t* __restrict__ dx, float* __restrict__ partial, int T, int D) {
  __shared__ float red[4 * 2 * 256 * NIT];  // [wave][dgamma|dbeta][NIT*256]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 gm[NIT], dg[NIT], db[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * l + 256 * it;
    gm[it] = c < D ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + w; row < T; row += gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NIT], gg[NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        const bf16x4 xv = *reinterpret_cast<const bf16x4*>(x + (size_t)row * D + c);
        const bf16x4 dv = *reinterpret_cast<const bf16x4*>(dy + (size_t)row * D + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xn = ((float)xv[k] - mu) * rs;
          const float d = (float)dv[k];
          const float g = d * gm[it][k];
          xh[it][k] = xn;
          gg[it][k] = g;
          s1 += g;
          s2 += g * xn;
          dg[it][k] += d * xn;
          db[it][k] += d;
        }
      } else {
        xh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        gg[it] = xh[it];
      }
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = 4 * l + 256 * it;
      if (c < D) {
        f32x4 o = (gg[it] - s1 - xh[it] * s2) * rs;
        if (dres) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(dres + (size_t)row * D + c);
          o += f32x4{(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
        }
        *reinterpret_cast<bf16x4*>(dx + (size_t)row * D + c) = pack4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  // block reduce of the per-wave column partials
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = 4 * l + 256 * it;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[(w * 2 + 0) * 256 * NIT + c + k] = dg[it][k];
      red[(w * 2 + 1) * 256 * NIT + c + k] = db[it][k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      a += red[(ww * 2 + 0) * 256 * NIT + c];
      b += red[(ww * 2 + 1) * 256 * NIT + c];
    }
    partial[(size_t)blockIdx.x * 2 * D + c] = a;
    partial[(size_t)blockIdx.x * 2 * D + D + c] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// D = 192 instances ("quad-row"): a wave handles FOUR rows at a time, 16 lanes per row, 12 columns per lane laid out as
// c = 64*j + 4*sub + k (j < 3, k < 4), so each of the three 8-byte loads of a tensor is a fully coalesced 128-byte segment
// per row.  Versus one row per wave (48 of 64 lanes active at D = 192): 4x the rows -- and bytes -- in flight per wave
// (these kernels are latency-bound at ~2.5-3 TB/s with one row per wave), all lanes busy, and the row reductions stay
// inside a 16-lane DPP row (4 v_add_dpp, no cross-row exchange).
// ---------------------------------------------------------------------------------------------------------------

// QD in {192, 384, 768}: LPR = QD / 12 lanes per row (16 / 32 / 64), RPW = 64 / LPR rows per wave (4 / 2 / 1)
template <int QD>
struct QuadRow {
  static constexpr int LPR = QD / 12, RPW = 64 / LPR;
  static_assert(QD == 192 || QD == 384 || QD == 768, "12 columns per lane");
  int sub, rg;
  __device__ __forceinline__ QuadRow() : sub(threadIdx.x & (LPR - 1)), rg((threadIdx.x & 63) / LPR) {}
  __device__ __forceinline__ int col(int j) const { return 4 * LPR * j + 4 * sub; }
  // sum over the LPR lanes of a row, result in every lane of the row
  static __device__ __forceinline__ float rsum(float v) {
    v = row16_sum(v);
    if constexpr (LPR >= 32) { float a, b; swap16(v, a, b); v = a + b; }
    if constexpr (LPR >= 64) { float a, b; swap32(v, a, b); v = a + b; }
    return v;
  }
};

__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ float hsum(const f32x4& v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// explicit operation order / fusion, so that every instance rounds identically (ln_fwd2 must equal two ln_fwd passes bit for bit)
__device__ __forceinline__ float sqdev(const f32x4 (&v)[3], float mean) {
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = v[j][k] - mean;
      q = __builtin_fmaf(d, d, q);
    }
  return q;
}
__device__ __forceinline__ f32x4 ln_apply(const f32x4& v, float mean, float rstd, const f32x4& g, const f32x4& b) {
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf((v[k] - mean) * rstd, g[k], b[k]);
  return o;
}

// OCP-MX quantisation of a freshly normalised row, for the fp8 weight path (gemm_mx8.hip): the lane's four consecutive columns of
// group j belong to the 32-column block of its aligned group of EIGHT lanes -- block maximum by three DPP steps, then the same scale
// rule, conversion and layout as mx8_quantize_kernel (bit-identical to quantising the stored bf16 row in a separate pass).
template <int QD>
__device__ __forceinline__ void mx_emit_row(const bf16x4 (&ob)[3], const QuadRow<QD>& q, uint8_t* __restrict__ yq, uint8_t* __restrict__ ys,
                                            int lds, int row, bool live) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float f[4], amax = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[k] = (float)ob[j][k]; amax = fmaxf(amax, fabsf(f[k])); }
    amax = fmaxf(amax, dpp_mov<0xB1>(amax));   // lane ^ 1
    amax = fmaxf(amax, dpp_mov<0x4E>(amax));   // lane ^ 2
    amax = fmaxf(amax, dpp_mov<0x141>(amax));  // row_half_mirror: the other quad of the eight
    int e8 = 127;
    float inv = 1.f;
    if (amax > 0.f) {
      int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xff) - 127 - 8;
      if (amax * __builtin_bit_cast(float, (unsigned)(127 - ex) << 23) > 448.f) ex += 1;
      ex = max(-127, min(127, ex));
      e8 = ex + 127;
      inv = __builtin_bit_cast(float, (unsigned)(127 - ex) << 23);
    }
    if (live) {
      int pk = 0;
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, pk, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, pk, true);
      *reinterpret_cast<unsigned*>(yq + (size_t)row * QD + q.col(j)) = (unsigned)pk;
      if ((q.sub & 7) == 0) ys[(size_t)(q.col(j) >> 5) * lds + row] = (uint8_t)e8;
    }
  }
}

template <int QD, bool EMITQ = false>
__global__ __launch_bounds__(256) void ln_fwd_quad_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out, int T, float eps,
                                                          uint8_t* __restrict__ yq = nullptr, uint8_t* __restrict__ ys = nullptr, int lds = 0) {
  using QR = QuadRow<QD>;
  const QR q;
  const int w = threadIdx.x >> 6;
  f32x4 gm[3], bt[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    gm[j] = *reinterpret_cast<const f32x4*>(gamma + q.col(j));
    bt[j] = *reinterpret_cast<const f32x4*>(beta + q.col(j));
  }
  constexpr float invD = 1.0f / QD;
  for (int row0 = (blockIdx.x * 4 + w) * QR::RPW; row0 < T; row0 += gridDim.x * 4 * QR::RPW) {
    const int row = row0 + q.rg;
    const bool live = row < T;
    const bf16_t* xr = x + (size_t)min(row, T - 1) * QD;
    f32x4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = ld4(xr + q.col(j));
    const float mean = QR::rsum(hsum(v[0]) + hsum(v[1]) + hsum(v[2])) * invD;
    const float rstd = rsqrtf(QR::rsum(sqdev(v, mean)) * invD + eps);
    bf16x4 ob[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 o = ln_apply(v[j], mean, rstd, gm[j], bt[j]);
      ob[j] = pack4(o[0], o[1], o[2], o[3]);
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < 3; ++j) *reinterpret_cast<bf16x4*>(y + (size_t)row * QD + q.col(j)) = ob[j];
      if (q.sub == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
      }
    }
    if constexpr (EMITQ) mx_emit_row<QD>(ob, q, yq, ys, lds, row, live);
  }
}

template <int QD, bool EMITQ = false>
__global__ __launch_bounds__(256) void ln_fwd2_quad_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ga,
                                                           const float* __restrict__ ba, const float* __restrict__ gb,
                                                           const float* __restrict__ bb, bf16_t* __restrict__ y1,
                                                           bf16_t* __restrict__ y2, float* __restrict__ mean1,
                                                           float* __restrict__ rstd1, float* __restrict__ mean2,
                                                           float* __restrict__ rstd2, int T, float eps_a, float eps_b,
                                                           uint8_t* __restrict__ yq = nullptr, uint8_t* __restrict__ ys = nullptr, int lds = 0) {
  using QR = QuadRow<QD>;
  const QR q;
  const int w = threadIdx.x >> 6;
  f32x4 g1[3], b1[3], g2[3], b2[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    g1[j] = *reinterpret_cast<const f32x4*>(ga + q.col(j));
    b1[j] = *reinterpret_cast<const f32x4*>(ba + q.col(j));
    g2[j] = *reinterpret_cast<const f32x4*>(gb + q.col(j));
    b2[j] = *reinterpret_cast<const f32x4*>(bb + q.col(j));
  }
  constexpr float invD = 1.0f / QD;
  for (int row0 = (blockIdx.x * 4 + w) * QR::RPW; row0 < T; row0 += gridDim.x * 4 * QR::RPW) {
    const int row = row0 + q.rg;
    const bool live = row < T;
    const bf16_t* xr = x + (size_t)min(row, T - 1) * QD;
    f32x4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = ld4(xr + q.col(j));
    const float m1 = QR::rsum(hsum(v[0]) + hsum(v[1]) + hsum(v[2])) * invD;
    const float r1 = rsqrtf(QR::rsum(sqdev(v, m1)) * invD + eps_a);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 o = ln_apply(v[j], m1, r1, g1[j], b1[j]);
      const bf16x4 ob = pack4(o[0], o[1], o[2], o[3]);
      if (live) *reinterpret_cast<bf16x4*>(y1 + (size_t)row * QD + q.col(j)) = ob;
      v[j] = f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};  // second LN sees the bf16-rounded x2
    }
    const float m2 = QR::rsum(hsum(v[0]) + hsum(v[1]) + hsum(v[2])) * invD;
    const float r2 = rsqrtf(QR::rsum(sqdev(v, m2)) * invD + eps_b);
    bf16x4 ob[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 o = ln_apply(v[j], m2, r2, g2[j], b2[j]);
      ob[j] = pack4(o[0], o[1], o[2], o[3]);
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < 3; ++j) *reinterpret_cast<bf16x4*>(y2 + (size_t)row * QD + q.col(j)) = ob[j];
      if (q.sub == 0) {
        if (mean1) { mean1[row] = m1; rstd1[row] = r1; }
        if (mean2) { mean2[row] = m2; rstd2[row] = r2; }
      }
    }
    if constexpr (EMITQ) mx_emit_row<QD>(ob, q, yq, ys, lds, row, live);  // the second norm's output: the next block's in_proj operand
  }
}

template <int QD>
__global__ __launch_bounds__(256) void ln_bwd_quad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const bf16_t* __restrict__ dres,
                                                          bf16_t* __restrict__ dx, float* __restrict__ partial, int T) {
  using QR = QuadRow<QD>;
  __shared__ float red[4 * QR::RPW][2][QD];  // [wave * RPW + row group][dgamma | dbeta][column]
  const QR q;
  const int w = threadIdx.x >> 6;
  f32x4 gm[3], dg[3], db[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    gm[j] = *reinterpret_cast<const f32x4*>(gamma + q.col(j));
    dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr float invD = 1.0f / QD;
  for (int row0 = (blockIdx.x * 4 + w) * QR::RPW; row0 < T; row0 += gridDim.x * 4 * QR::RPW) {
    const int row = row0 + q.rg;
    const bool live = row < T;
    const int rr = min(row, T - 1);
    const float mu = mean[rr], rs = rstd[rr];
    f32x4 xh[3], gg[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 xv = ld4(x + (size_t)rr * QD + q.col(j));
      f32x4 dv = ld4(dy + (size_t)rr * QD + q.col(j));
      if (!live) dv = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past T contribute nothing to dgamma / dbeta
      xh[j] = (xv - mu) * rs;
      gg[j] = dv * gm[j];
      s1 += hsum(gg[j]);
      s2 += hsum(gg[j] * xh[j]);
      dg[j] += dv * xh[j];
      db[j] += dv;
    }
    s1 = QR::rsum(s1) * invD;
    s2 = QR::rsum(s2) * invD;
    if (live) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 o = (gg[j] - s1 - xh[j] * s2) * rs;
        if (dres) o += ld4(dres + (size_t)row * QD + q.col(j));
        *reinterpret_cast<bf16x4*>(dx + (size_t)row * QD + q.col(j)) = pack4(o[0], o[1], o[2], o[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    *reinterpret_cast<f32x4*>(&red[w * QR::RPW + q.rg][0][q.col(j)]) = dg[j];
    *reinterpret_cast<f32x4*>(&red[w * QR::RPW + q.rg][1][q.col(j)]) = db[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * QD; c += 256) {
    const int which = c / QD, cc = c % QD;
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 4 * QR::RPW; ++g) a += red[g][which][cc];
    partial[(size_t)blockIdx.x * 2 * QD + c] = a;
  }
}

// Two chained LayerNorm backward passes in one sweep: the tail of block i's backward and the head of block i-1's,
//   dx = LN_a'(dy; x) + dres        (norm1 of block i applied to its input x, plus the residual branch's gradient)
//   dz = LN_b'(dx; z)               (norm2 of block i-1, whose output IS that input)
// dx is only ever consumed by the second pass: it stays in registers (rounded to bf16 as the two-launch chain would store it), which
// takes one write and one read of a [T, D] tensor out of every block boundary of the backward.  Partials: [a: dgamma | dbeta] in
// partial_a, [b: ...] in partial_b, same layout as ln_bwd_quad_kernel's.
// RECOMPUTE_X: x is not read but rebuilt from z as the forward produced it, x = bf16(LN_b(z)) with the saved statistics (the same
// expression, ln_apply, in every forward kernel): one stream of [T, D] less (5 -> 4).
template <int QD, bool RECOMPUTE_X>
__global__ __launch_bounds__(256) void ln_bwd_pair_quad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
                                                               const float* __restrict__ gamma_a, const bf16_t* __restrict__ dres,
                                                               const bf16_t* __restrict__ z, const float* __restrict__ mean_b,
                                                               const float* __restrict__ rstd_b, const float* __restrict__ gamma_b,
                                                               const float* __restrict__ beta_b, bf16_t* __restrict__ dz,
                                                               float* __restrict__ partial_a, float* __restrict__ partial_b, int T) {
  using QR = QuadRow<QD>;
  __shared__ float red[4 * QR::RPW][2][QD];
  const QR q;
  const int w = threadIdx.x >> 6;
  f32x4 ga[3], gb[3], bb[3], dga[3], dba[3], dgb[3], dbb[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    ga[j] = *reinterpret_cast<const f32x4*>(gamma_a + q.col(j));
    gb[j] = *reinterpret_cast<const f32x4*>(gamma_b + q.col(j));
    if constexpr (RECOMPUTE_X) bb[j] = *reinterpret_cast<const f32x4*>(beta_b + q.col(j));
    dga[j] = dba[j] = dgb[j] = dbb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr float invD = 1.0f / QD;
  for (int row0 = (blockIdx.x * 4 + w) * QR::RPW; row0 < T; row0 += gridDim.x * 4 * QR::RPW) {
    const int row = row0 + q.rg;
    const bool live = row < T;
    const int rr = min(row, T - 1);
    const float mua = mean_a[rr], rsa = rstd_a[rr], mub = mean_b[rr], rsb = rstd_b[rr];
    f32x4 xh[3], gg[3], zh[3], rv[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x4 dv = ld4(dy + (size_t)rr * QD + q.col(j));
      const f32x4 zv = ld4(z + (size_t)rr * QD + q.col(j));
      zh[j] = (zv - mub) * rsb;
      f32x4 xv;
      if constexpr (RECOMPUTE_X) {
        const f32x4 o = ln_apply(zv, mub, rsb, gb[j], bb[j]);
        const bf16x4 ob = pack4(o[0], o[1], o[2], o[3]);
        xv = f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};
      } else {
        xv = ld4(x + (size_t)rr * QD + q.col(j));
      }
      rv[j] = ld4(dres + (size_t)rr * QD + q.col(j));
      if (!live) { dv = f32x4{0.f, 0.f, 0.f, 0.f}; rv[j] = dv; }  // rows past T contribute nothing to either pair of column sums
      xh[j] = (xv - mua) * rsa;
      gg[j] = dv * ga[j];
      s1 += hsum(gg[j]);
      s2 += hsum(gg[j] * xh[j]);
      dga[j] += dv * xh[j];
      dba[j] += dv;
    }
    s1 = QR::rsum(s1) * invD;
    s2 = QR::rsum(s2) * invD;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 o = (gg[j] - s1 - xh[j] * s2) * rsa + rv[j];
      const bf16x4 ob = pack4(o[0], o[1], o[2], o[3]);
      const f32x4 dx = f32x4{(float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]};  // what the first launch would have stored
      gg[j] = dx * gb[j];
      t1 += hsum(gg[j]);
      t2 += hsum(gg[j] * zh[j]);
      dgb[j] += dx * zh[j];
      dbb[j] += dx;
    }
    t1 = QR::rsum(t1) * invD;
    t2 = QR::rsum(t2) * invD;
    if (live) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const f32x4 o = (gg[j] - t1 - zh[j] * t2) * rsb;
        *reinterpret_cast<bf16x4*>(dz + (size_t)row * QD + q.col(j)) = pack4(o[0], o[1], o[2], o[3]);
      }
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      *reinterpret_cast<f32x4*>(&red[w * QR::RPW + q.rg][0][q.col(j)]) = pass ? dgb[j] : dga[j];
      *reinterpret_cast<f32x4*>(&red[w * QR::RPW + q.rg][1][q.col(j)]) = pass ? dbb[j] : dba[j];
    }
    __syncthreads();
    float* partial = pass ? partial_b : partial_a;
    for (int c = threadIdx.x; c < 2 * QD; c += 256) {
      const int which = c / QD, cc = c % QD;
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < 4 * QR::RPW; ++g) a += red[g][which][cc];
      partial[(size_t)blockIdx.x * 2 * QD + c] = a;
    }
  }
}

// partial [nblk][2*D] -> dgamma|dbeta.  Block = 16 columns x 16 row groups (coalesced 64-byte row segments,
// 16 independent accumulation chains per column), combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int nblk, int D, int accumulate) {
  __shared__ float red[16][17];
  const int cx = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  float s = 0.f;
  if (c < 2 * D) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;  // independent chains: 4 loads in flight per thread
    int b = rg;
    for (; b + 48 < nblk; b += 64) {
      s += partial[(size_t)b * 2 * D + c];
      s1 += partial[(size_t)(b + 16) * 2 * D + c];
      s2 += partial[(size_t)(b + 32) * 2 * D + c];
      s3 += partial[(size_t)(b + 48) * 2 * D + c];
    }
    for (; b < nblk; b += 16) s += partial[(size_t)b * 2 * D + c];
    s = (s + s1) + (s2 + s3);
  }
  red[rg][cx] = s;
  __syncthreads();
  if (rg == 0 && c < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    float* dst = c < D ? dgamma + c : dbeta + (c - D);
    *dst = (accumulate ? *dst : 0.f) + t;
  }
}

}  // namespace

extern "C" int chadavit_layernorm_bwd_partials(void) { return LN_BWD_PARTIALS; }

extern "C" int chadavit_layernorm_fwd(const chada_bf16* x, const float* gamma, const float* beta, chada_bf16* y,
                                      float* mean, float* rstd, int T, int D, float eps, void* stream) {
  CHADA_ENTRY();
  if (!x || !gamma || !beta || !y || T <= 0) return 1;
  if (D % 4 != 0 || D > 1024 || D <= 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int grid = (T + 3) / 4;
  if (grid > 8192) grid = 8192;
  const int nit = (D + 255) / 256;
  const bf16_t* xx = reinterpret_cast<const bf16_t*>(x);
  bf16_t* yy = reinterpret_cast<bf16_t*>(y);
  if (D == 192 || D == 384 || D == 768) {
    const int rpb = 4 * (768 / D);  // rows per block and iteration
    int gq = (T + rpb - 1) / rpb;
    if (gq > 4096) gq = 4096;
    if (D == 192) hipLaunchKernelGGL(ln_fwd_quad_kernel<192>, dim3(gq), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, eps);
    else if (D == 384) hipLaunchKernelGGL(ln_fwd_quad_kernel<384>, dim3(gq), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, eps);
    else hipLaunchKernelGGL(ln_fwd_quad_kernel<768>, dim3(gq), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, eps);
    CHADA_CHECK_LAUNCH();
    return 0;
  }
  switch (nit) {
    case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, D, eps); break;
    case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, D, eps); break;
    case 3: hipLaunchKernelGGL(ln_fwd_kernel<3>, dim3(grid), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, D, eps); break;
    default: hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, D, eps); break;
  }
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_layernorm_bwd(const chada_bf16* dy, const chada_bf16* x, const float* mean, const float* rstd,
                                      const float* gamma, const chada_bf16* dres, chada_bf16* dx, float* dgamma,
                                      float* dbeta, int accumulate, int T, int D, float* workspace, void* stream) {
  CHADA_ENTRY();
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !workspace || T <= 0) return 1;
  if (D % 4 != 0 || D > 1024 || D <= 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int grid = (T + 3) / 4;
  if (grid > LN_BWD_PARTIALS) grid = LN_BWD_PARTIALS;
  const int nit = (D + 255) / 256;
  const bf16_t* dyy = reinterpret_cast<const bf16_t*>(dy);
  const bf16_t* xx = reinterpret_cast<const bf16_t*>(x);
  const bf16_t* rr = reinterpret_cast<const bf16_t*>(dres);
  bf16_t* dxx = reinterpret_cast<bf16_t*>(dx);
  if (D == 192 || D == 384 || D == 768) {
    const int rpb = 4 * (768 / D);
    int gq = (T + rpb - 1) / rpb;
    if (gq > LN_BWD_PARTIALS) gq = LN_BWD_PARTIALS;
    if (D == 192) hipLaunchKernelGGL(ln_bwd_quad_kernel<192>, dim3(gq), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T);
    else if (D == 384) hipLaunchKernelGGL(ln_bwd_quad_kernel<384>, dim3(gq), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T);
    else hipLaunchKernelGGL(ln_bwd_quad_kernel<768>, dim3(gq), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T);
    CHADA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, workspace, dgamma, dbeta, gq, D, accumulate);
    CHADA_CHECK_LAUNCH();
    return 0;
  }
  switch (nit) {
    case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T, D); break;
    case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(grid), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T, D); break;
    case 3: hipLaunchKernelGGL(ln_bwd_kernel<3>, dim3(grid), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T, D); break;
    default: hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, dyy, xx, mean, rstd, gamma, rr, dxx, workspace, T, D); break;
  }
  CHADA_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, workspace, dgamma, dbeta, grid, D,
                     accumulate);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_layernorm_bwd_pair(const chada_bf16* dy, const chada_bf16* x, const float* mean_a, const float* rstd_a,
                                           const float* gamma_a, const chada_bf16* dres, const chada_bf16* z, const float* mean_b,
                                           const float* rstd_b, const float* gamma_b, const float* beta_b, chada_bf16* dz, float* dgamma_a,
                                           float* dbeta_a, int accumulate_a, float* dgamma_b, float* dbeta_b, int accumulate_b, int T, int D,
                                           float* workspace, void* stream) {
  CHADA_ENTRY();
  if ((!x && !beta_b) || !dy || !mean_a || !rstd_a || !gamma_a || !dres || !z || !mean_b || !rstd_b || !gamma_b || !dz || !dgamma_a || !dbeta_a ||
      !dgamma_b || !dbeta_b || !workspace || T <= 0)
    return 1;
  if (D != 192 && D != 384 && D != 768) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpb = 4 * (768 / D);
  int gq = (T + rpb - 1) / rpb;
  if (gq > LN_BWD_PARTIALS) gq = LN_BWD_PARTIALS;
  float* pa = workspace;
  float* pb = workspace + (size_t)LN_BWD_PARTIALS * 2 * D;
#define LNPAIR(DV, RX)                                                                                                                  \
  hipLaunchKernelGGL((ln_bwd_pair_quad_kernel<DV, RX>), dim3(gq), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(dy),                 \
                     reinterpret_cast<const bf16_t*>(x), mean_a, rstd_a, gamma_a, reinterpret_cast<const bf16_t*>(dres),                \
                     reinterpret_cast<const bf16_t*>(z), mean_b, rstd_b, gamma_b, beta_b, reinterpret_cast<bf16_t*>(dz), pa, pb, T)
  if (x) { if (D == 192) LNPAIR(192, false); else if (D == 384) LNPAIR(384, false); else LNPAIR(768, false); }
  else { if (D == 192) LNPAIR(192, true); else if (D == 384) LNPAIR(384, true); else LNPAIR(768, true); }
#undef LNPAIR
  CHADA_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, pa, dgamma_a, dbeta_a, gq, D, accumulate_a);
  CHADA_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, s, pb, dgamma_b, dbeta_b, gq, D, accumulate_b);
  CHADA_CHECK_LAUNCH();
  return 0;
}

// LayerNorm forward that also emits its output as an OCP-MX fp8 operand (yq [T, D] e4m3, ys [D/32, lds] e8m0, lds >= T): what
// chadavit_mx8_quantize(y) would produce, without the pass over y.  D in {192, 384, 768}.
extern "C" int chadavit_layernorm_fwd_q(const chada_bf16* x, const float* gamma, const float* beta, chada_bf16* y, float* mean, float* rstd,
                                        void* yq, void* ys, int lds, int T, int D, float eps, void* stream) {
  CHADA_ENTRY();
  if (!x || !gamma || !beta || !y || !yq || !ys || T <= 0 || lds < T) return 1;
  if (D != 192 && D != 384 && D != 768) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpb = 4 * (768 / D);
  int gq = (T + rpb - 1) / rpb;
  if (gq > 4096) gq = 4096;
  const bf16_t* xx = reinterpret_cast<const bf16_t*>(x);
  bf16_t* yy = reinterpret_cast<bf16_t*>(y);
  uint8_t* q8 = reinterpret_cast<uint8_t*>(yq);
  uint8_t* s8 = reinterpret_cast<uint8_t*>(ys);
  if (D == 192) hipLaunchKernelGGL((ln_fwd_quad_kernel<192, true>), dim3(gq), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, eps, q8, s8, lds);
  else if (D == 384) hipLaunchKernelGGL((ln_fwd_quad_kernel<384, true>), dim3(gq), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, eps, q8, s8, lds);
  else hipLaunchKernelGGL((ln_fwd_quad_kernel<768, true>), dim3(gq), dim3(256), 0, s, xx, gamma, beta, yy, mean, rstd, T, eps, q8, s8, lds);
  CHADA_CHECK_LAUNCH();
  return 0;
}

// ... and the two chained LayerNorms with the SECOND output (the next block's norm1) emitted as an fp8 operand
extern "C" int chadavit_layernorm_fwd2_q(const chada_bf16* x, const float* gamma_a, const float* beta_a, const float* gamma_b,
                                         const float* beta_b, chada_bf16* y1, chada_bf16* y2, float* mean1, float* rstd1, float* mean2,
                                         float* rstd2, void* y2q, void* y2s, int lds, int T, int D, float eps_a, float eps_b, void* stream) {
  CHADA_ENTRY();
  if (!x || !gamma_a || !beta_a || !gamma_b || !beta_b || !y1 || !y2 || !y2q || !y2s || T <= 0 || lds < T) return 1;
  if ((mean1 == nullptr) != (rstd1 == nullptr) || (mean2 == nullptr) != (rstd2 == nullptr)) return 1;
  if (D != 192 && D != 384 && D != 768) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpb = 4 * (768 / D);
  int gq = (T + rpb - 1) / rpb;
  if (gq > 4096) gq = 4096;
  const bf16_t* xx = reinterpret_cast<const bf16_t*>(x);
  bf16_t* o1 = reinterpret_cast<bf16_t*>(y1);
  bf16_t* o2 = reinterpret_cast<bf16_t*>(y2);
  uint8_t* q8 = reinterpret_cast<uint8_t*>(y2q);
  uint8_t* s8 = reinterpret_cast<uint8_t*>(y2s);
#define LN2QQ(DV) hipLaunchKernelGGL((ln_fwd2_quad_kernel<DV, true>), dim3(gq), dim3(256), 0, s, xx, gamma_a, beta_a, gamma_b, beta_b, o1, o2, mean1, rstd1, mean2, rstd2, T, eps_a, eps_b, q8, s8, lds)
  if (D == 192) LN2QQ(192); else if (D == 384) LN2QQ(384); else LN2QQ(768);
#undef LN2QQ
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_layernorm_fwd2(const chada_bf16* x, const float* gamma_a, const float* beta_a, const float* gamma_b,
                                       const float* beta_b, chada_bf16* y1, chada_bf16* y2, float* mean1, float* rstd1,
                                       float* mean2, float* rstd2, int T, int D, float eps_a, float eps_b, void* stream) {
  CHADA_ENTRY();
  if (!x || !gamma_a || !beta_a || !gamma_b || !beta_b || !y1 || !y2 || T <= 0) return 1;
  if ((mean1 == nullptr) != (rstd1 == nullptr) || (mean2 == nullptr) != (rstd2 == nullptr)) return 1;
  if (D % 4 != 0 || D > 1024 || D <= 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int grid = (T + 3) / 4;
  if (grid > 8192) grid = 8192;
  const int nit = (D + 255) / 256;
  const bf16_t* xx = reinterpret_cast<const bf16_t*>(x);
  bf16_t* o1 = reinterpret_cast<bf16_t*>(y1);
  bf16_t* o2 = reinterpret_cast<bf16_t*>(y2);
  if (D == 192 || D == 384 || D == 768) {
    const int rpb = 4 * (768 / D);
    int gq = (T + rpb - 1) / rpb;
    if (gq > 4096) gq = 4096;
#define LN2Q(DV) hipLaunchKernelGGL(ln_fwd2_quad_kernel<DV>, dim3(gq), dim3(256), 0, s, xx, gamma_a, beta_a, gamma_b, beta_b, o1, o2, mean1, rstd1, mean2, rstd2, T, eps_a, eps_b)
    if (D == 192) LN2Q(192); else if (D == 384) LN2Q(384); else LN2Q(768);
#undef LN2Q
    CHADA_CHECK_LAUNCH();
    return 0;
  }
#define LN2_CASE(N) hipLaunchKernelGGL(ln_fwd2_kernel<N>, dim3(grid), dim3(256), 0, s, xx, gamma_a, beta_a, gamma_b, beta_b, o1, o2, mean1, rstd1, mean2, rstd2, T, D, eps_a, eps_b)
  switch (nit) {
    case 1: LN2_CASE(1); break;
    case 2: LN2_CASE(2); break;
    case 3: LN2_CASE(3); break;
    default: LN2_CASE(4); break;
  }
#undef LN2_CASE
  CHADA_CHECK_LAUNCH();
  return 0;
}
