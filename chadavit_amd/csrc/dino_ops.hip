// DINO head / loss / flat-parameter kernels (all HBM- or latency-bound; fp32 math).
//   l2norm, weight-norm            dino.py:98-111, :78-84
//   dino loss fwd+bwd, centre      losses/dino.py:69-118
//   EMA, AdamW, casts, per-tensor clip   momentum.py:63-74, base.py:67-72, dino.py:249-261
#include "common.h"

using namespace chada;

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {  // 256 threads, red[4]
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

inline int grid_for(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

// ---- F.normalize(x, dim=-1, eps=1e-12): one wave per row ----------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, bf16_t* __restrict__ y,
                                                         float* __restrict__ inv_norm, int M, int N) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * N;
    float s = 0.f;
    for (int c = l; c < N; c += 64) s += xr[c] * xr[c];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = l; c < N; c += 64) y[(size_t)row * N + c] = (bf16_t)(xr[c] * inv);
    if (l == 0) inv_norm[row] = inv;
  }
}
// dx = inv * (dy - yhat * <dy, yhat>),  yhat = x * inv
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ inv_norm, bf16_t* __restrict__ dx, int M,
                                                         int N) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
    const float inv = inv_norm[row];
    const float* xr = x + (size_t)row * N;
    const float* dr = dy + (size_t)row * N;
    float s = 0.f;
    for (int c = l; c < N; c += 64) s += dr[c] * xr[c] * inv;
    s = wave_sum(s);
    for (int c = l; c < N; c += 64) dx[(size_t)row * N + c] = (bf16_t)(inv * (dr[c] - xr[c] * inv * s));
  }
}
// ---- weight_norm (dim=0): w[p,:] = g[p] * v[p,:] / ||v[p,:]|| ; also the transposed bf16 copy ------
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             bf16_t* __restrict__ wout, bf16_t* __restrict__ wt,
                                                             float* __restrict__ inv_norm, int P, int K) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + w; row < P; row += gridDim.x * 4) {
    const float* vr = v + (size_t)row * K;
    float s = 0.f;
    for (int c = l; c < K; c += 64) s += vr[c] * vr[c];
    const float inv = 1.0f / sqrtf(wave_sum(s));
    const float sc = g[row] * inv;
    for (int c = l; c < K; c += 64) {
      const bf16_t o = (bf16_t)(vr[c] * sc);
      wout[(size_t)row * K + c] = o;
      if (wt) wt[(size_t)c * P + row] = o;
    }
    if (l == 0) inv_norm[row] = inv;
  }
}
// dv = g*inv * (dw - vhat <dw, vhat>);  dg = <dw, vhat> when the magnitudes are trained too (norm_last_layer = False; with the
// default they are frozen at 1, dino.py:83-84, and dg is NULL)
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                             const float* __restrict__ g, const float* __restrict__ inv_norm,
                                                             float* __restrict__ dv, float* __restrict__ dg, int accumulate, int P, int K) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + w; row < P; row += gridDim.x * 4) {
    const float inv = inv_norm[row], gg = g[row];
    const float* vr = v + (size_t)row * K;
    const float* dr = dw + (size_t)row * K;
    float s = 0.f;
    for (int c = l; c < K; c += 64) s += dr[c] * vr[c] * inv;
    s = wave_sum(s);
    if (dg && l == 0) dg[row] = (accumulate ? dg[row] : 0.f) + s;
    for (int c = l; c < K; c += 64) {
      const float o = gg * inv * (dr[c] - vr[c] * inv * s);
      float* dst = dv + (size_t)row * K + c;
      *dst = (accumulate ? *dst : 0.f) + o;
    }
  }
}

// ---- DINO loss: one block per image b; rows s0=student[b], s1=student[B+b], t0=teacher[b], t1=teacher[B+b]
__global__ __launch_bounds__(256) void dino_loss_kernel(const float* __restrict__ student, const float* __restrict__ teacher,
                                                        const float* __restrict__ center, float inv_ts, float inv_tt,
                                                        float* __restrict__ loss_rows, bf16_t* __restrict__ dstudent, int B,
                                                        int P, const float* __restrict__ teacher_temp_dev) {
  __shared__ float red[4];
  if (teacher_temp_dev) inv_tt = 1.0f / teacher_temp_dev[0];  // device-resident step scalar (graph-captured training step)
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* s0 = student + (size_t)b * P;
  const float* s1 = student + (size_t)(B + b) * P;
  const float* t0 = teacher + (size_t)b * P;
  const float* t1 = teacher + (size_t)(B + b) * P;
  float ms0 = -INFINITY, ms1 = -INFINITY, mt0 = -INFINITY, mt1 = -INFINITY;
  for (int c = tid; c < P; c += 256) {
    const float cc = center[c];
    ms0 = fmaxf(ms0, s0[c] * inv_ts);
    ms1 = fmaxf(ms1, s1[c] * inv_ts);
    mt0 = fmaxf(mt0, (t0[c] - cc) * inv_tt);
    mt1 = fmaxf(mt1, (t1[c] - cc) * inv_tt);
  }
  ms0 = block_max(ms0, red); ms1 = block_max(ms1, red); mt0 = block_max(mt0, red); mt1 = block_max(mt1, red);
  float zs0 = 0.f, zs1 = 0.f, zt0 = 0.f, zt1 = 0.f;
  for (int c = tid; c < P; c += 256) {
    const float cc = center[c];
    zs0 += expf(s0[c] * inv_ts - ms0);
    zs1 += expf(s1[c] * inv_ts - ms1);
    zt0 += expf((t0[c] - cc) * inv_tt - mt0);
    zt1 += expf((t1[c] - cc) * inv_tt - mt1);
  }
  zs0 = block_sum(zs0, red); zs1 = block_sum(zs1, red); zt0 = block_sum(zt0, red); zt1 = block_sum(zt1, red);
  const float lzs0 = logf(zs0), lzs1 = logf(zs1);
  const float izs0 = 1.0f / zs0, izs1 = 1.0f / zs1, izt0 = 1.0f / zt0, izt1 = 1.0f / zt1;
  const float gscale = 0.5f * inv_ts / (float)B;  // d(mean_b, 2 pairs)/ds = (softmax - q) / (2 B tau_s)
  float acc = 0.f;
  for (int c = tid; c < P; c += 256) {
    const float cc = center[c];
    const float a0 = s0[c] * inv_ts - ms0, a1 = s1[c] * inv_ts - ms1;
    const float q0 = expf((t0[c] - cc) * inv_tt - mt0) * izt0;
    const float q1 = expf((t1[c] - cc) * inv_tt - mt1) * izt1;
    acc -= q0 * (a1 - lzs1) + q1 * (a0 - lzs0);
    if (dstudent) {
      dstudent[(size_t)b * P + c] = (bf16_t)((expf(a0) * izs0 - q1) * gscale);
      dstudent[(size_t)(B + b) * P + c] = (bf16_t)((expf(a1) * izs1 - q0) * gscale);
    }
  }
  acc = block_sum(acc, red);
  if (tid == 0) loss_rows[b] = 0.5f * acc;
}

// ---- the same loss over V >= 2 student views (losses/dino.py:69-100 with `student_out.chunk(V)`: every teacher view iq < 2 against
// every student view v != iq, 2 V - 2 terms, mean): the standard-DINO multi-crop form, in which the local crops go through the head
// and reach the loss.  The reference's DINO never feeds them (its multicrop_forward returns backbone features only), so this is a
// separately flagged option, not the parity path; V = 2 is dino_loss_kernel's arithmetic.  Rows: student[v * B + b], teacher[iq * B + b].
// One block per image: teacher statistics once, then per student view its statistics and one pass for the loss terms and dL/ds_v
// (the teacher probabilities are recomputed from the logits: P floats x 2 do not fit LDS at P = 65536).
__global__ __launch_bounds__(256) void dino_loss_mc_kernel(const float* __restrict__ student, const float* __restrict__ teacher,
                                                           const float* __restrict__ center, float inv_ts, float inv_tt,
                                                           float* __restrict__ loss_rows, bf16_t* __restrict__ dstudent, int B, int V,
                                                           int P) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* t0 = teacher + (size_t)b * P;
  const float* t1 = teacher + (size_t)(B + b) * P;
  float mt0 = -INFINITY, mt1 = -INFINITY;
  for (int c = tid; c < P; c += 256) {
    const float cc = center[c];
    mt0 = fmaxf(mt0, (t0[c] - cc) * inv_tt);
    mt1 = fmaxf(mt1, (t1[c] - cc) * inv_tt);
  }
  mt0 = block_max(mt0, red); mt1 = block_max(mt1, red);
  float zt0 = 0.f, zt1 = 0.f;
  for (int c = tid; c < P; c += 256) {
    const float cc = center[c];
    zt0 += expf((t0[c] - cc) * inv_tt - mt0);
    zt1 += expf((t1[c] - cc) * inv_tt - mt1);
  }
  zt0 = block_sum(zt0, red); zt1 = block_sum(zt1, red);
  const float izt0 = 1.0f / zt0, izt1 = 1.0f / zt1;
  const float inv_terms = 1.0f / (float)(2 * V - 2);
  const float gscale = inv_terms * inv_ts / (float)B;
  float acc = 0.f;
  for (int v = 0; v < V; ++v) {
    const float* sv = student + (size_t)(v * B + b) * P;
    float ms = -INFINITY;
    for (int c = tid; c < P; c += 256) ms = fmaxf(ms, sv[c] * inv_ts);
    ms = block_max(ms, red);
    float zs = 0.f;
    for (int c = tid; c < P; c += 256) zs += expf(sv[c] * inv_ts - ms);
    zs = block_sum(zs, red);
    const float lzs = logf(zs), izs = 1.0f / zs;
    const float w0 = (v == 0) ? 0.f : 1.f, w1 = (v == 1) ? 0.f : 1.f;   // teacher view iq pairs with every student view but its own
    for (int c = tid; c < P; c += 256) {
      const float cc = center[c];
      const float a = sv[c] * inv_ts - ms;
      const float q = w0 * expf((t0[c] - cc) * inv_tt - mt0) * izt0 + w1 * expf((t1[c] - cc) * inv_tt - mt1) * izt1;
      acc -= q * (a - lzs);
      if (dstudent) dstudent[(size_t)(v * B + b) * P + c] = (bf16_t)(((w0 + w1) * expf(a) * izs - q) * gscale);
    }
  }
  acc = block_sum(acc, red);
  if (tid == 0) loss_rows[b] = acc * inv_terms;
}

// Column sums of a row-major [rows, cols] fp32 matrix (the centre's teacher-logit sum, losses/dino.py:106).  One block owns 64
// columns: 16 lanes x float4 = one 256-byte row segment per row group, 16 row groups per block each summing every 16th row
// with eight independent loads in flight, then a fixed-order LDS reduction over the row groups (deterministic; no atomics, no
// workspace).  Round 2's kernel (one thread per column looping over all rows, 16 blocks at P = 4096) took 237 us for 16 MB.
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                                       float scale, int vec) {
  __shared__ f32x4 part[16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + cl * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (vec && c0 + 3 < cols) {  // vec: cols % 4 == 0 and a 16-byte aligned base (host-checked)
    const float* base = x + c0;
    int r = rg;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (size_t)(r + 16 * u) * cols);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < rows; r += 16) acc += *reinterpret_cast<const f32x4*>(base + (size_t)r * cols);
  } else {
    for (int e = 0; e < 4; ++e)
      if (c0 + e < cols)
        for (int r = rg; r < rows; r += 16) acc[e] += x[(size_t)r * cols + c0 + e];
  }
  part[rg][cl] = acc;
  __syncthreads();
  if (rg == 0) {
    f32x4 t = part[0][cl];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += part[g][cl];
    for (int e = 0; e < 4; ++e)
      if (c0 + e < cols) out[c0 + e] = t[e] * scale;
  }
}

// ---- BatchNorm1d of the DINO head's projector (reference src/methods/dino.py:59-77, use_bn=True) ------------------------------
// z [N, C] bf16 (the Linear's output), statistics over the N rows of a column, training mode; C % 4 == 0.
// bn_colsums_kernel: per column the two sums a BatchNorm pass needs -- MODE 0: (sum z, sum z^2); MODE 1 (backward):
// (sum dy, sum dy * xhat) with xhat = (z - mean) * rstd.  64 columns per block (16 lanes x 4), 16 row groups, fixed-order LDS
// reduction (deterministic).
template <typename ZT> struct Z4 { typedef bf16x4 type; };
template <> struct Z4<float> { typedef f32x4 type; };
template <int MODE, typename ZT>
__global__ __launch_bounds__(256) void bn_colsums_kernel(const ZT* __restrict__ z, const bf16_t* __restrict__ dy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ s0, float* __restrict__ s1, int rows, int cols) {
  __shared__ f32x4 part[2][16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 64 + cl * 4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  if (c0 < cols) {
    f32x4 mu = a0, rs = a0;
    if (MODE == 1) { mu = *reinterpret_cast<const f32x4*>(mean + c0); rs = *reinterpret_cast<const f32x4*>(rstd + c0); }
    for (int r = rg; r < rows; r += 16) {
      const typename Z4<ZT>::type zv = *reinterpret_cast<const typename Z4<ZT>::type*>(z + (size_t)r * cols + c0);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float v = (float)zv[e]; a0[e] += v; a1[e] = __builtin_fmaf(v, v, a1[e]); }
      } else {
        const bf16x4 dv = *reinterpret_cast<const bf16x4*>(dy + (size_t)r * cols + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = (float)dv[e]; a0[e] += d; a1[e] = __builtin_fmaf(d, ((float)zv[e] - mu[e]) * rs[e], a1[e]); }
      }
    }
  }
  part[0][rg][cl] = a0;
  part[1][rg][cl] = a1;
  __syncthreads();
  if (rg == 0 && c0 < cols) {
    f32x4 t0 = part[0][0][cl], t1 = part[1][0][cl];
#pragma unroll
    for (int g = 1; g < 16; ++g) { t0 += part[0][g][cl]; t1 += part[1][g][cl]; }
    *reinterpret_cast<f32x4*>(s0 + c0) = t0;
    *reinterpret_cast<f32x4*>(s1 + c0) = t1;
  }
}
// (sum, sum of squares) -> mean, rstd (biased variance, as the normalisation uses it) and the running statistics
// (momentum m, UNBIASED variance, torch.nn.BatchNorm1d); one thread per column
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ s0, const float* __restrict__ s1, float* __restrict__ mean,
                                                        float* __restrict__ rstd, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                        int rows, int cols, float eps, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const float inv = 1.0f / (float)rows;
  const float mu = s0[c] * inv;
  const float var = fmaxf(s1[c] * inv - mu * mu, 0.f);
  mean[c] = mu;
  rstd[c] = rsqrtf(var + eps);
  if (run_mean) {
    run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * (rows > 1 ? (float)rows / (float)(rows - 1) : 1.0f);
  }
}
// y = (z - mean) * rstd * gamma + beta -> pre (bf16: the GELU's input, kept for the backward) and act = gelu(y)
template <typename ZT>
__global__ __launch_bounds__(256) void bn_apply_gelu_kernel(const ZT* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ pre,
                                                            bf16_t* __restrict__ act, long long n4, int cols) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256ll) {
    const int c = (int)((i * 4) % cols);
    const typename Z4<ZT>::type zv = *reinterpret_cast<const typename Z4<ZT>::type*>(z + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    bf16x4 p, a;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = __builtin_fmaf(((float)zv[e] - mu[e]) * rs[e], ga[e], be[e]);
      p[e] = (bf16_t)y;
      a[e] = (bf16_t)gelu_erf((float)p[e]);   // the GELU of the ROUNDED pre-activation: what the backward differentiates
    }
    *reinterpret_cast<bf16x4*>(pre + i * 4) = p;
    *reinterpret_cast<bf16x4*>(act + i * 4) = a;
  }
}
// dz = gamma * rstd * (dy - sum(dy) / N - xhat * sum(dy xhat) / N)
template <typename ZT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const ZT* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sdy,
                                                           const float* __restrict__ sdyx, bf16_t* __restrict__ dz, long long n4, int cols,
                                                           float inv_rows) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256ll) {
    const int c = (int)((i * 4) % cols);
    const bf16x4 dv = *reinterpret_cast<const bf16x4*>(dy + i * 4);
    const typename Z4<ZT>::type zv = *reinterpret_cast<const typename Z4<ZT>::type*>(z + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 a = *reinterpret_cast<const f32x4*>(sdy + c), b = *reinterpret_cast<const f32x4*>(sdyx + c);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = ((float)zv[e] - mu[e]) * rs[e];
      o[e] = (bf16_t)(ga[e] * rs[e] * ((float)dv[e] - a[e] * inv_rows - xh * b[e] * inv_rows));
    }
    *reinterpret_cast<bf16x4*>(dz + i * 4) = o;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, int accumulate, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (accumulate ? dst[i] : 0.f) + src[i];
}
__global__ __launch_bounds__(256) void center_ema_kernel(float* __restrict__ center, const float* __restrict__ colsum,
                                                         float inv_count, float momentum, int P) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < P) center[c] = center[c] * momentum + colsum[c] * inv_count * (1.0f - momentum);
}

// ---- flat parameter kernels ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, float tau, size_t n,
                                                  const float* __restrict__ tau_dev) {
  if (tau_dev) tau = tau_dev[0];
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(t)[i], b = reinterpret_cast<const f32x4*>(s)[i];
    reinterpret_cast<f32x4*>(t)[i] = tau * a + (1.0f - tau) * b;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    t[i] = tau * t[i] + (1.0f - tau) * s[i];
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, size_t n, const float* __restrict__ hyper, int l2) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }  // {lr, 1 - beta1^t, sqrt(1 - beta2^t)} of this step, on the device
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    // l2 = 0: AdamW (decoupled decay of the parameter);  l2 = 1: torch.optim.Adam (the decay enters the gradient: g + wd p)
    const float gg = l2 ? g[i] + wd * p[i] : g[i];
    float pp = l2 ? p[i] : p[i] * (1.0f - lr * wd);
    const float mm = b1 * m[i] + (1.0f - b1) * gg;
    const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pp - (lr / bc1) * mm / denom;
  }
}

// torch.optim.SGD: d = g + wd p;  buf = first ? d : mu buf + (1 - dampening) d;  d = nesterov ? d + mu buf : buf;  p -= lr d
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, float lr,
                                                  float mu, float dampening, float wd, int nesterov, int first, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float d = g[i] + wd * p[i];
    if (mu != 0.f) {
      const float b = first ? d : mu * buf[i] + (1.0f - dampening) * d;
      buf[i] = b;
      d = nesterov ? d + mu * b : b;
    }
    p[i] -= lr * d;
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[i];
    reinterpret_cast<bf16x4*>(dst)[i] = pack4(a[0], a[1], a[2], a[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    dst[i] = (bf16_t)src[i];
  }
}

// 32x32 tiles through LDS: dst = bf16(src) [rows, cols], dst_t = bf16(src)^T [cols, rows]
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             bf16_t* __restrict__ dst_t, int rows, int cols) {
  __shared__ bf16_t tile[32][34];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    bf16_t val = (bf16_t)0.f;
    if (r < rows && c < cols) {
      val = (bf16_t)src[(size_t)r * cols + c];
      if (dst) dst[(size_t)r * cols + c] = val;
    }
    tile[ty + 8 * k][tx] = val;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) dst_t[(size_t)c * rows + r] = tile[tx][ty + 8 * k];
  }
}

// the same for MANY matrices of one slab in a single launch: desc[t] = {src offset (floats), dst_t offset (bf16 elements), rows,
// cols}; blockIdx.y = matrix, blockIdx.x strides over its 32x32 tiles.  One optimiser step re-transposes ~50 weights of a few
// hundred KB each: as separate launches that is 50 x (4 us of kernel + the launch gap) on the critical path.
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst_t,
                                                                     const long long* __restrict__ desc) {
  __shared__ bf16_t tile[32][34];
  const long long* d = desc + 4 * blockIdx.y;
  const float* s = src + d[0];
  bf16_t* o = dst_t + d[1];
  const int rows = (int)d[2], cols = (int)d[3];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int tcols = (cols + 31) / 32, ntiles = tcols * ((rows + 31) / 32);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int c0 = (t % tcols) * 32, r0 = (t / tcols) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + ty + 8 * k, c = c0 + tx;
      tile[ty + 8 * k][tx] = (r < rows && c < cols) ? (bf16_t)s[(size_t)r * cols + c] : (bf16_t)0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < rows && c < cols) o[(size_t)c * rows + r] = tile[tx][ty + 8 * k];
    }
    __syncthreads();
  }
}

// per-tensor L2 clip: one block per tensor
__global__ __launch_bounds__(256) void clip_kernel(float* __restrict__ grads, const long long* __restrict__ offsets,
                                                   const long long* __restrict__ sizes, float clip) {
  __shared__ float red[4];
  float* gp = grads + offsets[blockIdx.x];
  const long long n = sizes[blockIdx.x];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) s += gp[i] * gp[i];
  s = block_sum(s, red);
  const float coef = clip / (sqrtf(s) + 1e-6f);
  if (coef < 1.0f)
    for (long long i = threadIdx.x; i < n; i += 256) gp[i] *= coef;
}


// ---- LARS (reference src/utils/lars.py:112-167): one block per tensor of a flat slab.
// flags[t] bit 0: layer-wise scaling applies (ndim != 1 or not exclude_bias_n_norm); bit 1: momentum buffer already initialised
__global__ __launch_bounds__(256) void lars_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                   float* __restrict__ bufs, const long long* __restrict__ offsets,
                                                   const long long* __restrict__ sizes, const int* __restrict__ flags, float lr,
                                                   float momentum, float dampening, float wd, float eta, float eps, int clip_lr,
                                                   int nesterov) {
  __shared__ float red[4];
  const long long off = offsets[blockIdx.x], n = sizes[blockIdx.x];
  const int fl = flags[blockIdx.x];
  float* p = params + off;
  const float* g = grads + off;
  float* b = bufs + off;
  float pn = 0.f, gn = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) {
    pn += p[i] * p[i];
    gn += g[i] * g[i];
  }
  pn = sqrtf(block_sum(pn, red));
  gn = sqrtf(block_sum(gn, red));
  float scale = 1.f, wdec = 0.f;
  if ((fl & 1) && pn != 0.f && gn != 0.f) {
    scale = pn / (gn + pn * wd + eps) * eta;
    if (clip_lr) scale = fminf(scale / lr, 1.f);
    wdec = wd;
  }
  for (long long i = threadIdx.x; i < n; i += 256) {
    float d = (g[i] + wdec * p[i]) * scale;
    if (momentum != 0.f) {
      const float bb = (fl & 2) ? b[i] * momentum + (1.f - dampening) * d : d;
      b[i] = bb;
      d = nesterov ? d + momentum * bb : bb;
    }
    p[i] -= lr * d;
  }
}


// ---- weighted k-NN vote (reference src/utils/knn.py:141-161).  One block per test sample: exact k-th largest similarity
// by an MSB-first radix select over the order-preserving integer image of the floats (4 passes of 8 bits, LDS histogram),
// then every train sample above the threshold -- plus the first (k - #above) samples equal to it, in index order -- votes
// for its class with weight exp(sim / T) (cosine) or sim (euclidean: sim = 1 / (dist + eps)); the `top` best classes by
// vote mass are written out.  The vote does not depend on the order inside the top-k, so no sort is needed.
__device__ __forceinline__ unsigned knn_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float <-> larger key
}

__global__ __launch_bounds__(256) void knn_vote_kernel(const float* __restrict__ sims, long long lds, const int* __restrict__ train_targets,
                                                       int n_train, int k, float inv_T, int use_exp, int num_classes, int top,
                                                       int* __restrict__ top_classes, float* __restrict__ votes_out) {
  extern __shared__ float sdyn[];         // votes[num_classes]
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_remaining;
  __shared__ int tie_budget;
  __shared__ float red_v[256];
  __shared__ int red_i[256];
  const int tid = threadIdx.x;
  const float* row = sims + (size_t)blockIdx.x * lds;
  // ---- radix select: after the 4 passes `prefix` is the key of the k-th largest element
  unsigned prefix = 0, mask = 0;
  int remaining = k;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    for (int j = tid; j < n_train; j += 256) {
      const unsigned key = knn_key(row[j]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = remaining;
      int d = 255;
      for (; d > 0; --d) {
        if ((int)hist[d] >= rem) break;
        rem -= (int)hist[d];
      }
      sel_prefix = prefix | ((unsigned)d << shift);
      sel_remaining = (unsigned)rem;
    }
    __syncthreads();
    prefix = sel_prefix;
    remaining = (int)sel_remaining;
    mask |= 255u << shift;
    __syncthreads();
  }
  // `remaining` = how many elements EQUAL to the threshold belong to the top-k
  for (int c = tid; c < num_classes; c += 256) sdyn[c] = 0.f;
  if (tid == 0) tie_budget = remaining;
  __syncthreads();
  for (int j0 = 0; j0 < n_train; j0 += 256) {  // index order matters only for ties at the threshold
    const int j = j0 + tid;
    bool vote = false;
    float v = 0.f;
    if (j < n_train) {
      v = row[j];
      const unsigned key = knn_key(v);
      if (key > prefix) vote = true;
      else if (key == prefix) vote = atomicSub(&tie_budget, 1) > 0;
    }
    if (vote) atomicAdd(&sdyn[train_targets[j]], use_exp ? __expf(v * inv_T) : v);
    __syncthreads();
  }
  if (votes_out)
    for (int c = tid; c < num_classes; c += 256) votes_out[(size_t)blockIdx.x * num_classes + c] = sdyn[c];
  // ---- `top` best classes (ties: lowest class index first)
  for (int t = 0; t < top; ++t) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = tid; c < num_classes; c += 256) {
      const float v = sdyn[c];
      if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
    red_v[tid] = bv;
    red_i[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const float v2 = red_v[tid + o];
        const int i2 = red_i[tid + o];
        if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      top_classes[(size_t)blockIdx.x * top + t] = red_i[0];
      if (red_i[0] < num_classes) sdyn[red_i[0]] = -INFINITY;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int chadavit_abi_version(void) { return 9; }

extern "C" int chadavit_l2norm_fwd(const float* x, chada_bf16* y, float* inv_norm, int M, int N, void* stream) {
  CHADA_ENTRY();
  if (!x || !y || !inv_norm || M <= 0 || N <= 0) return 1;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid_for((size_t)M * 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<bf16_t*>(y), inv_norm, M, N);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_l2norm_bwd(const float* dy, const float* x, const float* inv_norm, chada_bf16* dx, int M, int N,
                                   void* stream) {
  CHADA_ENTRY();
  if (!dy || !x || !inv_norm || !dx || M <= 0 || N <= 0) return 1;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid_for((size_t)M * 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy,
                     x, inv_norm, reinterpret_cast<bf16_t*>(dx), M, N);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_weightnorm_fwd(const float* v, const float* g, chada_bf16* w, chada_bf16* w_t, float* inv_norm, int P,
                                       int K, void* stream) {
  CHADA_ENTRY();
  if (!v || !g || !w || !inv_norm || P <= 0 || K <= 0) return 1;
  hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(grid_for((size_t)P * 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     v, g, reinterpret_cast<bf16_t*>(w), reinterpret_cast<bf16_t*>(w_t), inv_norm, P, K);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_weightnorm_bwd_g(const float* dw, const float* v, const float* g, const float* inv_norm, float* dv, float* dg,
                                         int accumulate, int P, int K, void* stream) {
  CHADA_ENTRY();
  if (!dw || !v || !g || !inv_norm || !dv || P <= 0 || K <= 0) return 1;
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(grid_for((size_t)P * 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     dw, v, g, inv_norm, dv, dg, accumulate, P, K);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, float* dv,
                                       int accumulate, int P, int K, void* stream) {
  return chadavit_weightnorm_bwd_g(dw, v, g, inv_norm, dv, nullptr, accumulate, P, K, stream);
}

extern "C" int chadavit_dino_loss(const float* student, const float* teacher, const float* center, float student_temp,
                                  float teacher_temp, float* loss_rows, chada_bf16* dstudent, float* teacher_colsum, int B,
                                  int P, void* stream) {
  CHADA_ENTRY();
  if (!student || !teacher || !center || !loss_rows || B <= 0 || P <= 0 || student_temp <= 0.f || teacher_temp <= 0.f) return 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dino_loss_kernel, dim3(B), dim3(256), 0, s, student, teacher, center, 1.0f / student_temp,
                     1.0f / teacher_temp, loss_rows, reinterpret_cast<bf16_t*>(dstudent), B, P, (const float*)nullptr);
  if (teacher_colsum)
    hipLaunchKernelGGL(sum_rows_kernel, dim3((P + 63) / 64), dim3(256), 0, s, teacher, teacher_colsum, 2 * B, P, 1.0f,
                       (int)((P & 3) == 0 && ((uintptr_t)teacher & 15) == 0));
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_dino_loss_multicrop(const float* student, const float* teacher, const float* center, float student_temp,
                                            float teacher_temp, float* loss_rows, chada_bf16* dstudent, float* teacher_colsum, int B,
                                            int V, int P, void* stream) {
  CHADA_ENTRY();
  if (!student || !teacher || !center || !loss_rows || B <= 0 || V < 2 || P <= 0 || student_temp <= 0.f || teacher_temp <= 0.f) return 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dino_loss_mc_kernel, dim3(B), dim3(256), 0, s, student, teacher, center, 1.0f / student_temp, 1.0f / teacher_temp,
                     loss_rows, reinterpret_cast<bf16_t*>(dstudent), B, V, P);
  if (teacher_colsum)
    hipLaunchKernelGGL(sum_rows_kernel, dim3((P + 63) / 64), dim3(256), 0, s, teacher, teacher_colsum, 2 * B, P, 1.0f,
                       (int)((P & 3) == 0 && ((uintptr_t)teacher & 15) == 0));
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_center_ema(float* center, const float* colsum, float inv_count, float momentum, int P, void* stream) {
  CHADA_ENTRY();
  if (!center || !colsum || P <= 0) return 1;
  hipLaunchKernelGGL(center_ema_kernel, dim3((P + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), center,
                     colsum, inv_count, momentum, P);
  CHADA_CHECK_LAUNCH();
  return 0;
}
// BatchNorm1d of the head's projector (src/methods/dino.py:59-77 with use_bn=True), training-mode statistics over the rows.
// z = the Linear's output, fp32 (z_f32 != 0: what the head uses -- the statistics of a handful of rows do not survive bf16 rounding of
// their inputs) or bf16.
// chadavit_bn_stats: mean / rstd of every column of z [N, C] (workspace: 2 C floats); running_mean / running_var (optional) get
// torch.nn.BatchNorm1d's update (momentum, unbiased variance).
extern "C" int chadavit_bn_stats(const void* z, int z_f32, int N, int C, float eps, float* mean, float* rstd, float* running_mean,
                                 float* running_var, float momentum, float* workspace, void* stream) {
  CHADA_ENTRY();
  if (!z || !mean || !rstd || !workspace || N <= 0 || C <= 0 || (running_mean == nullptr) != (running_var == nullptr)) return 1;
  if (C % 4 != 0 || ((uintptr_t)z & 15) != 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (z_f32)
    hipLaunchKernelGGL((bn_colsums_kernel<0, float>), dim3((C + 63) / 64), dim3(256), 0, s, reinterpret_cast<const float*>(z), (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, workspace, workspace + C, N, C);
  else
    hipLaunchKernelGGL((bn_colsums_kernel<0, bf16_t>), dim3((C + 63) / 64), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(z), (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, workspace, workspace + C, N, C);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, workspace + C, mean, rstd, running_mean, running_var, N,
                     C, eps, momentum);
  CHADA_CHECK_LAUNCH();
  return 0;
}
// pre = (z - mean) rstd gamma + beta (bf16), act = gelu(pre) (bf16)
extern "C" int chadavit_bn_apply_gelu(const void* z, int z_f32, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      chada_bf16* pre, chada_bf16* act, int N, int C, void* stream) {
  CHADA_ENTRY();
  if (!z || !mean || !rstd || !gamma || !beta || !pre || !act || N <= 0 || C <= 0) return 1;
  if (C % 4 != 0) return 2;
  const long long n4 = (long long)N * C / 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (z_f32)
    hipLaunchKernelGGL((bn_apply_gelu_kernel<float>), dim3(grid_for((size_t)n4, 4096)), dim3(256), 0, s, reinterpret_cast<const float*>(z), mean, rstd,
                       gamma, beta, reinterpret_cast<bf16_t*>(pre), reinterpret_cast<bf16_t*>(act), n4, C);
  else
    hipLaunchKernelGGL((bn_apply_gelu_kernel<bf16_t>), dim3(grid_for((size_t)n4, 4096)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(z), mean, rstd,
                       gamma, beta, reinterpret_cast<bf16_t*>(pre), reinterpret_cast<bf16_t*>(act), n4, C);
  CHADA_CHECK_LAUNCH();
  return 0;
}
// backward: dy = gradient w.r.t. the BatchNorm output.  dgamma / dbeta (+= when accumulate) and dz (bf16); workspace: 2 C floats
extern "C" int chadavit_bn_bwd(const chada_bf16* dy, const void* z, int z_f32, const float* mean, const float* rstd, const float* gamma,
                               float* dgamma, float* dbeta, int accumulate, chada_bf16* dz, int N, int C, float* workspace, void* stream) {
  CHADA_ENTRY();
  if (!dy || !z || !mean || !rstd || !gamma || !dgamma || !dbeta || !dz || !workspace || N <= 0 || C <= 0) return 1;
  if (C % 4 != 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long n4 = (long long)N * C / 4;
  if (z_f32) {
    hipLaunchKernelGGL((bn_colsums_kernel<1, float>), dim3((C + 63) / 64), dim3(256), 0, s, reinterpret_cast<const float*>(z),
                       reinterpret_cast<const bf16_t*>(dy), mean, rstd, workspace, workspace + C, N, C);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(grid_for((size_t)n4, 4096)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(dy),
                       reinterpret_cast<const float*>(z), mean, rstd, gamma, workspace, workspace + C, reinterpret_cast<bf16_t*>(dz), n4, C, 1.0f / (float)N);
  } else {
    hipLaunchKernelGGL((bn_colsums_kernel<1, bf16_t>), dim3((C + 63) / 64), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(z),
                       reinterpret_cast<const bf16_t*>(dy), mean, rstd, workspace, workspace + C, N, C);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), dim3(grid_for((size_t)n4, 4096)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(dy),
                       reinterpret_cast<const bf16_t*>(z), mean, rstd, gamma, workspace, workspace + C, reinterpret_cast<bf16_t*>(dz), n4, C, 1.0f / (float)N);
  }
  hipLaunchKernelGGL(axpy_kernel, dim3((C + 255) / 256), dim3(256), 0, s, dbeta, workspace, accumulate, C);
  hipLaunchKernelGGL(axpy_kernel, dim3((C + 255) / 256), dim3(256), 0, s, dgamma, workspace + C, accumulate, C);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_sum_rows_f32(const float* x, float* out, int rows, int cols, float scale, void* stream) {
  CHADA_ENTRY();
  if (!x || !out || rows <= 0 || cols <= 0) return 1;
  hipLaunchKernelGGL(sum_rows_kernel, dim3((cols + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, rows,
                     cols, scale, (int)((cols & 3) == 0 && ((uintptr_t)x & 15) == 0));
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_ema_update(float* teacher, const float* student, float tau, long long n, void* stream) {
  CHADA_ENTRY();
  if (!teacher || !student || n <= 0) return 1;
  if (((uintptr_t)teacher | (uintptr_t)student) & 15) return 2;
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for((size_t)n / 4 + 1, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     teacher, student, tau, (size_t)n, (const float*)nullptr);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, long long n,
                                   void* stream) {
  CHADA_ENTRY();
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return 1;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)n, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                     grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bias_corr1, sqrtf(bias_corr2), (size_t)n,
                     (const float*)nullptr, 0);
  CHADA_CHECK_LAUNCH();
  return 0;
}
// torch.optim.Adam (base.py:67-72 "adam"): as above with the weight decay added to the gradient instead of decaying the parameter
extern "C" int chadavit_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, long long n,
                                  void* stream) {
  CHADA_ENTRY();
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return 1;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)n, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                     grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bias_corr1, sqrtf(bias_corr2), (size_t)n,
                     (const float*)nullptr, 1);
  CHADA_CHECK_LAUNCH();
  return 0;
}
// torch.optim.SGD (base.py:67-72 "sgd"); momentum_buf may be NULL when momentum == 0; first = 1 on a parameter's first step
extern "C" int chadavit_sgd_step(float* param, const float* grad, float* momentum_buf, float lr, float momentum, float dampening,
                                 float weight_decay, int nesterov, int first, long long n, void* stream) {
  CHADA_ENTRY();
  if (!param || !grad || n <= 0 || (momentum != 0.f && !momentum_buf)) return 1;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for((size_t)n, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad,
                     momentum_buf, lr, momentum, dampening, weight_decay, nesterov, first, (size_t)n);
  CHADA_CHECK_LAUNCH();
  return 0;
}

// ---- the same three steps with their PER-STEP scalars read from device memory: what a hipGraph of the whole training step needs
// (a scalar passed by value is frozen into the captured launch; the learning rate, Adam's bias corrections, the EMA tau and the
// teacher temperature change every step / epoch).  The host writes them to a pinned buffer; the graph's first node copies it over.
extern "C" int chadavit_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* hyper,
                                       float beta1, float beta2, float eps, float weight_decay, long long n, void* stream) {
  CHADA_ENTRY();
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return 1;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)n, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param,
                     grad, exp_avg, exp_avg_sq, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, (size_t)n, hyper, 0);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_ema_update_dev(float* teacher, const float* student, const float* tau, long long n, void* stream) {
  CHADA_ENTRY();
  if (!teacher || !student || !tau || n <= 0) return 1;
  if (((uintptr_t)teacher | (uintptr_t)student) & 15) return 2;
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for((size_t)n / 4 + 1, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     teacher, student, 0.f, (size_t)n, tau);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_dino_loss_dev(const float* student, const float* teacher, const float* center, float student_temp,
                                      const float* teacher_temp, float* loss_rows, chada_bf16* dstudent, float* teacher_colsum, int B,
                                      int P, void* stream) {
  CHADA_ENTRY();
  if (!student || !teacher || !center || !loss_rows || !teacher_temp || B <= 0 || P <= 0 || student_temp <= 0.f) return 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dino_loss_kernel, dim3(B), dim3(256), 0, s, student, teacher, center, 1.0f / student_temp, 0.f, loss_rows,
                     reinterpret_cast<bf16_t*>(dstudent), B, P, teacher_temp);
  if (teacher_colsum)
    hipLaunchKernelGGL(sum_rows_kernel, dim3((P + 63) / 64), dim3(256), 0, s, teacher, teacher_colsum, 2 * B, P, 1.0f,
                       (int)((P & 3) == 0 && ((uintptr_t)teacher & 15) == 0));
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_cast_bf16(const float* src, chada_bf16* dst, long long n, void* stream) {
  CHADA_ENTRY();
  if (!src || !dst || n <= 0) return 1;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return 2;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for((size_t)n / 4 + 1, 2048)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src, reinterpret_cast<bf16_t*>(dst), (size_t)n);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_cast_transpose_bf16(const float* src, chada_bf16* dst, chada_bf16* dst_t, int rows, int cols,
                                            void* stream) {
  CHADA_ENTRY();
  if (!src || !dst_t || rows <= 0 || cols <= 0) return 1;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, reinterpret_cast<bf16_t*>(dst), reinterpret_cast<bf16_t*>(dst_t),
                     rows, cols);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_cast_transpose_batched(const float* src, chada_bf16* dst_t, const long long* desc, int n_mats,
                                               int max_tiles, void* stream) {
  CHADA_ENTRY();
  if (!src || !dst_t || !desc || n_mats <= 0 || max_tiles <= 0) return 1;
  hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(max_tiles < 256 ? max_tiles : 256, n_mats), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, reinterpret_cast<bf16_t*>(dst_t), desc);
  CHADA_CHECK_LAUNCH();
  return 0;
}
extern "C" int chadavit_clip_tensors(float* grads, const long long* offsets, const long long* sizes, int n_tensors, float clip,
                                     void* stream) {
  CHADA_ENTRY();
  if (!grads || !offsets || !sizes || n_tensors <= 0 || clip <= 0.f) return 1;
  hipLaunchKernelGGL(clip_kernel, dim3(n_tensors), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grads, offsets, sizes,
                     clip);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_lars_step(float* params, const float* grads, float* momentum_bufs, const long long* offsets,
                                  const long long* sizes, const int* flags, int n_tensors, float lr, float momentum,
                                  float dampening, float weight_decay, float eta, float eps, int clip_lr, int nesterov,
                                  void* stream) {
  CHADA_ENTRY();
  if (!params || !grads || !momentum_bufs || !offsets || !sizes || !flags || n_tensors <= 0) return 1;
  if (nesterov && (momentum <= 0.f || dampening != 0.f)) return 1;
  hipLaunchKernelGGL(lars_kernel, dim3(n_tensors), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, grads,
                     momentum_bufs, offsets, sizes, flags, lr, momentum, dampening, weight_decay, eta, eps, clip_lr, nesterov);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_knn_vote(const float* sims, long long ld_sims, const int* train_targets, int n_test, int n_train, int k,
                                 float temperature, int use_exp, int num_classes, int top, int* top_classes, float* votes,
                                 void* stream) {
  CHADA_ENTRY();
  if (!sims || !train_targets || !top_classes || n_test <= 0 || n_train <= 0 || k <= 0 || k > n_train || num_classes <= 0 ||
      top <= 0 || top > num_classes || ld_sims < n_train || (use_exp && temperature <= 0.f))
    return 1;
  if (num_classes > 12288) return 2;  // vote table lives in LDS
  hipLaunchKernelGGL(knn_vote_kernel, dim3(n_test), dim3(256), num_classes * sizeof(float), reinterpret_cast<hipStream_t>(stream),
                     sims, ld_sims, train_targets, n_train, k, use_exp ? 1.0f / temperature : 1.0f, use_exp, num_classes, top,
                     top_classes, votes);
  CHADA_CHECK_LAUNCH();
  return 0;
}
