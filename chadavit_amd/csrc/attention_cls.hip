// Attention of ONE query row per sequence -- the CLS row -- against all keys of its sequence, forward and backward.
//
// Why it exists: with return_all_tokens = False (the training configuration, args/pretrain.py:147) ChAdaViT.forward returns
// norm(x)[:, 0] (chada_vit.py:272-289): of the LAST encoder block's output only the CLS rows are ever read.  That block's
// attention output, out-projection, both LayerNorms and its FFN are therefore needed for one row per image, its K and V
// projections for all rows (the CLS query attends to every token).  The backbone runs the last block that way
// (ChAdaViT.cls_only_last_block); this file is its attention: per (image, head) one wave streams the K and V rows once.
// The numbers are those of nn.MultiheadAttention (chada_vit.py:105-111) for that row -- fp32 softmax, no bf16 rounding of
// the probabilities -- and the gradients those of autograd when only the CLS rows of the block's output carry gradient.
//
// Lane mapping: a key row's head slice (DH bf16 = DH/8 pieces of 16 bytes) is read by LPK consecutive lanes (LPK = the next
// power of two >= DH/8; pieces past DH/8 idle), so a wave covers 64/LPK keys per sweep step with whole-row coalesced reads;
// the dot product of a key is finished by xor-shuffles inside its lane group.  The softmax is online per lane group and merged
// across the groups and the block's four waves at the end; HBM-bound by construction (K and V read once: 2 * T * D * 2 bytes).
#include "common.h"

using namespace chada;

namespace {

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }

// sum over the LPK consecutive lanes of a key, in every one of them: DPP inside a 16-lane row, permlane swaps across rows (the
// xor-shuffle form goes through ds_bpermute: six dependent LDS round trips per key at dh = 384 -- measured 355 -> see DESIGN 5f)
template <int LPK>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (LPK >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (LPK >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (LPK >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if constexpr (LPK >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  if constexpr (LPK >= 32) {
    float a, b2;
    swap16(v, a, b2);
    v = a + b2;
  }
  if constexpr (LPK >= 64) {
    float a, b2;
    swap32(v, a, b2);
    v = a + b2;
  }
  return v;
}

__device__ __forceinline__ void load8(const bf16_t* p, float (&f)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}

__device__ __forceinline__ bf16x8 pack8f(const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
  return v;
}

// One BLOCK of NW waves per (image, head): the waves take the sweep steps round-robin (wave w: keys [(NW t + w) * KPW, +KPW)), so
// that long sequences on small batches (Base: 64 sequences of 1961 tokens -> NW = 16) still fill the chip -- the sweep is a chain of
// load round trips, only parallelism hides them; UNR sweep steps are loaded before any of them is used (independent 16-byte loads
// in flight per lane).  The partial softmax states / dQ sums of the waves meet in LDS.
constexpr int UNR = 4;

// out_cls [B, D] bf16, lse_cls [H, B] fp32 (natural log)
template <int LPK, int NW>
__global__ __launch_bounds__(64 * NW) void attn_cls_fwd_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ cu,
                                                           bf16_t* __restrict__ out_cls, float* __restrict__ lse_cls, int B, int D,
                                                           int H, float scale) {
  constexpr int KPW = 64 / LPK;  // keys per sweep step
  __shared__ float red[NW - 1][LPK][10];  // waves 1..: (m, l, acc[8]) per piece
  const int l = threadIdx.x & 63, c = l % LPK, grp = l / LPK;
  const int w = threadIdx.x >> 6;
  const int item = blockIdx.x;
  const int b = item / H, h = item % H, DH = D / H;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  const size_t ld = 3 * (size_t)D;
  const bool act = c * 8 < DH;
  const bf16_t* base = qkv + (size_t)seq0 * ld + h * DH + c * 8;
  float q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = 0.f;
  if (act) load8(base, q);  // the CLS row is the first row of the sequence
  const float sc = scale * LOG2E;
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] *= sc;
  float m = -INFINITY, ls = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int j0 = w * KPW + grp; j0 < len; j0 += NW * KPW * UNR) {
    bf16x8 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = min(j0 + u * NW * KPW, len - 1);  // clamped rows are dropped below
      if (act) {
        kr[u] = *reinterpret_cast<const bf16x8*>(base + (size_t)j * ld + D);
        vr[u] = *reinterpret_cast<const bf16x8*>(base + (size_t)j * ld + 2 * D);
      }
    }
    float s[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float part = 0.f;
      if (act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf(q[e], (float)kr[u][e], part);
      }
      s[u] = (j0 + u * NW * KPW < len) ? group_sum<LPK>(part) : -INFINITY;  // (group-uniform condition)
    }
    float mn = m;
#pragma unroll
    for (int u = 0; u < UNR; ++u) mn = fmaxf(mn, s[u]);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);  // s[0] is always a real key: mn is finite
    m = mn;
    ls *= alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float p = __builtin_amdgcn_exp2f(s[u] - mn);
      ls += p;
      if (act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, (float)vr[u][e], acc[e]);
      }
    }
  }
  // merge the KPW lane groups (same piece c, different keys): xor over the group bits
  auto merge = [&](float mo, float lo, const float (&ao)[8]) {
    const float mn = fmaxf(m, mo);
    const float a0 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float a1 = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
    ls = ls * a0 + lo * a1;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * a0 + ao[e] * a1;
    m = mn;
  };
#pragma unroll
  for (int sh = LPK; sh < 64; sh <<= 1) {
    float ao[8];
    const float mo = shfl_xor(m, sh), lo = shfl_xor(ls, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) ao[e] = shfl_xor(acc[e], sh);
    merge(mo, lo, ao);
  }
  // ... and the four waves
  if (w > 0 && grp == 0) {
    red[w - 1][c][0] = m;
    red[w - 1][c][1] = ls;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w - 1][c][2 + e] = acc[e];
  }
  __syncthreads();
  if (w == 0 && grp == 0) {
#pragma unroll
    for (int o = 0; o < NW - 1; ++o) {
      float ao[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ao[e] = red[o][c][2 + e];
      merge(red[o][c][0], red[o][c][1], ao);
    }
    const float inv = 1.0f / ls;
    if (act) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = acc[e] * inv;
      *reinterpret_cast<bf16x8*>(out_cls + (size_t)b * D + h * DH + c * 8) = pack8f(o);
    }
    if (c == 0) lse_cls[(size_t)h * B + b] = (m + log2f(ls)) * LN2;
  }
}

// dqkv [T, 3D]: fully written -- dQ is zero except on the CLS rows; dK / dV of every row = the CLS query's contribution
template <int LPK, int NW>
__global__ __launch_bounds__(64 * NW) void attn_cls_bwd_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ cu,
                                                           const bf16_t* __restrict__ out_cls, const bf16_t* __restrict__ dout_cls,
                                                           const float* __restrict__ lse_cls, bf16_t* __restrict__ dqkv, int B, int D,
                                                           int H, float scale) {
  constexpr int KPW = 64 / LPK;
  __shared__ float red[NW - 1][LPK][8];  // waves 1..: dQ partial sums per piece
  const int l = threadIdx.x & 63, c = l % LPK, grp = l / LPK;
  const int w = threadIdx.x >> 6;
  const int item = blockIdx.x;
  const int b = item / H, h = item % H, DH = D / H;
  const int seq0 = cu[b], len = cu[b + 1] - seq0;
  const size_t ld = 3 * (size_t)D;
  const bool act = c * 8 < DH;
  const bf16_t* base = qkv + (size_t)seq0 * ld + h * DH + c * 8;
  bf16_t* dbase = dqkv + (size_t)seq0 * ld + h * DH + c * 8;
  float q[8], go[8], o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = go[e] = o[e] = 0.f;
  if (act) {
    load8(base, q);
    load8(dout_cls + (size_t)b * D + h * DH + c * 8, go);
    load8(out_cls + (size_t)b * D + h * DH + c * 8, o);
  }
  float dpart = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) dpart = fmaf(go[e], o[e], dpart);
  const float delta = group_sum<LPK>(dpart);            // rowsum(dO * O) of the CLS row
  const float lse2 = lse_cls[(size_t)h * B + b] * LOG2E;  // log2 domain
  const float sc = scale * LOG2E;
  float dq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dq[e] = 0.f;
  const bf16x8 zero8 = pack8f(dq);
  for (int j0 = w * KPW + grp; j0 < len; j0 += NW * KPW * UNR) {
    bf16x8 kr[UNR], vr[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = min(j0 + u * NW * KPW, len - 1);
      if (act) {
        kr[u] = *reinterpret_cast<const bf16x8*>(base + (size_t)j * ld + D);
        vr[u] = *reinterpret_cast<const bf16x8*>(base + (size_t)j * ld + 2 * D);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * NW * KPW;
      if (j < len) {  // (group-uniform)
        float sp = 0.f, dp = 0.f;
        if (act) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sp = fmaf(q[e], (float)kr[u][e], sp);
            dp = fmaf(go[e], (float)vr[u][e], dp);
          }
        }
        const float s = group_sum<LPK>(sp), dP = group_sum<LPK>(dp);
        const float p = __builtin_amdgcn_exp2f(s * sc - lse2);
        const float ds = p * (dP - delta) * scale;  // d(score before the softmax scale) folded with the scale
        if (act) {
          float dk[8], dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            dv[e] = p * go[e];
            dk[e] = ds * q[e];
            dq[e] = fmaf(ds, (float)kr[u][e], dq[e]);
          }
          *reinterpret_cast<bf16x8*>(dbase + (size_t)j * ld + D) = pack8f(dk);
          *reinterpret_cast<bf16x8*>(dbase + (size_t)j * ld + 2 * D) = pack8f(dv);
          if (j > 0) *reinterpret_cast<bf16x8*>(dbase + (size_t)j * ld) = zero8;  // no query but the CLS row has a gradient
        }
      }
    }
  }
#pragma unroll
  for (int sh = LPK; sh < 64; sh <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] += shfl_xor(dq[e], sh);
  if (w > 0 && grp == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w - 1][c][e] = dq[e];
  }
  __syncthreads();
  if (w == 0 && grp == 0 && act) {
#pragma unroll
    for (int o = 0; o < NW - 1; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) dq[e] += red[o][c][e];
    *reinterpret_cast<bf16x8*>(dbase) = pack8f(dq);
  }
}

template <int LPK>
void launch_fwd(const bf16_t* qkv, const int* cu, bf16_t* out_cls, float* lse_cls, int B, int D, int H, float scale, hipStream_t s) {
  if (B * H < 1024)   // few (image, head) items: 16 waves each
    hipLaunchKernelGGL((attn_cls_fwd_kernel<LPK, 16>), dim3(B * H), dim3(1024), 0, s, qkv, cu, out_cls, lse_cls, B, D, H, scale);
  else
    hipLaunchKernelGGL((attn_cls_fwd_kernel<LPK, 4>), dim3(B * H), dim3(256), 0, s, qkv, cu, out_cls, lse_cls, B, D, H, scale);
}
template <int LPK>
void launch_bwd(const bf16_t* qkv, const int* cu, const bf16_t* out_cls, const bf16_t* dout_cls, const float* lse_cls, bf16_t* dqkv,
                int B, int D, int H, float scale, hipStream_t s) {
  if (B * H < 1024)
    hipLaunchKernelGGL((attn_cls_bwd_kernel<LPK, 8>), dim3(B * H), dim3(512), 0, s, qkv, cu, out_cls, dout_cls, lse_cls, dqkv, B, D, H,
                       scale);
  else
    hipLaunchKernelGGL((attn_cls_bwd_kernel<LPK, 4>), dim3(B * H), dim3(256), 0, s, qkv, cu, out_cls, dout_cls, lse_cls, dqkv, B, D, H,
                       scale);
}

int lanes_per_key(int dh) {
  const int pieces = dh / 8;
  int lpk = 1;
  while (lpk < pieces) lpk <<= 1;
  return lpk;
}

}  // namespace

#define CLS_DISPATCH(CALL)                    \
  switch (lanes_per_key(D / H)) {             \
    case 1: { constexpr int L_ = 1; CALL; break; }   \
    case 2: { constexpr int L_ = 2; CALL; break; }   \
    case 4: { constexpr int L_ = 4; CALL; break; }   \
    case 8: { constexpr int L_ = 8; CALL; break; }   \
    case 16: { constexpr int L_ = 16; CALL; break; } \
    case 32: { constexpr int L_ = 32; CALL; break; } \
    case 64: { constexpr int L_ = 64; CALL; break; } \
    default: return 2;                        \
  }

extern "C" int chadavit_attn_cls_fwd(const chada_bf16* qkv, const int* cu_seqlens, chada_bf16* out_cls, float* lse_cls, int B, int D,
                                     int H, float scale, void* stream) {
  CHADA_ENTRY();
  if (!qkv || !cu_seqlens || !out_cls || !lse_cls || B <= 0 || D <= 0 || H <= 0 || D % H != 0) return 1;
  if ((D / H) % 8 != 0 || D / H > 512) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  CLS_DISPATCH(launch_fwd<L_>(reinterpret_cast<const bf16_t*>(qkv), cu_seqlens, reinterpret_cast<bf16_t*>(out_cls), lse_cls, B, D, H, scale, s));
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_attn_cls_bwd(const chada_bf16* qkv, const int* cu_seqlens, const chada_bf16* out_cls, const chada_bf16* dout_cls,
                                     const float* lse_cls, chada_bf16* dqkv, int B, int D, int H, float scale, void* stream) {
  CHADA_ENTRY();
  if (!qkv || !cu_seqlens || !out_cls || !dout_cls || !lse_cls || !dqkv || B <= 0 || D <= 0 || H <= 0 || D % H != 0) return 1;
  if ((D / H) % 8 != 0 || D / H > 512) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  CLS_DISPATCH(launch_bwd<L_>(reinterpret_cast<const bf16_t*>(qkv), cu_seqlens, reinterpret_cast<const bf16_t*>(out_cls),
                              reinterpret_cast<const bf16_t*>(dout_cls), lse_cls, reinterpret_cast<bf16_t*>(dqkv), B, D, H, scale, s));
  CHADA_CHECK_LAUNCH();
  return 0;
}
