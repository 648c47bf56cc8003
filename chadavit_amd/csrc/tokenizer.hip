// Tokenizer-side data movement for the ragged packed token buffer (HBM-bound byte shuffling):
// im2col for the 1->D patch conv, CLS rows, CLS gather / scatter, and the tokenizer backward reductions.
// replaces chada_vit.py:128-133 (conv unfold), :226-268 (split / pad / stack / cat), :283-289 (select).
#include "common.h"

using namespace chada;

namespace {

// x fp32 [n_chan, S, S] -> patches bf16 [n_chan * g*g, P*P]; thread = 8 consecutive pixels of one image row.
// Thread order follows the IMAGE row (reads fully coalesced, 32 B per lane); writes are 16-byte chunks.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, bf16_t* __restrict__ patches, int n_chan,
                                                     int S, int P) {
  const int g = S / P;
  const int cpr = S / 8;  // 8-pixel chunks per image row
  const size_t total = (size_t)n_chan * S * cpr;
  for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
    const int xc = (int)(id % cpr);
    const size_t t = id / cpr;
    const int yy = (int)(t % S);
    const int ch = (int)(t / S);
    const float* src = x + ((size_t)ch * S + yy) * S + xc * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
    const int r = yy / P, u = yy % P;
    const int px = xc * 8;
    const int q = px / P, v = px % P;
    bf16x8 o;
    o[0] = (bf16_t)a[0]; o[1] = (bf16_t)a[1]; o[2] = (bf16_t)a[2]; o[3] = (bf16_t)a[3];
    o[4] = (bf16_t)b[0]; o[5] = (bf16_t)b[1]; o[6] = (bf16_t)b[2]; o[7] = (bf16_t)b[3];
    bf16_t* dst = patches + ((size_t)ch * g * g + (size_t)r * g + q) * (P * P) + u * P + v;
    *reinterpret_cast<bf16x8*>(dst) = o;
  }
}

__global__ __launch_bounds__(256) void write_cls_kernel(bf16_t* __restrict__ tokens, const int* __restrict__ cu,
                                                        const float* __restrict__ cls, const float* __restrict__ pos0, int B,
                                                        int D) {
  const int n = B * D;
  for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < n; id += gridDim.x * blockDim.x) {
    const int i = id / D, d = id % D;
    tokens[(size_t)cu[i] * D + d] = (bf16_t)(cls[d] + pos0[d]);
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                          bf16_t* __restrict__ dst, int n_rows, int D) {
  const int cpr = D / 4;
  const size_t total = (size_t)n_rows * cpr;
  for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(id / cpr), c = (int)(id % cpr) * 4;
    *reinterpret_cast<bf16x4*>(dst + (size_t)i * D + c) = *reinterpret_cast<const bf16x4*>(src + (size_t)rows[i] * D + c);
  }
}

__global__ __launch_bounds__(256) void zero_bf16_kernel(bf16_t* __restrict__ dst, size_t n8) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < n8; id += (size_t)gridDim.x * blockDim.x)
    *reinterpret_cast<u32x4*>(dst + id * 8) = z;
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ rows,
                                                           bf16_t* __restrict__ dst, int n_rows, int D) {
  const int cpr = D / 4;
  const size_t total = (size_t)n_rows * cpr;
  for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(id / cpr), c = (int)(id % cpr) * 4;
    *reinterpret_cast<bf16x4*>(dst + (size_t)rows[i] * D + c) = *reinterpret_cast<const bf16x4*>(src + (size_t)i * D + c);
  }
}

// ---- tokenizer backward -------------------------------------------------------------------------
// (1) compact the patch-token rows of dtok into dpatch_tok [Mp, D]
__global__ __launch_bounds__(256) void tokbwd_compact_kernel(const bf16_t* __restrict__ dtok, const int* __restrict__ chan_img,
                                                             bf16_t* __restrict__ dpatch, int Mp, int p, int D) {
  const int cpr = D / 8;
  const size_t total = (size_t)Mp * cpr;
  for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(id / cpr), c = (int)(id % cpr) * 8;
    const int row = m + chan_img[m / p] + 1;
    *reinterpret_cast<u32x4*>(dpatch + (size_t)m * D + c) = *reinterpret_cast<const u32x4*>(dtok + (size_t)row * D + c);
  }
}
// (2) ONE pass over the rows of dtok into a [max_c][p + 1][D] table:
//       slot[c][j][:]  (j < p) = sum over the channel instances gc with chan_idx[gc] == c of dtok[row(gc, j)][:],
//                                row(gc, j) = gc * p + j + chan_img[gc] + 1;
//       slot[c][p][:]          = sum over the images i = c, c + max_c, ... of dtok[cu[i]][:]   (the CLS rows, split max_c ways).
//     Block = (j, c).  The block first compacts its row list into LDS (one vectorised scan of chan_idx -- blocks of unused slots
//     leave after it), then the 4 waves walk the list 8 rows at a time with independent loads: the first version walked the
//     instance list with one dependent load per iteration and was latency-bound (168 us for 58 MB).  A lane owns 4 consecutive
//     columns (8-byte loads); partials meet in LDS.  Everything the tokenizer's parameter gradients need is a marginal:
//       dpos[j] = sum_c slot[c][j]      dchan[c] = sum_j slot[c][j]      dcls = sum_c slot[c][p]
constexpr int TOKB_MAXLIST = 4096;  // rows one (j, c) block lists at a time (larger batches go through in windows)
__global__ __launch_bounds__(256) void tokbwd_slot_kernel(const bf16_t* __restrict__ dtok, const int* __restrict__ cu,
                                                          const int* __restrict__ chan_img, const int* __restrict__ chan_idx,
                                                          float* __restrict__ slot, int B, int n_chan, int p, int D, int max_c) {
  __shared__ int rows[TOKB_MAXLIST];
  __shared__ int wcnt[4];
  __shared__ f32x4 red[4][64];
  const int j = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  // ordered compaction (ballot prefix, no atomics): the list -- and with it the summation order -- is the same on every run.
  // Items are taken in windows of TOKB_MAXLIST (a window's hits always fit the list); the sums carry over in registers.
  const int n_items = j < p ? n_chan : (B - c + max_c - 1) / max_c;
  f32x4 acc[4];  // D <= 1024: up to four 256-column slices per lane
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < n_items; base += TOKB_MAXLIST) {
    const int lim = min(n_items, base + TOKB_MAXLIST);
    int n = 0;
    for (int g0 = base; g0 < lim; g0 += 256) {
      const int idx = g0 + tid;
      bool hit = false;
      int row = 0;
      if (idx < lim) {
        if (j < p) {
          hit = chan_idx[idx] == c;
          row = idx * p + j + chan_img[idx] + 1;
        } else {
          hit = true;
          row = cu[c + idx * max_c];
        }
      }
      const unsigned long long bal = __ballot(hit);
      if (l == 0) wcnt[w] = __popcll(bal);
      __syncthreads();
      int woff = 0;
      for (int q = 0; q < w; ++q) woff += wcnt[q];
      const int at = n + woff + __popcll(bal & ((1ull << l) - 1ull));
      if (hit) rows[at] = row;
      n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
    }
#pragma unroll
    for (int di = 0; di < 4; ++di) {
      const int d = di * 256 + 4 * l;
      if (d < D) {
        f32x4 s = acc[di];
        int k = w;
        for (; k + 28 < n; k += 32) {  // 8 rows per wave and step: independent loads
          bf16x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const bf16x4*>(dtok + (size_t)rows[k + 4 * u] * D + d);
#pragma unroll
          for (int u = 0; u < 8; ++u) { s[0] += (float)v[u][0]; s[1] += (float)v[u][1]; s[2] += (float)v[u][2]; s[3] += (float)v[u][3]; }
        }
        for (; k < n; k += 4) {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>(dtok + (size_t)rows[k] * D + d);
          s[0] += (float)v[0]; s[1] += (float)v[1]; s[2] += (float)v[2]; s[3] += (float)v[3];
        }
        acc[di] = s;
      }
    }
    __syncthreads();  // the list is rebuilt by the next window
  }
#pragma unroll
  for (int di = 0; di < 4; ++di) {
    const int d = di * 256 + 4 * l;
    if (di * 256 < D) {
      red[w][l] = acc[di];
      __syncthreads();
      if (w == 0 && d < D)
        *reinterpret_cast<f32x4*>(slot + ((size_t)c * (p + 1) + j) * D + d) = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
      __syncthreads();
    }
  }
}
// (3) the marginals.  Blocks [0, p): dpos rows; [p, p + max_c): dchan rows; block p + max_c: dcls.
__global__ __launch_bounds__(256) void tokbwd_finish_kernel(const float* __restrict__ slot, float* __restrict__ dpos,
                                                            float* __restrict__ dchan, float* __restrict__ dcls, int p, int max_c,
                                                            int D) {
  __shared__ float red[4][64];
  const int b = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t cs = (size_t)(p + 1) * D;  // table stride between channel slots
  for (int d0 = 0; d0 < D; d0 += 64) {
    const int d = d0 + l;
    float s = 0.f;
    if (d < D) {
      if (b < p) {
        for (int c = w; c < max_c; c += 4) s += slot[c * cs + (size_t)b * D + d];
      } else if (b < p + max_c) {
        const float* base = slot + (size_t)(b - p) * cs + d;
        int jj = w;
        for (; jj + 28 < p; jj += 32) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(jj + 4 * u) * D];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; jj < p; jj += 4) s += base[(size_t)jj * D];
      } else {
        for (int c = w; c < max_c; c += 4) s += slot[c * cs + (size_t)p * D + d];
      }
    }
    red[w][l] = s;
    __syncthreads();
    if (w == 0 && d < D) {
      const float t = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
      if (b < p) dpos[(size_t)b * D + d] = t;
      else if (b < p + max_c) dchan[(size_t)(b - p) * D + d] = t;
      else dcls[d] = t;
    }
    __syncthreads();
  }
}

inline int grid_for(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int chadavit_im2col(const float* x, chada_bf16* patches, int n_chan, int S, int patch, void* stream) {
  CHADA_ENTRY();
  if (!x || !patches || n_chan <= 0 || S <= 0) return 1;
  if (patch % 8 != 0 || S % patch != 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)n_chan * S * (S / 8);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, x, reinterpret_cast<bf16_t*>(patches),
                     n_chan, S, patch);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_write_cls(chada_bf16* tokens, const int* cu_seqlens, const float* cls, const float* pos0, int B, int D,
                                  void* stream) {
  CHADA_ENTRY();
  if (!tokens || !cu_seqlens || !cls || !pos0 || B <= 0 || D <= 0) return 1;
  hipLaunchKernelGGL(write_cls_kernel, dim3(grid_for((size_t)B * D)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<bf16_t*>(tokens), cu_seqlens, cls, pos0, B, D);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_gather_rows(const chada_bf16* src, const int* rows, chada_bf16* dst, int n_rows, int D, void* stream) {
  CHADA_ENTRY();
  if (!src || !rows || !dst || n_rows <= 0 || D <= 0) return 1;
  if (D % 4 != 0) return 2;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)n_rows * D / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(src), rows,
                     reinterpret_cast<bf16_t*>(dst), n_rows, D);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_scatter_rows_zero(const chada_bf16* src, const int* rows, chada_bf16* dst, int n_rows, int T, int D,
                                          void* stream) {
  CHADA_ENTRY();
  if (!src || !rows || !dst || n_rows <= 0 || T <= 0 || D <= 0) return 1;
  if (D % 8 != 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n8 = (size_t)T * D / 8;
  hipLaunchKernelGGL(zero_bf16_kernel, dim3(grid_for(n8, 8192)), dim3(256), 0, s, reinterpret_cast<bf16_t*>(dst), n8);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((size_t)n_rows * D / 4)), dim3(256), 0, s,
                     reinterpret_cast<const bf16_t*>(src), rows, reinterpret_cast<bf16_t*>(dst), n_rows, D);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_tokenizer_bwd(const chada_bf16* dtok_, const int* cu_seqlens, const int* chan_img, const int* chan_idx,
                                      chada_bf16* dpatch_tok, float* dpos, float* dchan, float* dcls, float* workspace,
                                      int B, int n_chan, int p, int D, int max_channels, void* stream) {
  CHADA_ENTRY();
  if (!dtok_ || !cu_seqlens || !chan_img || !chan_idx || !dpatch_tok || !dpos || !dchan || !dcls || !workspace) return 1;
  if (B <= 0 || n_chan <= 0 || p <= 0 || D % 8 != 0 || D > 1024 || max_channels <= 0 || ((uintptr_t)workspace & 15) != 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* dtok = reinterpret_cast<const bf16_t*>(dtok_);
  bf16_t* dpatch = reinterpret_cast<bf16_t*>(dpatch_tok);
  const int Mp = n_chan * p;
  hipLaunchKernelGGL(tokbwd_compact_kernel, dim3(grid_for((size_t)Mp * D / 8, 8192)), dim3(256), 0, s, dtok, chan_img, dpatch,
                     Mp, p, D);
  hipLaunchKernelGGL(tokbwd_slot_kernel, dim3(p + 1, max_channels), dim3(256), 0, s, dtok, cu_seqlens, chan_img, chan_idx, workspace, B,
                     n_chan, p, D, max_channels);
  hipLaunchKernelGGL(tokbwd_finish_kernel, dim3(p + max_channels + 1), dim3(256), 0, s, workspace, dpos, dchan, dcls, p, max_channels, D);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" long long chadavit_tokenizer_bwd_workspace_floats(int p, int D, int max_channels) {
  if (p <= 0 || D <= 0 || max_channels <= 0) return -1;
  return (long long)max_channels * (p + 1) * D;
}


// ---- per-channel intensity jitter on the collated crop tensor (the tokenizer's input), in place.
// reference: CustomColorJitter.apply, src/data/custom_transforms.py:301-351 -- per channel image c:
// x <- clamp(gamma_c * (x + shift_c), 0, 1), with the optional horizontal flip of HorizontalFlip folded in (same pass; a
// flipped row is swapped pairwise inside the lane pair that owns it).  One block row per (channel image, image row).
namespace {
__global__ __launch_bounds__(256) void channel_jitter_kernel(float* __restrict__ x, const float* __restrict__ shift,
                                                             const float* __restrict__ gamma, const unsigned char* __restrict__ flip,
                                                             int S, long long rows) {
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const int c = (int)(r / S);
    const float sh = shift[c], gm = gamma[c];
    const bool fl = flip && flip[c];
    float* row = x + r * S;
    if (!fl) {
      for (int i = threadIdx.x; i < S; i += 256) row[i] = fminf(fmaxf(gm * (row[i] + sh), 0.f), 1.f);
    } else {
      for (int i = threadIdx.x; i < (S + 1) / 2; i += 256) {
        const int j = S - 1 - i;
        const float a = fminf(fmaxf(gm * (row[i] + sh), 0.f), 1.f);
        const float b = fminf(fmaxf(gm * (row[j] + sh), 0.f), 1.f);
        row[i] = b;
        row[j] = a;
      }
    }
  }
}
}  // namespace

extern "C" int chadavit_channel_jitter(float* x, const float* shift, const float* gamma, const unsigned char* flip,
                                       int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!x || !shift || !gamma || n_channel_images <= 0 || S <= 0) return 1;
  const long long rows = (long long)n_channel_images * S;
  const int grid = (int)(rows < 65536 ? rows : 65536);
  hipLaunchKernelGGL(channel_jitter_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, shift, gamma, flip, S,
                     rows);
  CHADA_CHECK_LAUNCH();
  return 0;
}
