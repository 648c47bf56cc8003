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
// (2) dpos[j, :] = sum over channel instances of dpatch[gc*p + j, :]   (block = patch position j)
__global__ __launch_bounds__(256) void tokbwd_dpos_kernel(const bf16_t* __restrict__ dpatch, float* __restrict__ dpos,
                                                          int n_chan, int p, int D) {
  // block = (patch position j, 64-column slice); the 4 waves split the channel instances, lanes own columns
  __shared__ float red[4][64];
  const int j = blockIdx.x, d = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float s = 0.f;
  if (d < D)
    for (int gc = w; gc < n_chan; gc += 4) s += (float)dpatch[((size_t)gc * p + j) * D + d];
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && d < D) dpos[(size_t)j * D + d] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// (3) per channel-slot sums: dchan[c, :] = sum_{gc: chan_idx[gc]==c} sum_j dpatch[gc*p+j, :]
//     grid = (max_channels, SPL); each block sums a strided share of the channel instances -> partial slabs
__global__ __launch_bounds__(256) void tokbwd_dchan_part_kernel(const bf16_t* __restrict__ dpatch,
                                                                const int* __restrict__ chan_idx, float* __restrict__ part,
                                                                int n_chan, int p, int D, int max_c) {
  const int c = blockIdx.x, sp = blockIdx.y, nsp = gridDim.y;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int gc = sp; gc < n_chan; gc += nsp) {
      if (chan_idx[gc] != c) continue;
      const bf16_t* base = dpatch + (size_t)gc * p * D + d;
      for (int j = 0; j < p; ++j) s += (float)base[(size_t)j * D];
    }
    part[((size_t)sp * max_c + c) * D + d] = s;
  }
}
__global__ __launch_bounds__(256) void tokbwd_finish_kernel(const float* __restrict__ part, const bf16_t* __restrict__ dtok,
                                                            const int* __restrict__ cu, float* __restrict__ dchan,
                                                            float* __restrict__ dcls, int nsp, int max_c, int B, int D) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id < max_c * D) {
    float s = 0.f;
    for (int sp = 0; sp < nsp; ++sp) s += part[(size_t)sp * max_c * D + id];
    dchan[id] = s;
  } else if (id < (max_c + 1) * D) {
    const int d = id - max_c * D;
    float s = 0.f;
    for (int i = 0; i < B; ++i) s += (float)dtok[(size_t)cu[i] * D + d];
    dcls[d] = s;
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
  if (B <= 0 || n_chan <= 0 || p <= 0 || D % 8 != 0 || max_channels <= 0) return 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* dtok = reinterpret_cast<const bf16_t*>(dtok_);
  bf16_t* dpatch = reinterpret_cast<bf16_t*>(dpatch_tok);
  const int Mp = n_chan * p;
  hipLaunchKernelGGL(tokbwd_compact_kernel, dim3(grid_for((size_t)Mp * D / 8, 8192)), dim3(256), 0, s, dtok, chan_img, dpatch,
                     Mp, p, D);
  hipLaunchKernelGGL(tokbwd_dpos_kernel, dim3(p, (D + 63) / 64), dim3(256), 0, s, dpatch, dpos, n_chan, p, D);
  const int nsp = chadavit_tokenizer_bwd_splits();
  hipLaunchKernelGGL(tokbwd_dchan_part_kernel, dim3(max_channels, nsp), dim3(256), 0, s, dpatch, chan_idx, workspace, n_chan, p,
                     D, max_channels);
  hipLaunchKernelGGL(tokbwd_finish_kernel, dim3(((max_channels + 1) * D + 255) / 256), dim3(256), 0, s, workspace, dtok,
                     cu_seqlens, dchan, dcls, nsp, max_channels, B, D);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_tokenizer_bwd_splits(void) { return 128; }


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
