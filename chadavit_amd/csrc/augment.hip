// Device side of the multi-crop augmentation contract: the per-crop arithmetic of
// build_transform_pipeline (src/data/pretrain_dataloader.py:272-328) for the IDRCell100k-style float images, emitted directly in
// the A1 collate layout (sum C, 1, S, S) the tokenizer consumes (src/data/channels_strategies.py:31-85), so that real data can
// replace the synthetic tensors without a per-channel Python loop on the host.
//
//   kernel 1 (crop_resize_kernel):  RandomResizedCrop / Resize with cv2.INTER_CUBIC  ->  CustomColorJitter  ->  HorizontalFlip
//   kernel 2 (blur_finish_kernel):  GaussianBlur (cv2, BORDER_REFLECT_101)           ->  Solarize           ->  Normalize
//
// (reference order: crop, jitter, [gray], blur, solarize, [equalize], flip, normalize.  The flip is pure indexing and commutes with
// the symmetric blur and the pointwise steps, so it is folded into kernel 1; ToGray / Equalize need 3-channel / uint8 images and do
// not apply to this path.)  The random parameters are drawn on the host (chadavit_amd/data/device_pipeline.py) in the order the
// reference's transforms draw them; the kernels are deterministic functions of their descriptors.
//
// cv2.INTER_CUBIC as OpenCV documents and implements it for float images (imgproc resize.cpp, interpolateCubic, A = -0.75):
//   fx = (dx + 0.5) * (src_w / dst_w) - 0.5 ; sx = floor(fx) ; t = fx - sx
//   w0 = ((A (t+1) - 5A)(t+1) + 8A)(t+1) - 4A ; w1 = ((A+2) t - (A+3)) t^2 + 1 ; w2 = ((A+2)(1-t) - (A+3))(1-t)^2 + 1 ; w3 = 1-w0-w1-w2
//   taps sx-1 .. sx+2, indices clamped to the source (= the crop window: cv2 resizes the cropped array), rows then columns.
// OpenCV itself is absent from the image: the kernel is pinned to the oracle's restatement of this algorithm, which is checked
// against torch's bicubic (documented to match OpenCV's: same A, same half-pixel mapping, clamped border) and closed forms.
#include "common.h"

namespace {
using namespace chada;

struct CropDesc {            // one per OUTPUT channel image (long long[8] on the host side)
  long long src_off;         // element offset of the source plane in the packed source buffer
  long long H, W;            // source plane size
  long long x0, y0, cw, ch;  // crop window (columns, rows)
  long long flip;            // horizontal flip of the output
};

__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
  const float A = -0.75f;
  w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
}

// ---- crop_resize --------------------------------------------------------------------------------------------------------------------
// One block = a band of `tile_rows` output rows of ONE channel image (blockIdx.y; the descriptor, the jitter pair and the scale factors are
// wave-uniform).  The interpolation's column half depends on the output column only and its row half on the output row only, so the block
// first builds, in LDS, the tap tables
//   column dxo -> { byte offsets of its 4 clamped source columns within a row, its 4 weights }     row dy -> { its 4 clamped source rows, weights }
// (the f64 half-pixel mapping and the cubic weights: once per column / row instead of once per pixel), then copies the source rows its band
// touches -- at most tile_rows * crop_h / S + 3 of them, crop_w wide -- into LDS with coalesced loads, and a pixel is 2 ds_read_b128 of
// tables + 16 x (add, ds_read_b32) + 20 FMA.  A thread produces four neighbouring pixels of a row (one f32x4 store when aligned).
// History (round 4, 1024 x 3-channel 256 x 256 images, the 224-pixel crop): the first version recomputed everything per pixel with 64-bit
// addresses, 1.45 ms; with the tables but the 16 taps still read from global memory, 1.35 ms although the VALU work had dropped 5x -- 16
// scattered dword loads per pixel are bound by the texture addresser (~16 cycles per wave64 dword load), not by VALU or HBM.  Per pixel
// the arithmetic -- mapping, weights, order of the 4 x 4 sum -- never changed, and the results are bit-identical across the versions.
// The band goes through LDS in sub-bands of as many output rows as fit a small budget (13 KB: occupancy beats band height); only a window so
// wide that not even one output row's four source rows fit reads its taps from global memory, with 32-bit byte offsets from the crop window's
// corner: a plane must be smaller than 2^30 pixels (host check in device_pipeline.py).
struct Taps {
  unsigned off[4];   // columns: byte offsets within a source row; rows: source row indices (relative to the crop window)
  float w[4];
};

// Pixels of output rows [ty0, ty0 + nrows) of the block's band: v = sum_j wy[j] (sum_i wx[i] src[yy[j]][xx[i]]) with the taps read through
// `ld` (byte offset -> value: the staged rows in LDS, or global memory); a source row yy sits at byte (yy - base) * pitch.  A thread owns ONE
// output column and four rows of it: neighbouring lanes read neighbouring source columns (stride crop_w / S words: no LDS bank pile-up, which
// a thread-per-4-columns mapping has at stride ~4), the column taps are read once per four pixels, the row taps are a broadcast, and a
// wave's stores are whole lines.
template <class Ld, class Jit>
__device__ __forceinline__ void resize_band(Ld ld, Jit jitter, const Taps* cols, const Taps* rows, float* __restrict__ o, int S, unsigned inv_S,
                                            int ty0, int nrows, unsigned base, unsigned pitch) {
  const int n_items = ((nrows + 3) >> 2) * S;
  for (int q = threadIdx.x; q < n_items; q += 256) {
    const int rg = S == 1 ? q : (int)__umulhi((unsigned)q, inv_S), x = q - rg * S;   // q / S (exact: q * S < 2^32)
    const Taps cx = cols[x];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ty = rg * 4 + e;
      if (ty < nrows) {
        const Taps ry = rows[ty0 + ty];
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned ro = (ry.off[j] - base) * pitch;
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) a += cx.w[i] * ld(ro + cx.off[i]);
          v += ry.w[j] * a;
        }
        o[(unsigned)((ty0 + ty) * S + x)] = jitter(v);
      }
    }
  }
}

// T: the element type of the source planes -- float, or the 8 / 16-bit unsigned integers image files store (IDRCell100k: the reference
// reader casts them to float32 on the host, custom_datasets.py:181-190; here they travel as stored and become float, exactly, where a
// value is first touched: while a band is staged, or per tap on the global-tap path).
template <typename T>
__global__ __launch_bounds__(256) void crop_resize_kernel(const T* __restrict__ src, const long long* __restrict__ desc_,
                                                          const float* __restrict__ shift, const float* __restrict__ gamma,
                                                          float* __restrict__ out, int S, int quads_per_row, unsigned inv_S, int tile_rows, int band_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Taps* cols = reinterpret_cast<Taps*>(smem_raw);   // [S]
  Taps* rows = cols + S;                             // [tile_rows]
  const char* band = reinterpret_cast<const char*>(rows + tile_rows);   // [band_bytes]
  const int c = blockIdx.y;
  const CropDesc d = *reinterpret_cast<const CropDesc*>(desc_ + 8 * (size_t)c);
  const int cw = (int)d.cw, chh = (int)d.ch, W = (int)d.W;
  const T* pf = src + d.src_off + d.y0 * d.W + d.x0;
  const char* p = reinterpret_cast<const char*>(pf);
  float* o = out + (size_t)c * S * S;
  const bool jit = shift != nullptr && gamma[c] >= 0.f;
  const float g = jit ? gamma[c] : 1.f, sh = jit ? shift[c] : 0.f;
  const bool copy = cw == S && chh == S;   // cv2.resize returns a copy when the size is unchanged
  const double scx = (double)cw / S, scy = (double)chh / S;
  const bool flip = d.flip != 0;
  const int y0 = blockIdx.x * tile_rows, nrows = min(tile_rows, S - y0);
  for (int t = threadIdx.x; t < S + nrows; t += 256) {
    const bool is_row = t >= S;
    const int dd = is_row ? y0 + t - S : (flip ? S - 1 - t : t);   // the flip is an index reversal of the output columns
    const int lim = is_row ? chh : cw;
    const unsigned unit = is_row ? 1u : 4u;
    Taps e;
    if (copy) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { e.off[i] = (unsigned)dd * unit; e.w[i] = 0.f; }
    } else {
      const float f = (float)((dd + 0.5) * (is_row ? scy : scx) - 0.5);
      const int s0 = (int)floorf(f);
      cubic_w(f - s0, e.w);
#pragma unroll
      for (int i = 0; i < 4; ++i) e.off[i] = (unsigned)min(max(s0 - 1 + i, 0), lim - 1) * unit;
    }
    (is_row ? rows : cols)[is_row ? t - S : t] = e;
  }
  __syncthreads();
  o += (size_t)y0 * S;
  // CustomColorJitter (custom_transforms.py:327-345); gamma < 0 marks a channel image whose sample did not draw the transform
  auto jitter = [&](float v) { return jit ? fminf(fmaxf(g * (v + sh), 0.f), 1.f) : v; };
  if (copy) {
    const int n_quads = quads_per_row * nrows;
    for (int q = threadIdx.x; q < n_quads; q += 256) {
      const int ty = q / quads_per_row, x4 = (q - ty * quads_per_row) * 4;
      const unsigned ro = rows[ty].off[0] * (unsigned)W;
      for (int e = 0; e < 4 && x4 + e < S; ++e) o[(unsigned)(ty * S + x4 + e)] = jitter((float)pf[ro + (cols[x4 + e].off[0] >> 2)]);
    }
    return;
  }
  // The band's source rows go through LDS in SUB-BANDS of as many output rows as fit `band_bytes` (the tables are monotonic: a sub-band
  // touches source rows first row's first tap .. last row's last tap, at most rows * crop_h / S + 4 of them, crop_w wide).  A small budget
  // is the point: ~13 KB per block keeps 7 blocks on a CU and the three phases of a sub-band (stage, barrier, taps) short -- 8-row bands in
  // 13 KB measured 1.84 ms per 1 024-image batch against 2.32 ms for 16-row bands in 32 KB and 4.1 ms for 32 rows in 48 KB.  Only when not even
  // one output row fits (a window wider than band_bytes / 16) do the taps come from global memory.
  const int fit_src = band_bytes / (4 * cw);                                             // source rows the budget holds
  const int sub = fit_src >= 5 ? max(1, min(nrows, (int)((fit_src - 4) / (scy > 1e-9 ? scy : 1e-9)))) : 0;   // (uniform)
  if (sub == 0) {
    if (sizeof(T) != 4) {   // the column records hold float byte offsets: in units of T for the global taps
      __syncthreads();
      for (int t = threadIdx.x; t < S; t += 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cols[t].off[i] = (cols[t].off[i] >> 2) * (unsigned)sizeof(T);
      }
      __syncthreads();
    }
    resize_band([&](unsigned off) { return (float)*reinterpret_cast<const T*>(p + off); }, jitter, cols, rows, o, S, inv_S, 0, nrows, 0u,
                (unsigned)sizeof(T) * (unsigned)W);
    return;
  }
  float* bw = reinterpret_cast<float*>(const_cast<char*>(band));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int s0 = 0; s0 < nrows; s0 += sub) {
    const int rs = min(sub, nrows - s0);
    const unsigned r0 = rows[s0].off[0];
    unsigned n_src = rows[s0 + rs - 1].off[3] - r0 + 1;
    if ((size_t)n_src * cw * 4 > (size_t)band_bytes) n_src = (unsigned)fit_src;   // (cannot happen with the bound above; never overrun the buffer)
    if (s0) __syncthreads();   // the previous sub-band's taps are done with the buffer
    for (unsigned r = wv; r < n_src; r += 4) {
      const T* srow = pf + (size_t)(r0 + r) * W;
      for (int x = lane; x < cw; x += 64) bw[r * cw + x] = (float)srow[x];
    }
    __syncthreads();
    resize_band([&](unsigned off) { return *reinterpret_cast<const float*>(band + off); }, jitter, cols, rows, o, S, inv_S, s0, rs, r0,
                4u * (unsigned)cw);
  }
}

// ---- blur_finish ----------------------------------------------------------------------------------------------------------------------
// per channel image: fin[c*12 + ..] = {ksize (0 = no blur), w0..w6 (1-D Gaussian taps, centred), solarize threshold, solarize max,
// normalise mean * max_pixel_value, 1 / (std * max_pixel_value)}
//
// One block = a band of `tile_rows` rows of ONE channel image (its descriptor is wave-uniform).  The blur is computed the way its
// definition nests -- out(y, x) = sum_j w_j (sum_i w_i in(y + j, x + i)) -- but each inner (horizontal) sum once: the band's rows plus
// r halo rows go through the horizontal pass into LDS, the vertical pass reads K of them per pixel.  Same sums in the same order as the
// first version (which evaluated all K x K taps per pixel, ~400 VALU instructions per pixel at K = 7): identical results.
__device__ __forceinline__ int reflect101(int v, int last) {   // BORDER_REFLECT_101, one bounce (K <= 7 and S >= 4)
  v = v < 0 ? -v : v;
  return v > last ? 2 * last - v : v;
}

__device__ __forceinline__ float finish_px(float v, float thr, float smax, float mean, float istd) {
  if (v >= thr) v = smax - v;   // Solarize: values at or above the threshold are inverted (threshold = +inf: off)
  return (v - mean) * istd;     // Normalize: (x - mean * max_pixel_value) / (std * max_pixel_value); identity = {0, 1}
}

template <int K>
__device__ __forceinline__ void blur_band(const float* __restrict__ img, float* __restrict__ o, const float* __restrict__ w, float* tmp,
                                          int S, unsigned inv_S, unsigned inv_Q, bool vec, int y0, int rows, float thr, float smax,
                                          float mean, float istd) {
  constexpr int R = K / 2;
  float wk[K];
#pragma unroll
  for (int i = 0; i < K; ++i) wk[i] = w[i];
  const int last = S - 1, rows_h = rows + 2 * R;   // horizontal pass: band rows y0 - R .. y0 + rows - 1 + R (reflected), every column
  if (vec) {
    // S % 4 == 0, 16-byte aligned planes: a thread produces four neighbouring columns of a row from the 4 + 2 R inputs under them (three
    // aligned 16-byte loads in the interior; the first and the last quad of a row take their reflected neighbours one by one)
    const int Q = S >> 2, n_h = rows_h * Q;
    for (int e = threadIdx.x; e < n_h; e += 256) {
      const int ry = (int)__umulhi((unsigned)e, inv_Q), xq = (e - ry * Q) * 4;   // e / Q (exact: e * Q < 2^32)
      const float* row = img + (unsigned)(reflect101(y0 - R + ry, last) * S);
      float in[12];   // columns xq - 4 .. xq + 7
      if (xq >= 4 && xq + 8 <= S) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + (unsigned)(xq - 4 + 4 * t));
          in[4 * t] = v[0]; in[4 * t + 1] = v[1]; in[4 * t + 2] = v[2]; in[4 * t + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int t = 4 - R; t < 8 + R; ++t) in[t] = row[(unsigned)reflect101(xq - 4 + t, last)];
      }
      float a[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        a[px] = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) a[px] += wk[i] * in[4 + px + i - R];
      }
      *reinterpret_cast<f32x4*>(tmp + ry * S + xq) = f32x4{a[0], a[1], a[2], a[3]};
    }
    __syncthreads();
    const int n_v = rows * Q;
    for (int e = threadIdx.x; e < n_v; e += 256) {
      const int ty = (int)__umulhi((unsigned)e, inv_Q), xq = (e - ty * Q) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tmp + (ty + j) * S + xq);
#pragma unroll
        for (int px = 0; px < 4; ++px) v[px] += wk[j] * t[px];
      }
      *reinterpret_cast<f32x4*>(o + (unsigned)((y0 + ty) * S + xq)) =
          f32x4{finish_px(v[0], thr, smax, mean, istd), finish_px(v[1], thr, smax, mean, istd), finish_px(v[2], thr, smax, mean, istd),
                finish_px(v[3], thr, smax, mean, istd)};
    }
    return;
  }
  const int n_h = rows_h * S;
  for (int e = threadIdx.x; e < n_h; e += 256) {
    const int ry = (int)__umulhi((unsigned)e, inv_S), x = e - ry * S;   // e / S (exact: e * S < 2^32)
    const float* row = img + (unsigned)(reflect101(y0 - R + ry, last) * S);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) a += wk[i] * row[(unsigned)reflect101(x + i - R, last)];
    tmp[e] = a;
  }
  __syncthreads();
  const int n_v = rows * S;
  for (int e = threadIdx.x; e < n_v; e += 256) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) v += wk[j] * tmp[e + j * S];
    o[(unsigned)(y0 * S + e)] = finish_px(v, thr, smax, mean, istd);
  }
}

__global__ __launch_bounds__(256) void blur_finish_kernel(const float* __restrict__ in, const float* __restrict__ fin,
                                                          float* __restrict__ out, int S, int tile_rows, unsigned inv_S, unsigned inv_Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tmp = reinterpret_cast<float*>(smem_raw);   // [(tile_rows + 6) * S]
  const int c = blockIdx.y;
  const float* f = fin + 12 * (size_t)c;
  const float* img = in + (size_t)c * S * S;
  float* o = out + (size_t)c * S * S;
  const int k = (int)f[0];
  const float thr = f[8], smax = f[9], mean = f[10], istd = f[11];
  const int y0 = blockIdx.x * tile_rows, rows = min(tile_rows, S - y0);
  const bool vec = (S & 3) == 0 && S >= 8 && (((uintptr_t)img | (uintptr_t)o) & 15) == 0;
  if (k <= 1) {   // no blur: the pointwise tail only
    const int n = rows * S, base = y0 * S;
    if (vec) {
      for (int e = threadIdx.x * 4; e < n; e += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(img + (unsigned)(base + e));
        *reinterpret_cast<f32x4*>(o + (unsigned)(base + e)) =
            f32x4{finish_px(v[0], thr, smax, mean, istd), finish_px(v[1], thr, smax, mean, istd), finish_px(v[2], thr, smax, mean, istd),
                  finish_px(v[3], thr, smax, mean, istd)};
      }
    } else {
      for (int e = threadIdx.x; e < n; e += 256) o[(unsigned)(base + e)] = finish_px(img[(unsigned)(base + e)], thr, smax, mean, istd);
    }
    return;
  }
  const float* w = f + 1;   // the 1-D taps (wave-uniform addresses)
  if (k == 3) blur_band<3>(img, o, w, tmp, S, inv_S, inv_Q, vec, y0, rows, thr, smax, mean, istd);
  else if (k == 5) blur_band<5>(img, o, w, tmp, S, inv_S, inv_Q, vec, y0, rows, thr, smax, mean, istd);
  else blur_band<7>(img, o, w, tmp, S, inv_S, inv_Q, vec, y0, rows, thr, smax, mean, istd);
}
}  // namespace

namespace {
template <typename T>
int crop_resize_launch(const T* src, const long long* desc, const float* shift, const float* gamma, float* out, int n_channel_images, int S,
                       void* stream) {
  const int qpr = (S + 3) / 4;
  // rows per block: ~6 pixels per thread and sub-band, so that the column table (rebuilt by every block of an image) is amortised; 13 KB of
  // staged source rows per block (see the kernel: occupancy beats band height)
  int tile_rows = 384 / qpr;
  tile_rows = tile_rows < 8 ? 8 : (tile_rows + 3) & ~3;
  if (tile_rows > S) tile_rows = S;
  const unsigned inv_S = (unsigned)(0x100000000ull / (unsigned)(S > 1 ? S : 2)) + 1u;   // q / S = umulhi(q, inv_S) for the q that occur (S = 1: unused)
  const int band_bytes = 13312;
  const int gx = (S + tile_rows - 1) / tile_rows;
  hipLaunchKernelGGL(crop_resize_kernel<T>, dim3((unsigned)gx, (unsigned)n_channel_images), dim3(256),
                     (size_t)(S + tile_rows) * sizeof(Taps) + band_bytes, reinterpret_cast<hipStream_t>(stream), src, desc, shift, gamma, out, S,
                     qpr, inv_S, tile_rows, band_bytes);
  CHADA_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int chadavit_crop_resize_src(const void* src, int src_kind, const long long* desc, const float* shift, const float* gamma, float* out,
                                        int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!src || !desc || !out || n_channel_images <= 0 || S <= 0 || (shift == nullptr) != (gamma == nullptr)) return 1;
  if (n_channel_images > 65535 || S > 1024) return 2;   // (S + band-height tap records of 32 bytes and the staged rows in LDS)
  switch (src_kind) {
    case 0: return crop_resize_launch(static_cast<const float*>(src), desc, shift, gamma, out, n_channel_images, S, stream);
    case 1: return crop_resize_launch(static_cast<const unsigned char*>(src), desc, shift, gamma, out, n_channel_images, S, stream);
    case 2: return crop_resize_launch(static_cast<const unsigned short*>(src), desc, shift, gamma, out, n_channel_images, S, stream);
    default: return 1;
  }
}

extern "C" int chadavit_crop_resize(const float* src, const long long* desc, const float* shift, const float* gamma, float* out,
                                    int n_channel_images, int S, void* stream) {
  return chadavit_crop_resize_src(src, 0, desc, shift, gamma, out, n_channel_images, S, stream);
}

extern "C" int chadavit_blur_finish(const float* in, const float* fin, float* out, int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!in || !fin || !out || in == out || n_channel_images <= 0 || S < 4) return 1;   // (one reflection per border: S > ksize / 2)
  if (n_channel_images > 65535 || S > 1024) return 2;
  // band height: 16 KB of horizontal-pass rows in LDS (band + 6 halo rows), at least 8 rows (measured per 1 024-image batch: 24 KB bands
  // 0.86 ms, 16 KB 0.82, 12 KB 0.86, 36 KB 1.07 -- halo recomputation against occupancy)
  int tile_rows = 4096 / S - 6;
  if (tile_rows < 8) tile_rows = 8;
  if (tile_rows > S) tile_rows = S;
  const int gx = (S + tile_rows - 1) / tile_rows;
  const unsigned inv_S = (unsigned)(0x100000000ull / (unsigned)S) + 1u;   // e / S = umulhi(e, inv_S) for the e that occur
  const unsigned inv_Q = (unsigned)(0x100000000ull / (unsigned)((S >> 2) > 0 ? (S >> 2) : 1)) + 1u;   // ... and e / (S / 4)
  hipLaunchKernelGGL(blur_finish_kernel, dim3((unsigned)gx, (unsigned)n_channel_images), dim3(256), (size_t)(tile_rows + 6) * S * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), in, fin, out, S, tile_rows, inv_S, inv_Q);
  CHADA_CHECK_LAUNCH();
  return 0;
}
