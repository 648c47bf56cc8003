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

// One block = one 4-pixel-wide strip set of ONE channel image (blockIdx.y): the descriptor, the jitter pair and the scale factors are
// wave-uniform (scalar loads, one f64 division per thread instead of two per pixel), no 64-bit index division per pixel, and a thread
// produces four neighbouring pixels of a row -- they share the row weights and most of their 4 x 4 source taps (round 4: the first
// version, one thread per output pixel with everything per pixel, ran at ~0.4 TB/s: ~7 ms per 512-image multi-crop batch).
// Per pixel the arithmetic -- and therefore the result, bit for bit -- is the first version's.
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ src, const long long* __restrict__ desc_,
                                                          const float* __restrict__ shift, const float* __restrict__ gamma,
                                                          float* __restrict__ out, int S, int quads_per_row) {
  const int c = blockIdx.y;
  const CropDesc d = *reinterpret_cast<const CropDesc*>(desc_ + 8 * (size_t)c);
  const int cw = (int)d.cw, chh = (int)d.ch, W = (int)d.W;
  const float* p = src + d.src_off + d.y0 * d.W + d.x0;
  float* o = out + (size_t)c * S * S;
  const bool jit = shift != nullptr && gamma[c] >= 0.f;
  const float g = jit ? gamma[c] : 1.f, sh = jit ? shift[c] : 0.f;
  const bool copy = cw == S && chh == S;   // cv2.resize returns a copy when the size is unchanged
  const double scx = (double)cw / S, scy = (double)chh / S;
  const bool vec = (S & 3) == 0 && ((uintptr_t)o & 15) == 0;
  const int n_quads = quads_per_row * S;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < n_quads; q += gridDim.x * 256) {
    const int dy = q / quads_per_row, x4 = (q - dy * quads_per_row) * 4;
    float wy[4];
    int yy[4];
    if (!copy) {
      const float fy = (float)((dy + 0.5) * scy - 0.5);
      const int sy = (int)floorf(fy);
      cubic_w(fy - sy, wy);
#pragma unroll
      for (int j = 0; j < 4; ++j) yy[j] = min(max(sy - 1 + j, 0), chh - 1);
    }
    float res[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int dxo = x4 + e;
      if (dxo >= S) { res[e] = 0.f; continue; }
      const int dx = d.flip ? S - 1 - dxo : dxo;
      float v;
      if (copy) {
        v = p[(size_t)dy * W + dx];
      } else {
        const float fx = (float)((dx + 0.5) * scx - 0.5);
        const int sx = (int)floorf(fx);
        float wx[4];
        cubic_w(fx - sx, wx);
        int xx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xx[i] = min(max(sx - 1 + i, 0), cw - 1);
        v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* row = p + (size_t)yy[j] * W;
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) a += wx[i] * row[xx[i]];
          v += wy[j] * a;
        }
      }
      // CustomColorJitter (custom_transforms.py:327-345); gamma < 0 marks a channel image whose sample did not draw the transform
      if (jit) v = fminf(fmaxf(g * (v + sh), 0.f), 1.f);
      res[e] = v;
    }
    float* dst = o + (size_t)dy * S + x4;
    if (vec) {
      *reinterpret_cast<f32x4*>(dst) = f32x4{res[0], res[1], res[2], res[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x4 + e < S) dst[e] = res[e];
    }
  }
}

// per channel image: fin[c*12 + ..] = {ksize (0 = no blur), w0..w6 (1-D Gaussian taps, centred), solarize threshold, solarize max,
// normalise mean * max_pixel_value, 1 / (std * max_pixel_value)}
__global__ __launch_bounds__(256) void blur_finish_kernel(const float* __restrict__ in, const float* __restrict__ fin,
                                                          float* __restrict__ out, int S) {
  const int c = blockIdx.y;   // one channel image per block row: its descriptor is wave-uniform
  const float* f = fin + 12 * (size_t)c;
  const float* img = in + (size_t)c * S * S;
  float* o = out + (size_t)c * S * S;
  const int k = (int)f[0], r = k >> 1;
  const float thr = f[8], smax = f[9], mean = f[10], istd = f[11];
  const float* w = f + 1;   // the 1-D taps (wave-uniform addresses)
  const int n = S * S;
  for (int rem = blockIdx.x * 256 + threadIdx.x; rem < n; rem += gridDim.x * 256) {
    const int y = rem / S, x = rem - y * S;
    float v;
    if (k <= 1) {
      v = img[rem];
    } else {
      v = 0.f;
      for (int j = -r; j <= r; ++j) {
        int yy = y + j;
        yy = yy < 0 ? -yy : (yy >= S ? 2 * S - 2 - yy : yy);  // BORDER_REFLECT_101
        float a = 0.f;
        for (int i = -r; i <= r; ++i) {
          int xx = x + i;
          xx = xx < 0 ? -xx : (xx >= S ? 2 * S - 2 - xx : xx);
          a += w[i + r] * img[yy * S + xx];
        }
        v += w[j + r] * a;
      }
    }
    if (v >= thr) v = smax - v;   // Solarize: values at or above the threshold are inverted (threshold = +inf: off)
    o[rem] = (v - mean) * istd;   // Normalize: (x - mean * max_pixel_value) / (std * max_pixel_value); identity = {0, 1}
  }
}
}  // namespace

extern "C" int chadavit_crop_resize(const float* src, const long long* desc, const float* shift, const float* gamma, float* out,
                                    int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!src || !desc || !out || n_channel_images <= 0 || S <= 0 || (shift == nullptr) != (gamma == nullptr)) return 1;
  if (n_channel_images > 65535) return 2;
  const int qpr = (S + 3) / 4;
  int gx = (qpr * S + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)gx, (unsigned)n_channel_images), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                     desc, shift, gamma, out, S, qpr);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_blur_finish(const float* in, const float* fin, float* out, int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!in || !fin || !out || in == out || n_channel_images <= 0 || S <= 1) return 1;
  if (n_channel_images > 65535) return 2;
  int gx = (S * S + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(blur_finish_kernel, dim3((unsigned)gx, (unsigned)n_channel_images), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, fin,
                     out, S);
  CHADA_CHECK_LAUNCH();
  return 0;
}
