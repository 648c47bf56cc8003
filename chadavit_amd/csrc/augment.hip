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

__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ src, const long long* __restrict__ desc_,
                                                          const float* __restrict__ shift, const float* __restrict__ gamma,
                                                          float* __restrict__ out, int S, long long n_pix) {
  for (long long id = blockIdx.x * 256ll + threadIdx.x; id < n_pix; id += (long long)gridDim.x * 256ll) {
    const int c = (int)(id / ((long long)S * S));
    const int rem = (int)(id - (long long)c * S * S);
    const int dy = rem / S, dxo = rem - dy * S;
    const CropDesc d = *reinterpret_cast<const CropDesc*>(desc_ + 8 * c);
    const int dx = d.flip ? S - 1 - dxo : dxo;
    const int cw = (int)d.cw, chh = (int)d.ch, W = (int)d.W;
    const float* p = src + d.src_off + d.y0 * d.W + d.x0;
    float v;
    if (cw == S && chh == S) {
      v = p[(size_t)dy * W + dx];  // cv2.resize returns a copy when the size is unchanged
    } else {
      const float fx = (float)((dx + 0.5) * ((double)cw / S) - 0.5), fy = (float)((dy + 0.5) * ((double)chh / S) - 0.5);
      const int sx = (int)floorf(fx), sy = (int)floorf(fy);
      float wx[4], wy[4];
      cubic_w(fx - sx, wx);
      cubic_w(fy - sy, wy);
      v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = min(max(sy - 1 + j, 0), chh - 1);
        const float* row = p + (size_t)yy * W;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) a += wx[i] * row[min(max(sx - 1 + i, 0), cw - 1)];
        v += wy[j] * a;
      }
    }
    // CustomColorJitter (custom_transforms.py:327-345); gamma < 0 marks a channel image whose sample did not draw the transform
    if (shift && gamma[c] >= 0.f) v = fminf(fmaxf(gamma[c] * (v + shift[c]), 0.f), 1.f);
    out[id] = v;
  }
}

// per channel image: fin[c*12 + ..] = {ksize (0 = no blur), w0..w6 (1-D Gaussian taps, centred), solarize threshold, solarize max,
// normalise mean * max_pixel_value, 1 / (std * max_pixel_value)}
__global__ __launch_bounds__(256) void blur_finish_kernel(const float* __restrict__ in, const float* __restrict__ fin,
                                                          float* __restrict__ out, int S, long long n_pix) {
  for (long long id = blockIdx.x * 256ll + threadIdx.x; id < n_pix; id += (long long)gridDim.x * 256ll) {
    const int c = (int)(id / ((long long)S * S));
    const int rem = (int)(id - (long long)c * S * S);
    const int y = rem / S, x = rem - y * S;
    const float* f = fin + 12 * c;
    const float* img = in + (size_t)c * S * S;
    const int k = (int)f[0];
    float v;
    if (k <= 1) {
      v = img[rem];
    } else {
      const int r = k >> 1;
      v = 0.f;
      for (int j = -r; j <= r; ++j) {
        int yy = y + j;
        yy = yy < 0 ? -yy : (yy >= S ? 2 * S - 2 - yy : yy);  // BORDER_REFLECT_101
        float a = 0.f;
        for (int i = -r; i <= r; ++i) {
          int xx = x + i;
          xx = xx < 0 ? -xx : (xx >= S ? 2 * S - 2 - xx : xx);
          a += f[1 + i + r] * img[(size_t)yy * S + xx];
        }
        v += f[1 + j + r] * a;
      }
    }
    if (v >= f[8]) v = f[9] - v;   // Solarize: values at or above the threshold are inverted (threshold = +inf: off)
    out[id] = (v - f[10]) * f[11];  // Normalize: (x - mean * max_pixel_value) / (std * max_pixel_value); identity = {0, 1}
  }
}
}  // namespace

extern "C" int chadavit_crop_resize(const float* src, const long long* desc, const float* shift, const float* gamma, float* out,
                                    int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!src || !desc || !out || n_channel_images <= 0 || S <= 0 || (shift == nullptr) != (gamma == nullptr)) return 1;
  const long long n = (long long)n_channel_images * S * S;
  long long grid = (n + 255) / 256;
  if (grid > 65536) grid = 65536;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, desc, shift, gamma,
                     out, S, n);
  CHADA_CHECK_LAUNCH();
  return 0;
}

extern "C" int chadavit_blur_finish(const float* in, const float* fin, float* out, int n_channel_images, int S, void* stream) {
  CHADA_ENTRY();
  if (!in || !fin || !out || in == out || n_channel_images <= 0 || S <= 1) return 1;
  const long long n = (long long)n_channel_images * S * S;
  long long grid = (n + 255) / 256;
  if (grid > 65536) grid = 65536;
  hipLaunchKernelGGL(blur_finish_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, fin, out, S, n);
  CHADA_CHECK_LAUNCH();
  return 0;
}
