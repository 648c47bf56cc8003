// Host side of the multi-crop augmentation contract: the per-sample random parameters of one crop, drawn in C.
//
// The reference's pipeline (build_transform_pipeline, src/data/pretrain_dataloader.py:272-328) is an albumentations 1.3.1 Compose: every
// transform consumes CPython's global `random` stream -- `random.random() < p` first, then its own parameters.  The build's producer thread
// (chadavit_amd/data/device_pipeline.py::_draw, the Python statement of that order, pinned against oracle/augment_ref.py) spent ~12 us of
// interpreter per sample and crop on it: 130 ms per 1 024-image 10-crop batch under the interpreter lock the training thread also needs for
// its ~700 launches per step.  This file is the same draw on the same stream without the interpreter: it continues a `random.Random`
// generator from its exported state (`getstate()[1]`: the 624 MT19937 words + the index) and hands the state back, so a pipeline can
// switch between the two implementations sample by sample and never leave the stream (tests/test_augment_cpu.py holds them equal).
//
// CPython arithmetic restated here (Lib/random.py, Modules/_randommodule.c of 3.10):
//   random()        = ((genrand() >> 5) * 2^26 + (genrand() >> 6)) / 2^53
//   _randbelow(n)   = k = n.bit_length(); r = genrand() >> (32 - k) until r < n                (n < 2^32)
//   randrange(a, b) = a + _randbelow(b - a);   uniform(a, b) = a + (b - a) * random();   round() = round-half-even
// No GPU work here: plain host code in the C-ABI library so that the data path needs no second shared object.
#include <cmath>
#include <cstdint>

#pragma clang fp contract(off)   // a * b + c stays two roundings, as the interpreter computes it

namespace {
struct MT {
  uint32_t* mt;   // 624 state words followed by the index (CPython's getstate() layout)
  uint32_t& idx() { return mt[624]; }
  uint32_t next() {
    constexpr int N = 624, M = 397;
    if (idx() >= (uint32_t)N) {
      auto tw = [](uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
      int k = 0;
      for (; k < N - M; ++k) mt[k] = mt[k + M] ^ tw(mt[k], mt[k + 1]);
      for (; k < N - 1; ++k) mt[k] = mt[k + (M - N)] ^ tw(mt[k], mt[k + 1]);
      mt[N - 1] = mt[M - 1] ^ tw(mt[N - 1], mt[0]);
      idx() = 0;
    }
    uint32_t y = mt[idx()++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  double random() {
    const uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
  uint32_t below(uint32_t n) {   // n >= 1
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t r = next() >> (32 - k);
    while (r >= n) r = next() >> (32 - k);
    return r;
  }
};

inline long long py_round(double x) { return (long long)std::nearbyint(x); }   // default rounding mode: half to even, as round()
}  // namespace

extern "C" int chadavit_draw_crop_params(unsigned int* mt_state, int n, const long long* hw, int rrc_enabled, double scale_lo, double scale_hi,
                                         double ratio_lo, double ratio_hi, double gray_p, double blur_p, int blur_lo, int blur_hi,
                                         double sigma_lo, double sigma_hi, double sol_p, double sol_threshold, double flip_p, int norm_on,
                                         double norm_p, long long* boxes, int* gray, int* blur_k, double* blur_sigma, int* sol_on,
                                         double* sol_value, int* flip, int* normed) {
  if (!mt_state || n < 0 || !hw || !boxes || !gray || !blur_k || !blur_sigma || !sol_on || !sol_value || !flip || !normed) return 1;
  if (mt_state[624] > 624u || blur_hi < blur_lo) return 1;
  MT g{mt_state};
  const double s0 = scale_lo, ds = scale_hi - scale_lo;
  const double l0 = std::log(ratio_lo), dl = std::log(ratio_hi) - l0;
  const double rmin = ratio_lo < ratio_hi ? ratio_lo : ratio_hi, rmax = ratio_lo < ratio_hi ? ratio_hi : ratio_lo;
  for (int s = 0; s < n; ++s) {
    const long long H = hw[2 * s], W = hw[2 * s + 1];
    if (H <= 0 || W <= 0 || H >= (1ll << 31) || W >= (1ll << 31)) return 2;
    g.random();   // RandomResizedCrop / Resize: p = 1.0, the draw still happens (BasicTransform.__call__)
    long long y0 = 0, x0 = 0, h = H, w = W;
    if (rrc_enabled) {
      // albumentations 1.3.1 RandomResizedCrop.get_params_dependent_on_targets: 10 attempts, central fallback
      const double area = (double)(H * W);
      bool found = false;
      long long i = 0, j = 0;
      for (int t = 0; t < 10; ++t) {
        const double target_area = (s0 + ds * g.random()) * area;
        const double aspect = std::exp(l0 + dl * g.random());
        w = py_round(std::sqrt(target_area * aspect));
        h = py_round(std::sqrt(target_area / aspect));
        if (0 < w && w <= W && 0 < h && h <= H) {
          i = (long long)g.below((uint32_t)(H - h + 1));
          j = (long long)g.below((uint32_t)(W - w + 1));
          found = true;
          break;
        }
      }
      if (!found) {
        const double in_ratio = (double)W / (double)H;
        if (in_ratio < rmin) { w = W; h = py_round((double)W / rmin); }
        else if (in_ratio > rmax) { h = H; w = py_round((double)H * rmax); }
        else { w = W; h = H; }
        i = (H - h) / 2; j = (W - w) / 2;   // (floor division of Python: the operands are non-negative whenever the fallback is sane)
        if (H - h < 0) i = -((h - H + 1) / 2);
        if (W - w < 0) j = -((w - W + 1) / 2);
      }
      // h_start = i / (H - h + 1e-10), then int((H - h) * h_start): the transform's float round trip, kept
      y0 = (long long)((double)(H - h) * ((double)i * 1.0 / ((double)(H - h) + 1e-10)));
      x0 = (long long)((double)(W - w) * ((double)j * 1.0 / ((double)(W - w) + 1e-10)));
    }
    boxes[4 * s] = y0; boxes[4 * s + 1] = x0; boxes[4 * s + 2] = h; boxes[4 * s + 3] = w;
    // (CustomColorJitter sits here in the reference's list and draws nothing from this stream: custom_transforms.py:309-311)
    gray[s] = (gray_p != 0.0 && g.random() < gray_p) ? 1 : 0;
    blur_k[s] = -1; blur_sigma[s] = 0.0;   // -1: the transform did not fire
    if (blur_p != 0.0 && g.random() < blur_p) {
      long long k = blur_lo + (long long)g.below((uint32_t)(blur_hi + 1 - blur_lo));
      if (k != 0 && (((k % 2) + 2) % 2) != 1) k = (k + 1) % (blur_hi + 1);   // A.GaussianBlur: an even size moves to the next odd one
      blur_k[s] = (int)k;
      blur_sigma[s] = sigma_lo + (sigma_hi - sigma_lo) * g.random();
    }
    sol_on[s] = 0; sol_value[s] = 0.0;
    if (sol_p != 0.0 && g.random() < sol_p) {
      sol_on[s] = 1;
      sol_value[s] = sol_threshold + (sol_threshold - sol_threshold) * g.random();
    }
    flip[s] = (flip_p != 0.0 && g.random() < flip_p) ? 1 : 0;
    g.random();   // ToTensorV2(always_apply=True): `random.random() < p or always_apply` still draws
    normed[s] = (norm_on && g.random() < norm_p) ? 1 : 0;
  }
  return 0;
}
