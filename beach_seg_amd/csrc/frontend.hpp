// Callers either side of the network, on device:
//   * tif_image: 4-band / 8-band surface-reflectance raster -> uint8 RGB (src/util/geo_util.py:449-470,
//     src/util/multichannel_img.py:7-29): two reductions (min over the valid pixels, max over all pixels) and a map,
//     every float operation its own IEEE op in numpy's order so that the 4-band output is bit-exact;
//   * the train-time augmentation chain of src/data.py:195-224 with EXPLICIT random parameters (flips, ColorJiggle,
//     RandomSharpness, RandomErasing, Gaussian noise, Normalize) and its backward to the prompt pixels;
//   * the confusion matrix behind MulticlassF1Score(ignore_index=0) (src/model.py:85-93, 256, 295).
#pragma once
#include "common.hpp"

// ---------------------------------------------------------------------------------------------- tif_image
struct TifArgs {
  const void* bands;      // (C, H, W), f32 (dtype 0) or u16 (dtype 1: promoted to f32, exact)
  const uint8_t* nodata;  // (H, W) or nullptr
  int* keys;              // scratch: [0..2] per-channel min keys over valid pixels, [4..6] per-channel max keys
  uint8_t* out;           // (H, W, 3)
  int dtype, C;
  long hw;
};
// monotone float <-> int key (atomicMin / atomicMax on floats of either sign)
DEVI int f2key(float f) { const int i = __builtin_bit_cast(int, f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
DEVI float key2f(int k) { return __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7FFFFFFF); }
DEVI float tif_band(const TifArgs& a, int c, long p) {
  return a.dtype == 0 ? ((const float*)a.bands)[c * a.hw + p] : (float)((const unsigned short*)a.bands)[c * a.hw + p];
}
// the three display channels before any stretch
DEVI void tif_channels(const TifArgs& a, long p, float (&v)[3]) {
  if (a.C == 4) {  // R = band 4, G = band 3, B = mean(band 1, band 2): (b0 + b1) / 2 in f32
    v[0] = tif_band(a, 3, p);
    v[1] = tif_band(a, 2, p);
    v[2] = __fmul_rn(__fadd_rn(tif_band(a, 0, p), tif_band(a, 1, p)), 0.5f);
  } else {  // 8 bands: log10(1 + mean of the band group); numpy's mean = left-to-right f32 sum / count
    const float r = __fdiv_rn(__fadd_rn(__fadd_rn(tif_band(a, 5, p), tif_band(a, 6, p)), tif_band(a, 7, p)), 3.0f);
    const float g = __fdiv_rn(__fadd_rn(__fadd_rn(tif_band(a, 2, p), tif_band(a, 3, p)), tif_band(a, 4, p)), 3.0f);
    const float b = __fmul_rn(__fadd_rn(tif_band(a, 0, p), tif_band(a, 1, p)), 0.5f);
    v[0] = (float)log10((double)__fadd_rn(1.0f, r));  // correctly rounded f32 log10 (via double)
    v[1] = (float)log10((double)__fadd_rn(1.0f, g));
    v[2] = (float)log10((double)__fadd_rn(1.0f, b));
  }
}
__global__ void tif_init_kernel(int* keys) {
  if (threadIdx.x < 3) keys[threadIdx.x] = 0x7FFFFFFF;
  else if (threadIdx.x >= 4 && threadIdx.x < 7) keys[threadIdx.x] = (int)0x80000000;
}
DEVI int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
DEVI int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
__global__ void tif_min_kernel(TifArgs a) {
  int k[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF};
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < a.hw; p += (long)gridDim.x * blockDim.x) {
    if (a.nodata && a.nodata[p]) continue;
    float v[3];
    tif_channels(a, p, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) k[c] = min(k[c], f2key(v[c]));
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int m = wave_min_i(k[c]);
    if ((threadIdx.x & 63) == 0) atomicMin(a.keys + c, m);
  }
}
// stretch of channel c given the minima: 4 bands: clip to [lo, lo + 3000] - lo with ONE lo = min over the three channels
// (the second `img -= img[valid].min()` of the reference subtracts an exact +0); 8 bands: v - lo_c
DEVI float tif_stretch(const TifArgs& a, float v, int c, const float (&lo)[3]) {
  if (a.C == 4) {
    const float l = fminf(fminf(lo[0], lo[1]), lo[2]), hi = __fadd_rn(3000.0f, l);
    return __fsub_rn(fminf(fmaxf(v, l), hi), l);
  }
  return __fsub_rn(v, lo[c]);
}
__global__ void tif_max_kernel(TifArgs a) {
  const float lo[3] = {key2f(a.keys[0]), key2f(a.keys[1]), key2f(a.keys[2])};
  int k[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < a.hw; p += (long)gridDim.x * blockDim.x) {
    float v[3];
    tif_channels(a, p, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) k[c] = max(k[c], f2key(tif_stretch(a, v[c], c, lo)));  // max over ALL pixels (nodata too)
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int m = wave_max_i(k[c]);
    if ((threadIdx.x & 63) == 0) atomicMax(a.keys + 4 + c, m);
  }
}
__global__ void tif_map_kernel(TifArgs a) {
  const float lo[3] = {key2f(a.keys[0]), key2f(a.keys[1]), key2f(a.keys[2])};
  const float hi[3] = {key2f(a.keys[4]), key2f(a.keys[5]), key2f(a.keys[6])};
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < a.hw; p += (long)gridDim.x * blockDim.x) {
    float v[3];
    tif_channels(a, p, v);
    const bool nd = a.nodata && a.nodata[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = nd ? 0.f : __fdiv_rn(tif_stretch(a, v[c], c, lo), hi[c]);
      a.out[p * 3 + c] = (uint8_t)(int)__fmul_rn(x, 255.0f);  // np.array(img * 255, dtype=np.uint8): truncation
    }
  }
}

// -------------------------------------------------------------------------------- train-time augmentation
// Per sample b: params[b] = {flags, ex0, ey0, ew, eh}, flags bit 0 vertical flip, bit 1 horizontal flip, bit 2 add noise,
// bit 3 RandomSharpness applied, bit 4 ColorJiggle applied; color[b] = {brightness, contrast, saturation, hue, sharpness
// factor, order code} (only read when bit 3 / bit 4 is set; order code = i0 | i1 << 2 | i2 << 4 | i3 << 6, i0 applied first,
// 0 brightness, 1 contrast, 2 saturation, 3 hue).
// Order of src/data.py:195-224: flips -> ColorJiggle -> RandomSharpness -> erase box to 0 -> + noise -> Normalize.
// mask_out (may be nullptr) follows the flips (the intensity operations leave masks alone); with flags bit 5 the erased box
// is also set to class 0 in the mask (what recent kornia does for masks under RandomErasing; parity unpinned, off by default).
// ColorJiggle / RandomSharpness follow kornia's published definitions (kornia.enhance.adjust_* / sharpness; kornia is not
// installed here: parity unpinned):
//   brightness: clamp(x + (f - 1), 0, 1);  contrast: clamp(x f, 0, 1);  saturation: HSV, s <- clamp(s f, 0, 1);
//   hue: HSV, h <- fmod(h + 2 pi f, 2 pi);  sharpness: d = clamp(conv3x3(x, [[1,1,1],[1,5,1],[1,1,1]] / 13), 0, 1) on the
//   interior, d = x on the one-pixel border; out = d + (x - d) f, clamped to [0,1] unless 0 <= f <= 1.
// The backward runs the per-pixel colour chain on forward-mode dual numbers (three tangents = the 3x3 Jacobian of one
// pixel) instead of a hand-derived adjoint of the HSV round trips.
struct Dual3 { float v, a, b, c; };
DEVI Dual3 operator+(Dual3 x, Dual3 y) { return {x.v + y.v, x.a + y.a, x.b + y.b, x.c + y.c}; }
DEVI Dual3 operator-(Dual3 x, Dual3 y) { return {x.v - y.v, x.a - y.a, x.b - y.b, x.c - y.c}; }
DEVI Dual3 operator*(Dual3 x, Dual3 y) { return {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.c * y.v + x.v * y.c}; }
DEVI Dual3 operator/(Dual3 x, Dual3 y) {
  const float q = x.v / y.v, iy = 1.f / y.v;
  return {q, (x.a - q * y.a) * iy, (x.b - q * y.b) * iy, (x.c - q * y.c) * iy};
}
DEVI Dual3 operator+(Dual3 x, float k) { return {x.v + k, x.a, x.b, x.c}; }
DEVI Dual3 operator-(Dual3 x, float k) { return {x.v - k, x.a, x.b, x.c}; }
DEVI Dual3 operator-(float k, Dual3 x) { return {k - x.v, -x.a, -x.b, -x.c}; }
DEVI Dual3 operator*(Dual3 x, float k) { return {x.v * k, x.a * k, x.b * k, x.c * k}; }
DEVI Dual3 operator*(float k, Dual3 x) { return x * k; }
DEVI Dual3 operator/(Dual3 x, float k) { return {x.v / k, x.a / k, x.b / k, x.c / k}; }
DEVI float nval(float x) { return x; }
DEVI float nval(Dual3 x) { return x.v; }
DEVI void nconst(float& x, float k) { x = k; }
DEVI void nconst(Dual3& x, float k) { x = Dual3{k, 0.f, 0.f, 0.f}; }
DEVI float nclamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
DEVI Dual3 nclamp01(Dual3 x) {  // torch.clamp: the gradient passes on the closed interval
  const bool in = x.v >= 0.f && x.v <= 1.f;
  return {fminf(fmaxf(x.v, 0.f), 1.f), in ? x.a : 0.f, in ? x.b : 0.f, in ? x.c : 0.f};
}
DEVI float pymod(float x, float m) { return x - m * floorf(x / m); }  // torch.remainder
DEVI float nshift(float x, float nv) { (void)x; return nv; }            // same tangents, new value (mod / fmod: slope 1)
DEVI Dual3 nshift(Dual3 x, float nv) { x.v = nv; return x; }

template <typename N> DEVI void rgb_to_hsv(N r, N g, N b, N& h, N& s, N& v) {
  const float rv = nval(r), gv = nval(g), bv = nval(b);
  const int imax = (rv >= gv && rv >= bv) ? 0 : (gv >= bv ? 1 : 2);
  const int imin = (rv <= gv && rv <= bv) ? 0 : (gv <= bv ? 1 : 2);
  const N mx = imax == 0 ? r : imax == 1 ? g : b;
  const N mn = imin == 0 ? r : imin == 1 ? g : b;
  N dc = mx - mn;
  v = mx;
  s = dc / (mx + 1e-8f);
  if (nval(dc) == 0.f) nconst(dc, 1.f);
  const N rc = mx - r, gc = mx - g, bc = mx - b;
  N hh = imax == 0 ? (bc - gc) : imax == 1 ? (rc - bc) + 2.f * dc : (gc - rc) + 4.f * dc;
  hh = hh / dc;
  hh = hh / 6.f;
  hh = nshift(hh, pymod(nval(hh), 1.f));
  h = 6.283185307179586f * hh;
}
template <typename N> DEVI void hsv_to_rgb(N h, N s, N v, N& r, N& g, N& b) {
  const N h6 = (h / 6.283185307179586f) * 6.f;
  const float h6v = nval(h6);
  const int hi = (int)pymod(floorf(h6v), 6.f);
  const N f = nshift(h6, pymod(h6v, 6.f)) - (float)hi;
  const N p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  switch (hi) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}
template <typename N> DEVI void color_jiggle(N& r, N& g, N& b, const float* cp) {
  const int order = (int)cp[5];
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int op = (order >> (2 * k)) & 3;
    if (op == 0) {
      const float d = cp[0] - 1.f;
      r = nclamp01(r + d); g = nclamp01(g + d); b = nclamp01(b + d);
    } else if (op == 1) {
      r = nclamp01(r * cp[1]); g = nclamp01(g * cp[1]); b = nclamp01(b * cp[1]);
    } else {
      N h, s, v;
      rgb_to_hsv(r, g, b, h, s, v);
      if (op == 2) s = nclamp01(s * cp[2]);
      else { h = h + cp[3] * 2.f * 3.141592653589793f; h = nshift(h, fmodf(nval(h), 6.283185307179586f)); }
      hsv_to_rgb(h, s, v, r, g, b);
    }
  }
}

struct AugArgs {
  const float* img;      // (B,3,H,W) in [0,1]
  const uint8_t* mask;   // (B,H,W) or nullptr
  const int* params;     // (B,5)
  const float* color;    // (B,6) or nullptr
  const float* noise;    // (B,3,H,W) or nullptr
  float* out;            // (B,3,H,W)
  uint8_t* mask_out;     // (B,H,W) or nullptr
  float* tmp;            // (B,3,H,W) scratch: the flipped + colour-jiggled image (needed when color != nullptr)
  int B, H, W;
  float mean[3], istd[3];
};
DEVI bool aug_erased(const int* pr, int x, int y) {
  return pr[3] > 0 && pr[4] > 0 && x >= pr[1] && x < pr[1] + pr[3] && y >= pr[2] && y < pr[2] + pr[4];
}
// stage 1 (only with colour parameters): tmp = ColorJiggle(flip(img))
__global__ void train_aug_color_kernel(AugArgs a) {
  const long hw = (long)a.H * a.W, n = (long)a.B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / hw;
    const long r = i % hw;
    const int y = r / a.W, x = r % a.W;
    const int fl = a.params[b * 5];
    const int ys = (fl & 1) ? a.H - 1 - y : y, xs = (fl & 2) ? a.W - 1 - x : x;
    const long o = (long)b * 3 * hw, sp = (long)ys * a.W + xs;
    float cr = a.img[o + sp], cg = a.img[o + hw + sp], cb = a.img[o + 2 * hw + sp];
    if (fl & 16) color_jiggle(cr, cg, cb, a.color + b * 6);
    a.tmp[o + r] = cr; a.tmp[o + hw + r] = cg; a.tmp[o + 2 * hw + r] = cb;
  }
}
// 3x3 blur of kornia.enhance.sharpness at an interior pixel of one channel plane
DEVI float sharp_blur(const float* pl, int W, int y, int x) {
  const float* c = pl + (long)y * W + x;
  const float k1 = 1.f / 13.f, k5 = 5.f / 13.f;
  return c[-W - 1] * k1 + c[-W] * k1 + c[-W + 1] * k1 + c[-1] * k1 + c[0] * k5 + c[1] * k1 + c[W - 1] * k1 + c[W] * k1 + c[W + 1] * k1;
}
__global__ void train_aug_fwd_kernel(AugArgs a) {
  const long hw = (long)a.H * a.W, n = (long)a.B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / hw;
    const long r = i % hw;
    const int y = r / a.W, x = r % a.W;
    const int* pr = a.params + b * 5;
    const int fl = pr[0];
    const int ys = (fl & 1) ? a.H - 1 - y : y, xs = (fl & 2) ? a.W - 1 - x : x;
    const bool erased = aug_erased(pr, x, y);
    if (a.mask_out) a.mask_out[i] = ((fl & 32) && erased) ? (uint8_t)0 : a.mask[(long)b * hw + (long)ys * a.W + xs];
    const bool interior = y > 0 && y < a.H - 1 && x > 0 && x < a.W - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long o = ((long)b * 3 + c) * hw;
      float v;
      if (a.color) {  // stage 1 already flipped
        v = a.tmp[o + r];
        if ((fl & 8) && interior) {
          const float f = a.color[b * 6 + 4];
          const float d = nclamp01(sharp_blur(a.tmp + o, a.W, y, x));
          v = d + (v - d) * f;
          if (!(f >= 0.f && f <= 1.f)) v = nclamp01(v);
        }
      } else {
        v = a.img[o + (long)ys * a.W + xs];
      }
      if (erased) v = 0.f;
      if ((fl & 4) && a.noise) v += a.noise[o + r];
      a.out[o + r] = (v - a.mean[c]) * a.istd[c];
    }
  }
}
// grad wrt img without colour parameters: undo Normalize, zero inside the erased box, undo the flips (noise is additive)
__global__ void train_aug_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ params, float* __restrict__ gin,
                                     int B, int H, int W, float is0, float is1, float is2) {
  const long hw = (long)H * W, n = (long)B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / hw;
    const long r = i % hw;
    const int y = r / W, x = r % W;
    const int* pr = params + b * 5;
    const int fl = pr[0];
    const int ys = (fl & 1) ? H - 1 - y : y, xs = (fl & 2) ? W - 1 - x : x;
    const bool erased = aug_erased(pr, x, y);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long o = ((long)b * 3 + c) * hw;
      gin[o + (long)ys * W + xs] = erased ? 0.f : gout[o + r] * (c == 0 ? is0 : c == 1 ? is1 : is2);
    }
  }
}
// With colour parameters the backward is three passes: (1) train_aug_color_kernel recomputes tmp, (2) this kernel takes
// grad_out back through Normalize / erase / sharpness to g2 = d loss / d tmp, (3) train_aug_color_bwd_kernel applies the
// transposed per-pixel Jacobian of ColorJiggle and undoes the flips.
struct AugBwdArgs {
  const float* gout;    // (B,3,H,W)
  const float* img;     // (B,3,H,W): the forward's input
  const int* params;
  const float* color;
  const float* tmp;     // recomputed by train_aug_color_kernel
  float* g2;            // (B,3,H,W) scratch
  float* gin;           // (B,3,H,W)
  int B, H, W;
  float istd[3];
};
// gradient arriving at the sharpness output of pixel (y, x), channel plane offset o: through Normalize and the erase box
DEVI float aug_g1(const AugBwdArgs& a, const int* pr, long o, int c, int y, int x) {
  return aug_erased(pr, x, y) ? 0.f : a.gout[o + (long)y * a.W + x] * a.istd[c];
}
__global__ void train_aug_sharp_bwd_kernel(AugBwdArgs a) {
  const long hw = (long)a.H * a.W, n = (long)a.B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / hw;
    const long r = i % hw;
    const int y = r / a.W, x = r % a.W;
    const int* pr = a.params + b * 5;
    const int fl = pr[0];
    const float f = a.color[b * 6 + 4];
    const bool clamp_res = !(f >= 0.f && f <= 1.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long o = ((long)b * 3 + c) * hw;
      const float* pl = a.tmp + o;
      float g;
      if (!(fl & 8)) {
        g = aug_g1(a, pr, o, c, y, x);
      } else {
        g = 0.f;
        // every interior pixel q of the 3x3 neighbourhood (q = p included): its blur reads tmp[p] with weight w(p - q)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int qy = y + dy, qx = x + dx;
            if (qy < 1 || qy >= a.H - 1 || qx < 1 || qx >= a.W - 1) continue;
            const float blur = sharp_blur(pl, a.W, qy, qx);
            const float d = nclamp01(blur), tq = pl[(long)qy * a.W + qx];
            float gres = aug_g1(a, pr, o, c, qy, qx);
            if (clamp_res) { const float res = d + (tq - d) * f; if (!(res >= 0.f && res <= 1.f)) gres = 0.f; }
            if (dy == 0 && dx == 0) g += f * gres;
            if (blur >= 0.f && blur <= 1.f) g += (1.f - f) * gres * ((dy == 0 && dx == 0) ? 5.f / 13.f : 1.f / 13.f);
          }
        const bool interior = y > 0 && y < a.H - 1 && x > 0 && x < a.W - 1;
        if (!interior) g += aug_g1(a, pr, o, c, y, x);  // the border keeps the input pixel
      }
      a.g2[o + r] = g;
    }
  }
}
__global__ void train_aug_color_bwd_kernel(AugBwdArgs a) {
  const long hw = (long)a.H * a.W, n = (long)a.B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / hw;
    const long r = i % hw;
    const int y = r / a.W, x = r % a.W;
    const int fl = a.params[b * 5];
    const int ys = (fl & 1) ? a.H - 1 - y : y, xs = (fl & 2) ? a.W - 1 - x : x;
    const long o = (long)b * 3 * hw, sp = (long)ys * a.W + xs;
    const float g0 = a.g2[o + r], g1 = a.g2[o + hw + r], g2 = a.g2[o + 2 * hw + r];
    if (fl & 16) {
      Dual3 cr{a.img[o + sp], 1.f, 0.f, 0.f}, cg{a.img[o + hw + sp], 0.f, 1.f, 0.f}, cb{a.img[o + 2 * hw + sp], 0.f, 0.f, 1.f};
      color_jiggle(cr, cg, cb, a.color + b * 6);
      a.gin[o + sp] = cr.a * g0 + cg.a * g1 + cb.a * g2;
      a.gin[o + hw + sp] = cr.b * g0 + cg.b * g1 + cb.b * g2;
      a.gin[o + 2 * hw + sp] = cr.c * g0 + cg.c * g1 + cb.c * g2;
    } else {
      a.gin[o + sp] = g0; a.gin[o + hw + sp] = g1; a.gin[o + 2 * hw + sp] = g2;
    }
  }
}

// ------------------------------------------------------------------------------------- confusion matrix
// confmat[t][p] += 1 over pixels with target t != ignore (torchmetrics' _multiclass_stat_scores_update with
// ignore_index: ignored targets are dropped, then bincount(target * K + pred)).  K <= 16.
__global__ __launch_bounds__(256) void confusion_kernel(const long long* __restrict__ pred_i64,
                                                         const uint8_t* __restrict__ pred_u8,
                                                         const uint8_t* __restrict__ target, long n, int K, int ignore,
                                                         unsigned long long* __restrict__ confmat) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = target[i];
    if (t == ignore || t >= K) continue;
    const long long p = pred_i64 ? pred_i64[i] : (long long)pred_u8[i];
    if (p >= 0 && p < K) atomicAdd(&h[t * K + (int)p], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < K * K && h[threadIdx.x]) atomicAdd(&confmat[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
