// Callers either side of the network, on device:
//   * tif_image: 4-band / 8-band surface-reflectance raster -> uint8 RGB (src/util/geo_util.py:449-470,
//     src/util/multichannel_img.py:7-29): two reductions (min over the valid pixels, max over all pixels) and a map,
//     every float operation its own IEEE op in numpy's order so that the 4-band output is bit-exact;
//   * the train-time augmentation chain of src/data.py:195-224 with EXPLICIT random parameters (flips, RandomErasing,
//     Gaussian noise, Normalize) and its backward to the prompt pixels;
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
// Per sample b: params[b] = {flags (bit 0 vertical flip, bit 1 horizontal flip, bit 2 add noise), ex0, ey0, ew, eh}.
// Order of src/data.py:195-224: flips -> (ColorJiggle, RandomSharpness: not built) -> erase box to 0 -> + noise ->
// Normalize.  mask_out (may be nullptr) follows the flips only (the intensity operations leave masks alone).
struct AugArgs {
  const float* img;      // (B,3,H,W) in [0,1]
  const uint8_t* mask;   // (B,H,W) or nullptr
  const int* params;     // (B,5)
  const float* noise;    // (B,3,H,W) or nullptr
  float* out;            // (B,3,H,W)
  uint8_t* mask_out;     // (B,H,W) or nullptr
  int B, H, W;
  float mean[3], istd[3];
};
__global__ void train_aug_fwd_kernel(AugArgs a) {
  const long hw = (long)a.H * a.W, n = (long)a.B * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / hw;
    const long r = i % hw;
    const int y = r / a.W, x = r % a.W;
    const int* pr = a.params + b * 5;
    const int fl = pr[0];
    const int ys = (fl & 1) ? a.H - 1 - y : y, xs = (fl & 2) ? a.W - 1 - x : x;
    const bool erased = pr[3] > 0 && pr[4] > 0 && x >= pr[1] && x < pr[1] + pr[3] && y >= pr[2] && y < pr[2] + pr[4];
    if (a.mask_out) a.mask_out[i] = a.mask[(long)b * hw + (long)ys * a.W + xs];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long o = ((long)b * 3 + c) * hw;
      float v = erased ? 0.f : a.img[o + (long)ys * a.W + xs];
      if ((fl & 4) && a.noise) v += a.noise[o + r];
      a.out[o + r] = (v - a.mean[c]) * a.istd[c];
    }
  }
}
// grad wrt img: undo Normalize, zero inside the erased box, undo the flips (noise is additive)
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
    const bool erased = pr[3] > 0 && pr[4] > 0 && x >= pr[1] && x < pr[1] + pr[3] && y >= pr[2] && y < pr[2] + pr[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long o = ((long)b * 3 + c) * hw;
      gin[o + (long)ys * W + xs] = erased ? 0.f : gout[o + r] * (c == 0 ? is0 : c == 1 ? is1 : is2);
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
