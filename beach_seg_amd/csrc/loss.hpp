// Reference wrapper arithmetic on device (all fp32 / integer, HBM-bound):
//   masked smooth-L1 loss + its gradient   (/root/reference/src/model.py:45-64, incl. the B x B broadcast of :61)
//   palette arg-min decode                  (src/model.py:155-175)  -- bit-exact vs torch: no FMA contraction
//   AdamW on the prompt pixels              (src/model.py:398, torch.optim.AdamW defaults)
//   predict-loop glue: nearest resize + one-hot vote + arg-max (src/predict.py:259-260, 120-159, 100)
#pragma once
#include "common.hpp"

// counts[p] = sum_i yes_i[p] over the batch (fp32), total[0] += number of kept pixels (exact integer)
__global__ void loss_prep_kernel(const uint8_t* __restrict__ yes, float* __restrict__ counts,
                                 unsigned long long* __restrict__ total, int B, long hw) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  unsigned c = 0;
  if (p < hw)
    for (int i = 0; i < B; ++i) c += yes[(long)i * hw + p] ? 1u : 0u;
  if (p < hw) counts[p] = (float)c;
  // wave-aggregate then one atomic per wave
  unsigned long long w = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(total, w);
}

// pred fp32 (B,3,2H,W); labels fp32 (B,3,H,W); yes u8 (B,H,W).  variant 0 = reference (weight = counts[p]),
// 1 = per-sample (weight = yes_j[p]).  Writes dpred (B,3,2H,W) (top half zero) and one partial sum per block.
__global__ __launch_bounds__(256) void loss_fwd_bwd_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ labels,
                                                            const uint8_t* __restrict__ yes,
                                                            const float* __restrict__ counts,
                                                            const unsigned long long* __restrict__ total,
                                                            float* __restrict__ dpred, float* __restrict__ partial,
                                                            int B, int H, int W, float beta, int variant,
                                                            const uint8_t* __restrict__ ids = nullptr,
                                                            const float* __restrict__ lut = nullptr, int K = 0) {
  const long hw = (long)H * W, n = (long)B * 3 * 2 * hw;
  const float denom = 3.f * (float)(*total);
  const float inv = 1.f / denom;
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long bc = i / (2 * hw), r = i % (2 * hw);
    float g = 0.f;
    if (r >= hw) {
      const long p = r - hw;
      const int b = bc / 3;
      const float wgt = variant == 0 ? counts[p] : (yes[(long)b * hw + p] ? 1.f : 0.f);
      if (wgt != 0.f) {
        // ids: the label image is never materialised -- its pixel IS lut[b][class id][c], the normalised palette entry
        // (same bits as Normalize(torch_apply_mask_rgb(...)), src/model.py:238-239)
        const float lab = ids ? lut[((long)b * K + min((int)ids[(long)b * hw + p], K - 1)) * 3 + (bc - 3 * b)] : labels[bc * hw + p];
        const float d = pred[i] - lab;
        const float ad = fabsf(d);
        float l, dl;
        if (ad < beta) { l = 0.5f * d * d / beta; dl = d / beta; }
        else { l = ad - 0.5f * beta; dl = d > 0.f ? 1.f : -1.f; }
        acc += wgt * l;
        g = wgt * dl * inv;
      }
    }
    if (dpred) dpred[i] = g;
  }
  acc = wave_sum(acc);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void loss_finalize_kernel(const float* __restrict__ partial, int n,
                                     const unsigned long long* __restrict__ total, float* __restrict__ loss) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += partial[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[0] = acc / (3.f * (float)(*total));
}

// torch_apply_mask_rgb (+ Normalize) on device (src/util/ml_util.py:114-132, src/data.py:345; callers src/model.py:211-212,
// 238-239): class ids u8 (B,H,W) + palette u8 (B,K,3) -> f32 (B,3,H,W) = (palette[b][id][c] / 255 - mean[c]) / std[c], every
// step its own correctly rounded IEEE operation like torch's (mean 0 / std 1 gives the un-normalised [0,1] image exactly).
// One block column per sample: the K x 3 table is formed once in LDS, then 4 pixels per thread (one u32 of ids, three
// 16-byte stores).  HBM-bound: 13 bytes per pixel.
__global__ __launch_bounds__(256) void mask_rgb_norm_kernel(const uint8_t* __restrict__ ids, const uint8_t* __restrict__ pal,
                                                             float* __restrict__ out, long hw, int K, float m0, float m1,
                                                             float m2, float s0, float s1, float s2) {
  __shared__ float lut[3 * 256];
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * K; t += 256) {
    const int k = t / 3, c = t - 3 * k;
    const float mean = c == 0 ? m0 : c == 1 ? m1 : m2, sd = c == 0 ? s0 : c == 1 ? s1 : s2;
    lut[c * 256 + k] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)pal[((long)b * K + k) * 3 + c], 255.0f), mean), sd);
  }
  __syncthreads();
  const uint8_t* src = ids + (long)b * hw;
  float* dst = out + (long)b * 3 * hw;
  const long nq = (hw & 3) ? 0 : hw >> 2;  // 4-pixel path needs 4-byte / 16-byte aligned planes
  for (long q = blockIdx.x * 256L + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    const unsigned v = *(const unsigned*)(src + 4 * q);
    const int i0 = min((int)(v & 255), K - 1), i1 = min((int)((v >> 8) & 255), K - 1), i2 = min((int)((v >> 16) & 255), K - 1),
              i3 = min((int)(v >> 24), K - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *(f32x4*)(dst + c * hw + 4 * q) = f32x4{lut[c * 256 + i0], lut[c * 256 + i1], lut[c * 256 + i2], lut[c * 256 + i3]};
  }
  if (blockIdx.x == 0)
    for (long p = 4 * nq + threadIdx.x; p < hw; p += 256) {
      const int i = min((int)src[p], K - 1);
      for (int c = 0; c < 3; ++c) dst[c * hw + p] = lut[c * 256 + i];
    }
}

// arg-min over K palette colours of sum_c (pred[c] - pal[k][c])^2 on the bottom half; first index wins ties.
// The products and sums are kept as separate IEEE operations in torch's order ((d0^2 + d1^2) + d2^2).
__global__ void decode_argmin_kernel(const float* __restrict__ pred, const float* __restrict__ pal_norm,
                                     long long* __restrict__ out_i64, uint8_t* __restrict__ out_u8, int B, int H,
                                     int W, int K) {
  const long hw = (long)H * W, n = (long)B * hw;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = i / hw;
  const long p = i % hw;
  const float* pr = pred + (long)b * 3 * 2 * hw + hw + p;
  const float v0 = pr[0], v1 = pr[2 * hw], v2 = pr[4 * hw];
  int best = 0;
  float bd = INFINITY;
  for (int k = 0; k < K; ++k) {
    const float* c = pal_norm + ((long)b * K + k) * 3;
    const float d0 = __fsub_rn(v0, c[0]), d1 = __fsub_rn(v1, c[1]), d2 = __fsub_rn(v2, c[2]);
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    if (d < bd) { bd = d; best = k; }
  }
  if (out_i64) out_i64[i] = best;
  if (out_u8) out_u8[i] = (uint8_t)best;
}

// HF decode (`SegGptImageProcessor.post_process_semantic_segmentation`, HF:image_processing_seggpt.py:300-332; used by
// src/predict_no_prompt.py:297-303): bottom half, x * std + mean, clip(x * 255, 0, 255), squared distance to the integer
// palette (K, 3), first arg-min.  Every product / sum is its own IEEE operation in torch's order.
__global__ void decode_hf_kernel(const float* __restrict__ pred, const float* __restrict__ palette,
                                 uint8_t* __restrict__ out_u8, int B, int H, int W, int K, float m0, float m1, float m2,
                                 float s0, float s1, float s2) {
  const long hw = (long)H * W, n = (long)B * hw;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = i / hw;
  const long p = i % hw;
  const float* pr = pred + (long)b * 3 * 2 * hw + hw + p;
  const float v0 = fminf(fmaxf(__fmul_rn(__fadd_rn(__fmul_rn(pr[0], s0), m0), 255.0f), 0.f), 255.f);
  const float v1 = fminf(fmaxf(__fmul_rn(__fadd_rn(__fmul_rn(pr[2 * hw], s1), m1), 255.0f), 0.f), 255.f);
  const float v2 = fminf(fmaxf(__fmul_rn(__fadd_rn(__fmul_rn(pr[4 * hw], s2), m2), 255.0f), 0.f), 255.f);
  int best = 0;
  float bd = INFINITY;
  for (int k = 0; k < K; ++k) {
    const float* c = palette + k * 3;
    const float d0 = __fsub_rn(v0, c[0]), d1 = __fsub_rn(v1, c[1]), d2 = __fsub_rn(v2, c[2]);
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    if (d < bd) { bd = d; best = k; }
  }
  out_u8[i] = (uint8_t)best;
}

// AdamW step on the active prompts (torch.optim.AdamW: decoupled decay, bias correction, eps outside sqrt-hat).
// idx[a] = prompt row; step_sizes[a] = lr / (1 - beta1^t), bc2_sqrts[a] = sqrt(1 - beta2^t) with t that
// parameter's own step count (torch keeps one `step` per Parameter and skips Parameters without a gradient);
// both are computed on the host in double, as torch does.
__global__ void adamw_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                             float* __restrict__ v, const int* __restrict__ idx, const uint8_t* __restrict__ touched,
                             const float* __restrict__ step_sizes, const float* __restrict__ bc2_sqrts, long n,
                             float lr, float beta1, float beta2, float eps, float wd, float grad_scale) {
  const int row = idx[blockIdx.y];
  if (touched && !touched[row]) return;  // a Parameter without a gradient is skipped entirely, as torch does
  const float step_size = step_sizes[blockIdx.y], bc2s = bc2_sqrts[blockIdx.y];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long o = (long)row * n + i;
    const float g = grad[o] * grad_scale;
    float p = param[o] * (1.f - lr * wd);
    const float mm = m[o] + (1.f - beta1) * (g - m[o]);  // lerp, as torch
    const float vv = beta2 * v[o] + (1.f - beta2) * g * g;
    m[o] = mm;
    v[o] = vv;
    p -= step_size * mm / (sqrtf(vv) / bc2s + eps);
    param[o] = p;
  }
}

// grad_param[idx[b]] += grad_prompt_pixels[b] / std[c]   (backward of Normalize + stack; duplicates accumulate)
__global__ void prompt_grad_scatter_kernel(const float* __restrict__ gpix, float* __restrict__ gparam,
                                           const int* __restrict__ idx, int B, long chw, long hw, float is0,
                                           float is1, float is2) {
  const long n = (long)B * chw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / chw;
    const long r = i % chw;
    const int c = r / hw;
    atomicAdd(gparam + (long)idx[b] * chw + r, gpix[i] * (c == 0 ? is0 : c == 1 ? is1 : is2));
  }
}

// prompt_pixels[b] = (param[idx[b]] - mean[c]) / std[c]   (stack + Normalize, src/model.py:197, data.py:224)
__global__ void prompt_gather_kernel(const float* __restrict__ param, float* __restrict__ out,
                                     const int* __restrict__ idx, int B, long chw, long hw, float m0, float m1,
                                     float m2, float s0, float s1, float s2) {
  const long n = (long)B * chw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = i / chw;
    const long r = i % chw;
    const int c = r / hw;
    const float mean = c == 0 ? m0 : c == 1 ? m1 : m2, sd = c == 0 ? s0 : c == 1 ? s1 : s2;
    out[i] = (param[(long)idx[b] * chw + r] - mean) / sd;
  }
}

// Predict glue: nearest-resize the (hin x win) class mask to (crop x crop), clip the window to the mosaic, add
// one vote to counter[y][x][cls] (uint8, wraps like numpy).  One thread per destination pixel of one crop.
__global__ void vote_paste_kernel(const uint8_t* __restrict__ masks, uint8_t* __restrict__ counter,
                                  const int* __restrict__ crops, int ncrops, int hin, int win, int crop, int mh,
                                  int mw, int K) {
  const int ci = blockIdx.y;
  const int xmin = crops[4 * ci], ymin = crops[4 * ci + 1], xmax = crops[4 * ci + 2], ymax = crops[4 * ci + 3];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= crop * crop) return;
  const int sy = t / crop, sx = t % crop;
  if (sy >= ymax - ymin || sx >= xmax - xmin) return;
  const int y = ymin + sy, x = xmin + sx;
  if (y < 0 || y >= mh || x < 0 || x >= mw) return;
  // cv2.resize(INTER_NEAREST) (src/predict.py:259): ifx = 1 / (dsize / (double)ssize); sx = min(cvFloor(x * ifx), ssize - 1), in double
  const int iy = min((int)floor((double)sy * (1.0 / ((double)crop / (double)hin))), hin - 1);
  const int ix = min((int)floor((double)sx * (1.0 / ((double)crop / (double)win))), win - 1);
  const int cls = masks[((long)ci * hin + iy) * win + ix];
  uint8_t* c = counter + ((long)y * mw + x) * K + cls;
  *c = (uint8_t)(*c + 1);  // crops of one launch must not overlap (callers launch overlapping crops separately)
}

__global__ void vote_argmax_kernel(const uint8_t* __restrict__ counter, uint8_t* __restrict__ out, long n, int K) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int best = 0, bv = counter[i * K];
  for (int k = 1; k < K; ++k) {
    const int v = counter[i * K + k];
    if (v > bv) { bv = v; best = k; }
  }
  out[i] = (uint8_t)best;
}

// ---------------------------------------------------------------------------------------- tile front-end
// Cut a crop x crop window out of a u8 HWC mosaic (zeros outside, src/util/geo_util.py:316-341), resize it to S x S
// exactly as Pillow's BICUBIC does for 8-bit images (src/data.py:93-96): horizontal integer pass -> u8 -> vertical
// integer pass -> u8, 22-bit fixed-point coefficients from the host (libImaging/Resample.c), then /255 and the
// ImageNet Normalize into NCHW f32.  One thread per output pixel (x fastest: the three plane writes are coalesced);
// the <= kmax x kmax source taps of a 4x up-scale come from L1/L2 (37 KB per window).
__global__ __launch_bounds__(256) void tile_frontend_kernel(const uint8_t* __restrict__ mosaic, int mh, int mw,
                                                            const int* __restrict__ crops, int S,
                                                            const int* __restrict__ coef, const int* __restrict__ bounds,
                                                            int kmax, float m0, float m1, float m2, float s0, float s1,
                                                            float s2, float* __restrict__ out, uint8_t* __restrict__ out_u8) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S) return;
  const int y = p / S, x = p - y * S;
  const int cx = crops[4 * n], cy = crops[4 * n + 1];
  const int x0 = bounds[2 * x], nx = bounds[2 * x + 1], y0 = bounds[2 * y], ny = bounds[2 * y + 1];
  const int* kx = coef + (long)x * kmax;
  const int* ky = coef + (long)y * kmax;
  constexpr int PB = 22, HALF = 1 << (PB - 1);
  int v0 = HALF, v1 = HALF, v2 = HALF;
  for (int j = 0; j < ny; ++j) {
    const int gy = cy + y0 + j;
    int h0 = HALF, h1 = HALF, h2 = HALF;
    if (gy >= 0 && gy < mh) {
      for (int i = 0; i < nx; ++i) {
        const int gx = cx + x0 + i;
        if (gx >= 0 && gx < mw) {
          const uint8_t* px = mosaic + ((long)gy * mw + gx) * 3;
          const int k = kx[i];
          h0 += px[0] * k; h1 += px[1] * k; h2 += px[2] * k;
        }
      }
    }
    const int k = ky[j];
    v0 += min(max(h0 >> PB, 0), 255) * k;
    v1 += min(max(h1 >> PB, 0), 255) * k;
    v2 += min(max(h2 >> PB, 0), 255) * k;
  }
  const int r0 = min(max(v0 >> PB, 0), 255), r1 = min(max(v1 >> PB, 0), 255), r2 = min(max(v2 >> PB, 0), 255);
  if (out_u8) {
    uint8_t* o = out_u8 + ((long)n * S * S + p) * 3;
    o[0] = (uint8_t)r0; o[1] = (uint8_t)r1; o[2] = (uint8_t)r2;
  }
  if (out) {
    const long plane = (long)S * S;
    float* o = out + (long)n * 3 * plane + p;  // (v / 255 - mean) / std, each step correctly rounded like numpy / torch
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r0, 255.0f), m0), s0);
    o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r1, 255.0f), m1), s1);
    o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)r2, 255.0f), m2), s2);
  }
}
