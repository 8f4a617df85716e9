// HBM-bound row kernels of the SegGPT hot path (wavefront reductions, vectorised 16-byte accesses):
// LayerNorm fwd/bwd (HF:397-398, 444, 475-476), stream merge (HF:470-473), canvas patch gather (HF:705-710 +
// HF:108 as an im2col-free k=s=16 gather), decomposed rel-pos tables (HF:236-311) and their gradient, head
// transposes, attention delta.
#pragma once
#include "common.hpp"

// ---------------------------------------------------------------------------------------------- LayerNorm
// one wave per row; x fp32 [rows][D], D <= 2048; y = T [rows][ldy] (column offset applied by the caller).
// Branch-free: lanes past D load column 0 and contribute zeros.
// NV = 16-byte pieces per lane: 4 covers D <= 1024 (ViT-L), 8 covers D <= 2048 (the loads past D used to be issued and masked:
// at D = 1024 that was every second load instruction of these kernels)
template <int NV> struct LnRow {
  f32x4 v[NV];
  float mean, rstd;
  DEVI void load(const float* xr, int lane, int D, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      const bool ok = c < D;
      const f32x4 t = *(const f32x4*)(xr + (ok ? c : 0));
      v[i] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool ok = (i * 64 + lane) * 4 < D;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = ok ? v[i][j] - mean : 0.f;
        q += v[i][j] * v[i][j];
      }
    }
    rstd = rsqrtf(wave_sum(q) / D + eps);
  }
};

template <typename T, int NV = 8>
__global__ __launch_bounds__(256, 4) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y, long ldy,
                                                         int rows, int D, float eps, int in_rpg = 0, long in_gstride = 0,
                                                         long in_off = 0) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  // in_rpg > 0: output row r (compact) reads input row (r / in_rpg) * in_gstride + in_off + r % in_rpg -- the row window of
  // every stream gathered into consecutive rows (the decoder's tap LayerNorms under bsg_forward_rows)
  const long xrow = in_rpg ? (long)(row / in_rpg) * in_gstride + in_off + row % in_rpg : row;
  LnRow<NV> r;
  r.load(x + xrow * D, lane, D, eps);
  T* yr = y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
      *(typename Traits<T>::Vec4*)(yr + c) =
          pack4<T>(r.v[i][0] * r.rstd * g[0] + b[0], r.v[i][1] * r.rstd * g[1] + b[1],
                   r.v[i][2] * r.rstd * g[2] + b[2], r.v[i][3] * r.rstd * g[3] + b[3]);
    }
  }
}

// dx[row] = dx_in[row] * in_scale + LN'(x[row]) applied to dy[row]; optional T copy for the next dgrad GEMM.
// Statistics are recomputed from x (read anyway), nothing is saved by the forward.
template <typename T, int NV = 8>
__global__ __launch_bounds__(256, 4) void ln_bwd_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* dx_in,
                                                         float in_scale, float* dx_out, T* dx_t, int rows, int D,
                                                         float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  LnRow<NV> r;
  r.load(x + (long)row * D, lane, D, eps);
  const T* dyr = dy + (long)row * lddy;
  f32x4 g[NV];
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const bool ok = c < D;
    const typename Traits<T>::Vec4 d = *(const typename Traits<T>::Vec4*)(dyr + (ok ? c : 0));
    const f32x4 gm = *(const f32x4*)(gamma + (ok ? c : 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r.v[i][j] *= r.rstd;  // xhat (0 past D)
      g[i][j] = ok ? to_f32(d[j]) * gm[j] : 0.f;
      sg += g[i][j];
      sgx += g[i][j] * r.v[i][j];
    }
  }
  const float mg = wave_sum(sg) / D, mgx = wave_sum(sgx) / D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = r.rstd * (g[i][j] - mg - r.v[i][j] * mgx);
      if (dx_in) {
        const f32x4 p = *(const f32x4*)(dx_in + (long)row * D + c);
        o += p * in_scale;
      }
      *(f32x4*)(dx_out + (long)row * D + c) = o;
      if (dx_t) *(typename Traits<T>::Vec4*)(dx_t + (long)row * D + c) = pack4<T>(o[0], o[1], o[2], o[3]);
    }
  }
}

// out[b] = (x[b] + x[B + b]) * 0.5     (HF:470-473), n4 = elements/4 of one half
__global__ void merge_halves_kernel(const float* __restrict__ x, float* __restrict__ out, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = ((const f32x4*)x)[i], b = ((const f32x4*)x)[i + n4];
    ((f32x4*)out)[i] = (a + b) * 0.5f;
  }
}

// fp32 -> T copy (used when a fp32 gradient stream feeds a dgrad GEMM without a LayerNorm in between)
template <typename T>
__global__ void cast_rows_kernel(const float* x, T* y, long n4, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = ((const f32x4*)x)[i] * scale;
    ((typename Traits<T>::Vec4*)y)[i] = pack4<T>(a[0], a[1], a[2], a[3]);
  }
}

// ------------------------------------------------------------------------------------ gradient scale (f16)
// IEEE half holds 6e-5 .. 65504 at full precision; the gradients of this path are ~1e-6.  The backward therefore runs on
// S * gradient with S = 2^k chosen from max |grad_pred| on device (no host round trip): exact in fp32, and the final
// patch-embed dgrad multiplies by 1 / S.  scale[0] = S, scale[1] = 1 / S; S = 1 when grad_pred is identically zero.
__global__ void absmax_kernel(const float* __restrict__ x, long n4, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = ((const f32x4*)x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    const float z = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);  // 0 for finite values, NaN otherwise
    if (!(z == 0.f)) m = INFINITY;  // a NaN would be dropped by fmaxf: report every non-finite input as +inf
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __builtin_bit_cast(unsigned, m));  // non-negative floats order like uints
}
// state (int32, in the same workspace region, persistent across steps while the caller keeps the workspace -- the C ABI never
// allocates device memory of its own, so the guard's memory is the caller's like everything else; one guard per workspace):
// [0] overflow flag of the LAST backward, [1] back-off exponent, [2] clean backwards since the last change, [3] dgrad overflows
// so far, [4] the LAST backward's input (grad_pred, or the forward's pred_masks) was itself non-finite, [5] backwards dropped
// for that reason so far, [6] raised by a train-mode forward whose pred_masks were non-finite, consumed by the next backward.
__global__ void grad_scale_kernel(const unsigned* __restrict__ absmax, float* __restrict__ scale, int target_exp,
                                  int* __restrict__ state) {
  const float m = __builtin_bit_cast(float, absmax[0]);
  int k = 0;
  // max |S * grad_pred| lands in [2^t, 2^(t+1)), t = target - back-off (the back-off grows when a backward overflowed)
  if (m > 0.f && m < INFINITY) k = target_exp - state[1] - (int)floorf(log2f(m));
  k = max(-100, min(100, k));
  scale[0] = exp2f((float)k);
  scale[1] = exp2f((float)-k);
  state[0] = 0;
  // non-finite INPUT -- grad_pred itself, or the forward that saved this backward's activations (state[6], raised by the
  // forward's own check of pred_masks: the reference smooth-L1 turns a NaN prediction into a FINITE gradient) -- is a bad
  // batch or a forward overflow, not the dgrad chain's doing
  state[4] = !(m < INFINITY) || state[6] != 0;
  state[6] = 0;
}
// forward, train mode, f16 / x3: any non-finite element of pred_masks raises state[6] for the backward that follows
// (plane4 > 0: only the last `keep4` float4 of every plane of `plane4` float4 are looked at -- bsg_forward_rows leaves the canvas
// rows above its window unwritten; n4 then counts the looked-at float4)
__global__ void nonfinite_flag_kernel(const float* __restrict__ x, long n4, int* __restrict__ flag, long plane4 = 0, long keep4 = 0) {
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = ((const f32x4*)x)[plane4 ? (i / keep4) * plane4 + (plane4 - keep4) + i % keep4 : i];
    const float z = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
    bad |= !(z == 0.f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
// Overflow guard of the half-precision dgrad chain (what torch's GradScaler does on the host): any non-finite element of
// the final prompt gradient raises state[0]; the caller skips its optimiser step (engine: the flag rides the gradient
// all-reduce), and the next backward runs with 4x more headroom.  After 1000 clean backwards one bit is given back.
// The back-off is raised ONLY when the incoming grad_pred was finite, i.e. when the half-precision chain really overflowed:
// a non-finite input (state[4]) drops the step as well but leaves the scale alone -- a few bad batches must not push the
// scale target down and flush small gradient elements for thousands of steps.
__global__ void grad_finite_kernel(const float* __restrict__ g, long n4, int* __restrict__ state) {
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = ((const f32x4*)g)[i];
    const float s = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);  // 0 for finite values, NaN otherwise
    bad |= !(s == 0.f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(state, 1);
}
__global__ void grad_state_kernel(int* __restrict__ state) {
  if (state[0] && state[4]) { state[5] += 1; }
  else if (state[0]) { state[1] = min(state[1] + 2, 14); state[2] = 0; state[3] += 1; }
  else if (++state[2] >= 1000 && state[1] > 0) { state[1] -= 1; state[2] = 0; }
}

// ---------------------------------------------------------------------------------- canvas patch gather
// A[m][k], m = s*N + t, k = c*256 + i*16 + j.  Stream s < B: image canvas = cat(prompt image, query image) on H
// (HF:705); s >= B: mask canvas, top half = prompt mask, bottom half is masked out by the default
// bool_masked_pos (HF:902-909) and never read: those rows are written as zeros (the token table supplies
// mask_token there).  One thread per (row, c, i): 64 B in, 16 elements out.
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ prompt_img, const float* __restrict__ query_img,
                                const float* __restrict__ prompt_mask, T* __restrict__ A, int B, int hp, int wp, int split) {
  // one thread per (row m, channel c, patch row i, half j8): 8 pixels in (32 B), 8 elements out per copy (16 B when T is
  // 16-bit): consecutive threads write consecutive 16-byte chunks of the A row, so every store instruction is one
  // contiguous run (an earlier 4 x 8-byte-per-lane form wrote partial lines: 3.6x write amplification in WRITE_SIZE)
  const int N = hp * wp, hh = hp / 2;
  const long total = (long)2 * B * N * 96;
  const int Hh = hh * 16, W = wp * 16;
  const long lda = split ? 3 * 768 : 768;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cj = idx % 96;
    const long m = idx / 96;
    const int c = cj / 32, i = (cj % 32) >> 1, j8 = cj & 1;
    const int s = m / N, t = m % N, ph = t / wp, pw = t % wp;
    const float* src = nullptr;
    if (s < B) src = (ph < hh ? prompt_img : query_img) + (((long)s * 3 + c) * Hh + (ph % hh) * 16 + i) * W + pw * 16 + j8 * 8;
    else if (ph < hh) src = prompt_mask + (((long)(s - B) * 3 + c) * Hh + ph * 16 + i) * W + pw * 16 + j8 * 8;
    T* dst = A + m * lda + c * 256 + i * 16 + j8 * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (src) { v0 = *(const f32x4*)src; v1 = *(const f32x4*)(src + 4); }
    typedef typename Traits<T>::Vec4 V4;
    const V4 h0 = pack4<T>(v0[0], v0[1], v0[2], v0[3]), h1 = pack4<T>(v1[0], v1[1], v1[2], v1[3]);
    struct alignas(2 * sizeof(V4)) Pair { V4 a, b; };
    *(Pair*)dst = Pair{h0, h1};
    if (split) {  // [hi | hi | lo]: with the weight laid out [W_hi | W_lo | W_hi] one K = 3*768 GEMM sums hi*W_hi + hi*W_lo + lo*W_hi
      *(Pair*)(dst + 768) = Pair{h0, h1};
      *(Pair*)(dst + 1536) = Pair{pack4<T>(v0[0] - to_f32(h0[0]), v0[1] - to_f32(h0[1]), v0[2] - to_f32(h0[2]), v0[3] - to_f32(h0[3])),
                                  pack4<T>(v1[0] - to_f32(h1[0]), v1[1] - to_f32(h1[1]), v1[2] - to_f32(h1[2]), v1[3] - to_f32(h1[3]))};
    }
  }
}

// fp32 rows -> [hi | hi | lo] in T (3 * D columns): the A operand of a split-precision GEMM against [W_hi | W_lo | W_hi].
// Row r of the output is physical input row (r / rpg) * gstride + r % rpg.
template <typename T>
__global__ void split_rows_kernel(const float* __restrict__ x, T* __restrict__ y, long rows, int D, int rpg, long gstride) {
  const long n4 = rows * (D / 4);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (D / 4);
    const int c = (int)(i % (D / 4)) * 4;
    const f32x4 v = *(const f32x4*)(x + ((r / rpg) * gstride + r % rpg) * D + c);
    const typename Traits<T>::Vec4 hi = pack4<T>(v[0], v[1], v[2], v[3]);
    T* dst = y + r * 3 * D + c;
    *(typename Traits<T>::Vec4*)dst = hi;
    *(typename Traits<T>::Vec4*)(dst + D) = hi;
    *(typename Traits<T>::Vec4*)(dst + 2 * D) =
        pack4<T>(v[0] - to_f32(hi[0]), v[1] - to_f32(hi[1]), v[2] - to_f32(hi[2]), v[3] - to_f32(hi[3]));
  }
}

// ------------------------------------------------------------------------------------- head transposes
// X [S*N][ld] (head columns) -> XT [S][nh][64][row_len]; one block per (group of 32 output columns, group of hb <= 4
// heads, stream).  Group g holds tokens g * tpg + kw, kw < tpg: tpg = Wp gives the row-padded layout of the key-side
// operands (group = grid row), tpg = 32 the dense token layout of the query-side operands of the dK/dV kernel.
// 16-byte loads of the contiguous hb*64-element row segments, 16-byte stores of 8 columns; padded columns = 0.
template <typename T>
__global__ __launch_bounds__(256) void head_transpose_kernel(const T* __restrict__ x, long ld, T* __restrict__ xt,
                                                              int N, int tpg, int row_len, int nh, int hb) {
  constexpr int EPC = Traits<T>::EPC;
  typedef typename Traits<T>::Chunk Chunk;
  __shared__ float tile[32][257];
  const int gr = blockIdx.x, head0 = blockIdx.y * hb, s = blockIdx.z, tid = threadIdx.x;
  const int cpt = hb * 64 / EPC;  // 16-byte chunks per token segment
  for (int i = tid; i < 32 * cpt; i += 256) {
    const int kw = i / cpt, c = i % cpt;
    float v[EPC];
    if (kw < tpg && gr * tpg + kw < N) {
      const Chunk ch = *(const Chunk*)(x + ((long)s * N + gr * tpg + kw) * ld + head0 * 64 + c * EPC);
#pragma unroll
      for (int j = 0; j < EPC; ++j) v[j] = to_f32(ch[j]);
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPC; ++j) tile[kw][c * EPC + j] = v[j];
  }
  __syncthreads();
  for (int i = tid; i < hb * 64 * 4; i += 256) {
    const int g = i & 3, hd = i >> 2;  // hd = head_local*64 + d; g = group of 8 output positions
    T* dst = xt + (((long)s * nh + head0 + (hd >> 6)) * 64 + (hd & 63)) * row_len + gr * 32 + 8 * g;
    // bf16: every 16-slot group is stored permuted (position 8h + 4a + i holds slot 8a + 4h + i) so that the MFMA
    // operand matching an S^T accumulator is one aligned 16-byte chunk (attention.hpp, lds_perm_chunk)
    const int s0 = sizeof(T) == 2 ? 16 * (g >> 1) + 4 * (g & 1) : 8 * g;       // slots of positions 0..3
    const int s1 = sizeof(T) == 2 ? s0 + 8 : s0 + 4;                           // slots of positions 4..7
    *(typename Traits<T>::Vec4*)dst = pack4<T>(tile[s0][hd], tile[s0 + 1][hd], tile[s0 + 2][hd], tile[s0 + 3][hd]);
    *(typename Traits<T>::Vec4*)(dst + 4) = pack4<T>(tile[s1][hd], tile[s1 + 1][hd], tile[s1 + 2][hd], tile[s1 + 3][hd]);
  }
}

// feature_ensemble (HF:414-423): on the bottom (query) half of the canvas replace the attention-block output a[s]
// by its mean over the streams of a group.  The proj GEMM has already written x_mid = x_in + a, so a is recovered
// as x_mid - x_in (fp32 residual stream).  groups = 2 (per stream kind, blocks before the merge) or 1.
__global__ void ensemble_fixup_kernel(const float* __restrict__ x_in, float* __restrict__ x_mid, int S, int groups, int N,
                                      int D) {
  const int per = S / groups, half = N / 2;
  const long n4 = (long)groups * half * D / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const int grp = e / ((long)half * D);
    const long r = e % ((long)half * D);  // offset inside the bottom half of one stream
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < per; ++m) {
      const long o = ((long)(grp * per + m) * N + half) * D + r;
      acc += *(const f32x4*)(x_mid + o) - *(const f32x4*)(x_in + o);
    }
    acc *= 1.0f / per;
    for (int m = 0; m < per; ++m) {
      const long o = ((long)(grp * per + m) * N + half) * D + r;
      *(f32x4*)(x_mid + o) = *(const f32x4*)(x_in + o) + acc;
    }
  }
}
