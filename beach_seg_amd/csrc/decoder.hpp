// SegGPT decoder head on gfx950 (HF:modeling_seggpt.py:525-546): direct 3x3 convolution 64->64 over the NHWC
// feature map (no im2col: nine shifted NT-GEMM steps over an LDS halo tile), fused in ONE pass with bias,
// per-pixel LayerNorm(64) (wavefront-lane reductions), exact GELU and the 1x1 head (64->3), and the matching
// backward: a per-pixel kernel through head/GELU/LayerNorm and the same conv kernel with flipped weights for dgrad.
#pragma once
#include "common.hpp"

enum ConvMode { CONV_FWD_FUSED = 0, CONV_PLAIN = 1 };

struct ConvArgs {
  const void* in;     // T NHWC [B][H][W][64]
  const void* w;      // T [64 co][9 taps][64 ci]
  const float* bias;  // [64] or nullptr (CONV_PLAIN)
  void* out;          // T NHWC: conv output (pre-LN) in FUSED mode (may be nullptr), plain output otherwise
  const float* ln_g; const float* ln_b;  // [64]
  const float* head_w;                   // [3][64]
  const float* head_b;                   // [3]
  float* pred;                           // fp32 NCHW [B][3][H][W]
  int H, W;
  float eps;
  int ty0;  // first 16-row output tile (dgrad over a row window)
};

// Workgroup = 16 x 32 output pixels; wave w owns rows 4w .. 4w+3 (128 pixels) x 64 output channels (4 x 8 MFMA 16x16
// tiles): every weight fragment fetched (L1/L2: the 73 KB filter bank is re-read by every wave) feeds 8 MFMAs.
constexpr int CONV_TR = 16, CONV_RW = 4, CONV_HALO = (CONV_TR + 2) * 34;  // tile rows, rows per wave, halo pixels
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs a) {
  typedef typename Traits<T>::Chunk Chunk;
  constexpr int EPC = Traits<T>::EPC, CPP = 64 / EPC;  // chunks per pixel: 8 (bf16) / 16 (f32)
  constexpr int PB = 64 * sizeof(T);                   // bytes per pixel
  constexpr int KS = CPP / 4;                          // 16x16 k-steps per tap
  constexpr int NM = 2 * CONV_RW;                      // pixel tiles per wave: (row, x half)
  extern __shared__ __attribute__((aligned(16))) char halo[];  // [TR + 2][34] pixels x PB, chunk-swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = blockIdx.x * 32, y0 = (blockIdx.y + a.ty0) * CONV_TR, b = blockIdx.z;
  const char* img = (const char*)a.in + (long)b * a.H * a.W * PB;

  for (int i = tid; i < CONV_HALO * CPP; i += 256) {
    const int pix = i / CPP, ch = i % CPP;
    const int hy = pix / 34, hx = pix % 34, y = y0 + hy - 1, x = x0 + hx - 1;
    Chunk v;
#pragma unroll
    for (int j = 0; j < EPC; ++j) v[j] = from_f32<T>(0.f);
    if (y >= 0 && y < a.H && x >= 0 && x < a.W) v = *(const Chunk*)(img + ((long)y * a.W + x) * PB + ch * 16);
    *(Chunk*)(halo + pix * PB + ((ch ^ (pix & (CPP - 1))) << 4)) = v;
  }
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  f32x4 acc[4][NM];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1  // rolled: with 128 accumulator registers a fully unrolled tap loop hoists operand loads into spills
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = fchunk + 4 * ks;
      Chunk fa[NM], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fw[i] = *(const Chunk*)((const char*)a.w + (((long)(i * 16 + frow) * 9 + tap) * 64) * sizeof(T) + c * 16);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int pix = (CONV_RW * wave + (i >> 1) + dy) * 34 + (i & 1) * 16 + frow + dx;
        fa[i] = *(const Chunk*)(halo + pix * PB + ((c ^ (pix & (CPP - 1))) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) mma16(acc[ni][mi], fw[ni], fa[mi]);
    }
  }

  // acc[ni][mi][r]: pixel (y0 + 4*wave + (mi>>1), x0 + (mi&1)*16 + frow), channel ni*16 + 4*fchunk + r
#pragma unroll
  for (int mi = 0; mi < NM; ++mi) {
    const int y = y0 + CONV_RW * wave + (mi >> 1), x = x0 + (mi & 1) * 16 + frow;
    const long pix = ((long)b * a.H + y) * a.W + x;
    if (MODE == CONV_PLAIN) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const f32x4 v = acc[ni][mi];
        *(typename Traits<T>::Vec4*)((T*)a.out + pix * 64 + ni * 16 + 4 * fchunk) = pack4<T>(v[0], v[1], v[2], v[3]);
      }
    } else {
      f32x4 v[4];
      float s = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        v[ni] = acc[ni][mi] + *(const f32x4*)(a.bias + ni * 16 + 4 * fchunk);
        if (a.out)
          *(typename Traits<T>::Vec4*)((T*)a.out + pix * 64 + ni * 16 + 4 * fchunk) =
              pack4<T>(v[ni][0], v[ni][1], v[ni][2], v[ni][3]);
        s += v[ni][0] + v[ni][1] + v[ni][2] + v[ni][3];
      }
      // the 64 channels of a pixel live in the 4 lanes {frow, frow+16, frow+32, frow+48}
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.f / 64.f);
      float q = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[ni][r] -= mean; q += v[ni][r] * v[ni][r]; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q * (1.f / 64.f) + a.eps);
      float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int c0 = ni * 16 + 4 * fchunk;
        const f32x4 g = *(const f32x4*)(a.ln_g + c0), be = *(const f32x4*)(a.ln_b + c0);
        const f32x4 w0 = *(const f32x4*)(a.head_w + c0), w1 = *(const f32x4*)(a.head_w + 64 + c0),
                    w2 = *(const f32x4*)(a.head_w + 128 + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = gelu_f(v[ni][r] * rstd * g[r] + be[r]);
          o0 += t * w0[r]; o1 += t * w1[r]; o2 += t * w2[r];
        }
      }
      o0 += __shfl_xor(o0, 16, 64); o0 += __shfl_xor(o0, 32, 64);
      o1 += __shfl_xor(o1, 16, 64); o1 += __shfl_xor(o1, 32, 64);
      o2 += __shfl_xor(o2, 16, 64); o2 += __shfl_xor(o2, 32, 64);
      if (fchunk == 0) {
        const long hw = (long)a.H * a.W, base = (long)b * 3 * hw + (long)y * a.W + x;
        a.pred[base] = o0 + a.head_b[0];
        a.pred[base + hw] = o1 + a.head_b[1];
        a.pred[base + 2 * hw] = o2 + a.head_b[2];
      }
    }
  }
}

// Per-pixel backward of head (1x1) -> GELU -> LayerNorm(64): dpred (fp32 NCHW) + saved conv output -> d conv out.
// Four lanes per pixel (16 channels each, 32/64 contiguous bytes per lane), reductions by two xor-shuffles.
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dpred, const T* __restrict__ conv_out,
                                                        const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                        const float* __restrict__ head_w, T* __restrict__ dconv, int B,
                                                        int H, int W, float eps, int row0) {
  const long hw = (long)H * W, win = (long)(H - row0) * W, total = (long)B * win;  // rows [row0, H) of every image
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long pl = gid >> 2;
  const int c0 = (int)(gid & 3) * 16;
  if (pl >= total) return;  // total*4 is a multiple of 64: whole waves exit together
  const int b = pl / win;
  const long yx = (long)row0 * W + pl % win;
  const long p = (long)b * hw + yx;
  const float d0 = dpred[(long)b * 3 * hw + yx], d1 = dpred[(long)b * 3 * hw + hw + yx],
              d2 = dpred[(long)b * 3 * hw + 2 * hw + yx];
  typedef typename Traits<T>::Vec4 V4;
  V4* dst = (V4*)(dconv + p * 64 + c0);
  float v[16], g[16];
  const V4* src = (const V4*)(conv_out + p * 64 + c0);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const V4 t = src[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[4 * i + j] = to_f32(t[j]); s += v[4 * i + j]; }
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) { v[c] -= mean; q += v[c] * v[c]; }
  q += __shfl_xor(q, 1, 64);
  q += __shfl_xor(q, 2, 64);
  const float rstd = rsqrtf(q * (1.f / 64.f) + eps);
  float mg = 0.f, mgx = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    v[c] *= rstd;  // xhat
    const float gm = ln_g[c0 + c];
    const float y = v[c] * gm + ln_b[c0 + c];
    const float dgl = d0 * head_w[c0 + c] + d1 * head_w[64 + c0 + c] + d2 * head_w[128 + c0 + c];
    g[c] = dgl * gelu_grad_f(y) * gm;
    mg += g[c];
    mgx += g[c] * v[c];
  }
  mg += __shfl_xor(mg, 1, 64);
  mg += __shfl_xor(mg, 2, 64);
  mgx += __shfl_xor(mgx, 1, 64);
  mgx += __shfl_xor(mgx, 2, 64);
  mg *= (1.f / 64.f);
  mgx *= (1.f / 64.f);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    dst[i] = pack4<T>(rstd * (g[4 * i] - mg - v[4 * i] * mgx), rstd * (g[4 * i + 1] - mg - v[4 * i + 1] * mgx),
                      rstd * (g[4 * i + 2] - mg - v[4 * i + 2] * mgx), rstd * (g[4 * i + 3] - mg - v[4 * i + 3] * mgx));
}
