// NT GEMM  C[M][N] = A[M][K] * W[N][K]^T  for gfx950, one source for bf16 (16x16x32 MFMA) and exact f32
// (16x16x4 MFMA).  Covers every dense Linear of the SegGPT hot path (HF:modeling_seggpt.py:108, 226-227,
// 355-356, 552-556) and, with pre-transposed weights, their dgrads.
//
// Tile 128x128 x 128 bytes of K, 4 waves (2x2), wave tile 64x64 = 4x4 MFMA accumulators.  Operands go
// HBM -> LDS by LDS-DMA (16 B per lane), double-buffered; the LDS image is lane-linear, so the bank
// swizzle (chunk ^= row & 7) is applied to the per-lane SOURCE address and again on the ds_read_b128.
// The MFMA "A" operand is the weight tile and "B" the activation tile, so an accumulator register quad
// is 4 consecutive output features of one row: the natural 8/16-byte row-major store.
#pragma once
#include "common.hpp"

enum AMode { A_PLAIN = 0, A_FEAT = 1 };
enum EpiMode {
  EPI_BIAS = 0,       // out[m][n] = T(acc + bias[n])
  EPI_BIAS_GELU = 1,  // out2[m][n] = T(h = acc + bias[n]) (if out2), out[m][n] = T(gelu(h))
  EPI_BIAS_RESID = 2, // outf[m][n] = resid[m][n] + acc + bias[n]            (fp32 residual stream)
  EPI_EMBED = 3,      // outf[m][n] = acc + table[kind(m)][tok(m)][n]        (patch embed + tokens)
  EPI_FEAT = 4,       // decoder_embed: pixel-shuffle store into NHWC feature map (+bias)
  EPI_PLAIN = 5,      // out[m][n] = T(acc)
  EPI_GELU_BWD = 6,   // out[m][n] = T(acc * gelu'(aux[m][n]))
  EPI_UNPATCH = 7,    // patch-embed dgrad: scatter rows into the (B,3,H/2,W) fp32 prompt-pixel gradient
};

struct GemmArgs {
  const void* A;
  const void* W;
  int M, N, K;
  long lda;  // elements
  // A_PLAIN row map: physical row = (m / a_rpg) * a_gstride + (m % a_rpg)
  int a_rpg;
  long a_gstride;
  // geometry for A_FEAT / EPI_FEAT / EPI_EMBED / EPI_UNPATCH
  int tokens;  // tokens per stream (Hp * Wp)
  int wp;      // token-grid width
  int himg, wimg;  // canvas pixels
  int batch;       // B (streams of kind 0)
  // epilogue
  const float* bias;
  void* out;
  long ldo;
  void* out2;
  const void* aux;  // T* (GELU_BWD pre-activation) or float* (resid / table)
  long ldaux;
};

template <typename T, int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BK = 8 * EPC;  // elements per 128-byte K tile
  typedef typename Traits<T>::Chunk Chunk;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][A|W][128 rows][128 B]
  char* lds_a0 = smem;
  char* lds_w0 = smem + 2 * 16384;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + 127) >> 7, tiles_m = (g.M + 127) >> 7;
  const int nwg = tiles_m * tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  // bands of 16 row-tiles: inside a band walk all column tiles for one row tile before the next, so the
  // activation tile is fetched from HBM once and the weight panel stays in the XCD's L2.
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm << 7, n0 = tn << 7;

  // ---- per-lane source pointers (constant over K): 4 row groups of 8 rows per wave for A and for W
  const int prow = lane >> 3, pchunk = lane & 7;
  const char* a_src[4];
  const char* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + prow;
    const int sc = pchunk ^ (r & 7);
    int m = m0 + r;
    if (m >= g.M) m = g.M - 1;
    long base;
    if (AMODE == A_PLAIN) {
      base = ((long)(m / g.a_rpg) * g.a_gstride + (m % g.a_rpg)) * g.lda;
    } else {  // NHWC feature map (B, himg, wimg, 64): row m = (b, ph, pw) starts at pixel (ph*16, pw*16)
      const int b = m / g.tokens, t = m % g.tokens;
      const int ph = t / g.wp, pw = t % g.wp;
      base = (((long)b * g.himg + ph * 16) * g.wimg + pw * 16) * 64;
    }
    a_src[i] = (const char*)g.A + base * sizeof(T) + sc * 16;
    int n = n0 + r;
    if (n >= g.N) n = g.N - 1;
    w_src[i] = (const char*)g.W + (long)n * g.K * sizeof(T) + sc * 16;
  }

  auto stage = [&](int kt, int buf) {
    const long k0 = (long)kt * BK;
    long ka;
    if (AMODE == A_PLAIN) ka = k0;
    else ka = (k0 >> 10) * ((long)g.wimg * 64) + (k0 & 1023);  // k = p1*1024 + (p2*64 + c)
    char* la = lds_a0 + buf * 16384 + wave * 4096;
    char* lw = lds_w0 + buf * 16384 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + ka * sizeof(T), la + i * 1024);
      glds16(w_src[i] + k0 * sizeof(T), lw + i * 1024);
    }
  };

  const int wr = wave >> 1, wc = wave & 1;  // wave tile: rows (m) wr*64.., cols (n) wc*64..
  const int frow = lane & 15, fchunk = lane >> 4;
  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    wait_vm0();
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const char* la = lds_a0 + buf * 16384;
    const char* lw = lds_w0 + buf * 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Chunk fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + frow;
        const int rw = wc * 64 + i * 16 + frow;
        const int c = fchunk + 4 * ks;
        fa[i] = *(const Chunk*)(la + ra * 128 + ((c ^ (ra & 7)) << 4));
        fw[i] = *(const Chunk*)(lw + rw * 128 + ((c ^ (rw & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) mma16(acc[ni][mi], fw[ni], fa[mi]);
    }
  }

  // ---- epilogue: acc[ni][mi][r] = C[m0 + wr*64 + mi*16 + (lane&15)][n0 + wc*64 + ni*16 + 4*(lane>>4) + r]
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wr * 64 + mi * 16 + frow;
    if (m >= g.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + 4 * fchunk;
      if (n >= g.N) continue;
      f32x4 v = acc[ni][mi];
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID || EPI == EPI_FEAT) {
        const f32x4 b = *(const f32x4*)(g.bias + n);
        v += b;
      }
      if (EPI == EPI_BIAS || EPI == EPI_PLAIN) {
        *(typename Traits<T>::Vec4*)((T*)g.out + (long)m * g.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_BIAS_GELU) {
        if (g.out2)
          *(typename Traits<T>::Vec4*)((T*)g.out2 + (long)m * g.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
        *(typename Traits<T>::Vec4*)((T*)g.out + (long)m * g.ldo + n) =
            pack4<T>(gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3]));
      } else if (EPI == EPI_BIAS_RESID) {
        const f32x4 r = *(const f32x4*)((const float*)g.aux + (long)m * g.ldaux + n);
        *(f32x4*)((float*)g.out + (long)m * g.ldo + n) = v + r;
      } else if (EPI == EPI_EMBED) {
        const int s = m / g.tokens, t = m % g.tokens;
        const int kind = s >= g.batch ? 1 : 0;
        const f32x4 r = *(const f32x4*)((const float*)g.aux + ((long)kind * g.tokens + t) * g.ldaux + n);
        *(f32x4*)((float*)g.out + (long)m * g.ldo + n) = v + r;
      } else if (EPI == EPI_FEAT) {
        // n = (p1*16 + p2)*64 + c  ->  pixel (ph*16 + p1, pw*16 + p2), channel c   (HF:559-572)
        const int b = m / g.tokens, t = m % g.tokens;
        const int ph = t / g.wp, pw = t % g.wp;
        const int p1 = n >> 10, p2 = (n >> 6) & 15, c = n & 63;
        const long o = (((long)b * g.himg + ph * 16 + p1) * g.wimg + pw * 16 + p2) * 64 + c;
        *(typename Traits<T>::Vec4*)((T*)g.out + o) = pack4<T>(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_GELU_BWD) {
        const typename Traits<T>::Vec4 h = *(const typename Traits<T>::Vec4*)((const T*)g.aux + (long)m * g.ldaux + n);
        *(typename Traits<T>::Vec4*)((T*)g.out + (long)m * g.ldo + n) =
            pack4<T>(v[0] * gelu_grad_f(to_f32(h[0])), v[1] * gelu_grad_f(to_f32(h[1])),
                     v[2] * gelu_grad_f(to_f32(h[2])), v[3] * gelu_grad_f(to_f32(h[3])));
      } else if (EPI == EPI_UNPATCH) {
        // rows m = (b, t) over the TOP half tokens only (a_rpg = tokens/2); n = c*256 + i*16 + j
        const int half = g.tokens >> 1;
        const int b = m / half, t = m % half;
        const int ph = t / g.wp, pw = t % g.wp;
        const int c = n >> 8, i = (n >> 4) & 15, j = n & 15;
        const long o = (((long)b * 3 + c) * (g.himg >> 1) + ph * 16 + i) * g.wimg + pw * 16 + j;
        *(f32x4*)((float*)g.out + o) = v;
      }
    }
  }
}

template <typename T, int AMODE, int EPI>
static inline void launch_gemm(const GemmArgs& g, hipStream_t st) {
  const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
  hipLaunchKernelGGL((gemm_nt_kernel<T, AMODE, EPI>), dim3(tiles), dim3(256), 65536, st, g);
}
