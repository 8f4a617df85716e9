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
#include <algorithm>

enum AMode { A_PLAIN = 0, A_FEAT = 1 };
enum EpiMode {
  EPI_BIAS = 0,       // out[m][n] = T(acc + bias[n])
  EPI_BIAS_GELU = 1,  // h = acc + bias[n]: out[m][n] = T(gelu(h)), out2[m][n] = T(gelu'(h)) (if out2: saved for dgrad)
  EPI_BIAS_RESID = 2, // outf[m][n] = resid[m][n] + acc + bias[n]            (fp32 residual stream)
  EPI_EMBED = 3,      // outf[m][n] = acc + table[kind(m)][tok(m)][n]        (patch embed + tokens)
  EPI_FEAT = 4,       // decoder_embed: pixel-shuffle store into NHWC feature map (+bias)
  EPI_PLAIN = 5,      // out[m][n] = T(acc)
  EPI_GELU_BWD = 6,   // out[m][n] = T(acc * aux[m][n]),  aux = gelu'(h) saved by EPI_BIAS_GELU
  EPI_UNPATCH = 7,    // patch-embed dgrad: scatter rows into the (B,3,H/2,W) fp32 prompt-pixel gradient
  EPI_BIAS_GELU_FWD = 8,  // inference: out[m][n] = T(gelu(acc + bias[n])) only -- no derivative is formed (a third less epilogue arithmetic)
  EPI_NONE = 9,       // diagnostics: no stores (accumulators kept alive), to price the epilogue
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
  int feat_lg;     // log2 of the decoder feature width (6: 64 channels, 7: 128) -- NHWC feature map addressing of A_FEAT / EPI_FEAT
  int batch;       // B (streams of kind 0)
  // epilogue
  const float* bias;
  void* out;
  long ldo;
  void* out2;
  const void* aux;  // T* (GELU_BWD pre-activation) or float* (resid / table)
  long ldaux;
  // A_FEAT / EPI_PLAIN row windows: GEMM row m <-> image m / a_rpg, token t_off + m % a_rpg (A_FEAT), and output
  // row (m / o_rpg) * o_gstride + o_off + m % o_rpg (EPI_PLAIN); o_rpg == 0 means the identity.  EPI_FEAT with o_rpg:
  // GEMM row m is token t_off + m % o_rpg of image m / o_rpg (the forward's decoder_embed over a token-row window)
  int t_off;
  int o_rpg;
  long o_gstride, o_off;
  const float* out_scale;  // EPI_UNPATCH: device scalar multiplied into the output (un-scaling of the fp16 gradient), or nullptr
  int x3;                  // f32 only: 1 = the three-term f16 split form of v3 (gemm_nt_kernel_v3<..., X3 = true>)
  int pack_hl;             // f32, EPI_BIAS only (the QKV projection in x3 mode): every output element is stored as the pair
                           // (f16 hi | f16 lo << 16), hi = f16(x), lo = f16(x - hi), in place of the float: the x3 attention
                           // kernels then take q / k / v operand halves with two byte-permutes per chunk instead of
                           // re-splitting every tile in every workgroup (attention.hpp x3_frag8_pk)
  float acc_scale;         // f32 only, 0 = off: the accumulator is multiplied by this power of two before the epilogue (x3 mode:
                           // the Linear weights are stored multiplied by its inverse so that their f16 hi / lo parts are normal)
  int group_m;  // v3: row tiles per L2 group (0 = 4)
};

// ---- shared epilogue: acc[ni][mi][r] = C[mw + mi*16 + (lane&15)][nw + ni*16 + 4*(lane>>4) + r]
// 16-bit outputs, N % 16 == 0: exchange register pairs between the four 16-lane rows (v_permlane16_swap) so every
// lane owns 8 consecutive columns -> 16-byte stores, 64 contiguous bytes per output row per instruction (the
// narrow path writes 32-byte segments and doubles the number of memory requests of the tile's store burst).
// Prefetched aux operand of one 64 x 64 sub-tile (EPI_BIAS_RESID: the fp32 residual; EPI_GELU_BWD: the saved gelu' in T, in
// the first half of each slot): a kernel with one wave per SIMD (v5) requests sub-tile q + 1's while it finishes sub-tile q,
// so that only the first request of an output tile waits for memory with nothing else to do.
template <typename T, int EPI>
DEVI void gemm_epilogue_aux_load(const GemmArgs& g, f32x4 (&ax)[4][4], int mw, int nw, int frow, int fchunk) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int mc = min(mw + mi * 16 + frow, g.M - 1);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = min(nw + ni * 16 + 4 * fchunk, g.N - 4);
      if constexpr (EPI == EPI_BIAS_RESID) {
        ax[ni][mi] = *(const f32x4*)((const float*)g.aux + (long)mc * g.ldaux + n);
      } else if constexpr (EPI == EPI_GELU_BWD) {
        const f32x2 h = *(const f32x2*)((const T*)g.aux + (long)mc * g.ldaux + n);
        ax[ni][mi] = f32x4{h[0], h[1], 0.f, 0.f};
      }
    }
  }
}

template <typename T, int EPI, int NMI = 4>
DEVI void gemm_epilogue_wide16(const GemmArgs& g, f32x4 (&acc)[4][4], int mw, int nw, int frow, int fchunk,
                               const f32x4 (*ax)[4] = nullptr, const f32x4* bv = nullptr) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int m = mw + mi * 16 + frow;
    const bool mok = m < g.M;
    const int mc = mok ? m : g.M - 1;
    unsigned lo[4], hi[4], lo2[4], hi2[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = min(nw + ni * 16 + 4 * fchunk, g.N - 4);
      f32x4 v = acc[ni][mi];
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD || EPI == EPI_FEAT) v += bv ? bv[ni] : *(const f32x4*)(g.bias + n);
      if (EPI == EPI_BIAS_GELU_FWD) {
        const f32x2 y0 = gelu_pk(f32x2{v[0], v[1]}), y1 = gelu_pk(f32x2{v[2], v[3]});
        v = f32x4{y0[0], y0[1], y1[0], y1[1]};
      } else if (EPI == EPI_BIAS_GELU) {
        f32x4 y, dy;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2 yy, dd;
          gelu_both_pk(f32x2{v[r], v[r + 1]}, yy, dd);
          y[r] = yy[0]; y[r + 1] = yy[1]; dy[r] = dd[0]; dy[r + 1] = dd[1];
        }
        lo2[ni] = pack2<T>(dy[0], dy[1]);
        hi2[ni] = pack2<T>(dy[2], dy[3]);
        v = y;
      } else if (EPI == EPI_GELU_BWD) {
        typedef typename Traits<T>::Vec4 HV;
        const HV hp = ax ? __builtin_bit_cast(HV, f32x2{ax[ni][mi][0], ax[ni][mi][1]}) : *(const HV*)((const T*)g.aux + (long)mc * g.ldaux + n);
        v = f32x4{v[0] * (float)hp[0], v[1] * (float)hp[1], v[2] * (float)hp[2], v[3] * (float)hp[3]};
      }
      lo[ni] = pack2<T>(v[0], v[1]);
      hi[ni] = pack2<T>(v[2], v[3]);
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int ia = 2 * pr, ib = 2 * pr + 1;
      auto r0 = __builtin_amdgcn_permlane16_swap(lo[ia], lo[ib], false, false);
      auto r1 = __builtin_amdgcn_permlane16_swap(hi[ia], hi[ib], false, false);
      const u32x4 out = u32x4{r0[0], r1[0], r0[1], r1[1]};
      const int col = nw + ((fchunk & 1) ? ib : ia) * 16 + (fchunk >> 1) * 8;
      const bool ok = mok && col < g.N;
      long off;
      if (EPI == EPI_FEAT) {
        const int rpg = g.o_rpg ? g.o_rpg : g.tokens;  // o_rpg: row window, GEMM row m = token t_off + m % o_rpg of image m / o_rpg
        const int b = m / rpg, t = (g.o_rpg ? g.t_off : 0) + m % rpg;
        const int ph = t / g.wp, pw = t % g.wp;
        const int lg = g.feat_lg, p1 = col >> (lg + 4), p2 = (col >> lg) & 15, c = col & ((1 << lg) - 1);
        off = ((((long)b * g.himg + ph * 16 + p1) * g.wimg + pw * 16 + p2) << lg) + c;
      } else {
        const long orow = (EPI == EPI_PLAIN && g.o_rpg) ? (long)(m / g.o_rpg) * g.o_gstride + g.o_off + m % g.o_rpg : m;
        off = orow * g.ldo + col;
      }
      if (ok) *(u32x4*)((T*)g.out + off) = out;
      if (EPI == EPI_BIAS_GELU) {
        auto q0 = __builtin_amdgcn_permlane16_swap(lo2[ia], lo2[ib], false, false);
        auto q1 = __builtin_amdgcn_permlane16_swap(hi2[ia], hi2[ib], false, false);
        if (ok && g.out2) *(u32x4*)((T*)g.out2 + off) = u32x4{q0[0], q1[0], q0[1], q1[1]};
      }
    }
  }
}

template <typename T, int EPI, int NMI = 4>
DEVI void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[4][4], int mw, int nw, int frow, int fchunk, const f32x4 (*ax)[4] = nullptr,
                        const f32x4* bv = nullptr) {  // bv: the sub-tile's four bias vectors (column blocks ni), loaded by the caller
  if constexpr (sizeof(T) == 2 && (EPI == EPI_PLAIN || EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD || EPI == EPI_GELU_BWD ||
                                   EPI == EPI_FEAT)) {
    if (!(g.N & 15)) {
      gemm_epilogue_wide16<T, EPI, NMI>(g, acc, mw, nw, frow, fchunk, ax, bv);
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < NMI; ++mi) {
    const int m = mw + mi * 16 + frow;
    if (m >= g.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = nw + ni * 16 + 4 * fchunk;
      if (n >= g.N) continue;
      f32x4 v = acc[ni][mi];
      if (EPI == EPI_NONE) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
      if constexpr (sizeof(T) == 4) { if (g.acc_scale != 0.f) v *= g.acc_scale; }
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD || EPI == EPI_BIAS_RESID || EPI == EPI_FEAT) {
        const f32x4 b = bv ? bv[ni] : *(const f32x4*)(g.bias + n);
        v += b;
      }
      if (EPI == EPI_BIAS || EPI == EPI_PLAIN) {
        const long orow = (EPI == EPI_PLAIN && g.o_rpg) ? (long)(m / g.o_rpg) * g.o_gstride + g.o_off + m % g.o_rpg : m;
        if constexpr (sizeof(T) == 4 && EPI == EPI_BIAS) {
          if (g.pack_hl) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const f16_t hi = (f16_t)v[r];
              pk[r] = __builtin_bit_cast(unsigned, f16x2{hi, (f16_t)(v[r] - (float)hi)});
            }
            *(u32x4*)((float*)g.out + orow * g.ldo + n) = pk;
            continue;
          }
        }
        *(typename Traits<T>::Vec4*)((T*)g.out + orow * g.ldo + n) = pack4<T>(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_BIAS_GELU_FWD) {
        const f32x2 y0 = gelu_pk(f32x2{v[0], v[1]}), y1 = gelu_pk(f32x2{v[2], v[3]});
        *(typename Traits<T>::Vec4*)((T*)g.out + (long)m * g.ldo + n) = pack4<T>(y0[0], y0[1], y1[0], y1[1]);
      } else if (EPI == EPI_BIAS_GELU) {
        f32x4 y, dy;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2 yy, dd;
          gelu_both_pk(f32x2{v[r], v[r + 1]}, yy, dd);
          y[r] = yy[0]; y[r + 1] = yy[1]; dy[r] = dd[0]; dy[r + 1] = dd[1];
        }
        if (g.out2)
          *(typename Traits<T>::Vec4*)((T*)g.out2 + (long)m * g.ldo + n) = pack4<T>(dy[0], dy[1], dy[2], dy[3]);
        *(typename Traits<T>::Vec4*)((T*)g.out + (long)m * g.ldo + n) = pack4<T>(y[0], y[1], y[2], y[3]);
      } else if (EPI == EPI_BIAS_RESID) {
        const f32x4 r = ax ? ax[ni][mi] : *(const f32x4*)((const float*)g.aux + (long)m * g.ldaux + n);
        *(f32x4*)((float*)g.out + (long)m * g.ldo + n) = v + r;
      } else if (EPI == EPI_EMBED) {
        const int s = m / g.tokens, t = m % g.tokens;
        const int kind = s >= g.batch ? 1 : 0;
        const f32x4 r = *(const f32x4*)((const float*)g.aux + ((long)kind * g.tokens + t) * g.ldaux + n);
        *(f32x4*)((float*)g.out + (long)m * g.ldo + n) = v + r;
      } else if (EPI == EPI_FEAT) {
        // n = (p1*16 + p2)*C + c  ->  pixel (ph*16 + p1, pw*16 + p2), channel c, C = decoder_hidden_size   (HF:559-572)
        const int rpg = g.o_rpg ? g.o_rpg : g.tokens;
        const int b = m / rpg, t = (g.o_rpg ? g.t_off : 0) + m % rpg;
        const int ph = t / g.wp, pw = t % g.wp;
        const int lg = g.feat_lg, p1 = n >> (lg + 4), p2 = (n >> lg) & 15, c = n & ((1 << lg) - 1);
        const long o = ((((long)b * g.himg + ph * 16 + p1) * g.wimg + pw * 16 + p2) << lg) + c;
        *(typename Traits<T>::Vec4*)((T*)g.out + o) = pack4<T>(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_GELU_BWD) {
        typedef typename Traits<T>::Vec4 HV;
        HV h;
        if constexpr (sizeof(T) == 2) h = ax ? __builtin_bit_cast(HV, f32x2{ax[ni][mi][0], ax[ni][mi][1]}) : *(const HV*)((const T*)g.aux + (long)m * g.ldaux + n);
        else h = *(const HV*)((const T*)g.aux + (long)m * g.ldaux + n);
        *(typename Traits<T>::Vec4*)((T*)g.out + (long)m * g.ldo + n) =
            pack4<T>(v[0] * to_f32(h[0]), v[1] * to_f32(h[1]), v[2] * to_f32(h[2]), v[3] * to_f32(h[3]));
      } else if (EPI == EPI_UNPATCH) {
        // rows m = (b, t) over the TOP half tokens only (a_rpg = tokens/2); n = c*256 + i*16 + j
        const int half = g.tokens >> 1;
        const int b = m / half, t = m % half;
        const int ph = t / g.wp, pw = t % g.wp;
        const int c = n >> 8, i = (n >> 4) & 15, j = n & 15;
        const long o = (((long)b * 3 + c) * (g.himg >> 1) + ph * 16 + i) * g.wimg + pw * 16 + j;
        *(f32x4*)((float*)g.out + o) = g.out_scale ? v * g.out_scale[0] : v;
      }
    }
  }
}

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
      const int b = m / g.a_rpg, t = g.t_off + m % g.a_rpg;
      const int ph = t / g.wp, pw = t % g.wp;
      base = (((long)b * g.himg + ph * 16) * g.wimg + pw * 16) << g.feat_lg;
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
    else ka = (k0 >> (g.feat_lg + 4)) * ((long)g.wimg << g.feat_lg) + (k0 & ((16 << g.feat_lg) - 1));  // k = p1*16C + (p2*C + c)
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

  gemm_epilogue<T, EPI>(g, acc, m0 + wr * 64, n0 + wc * 64, frow, fchunk);
}

// ------------------------------------------------------------------------------------------------------------
// The thin last round of a persistent 256 x 256 launch (launch_gemm_rows in seggpt_api.hip): the same 128 x 128 tile and wave
// layout as gemm_nt_kernel, but FOUR LDS stages with three K tiles of LDS-DMA in flight behind counted waits.  A tail is 128
// workgroups on 256 CUs -- one per CU, nothing else resident -- so the two-stage kernel above paid a full HBM round trip per
// K tile (measured 17 / 61 us at K = 1024 / 4096 where the MFMAs need 4 / 16).  The DMA is the asm form: hipcc drains the
// builtin one before the next LDS read.
// TMT = rows of the tile (columns: 128).  128: four waves of 64 x 64.  64: four waves of 32 x 64 -- twice the workgroups, for tails
// that would otherwise leave half of the CUs idle (2,048 rows x 1,024 columns = 128 tiles of 128 x 128, 256 of 64 x 128).
// X3 (T = float): the three-term f16 split of gemm_nt_kernel_v3<..., X3> on this tile -- activations split in registers, weights
// pre-split by the host, three `v_mfma_f32_16x16x32_f16` per K tile and accumulator in v3's order (the same bits as v3's rows).
// Without it an x3 GEMM of 6.125 rounds paid a whole seventh round of the 256 x 256 kernel (0.4 ms at K = 4096).
template <typename T, int EPI, int TMT = 128, bool X3 = false>
__global__ __launch_bounds__(256, 1) void gemm_nt_tail_kernel(GemmArgs g) {
  static_assert(!X3 || sizeof(T) == 4, "the three-term f16 split is the float32 kernel's option");
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BK = 8 * EPC;  // elements per 128-byte K tile
  constexpr int NS = 4, ABYTES = TMT * 128, STAGE = ABYTES + 16384;  // [stage][A TMT rows x 128 B | W 128 rows x 128 B]
  constexpr int NA = TMT / 32, MI = TMT / 32;  // A pieces (8 rows x 128 B) per wave and stage; 16-row fragments per wave tile
  constexpr int PIECES = NA + 4;               // LDS-DMA instructions per wave and stage
  typedef typename Traits<T>::Chunk Chunk;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + 127) >> 7, tiles_m = (g.M + TMT - 1) / TMT;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * TMT, n0 = tn << 7;

  const int prow = lane >> 3, pchunk = lane & 7;
  const char* a_src[NA];
  const char* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < NA) {
      const int r = (wave * NA + i) * 8 + prow;
      int m = m0 + r;
      if (m >= g.M) m = g.M - 1;
      a_src[i] = (const char*)g.A + ((long)(m / g.a_rpg) * g.a_gstride + (m % g.a_rpg)) * g.lda * sizeof(T) + ((pchunk ^ (r & 7)) << 4);
    }
    const int r = (wave * 4 + i) * 8 + prow;
    int n = n0 + r;
    if (n >= g.N) n = g.N - 1;
    w_src[i] = (const char*)g.W + (long)n * g.K * sizeof(T) + ((pchunk ^ (r & 7)) << 4);
  }
  auto stage = [&](int kt) {
    const long k0 = (long)kt * BK * sizeof(T);
    char* ls = smem + (kt & (NS - 1)) * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < NA) glds16_asm(a_src[i] + k0, ls + (wave * NA + i) * 1024);
      glds16_asm(w_src[i] + k0, ls + ABYTES + (wave * 4 + i) * 1024);
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  const int frow = lane & 15, fchunk = lane >> 4;
  f32x4 acc[4][4];  // [ni][mi], mi < MI
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;  // >= NS - 1 (the launcher checks)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) stage(s);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed when at most the PIECES of each younger tile are outstanding (vmcnt retires in order)
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PIECES) : "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIECES) : "memory");
    else wait_vm0();
    __syncthreads();  // ... for every wave's pieces; and every wave is done reading tile kt - 1, whose stage is refilled now
    if (kt + NS - 1 < nk) stage(kt + NS - 1);
    const char* la = smem + (kt & (NS - 1)) * STAGE;
    const char* lw = la + ABYTES;
    if constexpr (X3) {
      Chunk fa[2][MI], fw[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = fchunk + 4 * ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < MI) {
            const int ra = wr * (TMT / 2) + i * 16 + frow;
            fa[ks][i] = *(const Chunk*)(la + ra * 128 + ((c ^ (ra & 7)) << 4));
          }
          const int rw = wc * 64 + i * 16 + frow;
          fw[ks][i] = *(const Chunk*)(lw + rw * 128 + ((c ^ (rw & 7)) << 4));
        }
      }
      f16x8 ah8[MI], al8[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = fa[ks][i][e];
            const f16_t hi = (f16_t)x;
            ah8[i][4 * ks + e] = hi;
            al8[i][4 * ks + e] = (f16_t)(x - (float)hi);
          }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const f16x8 w0 = __builtin_bit_cast(f16x8, fw[0][ni]), w1 = __builtin_bit_cast(f16x8, fw[1][ni]);
        const f16x8 wh = f16x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        const f16x8 wl = f16x8{w0[4], w0[5], w0[6], w0[7], w1[4], w1[5], w1[6], w1[7]};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4& c = acc[ni][mi];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah8[mi], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al8[mi], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah8[mi], c, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Chunk fa[MI], fw[4];
        const int c = fchunk + 4 * ks;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < MI) {
            const int ra = wr * (TMT / 2) + i * 16 + frow;
            fa[i] = *(const Chunk*)(la + ra * 128 + ((c ^ (ra & 7)) << 4));
          }
          const int rw = wc * 64 + i * 16 + frow;
          fw[i] = *(const Chunk*)(lw + rw * 128 + ((c ^ (rw & 7)) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) mma16(acc[ni][mi], fw[ni], fa[mi]);
      }
    }
  }
  gemm_epilogue<T, EPI, MI>(g, acc, m0 + wr * (TMT / 2), n0 + wc * 64, frow, fchunk);
}

// ------------------------------------------------------------------------------------------------------------
// v2: 256 x 128 tile, 8 waves (4 x 2, same 64 x 64 wave tile), THREE LDS stages of 48 KB.  Two K tiles of LDS-DMA
// stay in flight across the (raw) barrier behind a counted s_waitcnt vmcnt(6): one block per CU, two waves per
// SIMD, never a vmcnt(0) inside the loop.
template <typename T, int AMODE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel_v2(GemmArgs g) {
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int STAGE = 49152;  // A 256 x 128 B, then W 128 x 128 B
  typedef typename Traits<T>::Chunk Chunk;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + 127) >> 7, tiles_m = (g.M + 255) >> 8;
  const int nwg = tiles_m * tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm << 8, n0 = tn << 7;

  const int prow = lane >> 3, pchunk = lane & 7;
  const char* a_src[4];
  const char* w_src[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + prow;
    const int sc = pchunk ^ (r & 7);
    int m = m0 + r;
    if (m >= g.M) m = g.M - 1;
    long base;
    if (AMODE == A_PLAIN) {
      base = ((long)(m / g.a_rpg) * g.a_gstride + (m % g.a_rpg)) * g.lda;
    } else {
      const int b = m / g.a_rpg, t = g.t_off + m % g.a_rpg;
      const int ph = t / g.wp, pw = t % g.wp;
      base = (((long)b * g.himg + ph * 16) * g.wimg + pw * 16) << g.feat_lg;
    }
    a_src[i] = (const char*)g.A + base * sizeof(T) + sc * 16;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + prow;
    const int sc = pchunk ^ (r & 7);
    int n = n0 + r;
    if (n >= g.N) n = g.N - 1;
    w_src[i] = (const char*)g.W + (long)n * g.K * sizeof(T) + sc * 16;
  }

  auto stage = [&](int kt, int st) {
    const long k0 = (long)kt * BK;
    long ka;
    if (AMODE == A_PLAIN) ka = k0;
    else ka = (k0 >> (g.feat_lg + 4)) * ((long)g.wimg << g.feat_lg) + (k0 & ((16 << g.feat_lg) - 1));
    char* la = smem + st * STAGE + wave * 4096;
    char* lw = smem + st * STAGE + 32768 + wave * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + ka * sizeof(T), la + i * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(w_src[i] + k0 * sizeof(T), lw + i * 1024);
  };

  const int wr = wave >> 1, wc = wave & 1;  // 4 x 2 waves
  const int frow = lane & 15, fchunk = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the 6 DMAs of tile kt+1 are still outstanding
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's part of tile kt is in LDS; every wave is done reading tile kt-1
    if (kt + 2 < nk) stage(kt + 2, st == 0 ? 2 : st - 1);  // stage of tile kt-1
    const char* la = smem + st * STAGE;
    const char* lw = la + 32768;
    Chunk fa[2][4], fw[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + frow;
        const int rw = wc * 64 + i * 16 + frow;
        const int c = fchunk + 4 * ks;
        fa[ks][i] = *(const Chunk*)(la + ra * 128 + ((c ^ (ra & 7)) << 4));
        fw[ks][i] = *(const Chunk*)(lw + rw * 128 + ((c ^ (rw & 7)) << 4));
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) mma16(acc[ni][mi], fw[ks][ni], fa[ks][mi]);
    st = st == 2 ? 0 : st + 1;
  }
  gemm_epilogue<T, EPI>(g, acc, m0 + wr * 64, n0 + wc * 64, frow, fchunk);
}

// ------------------------------------------------------------------------------------------------------------
// v3: 256 x 256 tile, 8 waves (2 x 4; wave tile 128 x 64 = four 64 x 32 quadrants), two 64 KB LDS buffers.
// A K tile is consumed in FOUR phases (one accumulator quadrant = 16 MFMAs each); every phase also issues one
// 16 KB quarter of the NEXT K tile by LDS-DMA (2 per lane) and reads only the fragments its quadrant adds:
//     phase 0: A rows mh0 (8 reads) + W rows nh0 (4)   | DMA Q0a = A rows mh0 of tile t+1
//     phase 1: W rows nh1 (4)                           | DMA Q0b = W rows nh0
//     phase 2: A rows mh1 (8)                           | DMA Q1  = W rows nh1
//     phase 3: --                                       | DMA Q2  = A rows mh1
// so every quarter is issued >= 3 phases before its first read and >= 5 phases after the last read of the
// region it overwrites.  One counted `s_waitcnt vmcnt(4)` per phase (two quarters stay in flight across the
// barriers; never vmcnt(0) in the loop), raw s_barrier, MFMA clusters under s_setprio.  Waves 4-7 run one barrier
// behind waves 0-3, so on every SIMD one wave is in its MFMA cluster while the other issues reads and DMA.
#ifdef BSG_DIAG_STAMPS
__device__ long long bsg_stamps[256 * 4];
#endif
// TM: rows of the output tile.  Only 256 is instantiated (a 224-row form -- whole rounds on 256 CUs for M = 100,352 -- was built in
// round 3 and measured slower on every shape of the step: DESIGN.md section 8).
// X3 (T = float only): "float32 at three f16 MFMAs".  Every f32 operand is used as hi = f16(x), lo = f16(x - hi) (22
// significant bits; the products hi*hi + hi*lo + lo*hi are exact in the fp32 accumulator, lo*lo ~2^-22 is dropped): the
// eight exact-f32 `v_mfma_f32_16x16x4_f32` of a K tile and accumulator tile (32 k-values, 8 x 32 cycles) become three
// `v_mfma_f32_16x16x32_f16` (3 x 16 cycles) on the same fragment geometry (lane = row, 4 + 4 k-values per lane).  LDS image,
// DMA, phases and epilogues are the f32 kernel's; activation fragments are split in registers once, at the head of the MFMA
// phase that first uses them; weights come pre-split from the host (first version: both split in registers with the K = 16
// f16 MFMA, which runs at half the rate on gfx950: 246 TFLOP/s against 117 for the exact kernel).
template <typename T, int AMODE, int EPI, int TM = 256, bool X3 = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel_v3(GemmArgs g) {
  static_assert(TM == 256, "row tile: 256");
  static_assert(!X3 || sizeof(T) == 4, "the three-term f16 split is the float32 kernel's option");
  constexpr int WR = TM / 2;                 // rows per wave row: mh0 = 64, mh1 = WR - 64
  constexpr int MH1 = (WR - 64) / 16;        // 16-row tiles of the second half: 4 or 3
  constexpr int G1 = (WR - 64) / 8;          // 8-row DMA groups of the second half per wave row: 8 or 6
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int BUF = 65536;  // A 256 x 128 B, then W 256 x 128 B
  typedef typename Traits<T>::Chunk Chunk;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + 255) >> 8, tiles_m = (g.M + TM - 1) / TM;
  const int nwg = tiles_m * tiles_n;
  // Persistent form (g.persist): the grid is one workgroup per CU and each walks the virtual block ids
  // blockIdx.x, blockIdx.x + gridDim.x, ... in the order the dispatcher would have started them; the first K tile of
  // the next output tile is requested BEFORE the epilogue of the current one, so its HBM/L2 latency and the
  // workgroup launch disappear behind the store burst.
  const bool wm1 = (wave >> 2) == 1;
#ifdef BSG_DIAG_STAMPS
  const long long st_wg_start = __builtin_amdgcn_s_memtime(), st_wg_real = __builtin_amdgcn_s_memrealtime();
#endif
  bool primed = false;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
#ifdef BSG_DIAG_STAMPS
  const long long st_tile_start = __builtin_amdgcn_s_memtime();
#endif
  int bid = xcd_remap(vb, nwg);
  // groups of 4 row tiles: the blocks resident on one XCD at a time share 4 activation panels and a few weight panels
  const int GM = g.group_m > 0 ? g.group_m : 4;
  const int gsz = GM * tiles_n, grp = bid / gsz, rem = bid - grp * gsz;
  const int gm = min(GM, tiles_m - grp * GM);
  const int tm = grp * GM + rem % gm, tn = rem / gm;
  const int m0 = tm * TM, n0 = tn << 8;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- DMA sources: quarter q, instruction j = 2*wave + i (i = 0, 1) covers 8 rows
  const int prow = lane >> 3, pchunk = lane & 7;
  const char* src[4][2];
  int ldsoff[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = 2 * wave + i;
      int row;  // row inside the 256-row A (q = 0, 3) or W (q = 1, 2) tile
      if (q == 0) row = (j >> 3) * WR + (j & 7) * 8;             // A, mh0: rows wm'*WR + [0, 64)
      else if (q == 3) row = (j >> 3) * WR + 64 + min(j & 7, G1 - 1) * 8;  // A, mh1 (TM = 224: groups 6, 7 repeat group 5 --
                                                                           // every wave keeps 2 DMA per quarter for vmcnt)
      else if (q == 1) row = (j >> 2) * 64 + (j & 3) * 8;        // W, nh0: rows wn'*64 + [0, 32)
      else row = (j >> 2) * 64 + 32 + (j & 3) * 8;               // W, nh1
      const int r = row + prow;
      const int sc = pchunk ^ (r & 7);
      const bool is_a = (q == 0 || q == 3);
      ldsoff[q][i] = (is_a ? 0 : 32768) + row * 128;
      if (is_a) {
        int m = m0 + r;
        if (m >= g.M) m = g.M - 1;
        long base;
        if (AMODE == A_PLAIN) {
          base = ((long)(m / g.a_rpg) * g.a_gstride + (m % g.a_rpg)) * g.lda;
        } else {
          const int b = m / g.a_rpg, t = g.t_off + m % g.a_rpg;
          const int ph = t / g.wp, pw = t % g.wp;
          base = (((long)b * g.himg + ph * 16) * g.wimg + pw * 16) << g.feat_lg;
        }
        src[q][i] = (const char*)g.A + base * sizeof(T) + sc * 16;
      } else {
        int n = n0 + r;
        if (n >= g.N) n = g.N - 1;
        src[q][i] = (const char*)g.W + (long)n * g.K * sizeof(T) + sc * 16;
      }
    }
  auto issue = [&](int q, int kt, int buf) {
    const long k0 = (long)kt * BK;
    long koff = k0;
    if (AMODE == A_FEAT && (q == 0 || q == 3)) koff = (k0 >> (g.feat_lg + 4)) * ((long)g.wimg << g.feat_lg) + (k0 & ((16 << g.feat_lg) - 1));
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(src[q][i] + koff * sizeof(T), smem + buf * BUF + ldsoff[q][i]);
  };

  const int frow = lane & 15, fchunk = lane >> 4;
  f32x4 acc[2][4][4];  // [mh][ni = nh*2 + j][mi]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  if (!primed) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(q, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  if (wm == 1) asm volatile("s_barrier" ::: "memory");  // stagger: waves 4-7 run one barrier behind

  Chunk af[2][4], bf[2][2][2];
  auto read_a = [&](const char* la, int mh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (mh && i >= MH1) continue;
        const int r = wm * WR + mh * 64 + i * 16 + frow, c = fchunk + 4 * ks;
        af[ks][i] = *(const Chunk*)(la + r * 128 + ((c ^ (r & 7)) << 4));
      }
  };
  auto read_b = [&](const char* lw, int nh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + nh * 32 + j * 16 + frow, c = fchunk + 4 * ks;
        bf[nh][ks][j] = *(const Chunk*)(lw + r * 128 + ((c ^ (r & 7)) << 4));
      }
  };
  // X3: hi / lo halves of the ACTIVATION fragments, both k-steps of the K tile side by side (8 k-values per lane = one
  // `v_mfma_f32_16x16x32_f16` operand; same register count as the f32 chunks they replace).  The WEIGHT operand arrives
  // pre-split from the host: each 16-byte chunk of a weight row holds [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] of its four values
  // (same bytes as four floats), so a W fragment needs no conversion at all.
  f16x8 ah8[4], al8[4];
  auto split_a = [&](int i) {
    if constexpr (X3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = af[ks][i][e];
          const f16_t hi = (f16_t)x;
          ah8[i][4 * ks + e] = hi;
          al8[i][4 * ks + e] = (f16_t)(x - (float)hi);
        }
    }
  };
  auto mfma_quadrant = [&](int mh, int nh, int p) {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (X3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f16x8 w0 = __builtin_bit_cast(f16x8, bf[nh][0][j]), w1 = __builtin_bit_cast(f16x8, bf[nh][1][j]);
        const f16x8 wh = f16x8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        const f16x8 wl = f16x8{w0[4], w0[5], w0[6], w0[7], w1[4], w1[5], w1[6], w1[7]};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!mh || i < MH1) {
            f32x4& c = acc[mh][nh * 2 + j][i];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah8[i], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al8[i], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah8[i], c, 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (!mh || i < MH1) mma16(acc[mh][nh * 2 + j][i], bf[nh][ks][j], af[ks][i]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const char* la = smem + buf * BUF;
    const char* lw = la + 32768;
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p == 0) { read_b(lw, 0); read_a(la, 0); }
      else if (p == 1) read_b(lw, 1);
      else if (p == 2) read_a(la, 1);
      if (more) {
        issue(p, kt + 1, buf ^ 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (X3) {
        // split the A fragments this phase's read step fetched (p0: rows mh0, p2: rows mh1) HERE, in the load segment: the
        // partner wave of this SIMD is in its MFMA cluster meanwhile, so the ~100 conversion instructions run under its MFMAs
        // instead of in front of this wave's own (in the MFMA phase they cost 127 -> measured below)
        if (p == 0 || p == 2)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (p == 0 || i < MH1) split_a(i);
      }
      asm volatile("s_barrier" ::: "memory");
      mfma_quadrant(p >> 1, (p == 1 || p == 2) ? 1 : 0, p);
      asm volatile("s_barrier" ::: "memory");
    }
  }
  if (wm == 0) asm volatile("s_barrier" ::: "memory");  // balance the stagger
#ifdef BSG_DIAG_STAMPS
  const long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
  primed = false;
  if (vb + (int)gridDim.x < nwg && !(nk & 1)) {
    // every wave is past its last LDS read of buffer 0 (K tile nk - 2); request K tile 0 of the next output tile into it
    // through a second copy of the address set-up (the epilogue below still needs this tile's m0 / n0 only)
    const int nb = xcd_remap(vb + gridDim.x, nwg);
    const int ngrp = nb / gsz, nrem = nb - ngrp * gsz, ngm = min(GM, tiles_m - ngrp * GM);
    const int nm0 = (ngrp * GM + nrem % ngm) * TM, nn0 = (nrem / ngm) << 8;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = 2 * wave + i;
        int row;
        if (q == 0) row = (j >> 3) * WR + (j & 7) * 8;
        else if (q == 3) row = (j >> 3) * WR + 64 + min(j & 7, G1 - 1) * 8;
        else if (q == 1) row = (j >> 2) * 64 + (j & 3) * 8;
        else row = (j >> 2) * 64 + 32 + (j & 3) * 8;
        const int r = row + prow;
        const int sc = pchunk ^ (r & 7);
        const bool is_a = (q == 0 || q == 3);
        const char* sp;
        if (is_a) {
          int m = nm0 + r;
          if (m >= g.M) m = g.M - 1;
          long base;
          if (AMODE == A_PLAIN) {
            base = ((long)(m / g.a_rpg) * g.a_gstride + (m % g.a_rpg)) * g.lda;
          } else {
            const int bb = m / g.a_rpg, t = g.t_off + m % g.a_rpg;
            const int ph = t / g.wp, pw = t % g.wp;
            base = (((long)bb * g.himg + ph * 16) * g.wimg + pw * 16) << g.feat_lg;
          }
          sp = (const char*)g.A + base * sizeof(T) + sc * 16;
        } else {
          int n = nn0 + r;
          if (n >= g.N) n = g.N - 1;
          sp = (const char*)g.W + (long)n * g.K * sizeof(T) + sc * 16;
        }
        glds16(sp, smem + ldsoff[q][i]);
      }
    primed = true;
  }
  gemm_epilogue<T, EPI>(g, acc[0], m0 + wm * WR, n0 + wn * 64, frow, fchunk);
  gemm_epilogue<T, EPI, MH1>(g, acc[1], m0 + wm * WR + 64, n0 + wn * 64, frow, fchunk);
#ifdef BSG_DIAG_STAMPS
  {
    const long long st_issued = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long st_done = __builtin_amdgcn_s_memtime();
    if (vb == (int)blockIdx.x + (int)gridDim.x && tid == 0 && blockIdx.x < 256) {  // second tile of each workgroup (steady state)
      bsg_stamps[blockIdx.x * 4 + 0] = st_loop_end - st_tile_start;
      bsg_stamps[blockIdx.x * 4 + 1] = st_issued - st_loop_end;
      bsg_stamps[blockIdx.x * 4 + 2] = st_done - st_issued;
    }
  }
#endif
  (void)wm1;
  }  // tile loop
#ifdef BSG_DIAG_STAMPS
  if (tid == 0 && blockIdx.x < 256)  // shader cycles per 100 MHz reference tick over the workgroup's lifetime, x 1000
    bsg_stamps[blockIdx.x * 4 + 3] = (__builtin_amdgcn_s_memtime() - st_wg_start) * 1000 / max(1LL, (long long)(__builtin_amdgcn_s_memrealtime() - st_wg_real));
#endif
}

// MFMA on accumulators pinned in the accumulator half of the register file (v4 / v5 experiment kernels)
DEVI void mfma16_agpr(f32x4& acc, const f32x4& a, const f32x4& b, bf16_t) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// an output tile's very last MFMA carries the settling nops in the SAME asm statement: nothing hipcc schedules can then land
// between it and them (the seven MFMAs before it finish under it)
DEVI void mfma16_agpr_settle(f32x4& acc, const f32x4& a, const f32x4& b, bf16_t) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+a"(acc) : "v"(a), "v"(b));
}
DEVI void mfma16_agpr_settle(f32x4& acc, const f32x4& a, const f32x4& b, f16_t) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+a"(acc) : "v"(a), "v"(b));
}
DEVI void mfma16_agpr(f32x4& acc, const f32x4& a, const f32x4& b, f16_t) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// v5's fp32 residual epilogue (out = acc + bias + aux, fp32, usually in place): the generic path's arithmetic with the same
// scalar addressing as gemm_epilogue_v5_wide -- the aux request of sub-tile q + 1 and the stores of sub-tile q then need no
// address registers, and the two 64-register aux buffers + the sub-tile fit without the scratch traffic the generic form had.
DEVI void gemm_v5_resid_load(const GemmArgs& g, f32x4 (&ax)[4][4], int mw, int nw, unsigned vo) {
  const char* ab = (const char*)g.aux + ((long)mw * g.ldaux + nw) * 4;
  const long rstep = 16L * g.ldaux * 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) ax[ni][mi] = *(const f32x4*)(ab + mi * rstep + ni * 64 + vo);
}
DEVI void gemm_v5_resid_store(const GemmArgs& g, f32x4 (&acc)[4][4], int mw, int nw, unsigned vo, const f32x4* bv, const f32x4 (&ax)[4][4]) {
  char* ob = (char*)g.out + ((long)mw * g.ldo + nw) * 4;
  const long rstep = 16L * g.ldo * 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) *(f32x4*)(ob + mi * rstep + ni * 64 + vo) = (acc[ni][mi] + bv[ni]) + ax[ni][mi];
}

// v5's epilogue for the row-major 16-bit outputs (plain / bias / bias + GELU with the saved derivative / gradient x saved
// gelu'): gemm_epilogue_wide16's arithmetic and store pattern (identical bits), minus everything a whole, in-range 256 x 256
// tile does not need -- no bounds masks, and the address of a store is a wave-uniform pointer (tile, sub-tile, 16-row block:
// scalar arithmetic) plus ONE per-lane 32-bit offset that is constant for the kernel (global_store saddr form), where the
// generic path spends ~10 vector instructions and a masked branch per store.
template <typename T, int EPI>
DEVI void gemm_epilogue_v5_wide(const GemmArgs& g, f32x4 (&acc)[4][4], int mw, int nw, unsigned vo, const f32x4* bv,
                                const f32x4 (*ax)[4]) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  char* ob = (char*)g.out + ((long)mw * g.ldo + nw) * sizeof(T);
  char* ob2 = (EPI == EPI_BIAS_GELU && g.out2) ? (char*)g.out2 + ((long)mw * g.ldo + nw) * sizeof(T) : nullptr;
  const long rstep = 16L * g.ldo * sizeof(T);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    unsigned lo[4], hi[4], lo2[4], hi2[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      f32x4 v = acc[ni][mi];
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD) v += bv[ni];
      if (EPI == EPI_BIAS_GELU_FWD) {
        const f32x2 y0 = gelu_pk(f32x2{v[0], v[1]}), y1 = gelu_pk(f32x2{v[2], v[3]});
        v = f32x4{y0[0], y0[1], y1[0], y1[1]};
      } else if (EPI == EPI_BIAS_GELU) {
        f32x4 y, dy;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2 yy, dd;
          gelu_both_pk(f32x2{v[r], v[r + 1]}, yy, dd);
          y[r] = yy[0]; y[r + 1] = yy[1]; dy[r] = dd[0]; dy[r + 1] = dd[1];
        }
        lo2[ni] = pack2<T>(dy[0], dy[1]);
        hi2[ni] = pack2<T>(dy[2], dy[3]);
        v = y;
      } else if (EPI == EPI_GELU_BWD) {
        typedef typename Traits<T>::Vec4 HV;
        const HV hp = __builtin_bit_cast(HV, f32x2{ax[ni][mi][0], ax[ni][mi][1]});
        v = f32x4{v[0] * (float)hp[0], v[1] * (float)hp[1], v[2] * (float)hp[2], v[3] * (float)hp[3]};
      }
      lo[ni] = pack2<T>(v[0], v[1]);
      hi[ni] = pack2<T>(v[2], v[3]);
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int ia = 2 * pr, ib = 2 * pr + 1;
      auto r0 = __builtin_amdgcn_permlane16_swap(lo[ia], lo[ib], false, false);
      auto r1 = __builtin_amdgcn_permlane16_swap(hi[ia], hi[ib], false, false);
      *(u32x4*)(ob + mi * rstep + pr * 64 + vo) = u32x4{r0[0], r1[0], r0[1], r1[1]};
      if (EPI == EPI_BIAS_GELU) {
        auto q0 = __builtin_amdgcn_permlane16_swap(lo2[ia], lo2[ib], false, false);
        auto q1 = __builtin_amdgcn_permlane16_swap(hi2[ia], hi2[ib], false, false);
        if (ob2) *(u32x4*)(ob2 + mi * rstep + pr * 64 + vo) = u32x4{q0[0], q1[0], q0[1], q1[1]};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// v5 (16-bit dtypes, plain rows, M and N multiples of 256, an even number of K tiles): the v4 tile shape (four waves, 128 x 128
// wave tiles, 256 accumulators pinned in the accumulator file) with the operand pipeline of a two-tiles-ahead LDS-DMA stream:
// buffer_load ... lds through an SGPR descriptor (the K advance is two scalar adds, no vector address math), waited for with
// COUNTED vmcnt one tile later; both k-steps' fragments of both operands live in registers (128 VGPRs), read a full phase
// (64 MFMAs) ahead, so the MFMA stream never waits on LDS except at the four barriers of a K tile:
//   #1 every wave has read all of tile kt's A     -> A pieces of tile kt+2 may overwrite it
//   #2 ... all of tile kt's W                     -> W pieces of tile kt+2
//   #3 tile kt+1's A pieces have landed (vmcnt)   -> A fragments (k-step 0) of tile kt+1
//   #4 tile kt+1's W pieces have landed           -> W fragments (k-step 0) of tile kt+1
typedef int i32x4_t __attribute__((ext_vector_type(4)));
DEVI i32x4_t raw_buffer_desc(const char* base_uniform) {
  const unsigned long v = (unsigned long)base_uniform;
  i32x4_t d;
  d[0] = (int)(unsigned)v; d[1] = (int)(unsigned)((v >> 32) & 0xffffu); d[2] = -1; d[3] = 0x00020000;
  return d;
}
// one LDS-DMA piece: lane l's 16 bytes at desc.base + voff land at LDS byte lds_uniform + 16 l
DEVI void dma16_buf(unsigned lds_uniform, unsigned voff, const i32x4_t& desc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_uniform), "v"(voff), "s"(desc) : "memory");
}
template <int N> DEVI void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
constexpr int v5_find(const int (&tab)[8], int n) {
  for (int i = 0; i < 8; ++i)
    if (tab[i] == n) return i;
  return -1;
}
// positions (index of the MFMA an action follows, 0..127 within a K tile)
constexpr int V5_A1[8] = {0, 3, 5, 7, 9, 11, 13, 15};             // A fragments, k-step 1
constexpr int V5_W1[8] = {24, 27, 30, 33, 36, 39, 41, 43};        // W fragments, k-step 1
constexpr int V5_A0[8] = {68, 70, 72, 74, 76, 78, 80, 82};        // A fragments, k-step 0 of the next tile
constexpr int V5_W0[8] = {105, 107, 109, 111, 113, 115, 117, 119};  // W fragments, k-step 0 of the next tile
// (measured: the same pieces spread one per 5 / 8 MFMAs +0.3 %; bunched right behind barriers #1 / #2 -7 %)
constexpr int V5_DA[8] = {22, 25, 28, 31, 34, 52, 55, 58};        // A pieces of tile kt + 2
constexpr int V5_DW[8] = {61, 64, 84, 86, 88, 94, 99, 123};       // W pieces of tile kt + 2
constexpr int V5_VMA = 18, V5_VMW = 15;  // pieces of this round issued before the waits at MFMA 66 / 103, plus the 8 / 0 of last round's that may still fly

template <typename T, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_kernel_v5(GemmArgs g) {
  static_assert(sizeof(T) == 2, "v5: 16-bit operands");
  constexpr int BK = 64, BUF = 65536;  // per stage: A 256 x 128 B, then W 256 x 128 B
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = g.N >> 8, tiles_m = g.M >> 8;
  const int nwg = tiles_m * tiles_n;
  const int wm = wave >> 1, wn = wave & 1, frow = lane & 15, fchunk = lane >> 4;
  const unsigned base = lds_addr(smem);
  // fragment rows: wm*128 + i*16 + frow (i*16 rows = +2048 B immediates); chunk (fchunk + 4 ks) ^ (row & 7), row & 7 = frow & 7
  unsigned aA[2][2], aW[2][2];  // [stage][k-step]
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aA[b][ks] = (base + b * BUF + (wm * 128 + frow) * 128 + ((fchunk ^ (frow & 7)) << 4)) ^ (ks * 64);
      aW[b][ks] = (base + b * BUF + 32768 + (wn * 128 + frow) * 128 + ((fchunk ^ (frow & 7)) << 4)) ^ (ks * 64);
    }
  const int nk = g.K / BK;
  // DMA piece i of this wave: rows wave*64 + i*8 + prow of the A (W) tile; lane (prow, pchunk) fetches source chunk
  // pchunk ^ prow of its row, which lands at chunk position pchunk (source-side swizzle)
  const int prow = lane >> 3, pchunk = lane & 7;
  unsigned voA[8], voW[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    voA[i] = (unsigned)((i * 8 + prow) * (long)g.lda * sizeof(T)) + ((pchunk ^ prow) << 4);
    voW[i] = (unsigned)((i * 8 + prow) * (long)g.K * sizeof(T)) + ((pchunk ^ prow) << 4);
  }
  const unsigned ldsA = __builtin_amdgcn_readfirstlane(base + wave * 64 * 128), ldsW = ldsA + 32768;
  // per-lane byte offset of the epilogue's 16-byte stores inside a 16-row block (after the lane-row exchange a lane owns 8
  // consecutive columns: column block fchunk & 1, half fchunk >> 1)
  const unsigned vo_out = (unsigned)((frow * (long)g.ldo + (fchunk & 1) * 16 + (fchunk >> 1) * 8) * sizeof(T));
  const unsigned vo_r_out = (unsigned)((frow * (long)g.ldo + 4 * fchunk) * 4), vo_r_aux = (unsigned)((frow * (long)g.ldaux + 4 * fchunk) * 4);  // fp32 residual form

  const int GM = g.group_m > 0 ? g.group_m : 4;
  auto coords = [&](int vb, int& m0, int& n0) {
    const int bid = xcd_remap(vb, nwg);
    const int gsz = GM * tiles_n, grp = bid / gsz, rem = bid - grp * gsz;
    const int gm = min(GM, tiles_m - grp * GM);
    m0 = (grp * GM + rem % gm) << 8;
    n0 = (rem / gm) << 8;
  };
  const char *pa = nullptr, *pw = nullptr;
  int kload = 0;  // K tile the next DMA round fetches (clamped to the last one: past the end the stream re-loads it, unread)
  auto dma_a = [&](auto ii, int stage) {
    constexpr int I = decltype(ii)::value;
    dma16_buf(ldsA + stage * BUF + I * 1024, voA[I], raw_buffer_desc(pa + (long)kload * 128));
  };
  auto dma_w = [&](auto ii, int stage) {
    constexpr int I = decltype(ii)::value;
    dma16_buf(ldsW + stage * BUF + I * 1024, voW[I], raw_buffer_desc(pw + (long)kload * 128));
  };
  auto advance = [&]() { kload = min(kload + 1, nk - 1); };
  // fill: K tiles 0 and 1 of output tile (m0, n0) requested into stages 0 and 1
  auto fill = [&](int m0, int n0) {
    pa = uniform_ptr((const char*)g.A + (long)(m0 + wave * 64) * g.lda * sizeof(T));
    pw = uniform_ptr((const char*)g.W + (long)(n0 + wave * 64) * g.K * sizeof(T));
    kload = 0;
    static_for<0, 8>([&](auto i) { dma_a(i, 0); });
    static_for<0, 8>([&](auto i) { dma_w(i, 0); });
    advance();
    static_for<0, 8>([&](auto i) { dma_a(i, 1); });
    static_for<0, 8>([&](auto i) { dma_w(i, 1); });
    advance();
  };
  bool primed = false;
#ifdef BSG_DIAG_STAMPS
  const long long st_wg_start = __builtin_amdgcn_s_memtime(), st_wg_real = __builtin_amdgcn_s_memrealtime();
#endif
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
#ifdef BSG_DIAG_STAMPS
    const long long st_tile_start = __builtin_amdgcn_s_memtime();
#endif
    int m0, n0;
    coords(vb, m0, n0);
    f32x4 acc[8][8];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2][8], fw[2][8];  // [k-step][row block]
    // the wave tile's eight bias vectors arrive under the K loop (inside the epilogue every re-load behind a store would
    // wait for L2 with nothing else on the SIMD)
    constexpr bool kBias = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD || EPI == EPI_BIAS_RESID || EPI == EPI_FEAT;
    f32x4 bias8[8];
    if constexpr (kBias) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bias8[i] = *(const f32x4*)(g.bias + min(n0 + wn * 128 + i * 16 + 4 * fchunk, g.N - 4));
    }

    if (!primed) {
      fill(m0, n0);
      vm_wait<16>();
    } else {
      vm_wait<0>();  // the fill went out before the previous tile's epilogue: behind its stores nothing is countable
    }
    __builtin_amdgcn_s_barrier();
    static_for<0, 8>([&](auto r) { fa[0][decltype(r)::value] = lds_read16_nw<decltype(r)::value * 2048>(aA[0][0]); });
    static_for<0, 8>([&](auto r) { fw[0][decltype(r)::value] = lds_read16_nw<decltype(r)::value * 2048>(aW[0][0]); });
    lds_wait<0>();

    constexpr bool kAux = EPI == EPI_BIAS_RESID || EPI == EPI_GELU_BWD;
    constexpr bool kAuxEarly = EPI == EPI_GELU_BWD;  // the fp32 residual (16 x 16-byte loads + their addresses in the MFMA stream): slower
    f32x4 ax[2][4][4];
    auto aux_load = [&](f32x4 (&a)[4][4], int mw, int nw) {
      if constexpr (EPI == EPI_BIAS_RESID) gemm_v5_resid_load(g, a, mw, nw, vo_r_aux);
      else gemm_epilogue_aux_load<T, EPI>(g, a, mw, nw, frow, fchunk);
    };
    // DMA: this K tile requests tile kt + 2 (false for the last two of an output tile); NEXT: it reads tile kt + 1's k-step-0
    // fragments (false for the last one, which instead requests the first sub-tile's epilogue operand under its MFMAs)
    auto ktile = [&](auto pp, auto dd, auto xx) {
      constexpr int P = decltype(pp)::value;  // stage of tile kt; tile kt + 1 sits in stage P ^ 1, tile kt + 2 goes to stage P
      constexpr bool DMA = decltype(dd)::value, NEXT = decltype(xx)::value;
      static_for<0, 128>([&](auto nn) {
        constexpr int n = decltype(nn)::value, ph = n >> 6, ni = (n & 63) >> 3, mi = n & 7;
        if constexpr (!NEXT && n == 127) mfma16_agpr_settle(acc[ni][mi], fw[ph][ni], fa[ph][mi], T());
        else mfma16_agpr(acc[ni][mi], fw[ph][ni], fa[ph][mi], T());
        constexpr int ra1 = v5_find(V5_A1, n), rw1 = v5_find(V5_W1, n), ra0 = v5_find(V5_A0, n), rw0 = v5_find(V5_W0, n);
        constexpr int da = v5_find(V5_DA, n), dw = v5_find(V5_DW, n);
        if constexpr (ra1 >= 0) fa[1][ra1] = lds_read16_nw<ra1 * 2048>(aA[P][1]);
        if constexpr (rw1 >= 0) fw[1][rw1] = lds_read16_nw<rw1 * 2048>(aW[P][1]);
        if constexpr (NEXT && ra0 >= 0) fa[0][ra0] = lds_read16_nw<ra0 * 2048>(aA[P ^ 1][0]);
        if constexpr (NEXT && rw0 >= 0) fw[0][rw0] = lds_read16_nw<rw0 * 2048>(aW[P ^ 1][0]);
        if constexpr (DMA && da >= 0) dma_a(std::integral_constant<int, da>{}, P);
        if constexpr (DMA && dw >= 0) dma_w(std::integral_constant<int, dw>{}, P);
        if constexpr (n == 19 || n == 50) lds_wait<0>();                          // k-step-1 fragments are in (phase 2 uses them)
        if constexpr (DMA && (n == 20 || n == 51)) __builtin_amdgcn_s_barrier();   // #1, #2
        if constexpr (NEXT && n == 66) vm_wait<(DMA ? V5_VMA : 8)>();              // tile kt + 1's 8 A pieces are in
        if constexpr (NEXT && n == 103) vm_wait<(DMA ? V5_VMW : 0)>();             // all of tile kt + 1
        if constexpr (NEXT && (n == 67 || n == 104)) __builtin_amdgcn_s_barrier();  // #3, #4
        if constexpr (!NEXT && kAuxEarly && n == 4) aux_load(ax[0], m0 + wm * 128, n0 + wn * 128);
      });
      if constexpr (DMA) advance();
      lds_wait<0>();  // next tile's k-step-0 fragments are in before the loop edge (and before any register copy there)
    };
    typedef std::true_type Yes;
    typedef std::false_type No;
    for (int kt = 0; kt < nk - 2; kt += 2) {
      ktile(std::integral_constant<int, 0>{}, Yes{}, Yes{});
      ktile(std::integral_constant<int, 1>{}, Yes{}, Yes{});
    }
    ktile(std::integral_constant<int, 0>{}, No{}, Yes{});
    ktile(std::integral_constant<int, 1>{}, No{}, No{});
    // MFMA results settle before anything reads the accumulators (see v4)
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]),
                   "+a"(acc[7][6]), "+a"(acc[7][7])
                 :
                 : "memory");
    __builtin_amdgcn_s_barrier();  // no wave's fill DMA of the next output tile lands under another wave's last fragment reads
#ifdef BSG_DIAG_STAMPS
    const long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (kAux && !kAuxEarly) aux_load(ax[0], m0 + wm * 128, n0 + wn * 128);
    // plain / bias / GELU epilogues: the four 64 x 64 sub-tiles go through ONE copy of the epilogue code (a rolled loop; the
    // accumulators are picked by a switch) -- unrolled, the GELU epilogue alone is 135 KB of instructions against a 64 KB
    // instruction cache (same box, ms per step, rolled vs unrolled: QKV 14.0 vs 14.5, fc1 + GELU 25.2 vs 26.3, plain dgrads
    // 28.4 vs 29.8).  The aux-operand epilogues keep the unrolled form: their double-buffered prefetch needs static register
    // names (rolled two by two, the buffers went to scratch: 263 vs 29 ms), and the pixel-shuffle one measured 9.8 vs 9.1.
    constexpr bool kRolled = EPI == EPI_PLAIN || EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD;
    if constexpr (kRolled) {
#pragma nounroll
      for (int q = 0; q < 4; ++q) {
        f32x4 sub[4][4], bv[4];
        static_for<0, 4>([&](auto qq) {
          constexpr int Q = decltype(qq)::value;
          if (q == Q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
              for (int j = 0; j < 4; ++j) sub[i][j] = acc[(Q >> 1) * 4 + i][(Q & 1) * 4 + j];
              if constexpr (kBias) bv[i] = bias8[(Q >> 1) * 4 + i];
            }
          }
        });
        gemm_epilogue_v5_wide<T, EPI>(g, sub, m0 + wm * 128 + (q & 1) * 64, n0 + wn * 128 + (q >> 1) * 64, vo_out, bv, nullptr);
      }
    } else {
      static_for<0, 4>([&](auto qq) {
        constexpr int q = decltype(qq)::value, qn = q >> 1, qm = q & 1;
        if constexpr (kAux && q < 3) aux_load(ax[(q + 1) & 1], m0 + wm * 128 + ((q + 1) & 1) * 64, n0 + wn * 128 + ((q + 1) >> 1) * 64);
        f32x4 sub[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) sub[i][j] = acc[qn * 4 + i][qm * 4 + j];
        if constexpr (EPI == EPI_GELU_BWD)
          gemm_epilogue_v5_wide<T, EPI>(g, sub, m0 + wm * 128 + qm * 64, n0 + wn * 128 + qn * 64, vo_out, nullptr, ax[q & 1]);
        else if constexpr (EPI == EPI_BIAS_RESID)
          gemm_v5_resid_store(g, sub, m0 + wm * 128 + qm * 64, n0 + wn * 128 + qn * 64, vo_r_out, &bias8[qn * 4], ax[q & 1]);
        else
          gemm_epilogue<T, EPI>(g, sub, m0 + wm * 128 + qm * 64, n0 + wn * 128 + qn * 64, frow, fchunk, kAux ? ax[q & 1] : nullptr,
                                kBias ? &bias8[qn * 4] : nullptr);
      });
    }
#ifdef BSG_DIAG_STAMPS
    {
      const long long st_issued = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long st_done = __builtin_amdgcn_s_memtime();
      if (vb == (int)blockIdx.x + (int)gridDim.x && tid == 0 && blockIdx.x < 256) {  // second tile of each workgroup (steady state)
        bsg_stamps[blockIdx.x * 4 + 0] = st_loop_end - st_tile_start;
        bsg_stamps[blockIdx.x * 4 + 1] = st_issued - st_loop_end;
        bsg_stamps[blockIdx.x * 4 + 2] = st_done - st_issued;
      }
    }
#endif
  }
#ifdef BSG_DIAG_STAMPS
  if (tid == 0 && blockIdx.x < 256)
    bsg_stamps[blockIdx.x * 4 + 3] = (__builtin_amdgcn_s_memtime() - st_wg_start) * 1000 / max(1LL, (long long)(__builtin_amdgcn_s_memrealtime() - st_wg_real));
#endif
}

// Where v5 is the default (BSG_GEMM unset or 5; 3 = never).  Same-box kernel trace of
// the B = 64 train step, ms per step, v5 vs v3: QKV (bias) 14.5 vs 16.7, fc1 + GELU (two outputs) 24.5 vs 27.0, proj / fc2 +
// fp32 residual 26.1 vs 28.3, plain dgrads 28.3 vs 31.0, dfc2 * gelu' 20.2 vs 21.7, decoder embed 9.2 vs 10.2; the two small
// one-per-step epilogues (patch-embed side) measured 1.04 vs 0.97 and 0.27 vs 0.25 and stay with v3.  With one wave per SIMD
// nothing runs under a wave's epilogue, so v5's first form lost the epilogue-heavy GEMMs; what turned them: bias vectors
// requested before the K loop, the aux operand of sub-tile q + 1 requested before sub-tile q is stored (gelu': the first one
// under the last K tile), one rolled copy of the epilogue code, scalar store addressing without bounds masks, and the GELU on
// element pairs (a lone wave issues one vector instruction per 4 cycles whatever its width).
// v5 is INSTANTIATED for these epilogues only: its MFMAs are asm statements hipcc cannot see through, and in the pixel-unshuffle
// variant it scheduled accumulator reads between the last MFMA and the settling nops (tests/test_kernel_isa.py checks the
// compiled code of every instance for that hazard).
template <int EPI> constexpr bool gemm_v5_pick() { return EPI != EPI_EMBED && EPI != EPI_UNPATCH && EPI != EPI_NONE; }
template <int AMODE, int EPI = EPI_PLAIN> static inline bool gemm_v5_ok(const GemmArgs& g, size_t es) {
  return AMODE == A_PLAIN && es == 2 && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 128 == 0 && g.a_rpg >= g.M && (g.o_rpg == 0 || EPI == EPI_FEAT) &&
         (long)g.ldo * 4 * 16 < (1L << 31) && (long)g.ldaux * 4 * 16 < (1L << 31) &&
         (long)g.lda * 2 * 64 < (1L << 31) && (long)g.K * 2 * 64 < (1L << 31);
}

template <typename T, int AMODE, int EPI>
static inline void launch_gemm(const GemmArgs& g, hipStream_t st) {
  static const int ver = getenv("BSG_GEMM") ? atoi(getenv("BSG_GEMM")) : 5;  // 5: v5 for the epilogues of gemm_v5_pick where its addressing applies, else v3; 3: v3 only (the one A/B switch left)
  const int tn3 = (g.N + 255) / 256, grid3 = std::min(((g.M + 255) / 256) * tn3, 256);
  if constexpr (sizeof(T) == 4) {
    if (g.x3) {  // pre-split (hi | lo) weights: ONLY the X3 instance of v3 reads that format, whatever BSG_GEMM says
      hipLaunchKernelGGL((gemm_nt_kernel_v3<T, AMODE, EPI, 256, true>), dim3(grid3), dim3(512), 131072, st, g);
      return;
    }
  }
  if (ver == 1) {
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    hipLaunchKernelGGL((gemm_nt_kernel<T, AMODE, EPI>), dim3(tiles), dim3(256), 65536, st, g);
  } else if (ver == 2 || g.N <= 192) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + 127) / 128);
    hipLaunchKernelGGL((gemm_nt_kernel_v2<T, AMODE, EPI>), dim3(tiles), dim3(512), 3 * 49152, st, g);
  } else if (ver >= 5 && gemm_v5_pick<EPI>() && gemm_v5_ok<AMODE, EPI>(g, sizeof(T))) {
    if constexpr (sizeof(T) == 2 && AMODE == A_PLAIN && gemm_v5_pick<EPI>()) {
      const int tiles = (g.M / 256) * (g.N / 256);
      hipLaunchKernelGGL((gemm_nt_kernel_v5<T, EPI>), dim3(std::min(tiles, 256)), dim3(256), 131072, st, g);
    }
  } else {
    // persistent: one workgroup per CU walks the tiles (one workgroup per tile / 512 workgroups: measured slower)
    hipLaunchKernelGGL((gemm_nt_kernel_v3<T, AMODE, EPI>), dim3(grid3), dim3(512), 131072, st, g);
  }
}
