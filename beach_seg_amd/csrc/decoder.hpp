// SegGPT decoder head on gfx950 (HF:modeling_seggpt.py:525-546): direct 3x3 convolution C->C (C = decoder_hidden_size: 64 in
// the reference checkpoint, 128 in BASELINE config 5) over the NHWC
// feature map (no im2col: nine shifted NT-GEMM steps over an LDS halo tile), fused in ONE pass with bias,
// per-pixel LayerNorm(C) (wavefront-lane reductions), exact GELU and the 1x1 head (64->3), and the matching
// backward: a per-pixel kernel through head/GELU/LayerNorm and the same conv kernel with flipped weights for dgrad.
#pragma once
#include "common.hpp"

enum ConvMode { CONV_FWD_FUSED = 0, CONV_PLAIN = 1 };

struct ConvArgs {
  const void* in;     // T NHWC [B][H][W][DC], DC = decoder_hidden_size (64 or 128)
  const void* w;      // T [DC co][9 taps][DC ci]
  const float* bias;  // [DC] or nullptr (CONV_PLAIN)
  void* out;          // T NHWC: conv output (pre-LN) in FUSED mode (may be nullptr), plain output otherwise
  const float* ln_g; const float* ln_b;  // [DC]
  const float* head_w;                   // [3][DC]
  const float* head_b;                   // [3]
  float* pred;                           // fp32 NCHW [B][3][H][W]
  int H, W;
  float eps;
  int ty0;  // first output row / 16 (dgrad over a row window: token rows)
};

// Workgroup = TR x 32 output pixels; wave w owns rows RW*w .. RW*w + RW-1 x all DC output channels (DC/16 x 2 RW MFMA 16x16
// tiles): every weight fragment fetched (L1/L2: the filter bank is re-read by every wave) feeds 2 RW MFMAs.  DC = 64 (the
// reference checkpoint, HF:configuration_seggpt.py decoder_hidden_size): TR = 16, RW = 4, one pass over the 64 input channels.
// DC = 128 (BASELINE config 5): TR = 8, RW = 2 (the 8 x 4 accumulator tiles fill 128 registers) and the contraction runs in
// two passes of 64 input channels through the same 128-byte-per-pixel halo tile.  CONV_TR (16 rows = one token row) is the
// unit of `ty0` for every variant.
constexpr int CONV_TR = 16;
template <int DC> struct ConvGeo {
  static_assert(DC == 64 || DC == 128, "decoder width: 64 or 128 channels");
  static constexpr int RW = DC == 64 ? 4 : 2, TR = 4 * RW, HALO = (TR + 2) * 34, NO = DC / 16, HALVES = DC / 64;
};
// X3 (T = float only, bsg_config.gemm_x3): "float32 at three f16 MFMAs" as in gemm_nt_kernel_v3<..., X3> -- two f32 k-steps
// (2 x 16 channels) side by side form one 32-deep f16 operand, each fragment split hi = f16(x), lo = f16(x - hi); hi*hi + hi*lo +
// lo*hi on v_mfma_f32_16x16x32_f16 (3 x 16 cycles) in place of eight v_mfma_f32_16x16x4_f32 (8 x 32).  Weights are split
// in-kernel, x 2^5 so that the lo halves of sigma = 0.02 filters stay normal f16 numbers; the accumulator is scaled back exactly.
DEVI void x3_split8(const f32x4& x0, const f32x4& x1, float s, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a0 = x0[e] * s, a1 = x1[e] * s;
    const f16_t h0 = (f16_t)a0, h1 = (f16_t)a1;
    hi[e] = h0; hi[4 + e] = h1;
    lo[e] = (f16_t)(a0 - (float)h0); lo[4 + e] = (f16_t)(a1 - (float)h1);
  }
}
template <typename T, int MODE, int DC, bool X3 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs a) {
  static_assert(!X3 || sizeof(T) == 4, "the three-term f16 split is the float32 kernel's option");
  typedef typename Traits<T>::Chunk Chunk;
  typedef ConvGeo<DC> G;
  constexpr int EPC = Traits<T>::EPC, CPP = 64 / EPC;  // chunks per pixel of one 64-channel pass: 8 (16-bit) / 16 (f32)
  constexpr int PB = 64 * sizeof(T);                   // LDS bytes per pixel (one pass)
  constexpr int GPB = DC * sizeof(T);                  // global bytes per pixel
  constexpr int KS = CPP / 4;                          // 16x16 k-steps per tap and pass
  constexpr int NM = 2 * G::RW;                        // pixel tiles per wave: (row, x half)
  constexpr int NO = G::NO;                            // output-channel tiles
  extern __shared__ __attribute__((aligned(16))) char halo[];  // [TR + 2][34] pixels x PB, chunk-swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = blockIdx.x * 32, y0 = a.ty0 * CONV_TR + blockIdx.y * G::TR, b = blockIdx.z;
  const char* img = (const char*)a.in + (long)b * a.H * a.W * GPB;

  const int frow = lane & 15, fchunk = lane >> 4;
  f32x4 acc[NO][NM];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < NO; ++i)
#pragma unroll
    for (int j = 0; j < NM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int half = 0; half < G::HALVES; ++half) {
    if (half) __syncthreads();  // every wave is done with the previous pass's halo
    for (int i = tid; i < G::HALO * CPP; i += 256) {
      const int pix = i / CPP, ch = i % CPP;
      const int hy = pix / 34, hx = pix % 34, y = y0 + hy - 1, x = x0 + hx - 1;
      Chunk v;
#pragma unroll
      for (int j = 0; j < EPC; ++j) v[j] = from_f32<T>(0.f);
      if (y >= 0 && y < a.H && x >= 0 && x < a.W) v = *(const Chunk*)(img + ((long)y * a.W + x) * GPB + half * PB + ch * 16);
      *(Chunk*)(halo + pix * PB + ((ch ^ (pix & (CPP - 1))) << 4)) = v;
    }
    __syncthreads();

#pragma unroll 1  // rolled: with 128 accumulator registers a fully unrolled tap loop hoists operand loads into spills
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - 3 * dy;
      if constexpr (X3) {
#pragma unroll
        for (int kp = 0; kp < KS / 2; ++kp) {
          const int c0 = fchunk + 8 * kp, c1 = c0 + 4;
#pragma unroll
          for (int n0 = 0; n0 < NO; n0 += 4) {  // four output-channel tiles at a time: their split filters are 64 registers
            f16x8 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const char* wp = (const char*)a.w + (((long)((n0 + i) * 16 + frow) * 9 + tap) * DC + half * 64) * sizeof(T);
              x3_split8(*(const f32x4*)(wp + c0 * 16), *(const f32x4*)(wp + c1 * 16), 32.0f, wh[i], wl[i]);
            }
#pragma unroll
            for (int mi = 0; mi < NM; ++mi) {
              const int pix = (G::RW * wave + (mi >> 1) + dy) * 34 + (mi & 1) * 16 + frow + dx;
              f16x8 ah, al;
              x3_split8(*(const f32x4*)(halo + pix * PB + ((c0 ^ (pix & (CPP - 1))) << 4)),
                        *(const f32x4*)(halo + pix * PB + ((c1 ^ (pix & (CPP - 1))) << 4)), 1.0f, ah, al);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                mma16(acc[n0 + i][mi], wh[i], ah);
                mma16(acc[n0 + i][mi], wh[i], al);
                mma16(acc[n0 + i][mi], wl[i], ah);
              }
              __builtin_amdgcn_sched_barrier(0);  // (as in attention.hpp mma32_x3: keeps hipcc from hoisting all the splits)
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = fchunk + 4 * ks;
        Chunk fa[NM], fw[NO];
#pragma unroll
        for (int i = 0; i < NO; ++i)
          fw[i] = *(const Chunk*)((const char*)a.w + (((long)(i * 16 + frow) * 9 + tap) * DC + half * 64) * sizeof(T) + c * 16);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          const int pix = (G::RW * wave + (i >> 1) + dy) * 34 + (i & 1) * 16 + frow + dx;
          fa[i] = *(const Chunk*)(halo + pix * PB + ((c ^ (pix & (CPP - 1))) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < NO; ++ni)
#pragma unroll
          for (int mi = 0; mi < NM; ++mi) mma16(acc[ni][mi], fw[ni], fa[mi]);
      }
    }
  }

  if constexpr (X3) {  // the filters were multiplied by 2^5
#pragma unroll
    for (int i = 0; i < NO; ++i)
#pragma unroll
      for (int j = 0; j < NM; ++j) acc[i][j] *= 0.03125f;
  }
  // acc[ni][mi][r]: pixel (y0 + RW*wave + (mi>>1), x0 + (mi&1)*16 + frow), channel ni*16 + 4*fchunk + r
#pragma unroll
  for (int mi = 0; mi < NM; ++mi) {
    const int y = y0 + G::RW * wave + (mi >> 1), x = x0 + (mi & 1) * 16 + frow;
    const long pix = ((long)b * a.H + y) * a.W + x;
    if (MODE == CONV_PLAIN) {
#pragma unroll
      for (int ni = 0; ni < NO; ++ni) {
        const f32x4 v = acc[ni][mi];
        *(typename Traits<T>::Vec4*)((T*)a.out + pix * DC + ni * 16 + 4 * fchunk) = pack4<T>(v[0], v[1], v[2], v[3]);
      }
    } else {
      f32x4 v[NO];
      float s = 0.f;
#pragma unroll
      for (int ni = 0; ni < NO; ++ni) {
        v[ni] = acc[ni][mi] + *(const f32x4*)(a.bias + ni * 16 + 4 * fchunk);
        if (a.out)
          *(typename Traits<T>::Vec4*)((T*)a.out + pix * DC + ni * 16 + 4 * fchunk) =
              pack4<T>(v[ni][0], v[ni][1], v[ni][2], v[ni][3]);
        s += v[ni][0] + v[ni][1] + v[ni][2] + v[ni][3];
      }
      // the DC channels of a pixel live in the 4 lanes {frow, frow+16, frow+32, frow+48}
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.f / DC);
      float q = 0.f;
#pragma unroll
      for (int ni = 0; ni < NO; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[ni][r] -= mean; q += v[ni][r] * v[ni][r]; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q * (1.f / DC) + a.eps);
      float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
      for (int ni = 0; ni < NO; ++ni) {
        const int c0 = ni * 16 + 4 * fchunk;
        const f32x4 g = *(const f32x4*)(a.ln_g + c0), be = *(const f32x4*)(a.ln_b + c0);
        const f32x4 w0 = *(const f32x4*)(a.head_w + c0), w1 = *(const f32x4*)(a.head_w + DC + c0),
                    w2 = *(const f32x4*)(a.head_w + 2 * DC + c0);
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

// ------------------------------------------------------------------------------------------------------------
// 16-bit dtypes: persistent "column-strip walker" (conv3x3_ring8_kernel below).  One workgroup per CU (the whole 160 KB of LDS)
// owns a work item = (image, 32-pixel column strip, row range) and walks it top to bottom in steps of 8 output rows:
//   * the 72 KB filter bank [co][tap][ci] is staged in LDS ONCE per workgroup, not re-read from L2 by every wave for every
//     tap.  Rows are padded to 1184 B = 74 sixteen-byte slots: a ds_read_b128 is served in four groups of 16 lanes that mix
//     two fragment chunks ({0-3, 12-15, 20-27}: filter rows 0-3, 12-15 of chunk c with rows 4-11 of chunk c + 1), and with a
//     pitch of 10 mod 16 slots the 16 (row, chunk) pairs of every group fall on 16 different slots of the 256-byte bank row
//     (round 3's 1168 B = 9 mod 16 made 7 of them collide two-way: SQ_LDS_BANK_CONFLICT 0.36 per active LDS cycle);
//   * the input rows live in an 18-row LDS ring (row slot = row mod 18, 36-pixel slots, chunk-swizzled by pixel); the 8
//     rows of the NEXT step arrive by LDS-DMA (zero page for pixels outside the image) while the MFMAs of the current
//     step run, so every input row is fetched once per strip (no vertical halo overlap) and the halo fill does not
//     serialise with the compute;
//   * the wait for the DMA sits BEFORE the epilogue's stores (vmcnt counts stores too).
// (Round 2's four-wave form of the walker -- one wave per SIMD, MFMA phase and epilogue one after the other -- was superseded
// by the eight-wave anti-phase form in round 3 and removed in round 4: DESIGN.md section 8.)
constexpr int CR_ROWS = 8, CR_RING = 18, CR_SLOT_PX = 36, CR_WROW = 1184;
constexpr int CR_W_BYTES = 64 * CR_WROW, CR_RING_BYTES = CR_RING * CR_SLOT_PX * 128, CR_LDS = CR_W_BYTES + CR_RING_BYTES;
__device__ __attribute__((aligned(256))) const unsigned char bsg_conv_zero_page[256] = {0};

struct ConvRingArgs {
  ConvArgs c;
  int y_begin;   // first output row (multiple of 8); rows [y_begin, H) are produced
  int nsplit;    // row ranges per (image, strip)
  int batch;
};

// ------------------------------------------------------------------------------------------------------------
// EIGHT waves in two groups that run in ANTI-PHASE (round 3).  With four waves (one per SIMD) the 288-MFMA phase of a step
// (MFMA pipe busy, vector ALU idle) and its fused bias + LayerNorm + GELU + 1x1-head epilogue (vector ALU busy --
// SQ_ACTIVE_INST_VALU 0.52 of the kernel -- MFMA pipe idle) ran one after the other: mfma_busy 0.26.  Here a wave owns ONE row of the 8-row step (2 x 4 accumulator tiles, 144 MFMAs); group 0 (waves 0-3,
// rows 0-3) and group 1 (waves 4-7, rows 4-7) share the SIMDs pairwise and are held half a step apart by two barriers per
// step: while one group is in its MFMA phase the other is in its epilogue, so the two pipes of a SIMD work at the same time:
//   half-period 2k   : group 0 MFMA(step k)   | group 1 epilogue(step k-1)     [all waves: DMA of step k+1's 8 new rows]
//   half-period 2k+1 : group 0 epilogue(k)    | group 1 MFMA(step k)
// The ring has 18 rows: the rows of step k+1 replace interior rows of step k-1, whose last reader
// (group 1's MFMA phase, half-period 2k-1) is a barrier behind the issue (start of half-period 2k).  Filter bank, ring image,
// swizzles, DMA pieces and the epilogue arithmetic are unchanged.
template <typename T, int MODE>
__global__ __launch_bounds__(512, 1) void conv3x3_ring8_kernel(ConvRingArgs ra) {
  static_assert(sizeof(T) == 2, "ring kernel: 16-bit activations (128-byte pixels)");
  typedef typename Traits<T>::Chunk Chunk;
  const ConvArgs& a = ra.c;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                 // filter bank, [64 co] rows of CR_WROW bytes: [tap][ci]
  char* ring = smem + CR_W_BYTES;  // [CR_RING][CR_SLOT_PX] pixels x 128 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wrow = wave;  // group; row of the 8-row step this wave owns (group g: rows 4g .. 4g+3)
  const int frow = lane & 15, fchunk = lane >> 4;

  for (int i = tid; i < 64 * 72; i += 512) {
    const int co = i / 72, ch = i % 72;
    *(Chunk*)(wl + co * CR_WROW + ch * 16) = *(const Chunk*)((const char*)a.w + ((long)co * 72 + ch) * 16);
  }

  const int strips = a.W / 32, steps_total = (a.H - ra.y_begin) / CR_ROWS;
  const int items = ra.batch * strips * ra.nsplit;
  const char* zero = (const char*)bsg_conv_zero_page;
  const int lp = lane >> 3, lq = lane & 7;
  unsigned col_off[5];
  bool col_ok[5];
  auto issue_row = [&](const char* img, int row) {
    const int slot = (row + CR_RING) % CR_RING;
    const bool row_ok = row >= 0 && row < a.H;
    const char* rp = img + (long)row * a.W * 128;
#pragma unroll
    for (int part = 0; part < 5; ++part) {
      const char* src = (row_ok && col_ok[part]) ? rp + col_off[part] : zero + (lq << 4);
      if (part < 4 || lane < 16) glds16(src, ring + (slot * CR_SLOT_PX + part * 8) * 128);
    }
  };

  f32x4 cb[4], cg[4], cbe[4], cw0[4], cw1[4], cw2[4];
  if (MODE == CONV_FWD_FUSED) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c0 = ni * 16 + 4 * fchunk;
      cb[ni] = *(const f32x4*)(a.bias + c0);
      cg[ni] = *(const f32x4*)(a.ln_g + c0);
      cbe[ni] = *(const f32x4*)(a.ln_b + c0);
      cw0[ni] = *(const f32x4*)(a.head_w + c0);
      cw1[ni] = *(const f32x4*)(a.head_w + 64 + c0);
      cw2[ni] = *(const f32x4*)(a.head_w + 128 + c0);
    }
  }
  const float hb0 = MODE == CONV_FWD_FUSED ? a.head_b[0] : 0.f, hb1 = MODE == CONV_FWD_FUSED ? a.head_b[1] : 0.f,
              hb2 = MODE == CONV_FWD_FUSED ? a.head_b[2] : 0.f;
  const unsigned wl_a = lds_addr(wl) + frow * CR_WROW + fchunk * 16, ring_a = lds_addr(ring);

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int part = item % ra.nsplit, t = item / ra.nsplit;
    const int strip = t % strips, b = t / strips;
    const int s0 = (int)((long)steps_total * part / ra.nsplit), s1 = (int)((long)steps_total * (part + 1) / ra.nsplit);
    if (s1 <= s0) continue;
    const int ns = s1 - s0, x0 = strip * 32;
    const char* img = (const char*)a.in + (long)b * a.H * a.W * 128;
#pragma unroll
    for (int pt = 0; pt < 5; ++pt) {
      const int p = pt * 8 + lp, x = x0 - 1 + p;
      col_ok[pt] = x >= 0 && x < a.W;
      col_off[pt] = (unsigned)(x * 128 + ((lq ^ (p & 7)) << 4));
    }
    __syncthreads();  // every wave is done with the ring of the previous item (and the filter bank is staged)
    {  // rows y - 1 .. y + 8 of the first step: one per wave, the last two by waves 0 and 1
      const int r0 = ra.y_begin + s0 * CR_ROWS - 1;
      issue_row(img, r0 + wave);
      if (wave < 2) issue_row(img, r0 + 8 + wave);
    }
    wait_vm0();
    __syncthreads();

    f32x4 acc[4][2];  // [ni][x half]: this wave's row of the step it last ran the MFMA phase of
    for (int hp = 0; hp < 2 * ns + 1; ++hp) {
      const bool even = !(hp & 1);
      if (even && hp / 2 + 1 < ns)  // the next step's 8 new rows, one per wave
        issue_row(img, ra.y_begin + (s0 + hp / 2) * CR_ROWS + CR_ROWS + 1 + wave);
      const bool do_mma = grp == 0 ? (even && hp / 2 < ns) : !even;
      const bool do_epi = grp == 0 ? !even : (even && hp >= 2);
      const int st = s0 + (grp == 0 ? hp / 2 : (even ? hp / 2 - 1 : (hp - 1) / 2));
      const int y0 = ra.y_begin + st * CR_ROWS;
      if (do_mma) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int rbase[3];  // ring bases of the 3 input rows this wave touches: y0 + wrow - 1 + {0, 1, 2}
#pragma unroll
        for (int i = 0; i < 3; ++i) rbase[i] = ((y0 + wrow - 1 + i + CR_RING) % CR_RING) * (CR_SLOT_PX * 128);
        f32x4 fa[2][2], fw[2][4];
        auto load_frags = [&](int it, f32x4 (&A)[2], f32x4 (&Wf)[4]) {
          const int tap = it >> 1, ks = it & 1, dy = tap / 3, dx = tap - 3 * dy, c = fchunk + 4 * ks;
#pragma unroll
          for (int i = 0; i < 4; ++i) Wf[i] = lds_read16_nowait(wl_a + i * 16 * CR_WROW + tap * 128 + ks * 64);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int p = i * 16 + frow + dx;  // pixel inside the slot
            A[i] = lds_read16_nowait(ring_a + rbase[dy] + p * 128 + ((c ^ (p & 7)) << 4));
          }
        };
        load_frags(0, fa[0], fw[0]);
#pragma unroll
        for (int it = 0; it < 18; ++it) {
          if (it + 1 < 18) {
            load_frags(it + 1, fa[(it + 1) & 1], fw[(it + 1) & 1]);
            lds_wait<6>();  // iteration it's 6 fragments are in; the 6 just issued stay in flight under the MFMAs
          } else {
            lds_wait<0>();
          }
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              mma16(acc[ni][mi], __builtin_bit_cast(Chunk, fw[it & 1][ni]), __builtin_bit_cast(Chunk, fa[it & 1][mi]));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 1) wait_vm0();  // group 1: its DMA piece (issued a half-period ago, behind it only old stores) has landed
      }
      if (do_epi) {
        if (grp == 0) wait_vm0();  // group 0: its DMA piece of this step's start has landed; waited BEFORE the stores below
        // acc[ni][mi][r]: pixel (y0 + wrow, x0 + mi * 16 + frow), channel ni*16 + 4*fchunk + r
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int y = y0 + wrow, x = x0 + mi * 16 + frow;
          const long pix = ((long)b * a.H + y) * a.W + x;
          f32x4 v[4];
          float s = 0.f;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            v[ni] = acc[ni][mi];
            if (MODE == CONV_FWD_FUSED) {
              v[ni] += cb[ni];
              s += v[ni][0] + v[ni][1] + v[ni][2] + v[ni][3];
            }
          }
          if (MODE == CONV_PLAIN || a.out) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            unsigned lo[4], hi[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              lo[ni] = pack2<T>(v[ni][0], v[ni][1]);
              hi[ni] = pack2<T>(v[ni][2], v[ni][3]);
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const int ia = 2 * pr, ib = 2 * pr + 1;
              auto r0 = __builtin_amdgcn_permlane16_swap(lo[ia], lo[ib], false, false);
              auto r1 = __builtin_amdgcn_permlane16_swap(hi[ia], hi[ib], false, false);
              const int ch = ((fchunk & 1) ? ib : ia) * 16 + (fchunk >> 1) * 8;
              *(u32x4*)((T*)a.out + pix * 64 + ch) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
          }
          if (MODE == CONV_FWD_FUSED) {
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
#ifdef BSG_CONV_PK  // measured (round 4): the LayerNorm affine + GELU + head on v_pk_* pairs: 3.02 vs 2.96 ms -- beside the other wave
            // group's MFMAs packed f32 instructions are no faster than the scalar pair (guide: an anti-lever beside MFMAs), and
            // the pair alignment costs a spilled register
            const f32x2 rs2 = f32x2{rstd, rstd};
            f32x2 p0 = f32x2{0.f, 0.f}, p1 = p0, p2 = p0;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
              for (int r = 0; r < 4; r += 2) {
                const f32x2 xn = f32x2{v[ni][r], v[ni][r + 1]} * rs2;
                const f32x2 tt = gelu_pk(fma2(xn, f32x2{cg[ni][r], cg[ni][r + 1]}, f32x2{cbe[ni][r], cbe[ni][r + 1]}));
                p0 = fma2(tt, f32x2{cw0[ni][r], cw0[ni][r + 1]}, p0);
                p1 = fma2(tt, f32x2{cw1[ni][r], cw1[ni][r + 1]}, p1);
                p2 = fma2(tt, f32x2{cw2[ni][r], cw2[ni][r + 1]}, p2);
              }
            }
            float o0 = p0[0] + p0[1], o1 = p1[0] + p1[1], o2 = p2[0] + p2[1];
#else
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float tt = gelu_f(v[ni][r] * rstd * cg[ni][r] + cbe[ni][r]);
                o0 += tt * cw0[ni][r]; o1 += tt * cw1[ni][r]; o2 += tt * cw2[ni][r];
              }
            }
#endif
            o0 += __shfl_xor(o0, 16, 64); o0 += __shfl_xor(o0, 32, 64);
            o1 += __shfl_xor(o1, 16, 64); o1 += __shfl_xor(o1, 32, 64);
            o2 += __shfl_xor(o2, 16, 64); o2 += __shfl_xor(o2, 32, 64);
            if (fchunk == 0) {
              const long hw = (long)a.H * a.W, base = (long)b * 3 * hw + (long)y * a.W + x;
              a.pred[base] = o0 + hb0;
              a.pred[base + hw] = o1 + hb1;
              a.pred[base + 2 * hw] = o2 + hb2;
            }
          }
        }
      }
      __syncthreads();  // the half-period boundary: roles swap; after an odd one the next step's rows are in for everybody
    }
  }
}

// Per-pixel backward of head (1x1) -> GELU -> LayerNorm(DC): dpred (fp32 NCHW) + saved conv output -> d conv out.
// DC / 16 lanes per pixel (16 channels each, 32/64 contiguous bytes per lane), reductions by xor-shuffles inside the group.
template <typename T, int DC>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dpred, const T* __restrict__ conv_out,
                                                        const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                        const float* __restrict__ head_w, T* __restrict__ dconv, int B,
                                                        int H, int W, float eps, int row0, const float* gscale) {
  constexpr int LP = DC / 16;  // lanes per pixel: 4 or 8
  const long hw = (long)H * W, win = (long)(H - row0) * W, total = (long)B * win;  // rows [row0, H) of every image
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long pl = gid / LP;
  const int c0 = (int)(gid % LP) * 16;
  if (pl >= total) return;  // total*LP is a multiple of 64: whole waves exit together
  const int b = pl / win;
  const long yx = (long)row0 * W + pl % win;
  const long p = (long)b * hw + yx;
  const float gs = gscale ? gscale[0] : 1.0f;  // f16 mode: the dgrad chain runs on a power-of-two multiple of the gradient
  const float d0 = dpred[(long)b * 3 * hw + yx] * gs, d1 = dpred[(long)b * 3 * hw + hw + yx] * gs,
              d2 = dpred[(long)b * 3 * hw + 2 * hw + yx] * gs;
  typedef typename Traits<T>::Vec4 V4;
  V4* dst = (V4*)(dconv + p * DC + c0);
  float v[16], g[16];
  const V4* src = (const V4*)(conv_out + p * DC + c0);
  auto group_sum = [](float x) {
    x += __shfl_xor(x, 1, 64);
    x += __shfl_xor(x, 2, 64);
    if (LP == 8) x += __shfl_xor(x, 4, 64);
    return x;
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const V4 t = src[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[4 * i + j] = to_f32(t[j]); s += v[4 * i + j]; }
  }
  const float mean = group_sum(s) * (1.f / DC);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) { v[c] -= mean; q += v[c] * v[c]; }
  const float rstd = rsqrtf(group_sum(q) * (1.f / DC) + eps);
  float mg = 0.f, mgx = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    v[c] *= rstd;  // xhat
    const float gm = ln_g[c0 + c];
    const float y = v[c] * gm + ln_b[c0 + c];
    const float dgl = d0 * head_w[c0 + c] + d1 * head_w[DC + c0 + c] + d2 * head_w[2 * DC + c0 + c];
    g[c] = dgl * gelu_grad_f(y) * gm;
    mg += g[c];
    mgx += g[c] * v[c];
  }
  mg = group_sum(mg) * (1.f / DC);
  mgx = group_sum(mgx) * (1.f / DC);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    dst[i] = pack4<T>(rstd * (g[4 * i] - mg - v[4 * i] * mgx), rstd * (g[4 * i + 1] - mg - v[4 * i + 1] * mgx),
                      rstd * (g[4 * i + 2] - mg - v[4 * i + 2] * mgx), rstd * (g[4 * i + 3] - mg - v[4 * i + 3] * mgx));
}
