// Fused SegGPT attention for gfx950: softmax(scale * q k^T + relh[q, kh] + relw[q, kw]) v with the decomposed
// relative-position bias of HF:modeling_seggpt.py:268-311 applied in-kernel, forward and backward, no N x N
// score tensor in HBM.  head_dim is fixed at 64.  One source for bf16 (32x32x16 MFMA) and f32 (32x32x2 MFMA).
//
// Geometry.  Tokens form an Hp x Wp grid (Wp <= 32).  A key tile is TWO grid rows, each padded to 32 slots, so
// a 64-slot tile is two 32-row MFMA blocks: block b <-> grid row 2t + b, slot-in-block <-> kw.  Consequences:
//   * the column part of the bias depends only on the accumulator register (16 loop-invariant values per lane)
//     and the row part is one scalar per (query, block): the bias is the INITIAL ACCUMULATOR of the QK^T MFMAs;
//   * "transposed" operands (V^T for PV; K^T, Q^T, dO^T in backward) live in HBM in a row-padded layout
//     [stream][head][64][Hp*32], so every MFMA operand of every product is read along its contraction index.
// QK^T is computed "swapped" (S^T = K Q^T): a lane owns ONE query column, so softmax statistics, the bias and
// the O^T rescale are lane-local, and the S^T accumulator is directly the B operand of O^T += V^T P^T.
#pragma once
#include "common.hpp"
#ifndef BSG_DIAG_DQ
#define BSG_DIAG_DQ 0  // timing-only ablations of the dQ kernel (wrong results): 1 no softmax VALU, 2 no MFMA, 3 no LDS reads, 4 no DMA/barrier
#endif

// "float32 at three f16 MFMAs" for the exact-f32 attention kernels (X3 = bsg_config.gemm_x3): each f32 fragment as
// hi = f16(x), lo = f16(x - hi) (22 significant bits), hi*hi + hi*lo + lo*hi accumulated in fp32 by three
// v_mfma_f32_32x32x8_f16 (3 x 32 cycles) in place of four v_mfma_f32_32x32x2_f32 (4 x 64 cycles); same fragment geometry
// (lane = row / column, 4 k-values per lane and half-wave).  Loop-invariant fragments (Q, K, V, dO held in registers) are
// converted once by the compiler's hoisting; tile fragments once per use.
DEVI void x3_split(const f32x4& x, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) hi[e] = (f16_t)x[e];
#pragma unroll
  for (int e = 0; e < 4; ++e) lo[e] = (f16_t)(x[e] - (float)hi[e]);
}
DEVI void mma32_x3(f32x16& acc, const f32x4& a, const f32x4& b) {
  f16x4 ah, al, bh, bl;
  x3_split(a, ah, al);
  x3_split(b, bh, bl);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(al, bh, acc, 0, 0, 0);
#ifndef BSG_X3_NO_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
}
// a fragment that lives in registers across the key / query loop, split ONCE (same register count as the f32x4 it replaces)
struct X3Frag { f16x4 hi, lo; };
DEVI X3Frag x3_frag(const f32x4& x) { X3Frag f; x3_split(x, f.hi, f.lo); return f; }
template <typename Ch> DEVI Ch x3_frag(const Ch& x) { return x; }  // 16-bit dtypes: the fragment itself
DEVI void mma32_x3(f32x16& acc, const f32x4& a, const X3Frag& b) {
  f16x4 ah, al;
  x3_split(a, ah, al);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ah, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x8f16(al, b.hi, acc, 0, 0, 0);
#ifndef BSG_X3_NO_FENCE
  // keep the scheduler from hoisting the NEXT iterations' tile reads + splits above these MFMAs: with 16 unrolled (block,
  // k-step) iterations that parks ~120 split registers and spills (dQ kernel: 480 B of scratch per lane, slower than exact f32)
  __builtin_amdgcn_sched_barrier(0);
#endif
}
template <bool X3> DEVI void mm32(f32x16& acc, const f32x4& a, const X3Frag& b) { mma32_x3(acc, a, b); }
// Two f32 k-steps (8 + 8 k-values) side by side = one operand of `v_mfma_f32_32x32x16_f16`, which runs at the full f16 rate on
// gfx950 (the K = 8 form above takes the same 32 cycles for half the k-values).  The k-order inside the instruction is
// (first chunk | second chunk) on BOTH operands, so the contraction is the same sum.
struct X3Frag8 { f16x8 hi, lo; };
DEVI X3Frag8 x3_frag8(const f32x4& x0, const f32x4& x1) {
  X3Frag8 f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f16_t h0 = (f16_t)x0[e], h1 = (f16_t)x1[e];
    f.hi[e] = h0; f.hi[4 + e] = h1;
    f.lo[e] = (f16_t)(x0[e] - (float)h0); f.lo[4 + e] = (f16_t)(x1[e] - (float)h1);
  }
  return f;
}
template <typename Ch> DEVI Ch x3_frag8(const Ch& x, const Ch&) { return x; }  // never used for the 16-bit dtypes (keeps `if constexpr` bodies well-formed)
// x3 mode: q / k / v live in HBM as PACKED pairs -- each float slot of the qkv buffer holds (f16 hi | f16 lo << 16), written by
// the QKV projection's epilogue (gemm.hpp pack_hl) -- so every copy or transpose of 4-byte elements keeps the pairs, and an
// operand half is two byte-permutes per chunk instead of a 12-instruction split repeated by every workgroup that reads the tile.
DEVI X3Frag8 x3_frag8_pk(const f32x4& c0, const f32x4& c1) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 a = __builtin_bit_cast(u32x4, c0), b = __builtin_bit_cast(u32x4, c1);
  u32x4 hi, lo;  // 8 halves each: (a0 a1 | a2 a3 | b0 b1 | b2 b3)
  hi[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100u); lo[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u);
  hi[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100u); lo[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
  hi[2] = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u); lo[2] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u);
  hi[3] = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u); lo[3] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
  X3Frag8 f;
  f.hi = __builtin_bit_cast(f16x8, hi);
  f.lo = __builtin_bit_cast(f16x8, lo);
  return f;
}
template <typename Ch> DEVI Ch x3_frag8_pk(const Ch& x, const Ch&) { return x; }
DEVI f32x4 x3_unpack(const f32x4& c) {  // packed pairs -> the float values (hi + lo is exact)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 a = __builtin_bit_cast(u32x4, c);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned u = a[e];  // (bit_cast of the vector element itself to f16x2 miscompiles: every lane reads element 0)
    v[e] = (float)__builtin_bit_cast(f16_t, (unsigned short)(u & 0xffffu)) + (float)__builtin_bit_cast(f16_t, (unsigned short)(u >> 16));
  }
  return v;
}
template <typename Ch> DEVI Ch x3_unpack(const Ch& x) { return x; }
// "use" of a register value that emits nothing: pins the point where hipcc has to have waited for the load that produces it
template <typename V> DEVI void reg_consume(const V& v) { asm volatile("" ::"v"(v)); }
DEVI void reg_consume(const X3Frag8& f) { asm volatile("" ::"v"(f.hi), "v"(f.lo)); }
DEVI void mma32_x3_pair(f32x16& acc, const X3Frag8& a, const X3Frag8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, acc, 0, 0, 0);
#ifndef BSG_X3_NO_FENCE
  __builtin_amdgcn_sched_barrier(0);
#endif
}
template <bool X3, typename Ch> DEVI void mm32(f32x16& acc, const Ch& a, const Ch& b) {
  if constexpr (X3 && std::is_same<Ch, f32x4>::value) mma32_x3(acc, a, b);
  else mma32(acc, a, b);
}

struct AttnArgs {
  const void* q;   // T [S*N][ld] (+ column offset applied by caller) ; head h at columns h*64..
  const void* k;
  const void* v;   // backward only (row-major V)
  long ld;         // row stride in elements (3*D for the fused qkv buffer)
  const void* vt;  // T [S][nh][64][Hp*32]
  const void* kt;  // backward
  const void* qt;  // backward
  const void* dot; // backward: dO^T
  const void* dout;  // backward: dO, T [S*N][ldo]
  const void* rel_cat;   // T [LH + LW][64]: rel_pos_h rows at 0.., rel_pos_w rows at LH = roundup16(2 Hp)..
  const void* rel_catT;  // backward: T [64][LH + LW]
  float* relhT;          // backward, WRITTEN by the dQ kernel for dK/dV: [S][nh][Hp][Hp*32], column = token; holds relh * c2 - lse2
  float* relwT;          // same: [S][nh][32][Hp*32]; -inf for kw >= Wp and in columns N .. 64 ceil(N/64) - 1
  void* out;          // fwd: O, T [S*N][ldo]
  long ldo;
  float* lse2;        // [S][nh][Hp*32] (entries 0..N-1 used, the rest stay 0)  log2-domain logsumexp of the logits
  float* delta;       // backward: [S][nh][Hp*32] (same indexing)  MINUS rowsum(dO * O): WRITTEN by the dQ kernel, read by dK/dV
  void* dq;  // backward outputs, T [S*N][ld] at the q/k/v column offsets of the dqkv buffer
  void* dk;
  void* dv;
  int S, nh, N, hp, wp;
  float scale;
  // backward, dQ kernel: the caller reads dq of the queries [dq_begin, dq_end) only (dq_end = 0: up to N).  Workgroups whose 128
  // queries lie outside still publish their relwT / relhT / delta columns (dK/dV reads them) and store zero dq rows.
  int dq_begin, dq_end;
};

// 1-D grid -> (x, head, stream) with all x-blocks of one (stream, head) on the SAME XCD: they re-read the same
// K / V^T (or Q / dO) tiles, which then come from that XCD's L2 instead of eight separate fetches from HBM.
DEVI void attn_block_ids(int nx, int nh, int S, int& x, int& head, int& s) {
  const int lid = xcd_remap(blockIdx.x, nx * nh * S);
  x = lid % nx;
  const int t = lid / nx;
  head = t % nh;
  s = t / nh;
}


// LDS chunk swizzle (position = chunk ^ swz(row)).  128-byte (bf16) rows are read two ways: along the row
// (ds_read_b128, lane = row) and TRANSPOSED (ds_read_b64_tr_b16, four rows x 64 bytes per half-wave).  With
// u = row >> 1, (u & 7) ^ ((u & 1) << 2) keeps both conflict-free: it is a bijection on the eight row pairs a
// b128 phase touches, and rows r, r+2 of a transposed 4-row block land in different 64-byte bank groups.
template <int RB> DEVI int swz(int row) {
  return RB == 128 ? (((row >> 1) & 7) ^ (((row >> 1) & 1) << 2)) : (row & 15);
}

// token index of slot `kw` of grid row `gr` (clamped to a valid token: padded slots carry weight 0)
DEVI int slot_token(int gr, int kw, int wp) { return gr * wp + (kw < wp ? kw : wp - 1); }

// Issue the LDS-DMA of one 64-row x RB-byte tile.  row_src(r) = global address of the start of tile row r.
template <typename T, typename RowSrc>
DEVI void dma_tile(char* lds_tile, int wave, int lane, RowSrc row_src) {
  constexpr int RB = 64 * sizeof(T), CPR = RB / 16, RPI = 64 / CPR, IPW = 64 / (4 * RPI);
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int r = (wave * IPW + i) * RPI + lane / CPR, p = lane % CPR;
    const char* src = row_src(r) + ((p ^ swz<RB>(r)) << 4);
    glds16_asm(src, lds_tile + (wave * IPW + i) * 1024);
  }
}

// The same tile copy with the per-lane part of the source address hoisted out of the key loop: 32-bit byte offsets
// (row offset at tile 0 + swizzled chunk) computed once, and a wave-uniform base that advances by a constant per tile,
// so a tile costs one scalar add and the DMA instructions (SGPR base + VGPR offset) instead of re-deriving 64-bit
// per-lane pointers.
template <typename T> struct TileDma {
  static constexpr int RB = 64 * sizeof(T), CPR = RB / 16, RPI = 64 / CPR, IPW = 64 / (4 * RPI);
};
template <typename T, typename RowOff>
DEVI void dma_tile_offsets(unsigned (&off)[TileDma<T>::IPW], int wave, int lane, RowOff row_off) {
  typedef TileDma<T> D;
#pragma unroll
  for (int i = 0; i < D::IPW; ++i) {
    const int r = (wave * D::IPW + i) * D::RPI + lane / D::CPR, p = lane % D::CPR;
    off[i] = (unsigned)row_off(r) + ((p ^ swz<D::RB>(r)) << 4);
  }
}
template <typename T>
DEVI void dma_tile_issue(char* lds_tile, int wave, const char* base, const unsigned (&off)[TileDma<T>::IPW]) {
  typedef TileDma<T> D;
#pragma unroll
  for (int i = 0; i < D::IPW; ++i) glds16_asm(base + off[i], lds_tile + (wave * D::IPW + i) * 1024);
}

template <typename T> DEVI typename Traits<T>::Chunk lds_chunk(const char* tile, int row, int chunk) {
  constexpr int RB = 64 * sizeof(T);
  return *(const typename Traits<T>::Chunk*)(tile + row * RB + ((chunk ^ swz<RB>(row)) << 4));
}

// A-operand chunk of a [row][64 contraction slots] LDS tile matching an S^T-layout accumulator block used as
// the B operand.  bf16: k-step s (16 slots) of 32-slot block b, element j <-> slot 16s + 8(j>>2) + 4h + (j&3);
// f32: k-step s (8 slots), element j <-> slot 8s + 4h + j.  In both cases one conflict-free ds_read_b128.
// (16-bit dtypes never take this path: their "transposed" operands come out of the row-major tiles, lds_tr_chunk)
template <typename T16> DEVI typename Traits<T16>::Chunk lds_perm_chunk(const char*, int, int, int, int, T16) {
  return typename Traits<T16>::Chunk{};
}
DEVI f32x4 lds_perm_chunk(const char* tile, int row, int b, int s, int h, float) {
  const int c = 8 * b + 2 * s + h;
  return *(const f32x4*)(tile + row * 256 + ((c ^ swz<256>(row)) << 4));
}
// The same bf16 A operand fetched from the ROW-MAJOR [slot][64] tile with the gfx950 transposing
// LDS read: per 16-lane group, ds_read_b64_tr_b16 takes a 4-row x 16-column block (lane 4q+p addresses row q,
// columns 4p..4p+3) and hands lane i column i of the four rows.  Rows = slots 16s + 4h + {0..3} (+8 for the second
// read), columns = d 32*db + 16*(group & 1) + i = this lane's A row.  No transposed copy in HBM, no permutation.
typedef __attribute__((address_space(3))) bf16x4* lds_b4_ptr;
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_raw* lds_h4_ptr;
DEVI f32x2 lds_tr_read(const char* p, bf16_t) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_ptr)p));
}
DEVI f32x2 lds_tr_read(const char* p, f16_t) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4_ptr)p));
}
template <typename T> DEVI typename Traits<T>::Chunk lds_tr_chunk(const char* tile, int db, int b, int s, int lane) {
  const int gi = lane >> 4, i = lane & 15, qd = i >> 2, p = i & 3, h = gi >> 1;
  const int r0 = 32 * b + 16 * s + 4 * h + qd;
  const int c = 4 * db + 2 * (gi & 1) + (p >> 1);
  const char* a0 = tile + r0 * 128 + ((c ^ swz<128>(r0)) << 4) + 8 * (p & 1);
  const char* a1 = tile + (r0 + 8) * 128 + ((c ^ swz<128>(r0 + 8)) << 4) + 8 * (p & 1);
  const f32x2 lo = lds_tr_read(a0, T()), hi = lds_tr_read(a1, T());
  return __builtin_bit_cast(typename Traits<T>::Chunk, (f32x4{lo[0], lo[1], hi[0], hi[1]}));
}
// The matching B-operand chunk built from accumulator block `p` (already exponentiated / scaled).
DEVI bf16x8 acc_chunk(const f32x16& p, int s, bf16_t) {
  bf16x8 c;
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = (bf16_t)p[8 * s + j];
  return c;
}
DEVI f16x8 acc_chunk(const f32x16& p, int s, f16_t) {
  f16x8 c;
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = (f16_t)p[8 * s + j];
  return c;
}
DEVI f32x4 acc_chunk(const f32x16& p, int s, float) { return f32x4{p[4 * s], p[4 * s + 1], p[4 * s + 2], p[4 * s + 3]}; }

template <typename T> struct AttnK {
  static constexpr int EPC = Traits<T>::EPC;
  static constexpr int KS_D = 64 / (2 * EPC);   // k-steps over head_dim for a 32x32 MFMA (4 bf16 / 8 f32)
  static constexpr int KS_B = 32 / (2 * EPC);   // k-steps over one 32-slot block (2 bf16 / 4 f32)
  static constexpr int RB = 64 * sizeof(T);
  static constexpr int TILE = 64 * RB;
};


// ------------------------------------------------------------------------- decomposed rel-pos bias, in-kernel
// relh[q][kh] = q . rel_pos_h[qh - kh + Hp - 1] / scale,  relw[q][kw] = q . rel_pos_w[qw - kw + Wp - 1] / scale
// (unscaled q, HF:268-311, HF:326-329) for the 32 queries of one wave, from the Q fragments the wave already
// holds: G^T[rel][q] = rel_cat[rel] . q on MFMA (lane = query, register = rel row), then the per-query shift
// goes through LDS.  relh lands in a per-wave table [32][HS] that stays resident for the key loop; relw is
// returned as the loop-invariant accumulator-init vector (-inf on padded key slots: no masking anywhere).
DEVI int relh_stride(int hp) { return hp | 1; }  // odd: lane-per-row accesses are conflict-free

template <typename T>
DEVI void relpos_wave_h(const typename Traits<T>::Chunk* qf, const void* rel_cat, int hp, int qh, int qh_first,
                        int qh_last /* wave-uniform range of qh */, float alpha, float* table_h /* 32 x HS */, int lane) {
  typedef typename Traits<T>::Chunk Chunk;
  typedef AttnK<T> C;
  const int h = lane >> 5, col = lane & 31;
  const int HS = relh_stride(hp), nrelh = 2 * hp - 1;
  const char* rc = (const char*)rel_cat + h * 16;  // chunk 2 ks + h of a 128-byte row
  // h part: query row qh needs rel rows qh .. qh + hp - 1, so the wave's 32 queries need the window
  // [qh_first, qh_last + hp - 1] only (two 32-row blocks for the 56 x 28 grid).  All A chunks of a group are fetched
  // (L2) before the first MFMA; rows past the table are clamped and results outside a lane's own window land in the
  // spare slot `hp` of its row (no exec juggling per store).
  const int nrows = hp + qh_last - qh_first;
  for (int blk0 = 0; blk0 * 32 < nrows; blk0 += 2) {
    Chunk af[2][C::KS_D];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = min(qh_first + (blk0 + i) * 32 + col, nrelh - 1);
#pragma unroll
      for (int ks = 0; ks < C::KS_D; ++ks) af[i][ks] = *(const Chunk*)(rc + ((long)rr * 64) * sizeof(T) + ks * 32);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS_D; ++ks) mma32(acc, af[i][ks], qf[ks]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kh = qh + hp - 1 - (qh_first + (blk0 + i) * 32 + acc32_row(r, h));
        table_h[col * HS + ((unsigned)kh < (unsigned)hp ? kh : hp)] = acc[r] * alpha;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): single-wave image
  __builtin_amdgcn_wave_barrier();
}

template <typename T>
DEVI void relpos_wave_w(const typename Traits<T>::Chunk* qf, const void* rel_cat, int hp, int wp, int qw, float alpha,
                        float* scratch_w /* 32 x 64 floats */, f32x16& rwv, int lane) {
  typedef typename Traits<T>::Chunk Chunk;
  typedef AttnK<T> C;
  const int h = lane >> 5, col = lane & 31;
  const int nrelw = 2 * wp - 1, LH = (2 * hp + 15) & ~15;
  const char* rc = (const char*)rel_cat + h * 16;
  {  // 2 wp - 1 <= 63 rows
    Chunk af[2][C::KS_D];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = LH + min(i * 32 + col, nrelw - 1);
#pragma unroll
      for (int ks = 0; ks < C::KS_D; ++ks) af[i][ks] = *(const Chunk*)(rc + ((long)rr * 64) * sizeof(T) + ks * 32);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS_D; ++ks) mma32(acc, af[i][ks], qf[ks]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rel = i * 32 + acc32_row(r, h);  // < 64: column rotated by the row keeps both accesses spread
        scratch_w[col * 64 + ((rel + col) & 63)] = acc[r] * alpha;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): single-wave image
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kw = acc32_row(r, h);
    const float v = scratch_w[col * 64 + ((qw + wp - 1 - min(kw, wp - 1) + col) & 63)];
    rwv[r] = kw < wp ? v : -INFINITY;
  }
}

template <typename T>
DEVI void relpos_wave_tables(const typename Traits<T>::Chunk* qf, const void* rel_cat, int hp, int wp, int qh, int qw,
                             int qh_first, int qh_last, float alpha, float* scratch_w, float* table_h, f32x16& rwv,
                             int lane) {
  relpos_wave_h<T>(qf, rel_cat, hp, qh, qh_first, qh_last, alpha, table_h, lane);
  relpos_wave_w<T>(qf, rel_cat, hp, wp, qw, alpha, scratch_w, rwv, lane);
}

// Backward of the above into dq: acc[blk][.] (d = 32 blk + ..., lane = query) = sum_rel rel_catT[d][rel] X[rel][q],
// X[rel][q] = drel[q][pos + size - 1 - rel] gathered per lane from the per-wave LDS images of d relh / d relw.
template <typename T>
DEVI void relpos_wave_bwd(f32x16 (&acc)[2], const void* rel_catT, int hp, int wp, int qh, int qw, int qh_first, int qh_last,
                          const float* table_h, const float* image_w /* 32 x 33 */, int lane) {
  typedef typename Traits<T>::Chunk Chunk;
  constexpr int EPC = Traits<T>::EPC, KG = 4;  // k-steps whose A chunks are fetched together
  const int h = lane >> 5, col = lane & 31;
  const int HS = relh_stride(hp), LH = (2 * hp + 15) & ~15, LW = (2 * wp + 15) & ~15, RC = LH + LW;
  const char* rt = (const char*)rel_catT;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  for (int part = 0; part < 2; ++part) {
    const int len = part == 0 ? LH : LW, size = part == 0 ? hp : wp, pos = part == 0 ? qh : qw;
    const float* src = part == 0 ? table_h + col * HS : image_w + col * 33;
    // contraction k-steps: the h part only over the wave's window of rel rows [qh_first, qh_last + hp - 1]
    const int cbase = part == 0 ? 0 : LH;
    const int nks = part == 0 ? min((qh_last + hp - 1) / (2 * EPC) + 1, len / (2 * EPC)) : len / (2 * EPC);
    for (int ks0 = part == 0 ? qh_first / (2 * EPC) : 0; ks0 < nks; ks0 += KG) {
      Chunk af[KG][2], bf[KG];
#pragma unroll
      for (int i = 0; i < KG; ++i) {
        const int r0 = (2 * min(ks0 + i, nks - 1) + h) * EPC;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
          af[i][blk] = *(const Chunk*)(rt + ((long)(32 * blk + col) * RC + cbase + r0) * sizeof(T));
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const int k = pos + size - 1 - (r0 + j);
          const bool ok = (unsigned)k < (unsigned)size && ks0 + i < nks;
          const float v = src[ok ? k : 0];
          bf[i][j] = from_f32<T>(ok ? v : 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < KG; ++i)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) mma32(acc[blk], af[i][blk], bf[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T, bool TR, bool X3 = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  typedef typename Traits<T>::Chunk Chunk;
  typedef AttnK<T> C;
  static_assert(TR == (sizeof(T) == 2), "16-bit dtypes take the transposing-read path, f32 the transposed copies");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [buf][K | VT (TR: V)][TILE]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, head, s;
  attn_block_ids((a.N + 127) / 128, a.nh, a.S, bx, head, s);
  const int q0 = bx * 128 + wave * 32;
  const int q = min(q0 + col, a.N - 1);
  const long sh = (long)s * a.nh + head;
  const int npad = a.hp * 32;
  const char* kbase = (const char*)a.k + ((long)s * a.N * a.ld + head * 64) * sizeof(T);
  const char* vtbase = TR ? (const char*)a.v + ((long)s * a.N * a.ld + head * 64) * sizeof(T)
                          : (const char*)a.vt + sh * 64 * npad * sizeof(T);

  // loop-invariant per-lane state
  Chunk qf[C::KS_D];
  {
    const char* qrow = (const char*)a.q + (((long)s * a.N + q) * a.ld + head * 64) * sizeof(T);
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) {
      qf[ks] = *(const Chunk*)(qrow + (2 * ks + h) * 16);
      if constexpr (X3 && sizeof(T) == 4) qf[ks] = x3_unpack(qf[ks]);  // x3: qkv holds packed (hi, lo) pairs
    }
  }
  // rel-pos bias of this wave's 32 queries.  The tile area doubles as the shear scratch before the first DMA; the
  // row part goes out to a key-major global scratch [kh][token] (coalesced both ways, L2-resident: written here,
  // read back two floats per key tile) so that LDS holds only the K / V tiles and more workgroups fit per CU.
  f32x16 rwv;
  float* relh_g = a.relhT + sh * a.hp * npad + q0 + col;
  const bool active = q0 < a.N;  // wave-uniform: a wave past the last query only helps with the DMA and the barriers
  if (active) {
    float* slice = (float*)(smem + wave * max(8192, 128 * relh_stride(a.hp)));  // 32 x 64 shear image or 32 x HS relh table
    relpos_wave_h<T>(qf, a.rel_cat, a.hp, q / a.wp, q0 / a.wp, min(q0 + 31, a.N - 1) / a.wp, 1.0f / a.scale, slice, lane);
    if (q0 + col < a.N)
      for (int kh = h; kh < a.hp; kh += 2) relh_g[(long)kh * npad] = slice[col * relh_stride(a.hp) + kh];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    relpos_wave_w<T>(qf, a.rel_cat, a.hp, a.wp, q % a.wp, 1.0f / a.scale, slice, rwv, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the scratch stores have left before this wave reads them back
  __syncthreads();  // scratch reads done before any wave's DMA lands in the tile area
  if (q0 + col >= a.N) relh_g = a.relhT + sh * a.hp * npad + a.N - 1;  // clamped duplicate lanes read a valid column
  const float c2 = a.scale * 1.44269504088896340736f;
  const float lazy_thr = 8.0f / c2;  // 2^8 in the exponent, in score units
  float m = -INFINITY, l = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }

  const int nt = a.hp >> 1;
  // tile t = grid rows 2t, 2t + 1: every source row advances by 2 Wp tokens per tile
  unsigned koff[TileDma<T>::IPW], voff[TileDma<T>::IPW];
  const long es = sizeof(T), kstep = 2L * a.wp * a.ld * es, vstep = TR ? kstep : 64 * es;
  dma_tile_offsets<T>(koff, wave, lane, [&](int r) { return (long)slot_token(r >> 5, r & 31, a.wp) * a.ld * es; });
  if constexpr (TR) {
#pragma unroll
    for (int i = 0; i < TileDma<T>::IPW; ++i) voff[i] = koff[i];
  } else {
    dma_tile_offsets<T>(voff, wave, lane, [&](int r) { return (long)r * npad * es; });
  }
  auto issue = [&](int t, int buf) {
    char* kt_l = smem + buf * 2 * C::TILE;
    dma_tile_issue<T>(kt_l, wave, kbase + t * kstep, koff);
    dma_tile_issue<T>(kt_l + C::TILE, wave, vtbase + t * vstep, voff);
  };

  // X3: the loop-invariant Q fragments split once; the raw f32 copies die here
  constexpr bool X3f = X3 && sizeof(T) == 4;
  typedef typename std::conditional<X3f, X3Frag8, Chunk>::type LFrag;
  LFrag qx[X3f ? C::KS_D / 2 : C::KS_D];
  if constexpr (X3f) {
#pragma unroll
    for (int j = 0; j < C::KS_D / 2; ++j) qx[j] = x3_frag8(qf[2 * j], qf[2 * j + 1]);
  } else {
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) qx[ks] = qf[ks];
  }
  // (the two row-bias scalars of the next tile stay ORDINARY loads: hipcc waits for them -- `vmcnt(0)`, i.e. for the next tile's
  // DMA as well -- at the loop's bottom edge, a whole tile after the request.  As uncounted asm loads they were wrong: hipcc
  // copied their destination registers at the loop edge, before the top-of-tile wait.)
  f32x2 rh_next = f32x2{relh_g[0], relh_g[npad]};
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const f32x2 rh = rh_next;
    wait_vm0();
    __syncthreads();
    if (t + 1 < nt) {
      rh_next = f32x2{relh_g[(long)(2 * t + 2) * npad], relh_g[(long)(2 * t + 3) * npad]};
      issue(t + 1, buf ^ 1);
    }
    if (!active) continue;
    const char* kt_l = smem + buf * 2 * C::TILE;
    const char* vt_l = kt_l + C::TILE;

    // S^T[slot][q] = relw (initial accumulator = the loop-invariant register vector) + K q^T; the per-block
    // row bias relh[b] is one scalar per lane and is folded into the max and the exponent instead of 32 adds.
    f32x16 st[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      st[b] = rwv;
      if constexpr (X3f) {
#pragma unroll
        for (int j = 0; j < C::KS_D / 2; ++j)
          mma32_x3_pair(st[b], x3_frag8_pk(lds_chunk<T>(kt_l, 32 * b + col, 4 * j + h), lds_chunk<T>(kt_l, 32 * b + col, 4 * j + 2 + h)), qx[j]);
      } else {
#pragma unroll
        for (int ks = 0; ks < C::KS_D; ++ks)
          mm32<false>(st[b], lds_chunk<T>(kt_l, 32 * b + col, 2 * ks + h), qx[ks]);
      }
    }
    // online softmax over this lane's 32 slots (+ partner half-wave); padded key slots carry relw = -inf
    float mx0 = -INFINITY, mx1 = -INFINITY;  // a constant seed: no canonicalising v_max of the first MFMA outputs
#pragma unroll
    for (int r = 0; r < 16; ++r) { mx0 = fmaxf(mx0, st[0][r]); mx1 = fmaxf(mx1, st[1][r]); }
    float mx = fmaxf(mx0 + rh[0], mx1 + rh[1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    // Lazy reference maximum: the 34-multiply rescale of O and l runs only when some row of the wave overshoots its
    // reference by more than 2^8 in the exponent (wave-uniform branch).  A new row maximum shows up at key tile t with
    // probability ~1/t PER ROW, so with 32 rows per wave "rescale whenever any row has a new maximum" fired in ~85 % of
    // the tiles; below the threshold P simply reaches up to 2^8 instead of 1 (fp32 sums; bf16 / f16 P operands have the
    // range), and m stays a valid reference for lse2 = m c2 + log2(l).
    if (__builtin_amdgcn_ballot_w64(mn > m + lazy_thr)) {
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * c2);
      l *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
      m = mn;
    }
    const float nb0 = (rh[0] - m) * c2, nb1 = (rh[1] - m) * c2;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#ifdef BSG_DIAG_NOEXP  // timing-only build: prices the transcendental
      const float p0 = fmaf(st[0][r], c2, nb0), p1 = fmaf(st[1][r], c2, nb1);
#else
      const float p0 = __builtin_amdgcn_exp2f(fmaf(st[0][r], c2, nb0));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(st[1][r], c2, nb1));
#endif
      st[0][r] = p0;
      st[1][r] = p1;
      ps += p0 + p1;
    }
    l += ps;
    // O^T[d][q] += V^T[d][slot] P^T[slot][q]
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if constexpr (X3f) {
#pragma unroll
        for (int j = 0; j < C::KS_B / 2; ++j) {
          const X3Frag8 pb = x3_frag8(acc_chunk(st[b], 2 * j, T()), acc_chunk(st[b], 2 * j + 1, T()));
#pragma unroll
          for (int db = 0; db < 2; ++db)
            mma32_x3_pair(o[db], x3_frag8_pk(lds_perm_chunk(vt_l, 32 * db + col, b, 2 * j, h, T()),
                                             lds_perm_chunk(vt_l, 32 * db + col, b, 2 * j + 1, h, T())), pb);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < C::KS_B; ++ks) {
          const Chunk pb = acc_chunk(st[b], ks, T());
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            if constexpr (TR) mm32<false>(o[db], lds_tr_chunk<T>(vt_l, db, b, ks, lane), pb);
            else mm32<false>(o[db], lds_perm_chunk(vt_l, 32 * db + col, b, ks, h, T()), pb);
          }
        }
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  if (q0 + col < a.N) {
    const float inv = 1.f / l;
    T* orow = (T*)a.out + ((long)s * a.N + q) * a.ldo + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(typename Traits<T>::Vec4*)(orow + 32 * db + 8 * i + 4 * h) =
            pack4<T>(o[db][4 * i] * inv, o[db][4 * i + 1] * inv, o[db][4 * i + 2] * inv, o[db][4 * i + 3] * inv);
    if (h == 0 && a.lse2) a.lse2[sh * npad + q] = m * c2 + log2f(l);
  }
}

// ------------------------------------------------------------------------------------------- backward: dQ
// Query-stationary, same swapped layout as forward.  Per key tile: S^T (bias as initial accumulator) ->
// P^T = exp2(S^T c2 - lse2) -> dP^T = V dO^T -> dS^T = P^T (dP^T - delta) -> dQ^T += K^T dS^T.
// The rel-pos gradients fall out of the layout: d relw[q][kw] accumulates per accumulator REGISTER over the
// whole key loop (register <-> kw), d relh[q][2t+b] is the per-block sum.
#ifdef BSG_DIAG_STAMPS_ATTN
__device__ unsigned long long bsg_attn_stamps[8];
#define ATTN_STAMP(i) do { if (threadIdx.x == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&bsg_attn_stamps[i], (unsigned long long)(t_ - st_prev)); st_prev = t_; } } while (0)
#else
#define ATTN_STAMP(i) do {} while (0)
#endif
// f32: 6 tiles of 16 KB + the relh table = one workgroup per CU, i.e. one wave per SIMD anyway -> the whole 512-register
// file is this wave's (the 256-register bound of the 16-bit kernels made the f32 forms spill: 60 B exact, 100 B x3)
template <typename T, bool TR, bool X3 = false>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 1 : 2) void attn_bwd_dq_kernel(AttnArgs a) {
#ifdef BSG_DIAG_STAMPS_ATTN
  long long st_prev = __builtin_amdgcn_s_memtime();
#endif
  typedef typename Traits<T>::Chunk Chunk;
  typedef AttnK<T> C;
  static_assert(TR == (sizeof(T) == 2), "16-bit dtypes take the transposing-read path, f32 the transposed copies");
  constexpr int NTILE = TR ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [buf][K | V | KT (not TR)][TILE] | relh tables [4][32][HS]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, head, s;
  attn_block_ids((a.N + 127) / 128, a.nh, a.S, bx, head, s);
  const int q0 = bx * 128 + wave * 32;
  const int q = min(q0 + col, a.N - 1);
  const long sh = (long)s * a.nh + head;
  const int npad = a.hp * 32;
  const char* kbase = (const char*)a.k + ((long)s * a.N * a.ld + head * 64) * sizeof(T);
  const char* vbase = (const char*)a.v + ((long)s * a.N * a.ld + head * 64) * sizeof(T);
  const char* ktbase = (const char*)a.kt + sh * 64 * npad * sizeof(T);

  Chunk qf[C::KS_D], dof[C::KS_D];
  {
    const char* qrow = (const char*)a.q + (((long)s * a.N + q) * a.ld + head * 64) * sizeof(T);
    const char* drow = (const char*)a.dout + (((long)s * a.N + q) * a.ldo + head * 64) * sizeof(T);
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) {
      qf[ks] = *(const Chunk*)(qrow + (2 * ks + h) * 16);
      if constexpr (X3 && sizeof(T) == 4) qf[ks] = x3_unpack(qf[ks]);  // x3: qkv holds packed (hi, lo) pairs; dO is plain f32
      dof[ks] = *(const Chunk*)(drow + (2 * ks + h) * 16);
    }
  }
  // rel-pos bias of this wave's queries, computed in-kernel; d relh overwrites the relh table entry by entry
  f32x16 rwv;
  float drw[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) drw[i] = 0.f;
  const int qh = q / a.wp, qw = q % a.wp;
  float* relh_q = (float*)(smem + 2 * NTILE * C::TILE) + (wave * 32 + col) * relh_stride(a.hp);
  const bool active = q0 < a.N;  // wave-uniform: a wave past the last query only helps with the DMA and the barriers
  if (active)
    relpos_wave_tables<T>(qf, a.rel_cat, a.hp, a.wp, qh, qw, q0 / a.wp, min(q0 + 31, a.N - 1) / a.wp, 1.0f / a.scale,
                          (float*)(smem + wave * 8192), relh_q - col * relh_stride(a.hp), rwv, lane);
  ATTN_STAMP(0);  // loads of q / dO + rel-pos tables
  // publish the key-major copies the dK/dV kernel streams (lane = key there): 128-byte row segments per half-wave
  if (q0 + col < min((a.N + 127) & ~127, npad)) {
    const bool real = q0 + col < a.N;  // tail columns up to the dK/dV tile boundary (64 or 128 queries): -inf bias, i.e. P = 0
    float* wT = a.relwT + sh * 32 * npad + q0 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) wT[(long)acc32_row(r, h) * npad] = real ? rwv[r] : -INFINITY;
    // row bias pre-folded with the softmax statistics: relhT[kh][q] = relh[q][kh] * c2 - lse2[q], so that the dK/dV
    // kernel's exponent is one fma on top of an accumulator that starts from relwT alone
    float* hT = a.relhT + sh * a.hp * npad + q0 + col;
    const float c2p = a.scale * 1.44269504088896340736f, lsep = a.lse2[sh * npad + q];
    for (int kh = h; kh < a.hp; kh += 2) hT[(long)kh * npad] = real ? fmaf(relh_q[kh], c2p, -lsep) : 0.f;
    if (!real && h == 0) {  // tail columns of the per-query statistics: finite whatever the workspace held before
      a.lse2[sh * npad + q0 + col] = 0.f;
      a.delta[sh * npad + q0 + col] = 0.f;
    }
  }
  ATTN_STAMP(1);  // publish relwT / relhT
  __syncthreads();  // scratch reads done before any wave's DMA lands in the tile area
  const float c2 = a.scale * 1.44269504088896340736f;
  const float lse = a.lse2[sh * npad + q];
  // delta[q] = sum_d dO[q][d] O[q][d]: each half-wave lane holds half of the row; published for the dK/dV kernel
  float dl = 0.f;
  {
    const char* orow = (const char*)a.out + (((long)s * a.N + q) * a.ldo + head * 64) * sizeof(T);
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) {
      const Chunk oc = *(const Chunk*)(orow + (2 * ks + h) * 16);
#pragma unroll
      for (int j = 0; j < Traits<T>::EPC; ++j) dl += to_f32(dof[ks][j]) * to_f32(oc[j]);
    }
    dl += __shfl_xor(dl, 32, 64);
    if (h == 0 && q0 + col < a.N) a.delta[sh * npad + q] = -dl;  // negated: the dK/dV kernel starts its dP accumulator from it
  }
  if ((bx + 1) * 128 <= a.dq_begin || (a.dq_end > 0 && bx * 128 >= a.dq_end)) {
    // workgroup-uniform: nobody reads dq of these queries (bsg_backward_rows: their dO rows are exactly zero in the block of the
    // top tap, so dq is; block 0 feeds the prompt half of the canvas only) -- the tables are out, the rows are zeroed
    if (q0 + col < a.N) {
      T* orow = (T*)a.dq + ((long)s * a.N + q) * a.ld + head * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) *(typename Traits<T>::Vec4*)(orow + 32 * d + 8 * i + 4 * h) = pack4<T>(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  f32x16 dqt[2], ndl;  // ndl: loop-invariant initial accumulator of dP^T (all entries -delta[q])
#pragma unroll
  for (int i = 0; i < 16; ++i) { dqt[0][i] = 0.f; dqt[1][i] = 0.f; ndl[i] = -dl; }

  const int nt = a.hp >> 1;
  unsigned koff[TileDma<T>::IPW], ktoff[TileDma<T>::IPW];
  const long es = sizeof(T), kstep = 2L * a.wp * a.ld * es;
  dma_tile_offsets<T>(koff, wave, lane, [&](int r) { return (long)slot_token(r >> 5, r & 31, a.wp) * a.ld * es; });
  if constexpr (!TR) dma_tile_offsets<T>(ktoff, wave, lane, [&](int r) { return (long)r * npad * es; });
  auto issue = [&](int t, int buf) {
    char* k_l = smem + buf * NTILE * C::TILE;
    dma_tile_issue<T>(k_l, wave, kbase + t * kstep, koff);
    dma_tile_issue<T>(k_l + C::TILE, wave, vbase + t * kstep, koff);
    if constexpr (!TR) dma_tile_issue<T>(k_l + 2 * C::TILE, wave, ktbase + t * 64 * es, ktoff);
  };

  ATTN_STAMP(2);  // delta, DMA offsets
  f32x2 rh_next = f32x2{relh_q[0], relh_q[1]};
  // X3: the loop-invariant Q / dO fragments split once; the raw f32 copies die here
  constexpr bool X3f = X3 && sizeof(T) == 4;
  typedef typename std::conditional<X3f, X3Frag8, Chunk>::type LFrag;
  LFrag qx[X3f ? C::KS_D / 2 : C::KS_D], dox[X3f ? C::KS_D / 2 : C::KS_D];
  if constexpr (X3f) {
#pragma unroll
    for (int j = 0; j < C::KS_D / 2; ++j) { qx[j] = x3_frag8(qf[2 * j], qf[2 * j + 1]); dox[j] = x3_frag8(dof[2 * j], dof[2 * j + 1]); }
  } else {
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) { qx[ks] = qf[ks]; dox[ks] = dof[ks]; }
  }
#pragma unroll
  for (int i = 0; i < (int)(sizeof(qx) / sizeof(qx[0])); ++i) { reg_consume(qx[i]); reg_consume(dox[i]); }  // waits for the Q / dO loads stay out of the loop
  reg_consume(lse);
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const f32x2 rh = rh_next;
#if BSG_DIAG_DQ == 4
    if (t == 0) { wait_vm0(); __syncthreads(); }
    if (t + 1 < nt) rh_next = f32x2{relh_q[2 * t + 2], relh_q[2 * t + 3]};
#else
    wait_vm0();
    __syncthreads();
    if (t + 1 < nt) {
      rh_next = f32x2{relh_q[2 * t + 2], relh_q[2 * t + 3]};
      issue(t + 1, buf ^ 1);
    }
#endif
    if (!active) continue;
    const char* k_l = smem + buf * NTILE * C::TILE;
    const char* v_l = k_l + C::TILE;
    const char* kt_l = k_l + 2 * C::TILE;  // !TR only
    float drh[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x16 st = rwv, dp = ndl;  // S^T accumulator starts at relw (column bias), dP^T at -delta
      if constexpr (X3f) {
#pragma unroll
        for (int j = 0; j < C::KS_D / 2; ++j) {
          mma32_x3_pair(st, x3_frag8_pk(lds_chunk<T>(k_l, 32 * b + col, 4 * j + h), lds_chunk<T>(k_l, 32 * b + col, 4 * j + 2 + h)), qx[j]);
          mma32_x3_pair(dp, x3_frag8_pk(lds_chunk<T>(v_l, 32 * b + col, 4 * j + h), lds_chunk<T>(v_l, 32 * b + col, 4 * j + 2 + h)), dox[j]);
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_D; ++ks) {
#if BSG_DIAG_DQ == 3
        mm32<false>(st, dof[(ks + b) & 3], qf[ks]);
        mm32<false>(dp, qf[(ks + b) & 3], dof[ks]);
#elif BSG_DIAG_DQ == 2
        st[ks] += to_f32(lds_chunk<T>(k_l, 32 * b + col, 2 * ks + h)[0]);
        dp[ks] += to_f32(lds_chunk<T>(v_l, 32 * b + col, 2 * ks + h)[0]);
#else
        mm32<false>(st, lds_chunk<T>(k_l, 32 * b + col, 2 * ks + h), qx[ks]);
        mm32<false>(dp, lds_chunk<T>(v_l, 32 * b + col, 2 * ks + h), dox[ks]);
#endif
      }
      }
      const float nb = fmaf(rh[b], c2, -lse);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#if BSG_DIAG_DQ == 1
        const float ds = st[r] + dp[r];
        st[r] = ds;
        sum = nb;
#else
        const float p = __builtin_amdgcn_exp2f(fmaf(st[r], c2, nb));  // 0 on padded key slots (bias -inf)
        const float ds = p * dp[r];
        st[r] = ds;
        drw[r] += ds;
        sum += ds;
#endif
      }
      drh[b] = sum + __shfl_xor(sum, 32, 64);
      if constexpr (X3f) {
#pragma unroll
        for (int j = 0; j < C::KS_B / 2; ++j) {
          const X3Frag8 db8 = x3_frag8(acc_chunk(st, 2 * j, T()), acc_chunk(st, 2 * j + 1, T()));
#pragma unroll
          for (int d = 0; d < 2; ++d)
            mma32_x3_pair(dqt[d], x3_frag8_pk(lds_perm_chunk(kt_l, 32 * d + col, b, 2 * j, h, T()),
                                              lds_perm_chunk(kt_l, 32 * d + col, b, 2 * j + 1, h, T())), db8);
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_B; ++ks) {
        const Chunk db_ = acc_chunk(st, ks, T());
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#if BSG_DIAG_DQ == 3
          if constexpr (TR) mm32<false>(dqt[d], qf[(ks + d) & 3], db_);
#elif BSG_DIAG_DQ == 2
          if constexpr (TR) dqt[d][ks] += to_f32(lds_tr_chunk<T>(k_l, d, b, ks, lane)[0]) * to_f32(db_[0]);
#else
          if constexpr (TR) mm32<false>(dqt[d], lds_tr_chunk<T>(k_l, d, b, ks, lane), db_);
#endif
          else mm32<false>(dqt[d], lds_perm_chunk(kt_l, 32 * d + col, b, ks, h, T()), db_);
        }
      }
      }
    }
    if (h == 0) { relh_q[2 * t] = drh[0]; relh_q[2 * t + 1] = drh[1]; }  // both were read a tile ago
  }
  ATTN_STAMP(3);  // key loop
  // rel-pos gradient straight into dq: d relw goes to a [32][33] image in this wave's slice of the (now idle) tile area
  __syncthreads();
  if (!active) return;
  float* image_w = (float*)(smem + wave * 8192);
#pragma unroll
  for (int r = 0; r < 16; ++r) image_w[col * 33 + acc32_row(r, h)] = drw[r];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  f32x16 racc[2];
  relpos_wave_bwd<T>(racc, a.rel_catT, a.hp, a.wp, qh, qw, q0 / a.wp, min(q0 + 31, a.N - 1) / a.wp,
                     relh_q - col * relh_stride(a.hp), image_w, lane);
  ATTN_STAMP(4);  // rel-pos gradient
  if (q0 + col < a.N) {
    T* orow = (T*)a.dq + ((long)s * a.N + q) * a.ld + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(typename Traits<T>::Vec4*)(orow + 32 * d + 8 * i + 4 * h) =
            pack4<T>(fmaf(dqt[d][4 * i], a.scale, racc[d][4 * i]), fmaf(dqt[d][4 * i + 1], a.scale, racc[d][4 * i + 1]),
                     fmaf(dqt[d][4 * i + 2], a.scale, racc[d][4 * i + 2]), fmaf(dqt[d][4 * i + 3], a.scale, racc[d][4 * i + 3]));
  }
#ifdef BSG_DIAG_STAMPS_ATTN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(5);  // dq store
  if (threadIdx.x == 0) atomicAdd(&bsg_attn_stamps[7], 1ull);
#endif
}

// ------------------------------------------------------------------------------------- backward: dK, dV
// Key-stationary: each wave owns one key tile (64 slots = 2 grid rows) and keeps dK^T, dV^T for it in
// accumulators while the workgroup streams query tiles through LDS: 64 CONSECUTIVE tokens each (queries are the
// register / contraction index here, so they need no row padding: ceil(N / 64) tiles instead of Hp / 2).
// Here the KEY is on the lane (S = Q K^T un-swapped), so P and dS accumulators are the B operands of
// dV^T += dO^T P and dK^T += Q^T dS.  Bias / lse2 / delta arrive in query-slot-major ("T") layouts:
//   relwT [S][nh][32 kw][Hp*32],  relhT [S][nh][Hp key rows][Hp*32],  lse2/delta [S][nh][Hp*32], column = token;
//   columns N .. 64 ceil(N/64) - 1 of relwT hold -inf (P = 0 there), of the others 0 (all written by the dQ kernel).
struct AttnBwdKvArgs {
  const void* k; const void* v; const void* q; const void* dout;  // row-major T (q/k/v with ld, dout with ldo)
  const void* qt; const void* dot;                                // [S][nh][64][Hp*32], column = token (f32 path)
  long ld, ldo;
  const float* relwT; const float* relhT; const float* lse2; const float* delta;  // delta = MINUS rowsum(dO * O); relhT = relh * c2 - lse2
  void* dk; void* dv;  // T, row stride ld
  int S, nh, N, hp, wp;
  float scale;
  int kr_begin, kr_count;  // attention_kv4.hpp only: the key rows [kr_begin, kr_begin + kr_count) this launch covers
  // row windows (bsg_backward_rows): queries < q_begin (a multiple of 64) contribute exact zeros and are not streamed; only the
  // first key_rows key rows are wanted (0 = all Hp) -- rows past them (rounded up to a launch's row group) are NOT written
  int q_begin, key_rows;
};

// LDS-DMA of a ROWS x RB-byte tile by NW waves (1 KiB per wave-instruction), optional chunk swizzle.
template <int RB, int ROWS, int NW, bool SWZ, typename RowSrc>
DEVI void dma_rows(char* lds_tile, int wave, int lane, RowSrc row_src) {
  constexpr int CPR = RB / 16, RPI = 64 / CPR, NI = ROWS / RPI;
#pragma unroll
  for (int j0 = 0; j0 < NI; j0 += NW) {
    const int j = j0 + wave;
    if (NI % NW == 0 || j < NI) {
      const int r = j * RPI + lane / CPR, p = lane % CPR;
      const char* src = row_src(r) + ((SWZ ? (p ^ swz<RB>(r)) : p) << 4);
      glds16_asm(src, lds_tile + j * 1024);
    }
  }
}

template <typename T, bool TR, int QT = 64> struct DkvK {
  static constexpr int TILE = QT * AttnK<T>::RB;  // QT query rows
  static constexpr int NTILE = TR ? 2 : 4;
  static constexpr int RW = 32 * QT * 4, STATS = 16 * QT * 4;
  static constexpr int STAGE = NTILE * TILE + RW + STATS;  // Q | dO | (QT | dOT) | relwT [32][QT f32] | stats [16][QT f32]
};

// Eight waves per workgroup, each owning ONE grid row of keys (32 slots): dK^T, dV^T of that row stay in 64
// accumulator registers, so two waves fit per SIMD.  The per-query statistics (lse2, delta, rel-pos rows) ride the
// same LDS-DMA stream as the Q / dO tiles instead of occupying 128 registers.
// QT = queries per streamed tile (64 or 128): a tile costs one workgroup barrier + one DMA wait for all 8 waves, so the
// 128-query form halves the synchronisations (112 KB of LDS for the two stages; 16-bit dtypes only).
// NW = waves per workgroup = key rows per workgroup.  8 (round 1): one workgroup per CU, two waves per SIMD that meet at the SAME
// barrier every query tile, so both are in their MFMA phase, then both in their exp phase.  4 (round 4, 16-bit dtypes): two
// independent workgroups per CU (launch bound of two waves per SIMD, 56 KB of LDS each at QT = 64) -- the two waves of a SIMD
// belong to different workgroups, drift apart, and one's vector phase runs under the other's MFMAs.
template <typename T, bool TR, int QT = 64, bool X3 = false, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_kernel(AttnBwdKvArgs a) {
  typedef typename Traits<T>::Chunk Chunk;
  typedef AttnK<T> C;
  typedef DkvK<T, TR, QT> K_;
  static_assert(TR == (sizeof(T) == 2), "16-bit dtypes take the transposing-read path, f32 the transposed copies");
  static_assert(QT == 64 || (QT == 128 && TR), "128-query tiles: transposing-read path only");
  constexpr int RB = C::RB, STAGE = K_::STAGE, NTILE = K_::NTILE, QB = QT * 4;  // QB: bytes of one f32 table row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, head, s;
  const int key_rows = a.key_rows > 0 && a.key_rows < a.hp ? a.key_rows : a.hp;  // the launcher's grid covers these rows (rounded up to NW)
  attn_block_ids((key_rows + NW - 1) / NW, a.nh, a.S, bx, head, s);
  const int nt = (a.N + QT - 1) / QT, t0 = a.q_begin / QT;  // queries below q_begin contribute exact zeros (their dO rows are zero)
  const int kr0 = bx * NW;
  const bool wave_valid = kr0 + wave < a.hp;
  const int kr = min(kr0 + wave, a.hp - 1);  // this wave's key grid row (clamped duplicates do not store)
  const long sh = (long)s * a.nh + head;
  const int npad = a.hp * 32;
  const char* qbase = (const char*)a.q + ((long)s * a.N * a.ld + head * 64) * sizeof(T);
  const char* dobase = (const char*)a.dout + ((long)s * a.N * a.ldo + head * 64) * sizeof(T);
  const char* qtbase = (const char*)a.qt + sh * 64 * npad * sizeof(T);
  const char* dotbase = (const char*)a.dot + sh * 64 * npad * sizeof(T);

  Chunk kf[C::KS_D], vf[C::KS_D];
  {
    const long tok = (long)s * a.N + slot_token(kr, col, a.wp);
    const char* krow = (const char*)a.k + (tok * a.ld + head * 64) * sizeof(T);
    const char* vrow = (const char*)a.v + (tok * a.ld + head * 64) * sizeof(T);
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) {
      kf[ks] = *(const Chunk*)(krow + (2 * ks + h) * 16);
      vf[ks] = *(const Chunk*)(vrow + (2 * ks + h) * 16);
      if constexpr (X3 && sizeof(T) == 4) { kf[ks] = x3_unpack(kf[ks]); vf[ks] = x3_unpack(vf[ks]); }  // x3: packed (hi, lo) pairs
    }
  }
  // X3: the loop-invariant K / V fragments split once; the raw f32 copies die here
  constexpr bool X3f = X3 && sizeof(T) == 4;
  typedef typename std::conditional<X3f, X3Frag8, Chunk>::type LFrag;
  LFrag kx[X3f ? C::KS_D / 2 : C::KS_D], vx[X3f ? C::KS_D / 2 : C::KS_D];
  if constexpr (X3f) {
#pragma unroll
    for (int j = 0; j < C::KS_D / 2; ++j) { kx[j] = x3_frag8(kf[2 * j], kf[2 * j + 1]); vx[j] = x3_frag8(vf[2 * j], vf[2 * j + 1]); }
  } else {
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) { kx[ks] = kf[ks]; vx[ks] = vf[ks]; }
  }
  // consumed here, before the loop: hipcc then waits for the K / V loads NOW and not at their first use inside the loop, where
  // its `s_waitcnt vmcnt(0)` would also wait for the (to hipcc invisible) LDS-DMA of the next query tile
#pragma unroll
  for (int i = 0; i < (int)(sizeof(kx) / sizeof(kx[0])); ++i) { reg_consume(kx[i]); reg_consume(vx[i]); }
  const float c2 = a.scale * 1.44269504088896340736f;
  const bool key_valid = col < a.wp;
  f32x16 dkt[2], dvt[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkt[0][i] = dkt[1][i] = dvt[0][i] = dvt[1][i] = 0.f; }

  auto issue = [&](int t, int buf) {
    char* base = smem + buf * STAGE;
    dma_rows<RB, QT, NW, true>(base, wave, lane, [&](int r) {
      return qbase + (long)min(t * QT + r, a.N - 1) * a.ld * sizeof(T);
    });
    dma_rows<RB, QT, NW, true>(base + K_::TILE, wave, lane, [&](int r) {
      return dobase + (long)min(t * QT + r, a.N - 1) * a.ldo * sizeof(T);
    });
    if constexpr (!TR) {
      dma_rows<RB, 64, NW, true>(base + 2 * K_::TILE, wave, lane,
                                [&](int r) { return qtbase + ((long)r * npad + t * 64) * sizeof(T); });
      dma_rows<RB, 64, NW, true>(base + 3 * K_::TILE, wave, lane,
                                [&](int r) { return dotbase + ((long)r * npad + t * 64) * sizeof(T); });
    }
    dma_rows<QB, 32, NW, true>(base + NTILE * K_::TILE, wave, lane, [&](int r) {
      return (const char*)(a.relwT + (sh * 32 + r) * npad + t * QT);
    });
    dma_rows<QB, 16, NW, false>(base + NTILE * K_::TILE + K_::RW, wave, lane, [&](int r) {
      const float* p = r == 0 ? a.lse2 + sh * npad
                     : r == 1 ? a.delta + sh * npad
                              : a.relhT + (sh * a.hp + min(kr0 + max(r - 2, 0), a.hp - 1)) * npad;
      return (const char*)(p + t * QT);
    });
  };

  issue(t0, t0 & 1);
  for (int t = t0; t < nt; ++t) {
    const int buf = t & 1;
    wait_vm0();
    __syncthreads();
    if (t + 1 < nt) issue(t + 1, buf ^ 1);
    const char* q_l = smem + buf * STAGE;
    const char* do_l = q_l + K_::TILE;
    const char* qt_l = q_l + 2 * K_::TILE;   // !TR only
    const char* dot_l = q_l + 3 * K_::TILE;  // !TR only
    const char* rw_l = q_l + NTILE * K_::TILE + col * QB;
    const char* st_l = q_l + NTILE * K_::TILE + K_::RW;
#pragma unroll
    for (int qa = 0; qa < QT / 32; ++qa) {
      f32x16 st, dp;
      f32x4 rhl4[4];  // relh * c2 - lse2 of this wave's key row, per query slot (pre-folded by the dQ kernel)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = qa * 8 + 2 * i + h;  // 16-byte chunk holding query slots qa*32 + 8i + 4h .. +3
        const f32x4 dl4 = *(const f32x4*)(st_l + QB + c * 16);
        rhl4[i] = *(const f32x4*)(st_l + (2 + wave) * QB + c * 16);
        const f32x4 rw = *(const f32x4*)(rw_l + ((c ^ (col & 15)) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) { st[4 * i + j] = rw[j]; dp[4 * i + j] = dl4[j]; }  // S starts from relwT, dP from -delta
      }
      if constexpr (X3f) {
#pragma unroll
        for (int j = 0; j < C::KS_D / 2; ++j) {
          mma32_x3_pair(st, x3_frag8_pk(lds_chunk<T>(q_l, 32 * qa + col, 4 * j + h), lds_chunk<T>(q_l, 32 * qa + col, 4 * j + 2 + h)), kx[j]);
          mma32_x3_pair(dp, x3_frag8(lds_chunk<T>(do_l, 32 * qa + col, 4 * j + h), lds_chunk<T>(do_l, 32 * qa + col, 4 * j + 2 + h)), vx[j]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < C::KS_D; ++ks) {
          mm32<false>(st, lds_chunk<T>(q_l, 32 * qa + col, 2 * ks + h), kx[ks]);
          mm32<false>(dp, lds_chunk<T>(do_l, 32 * qa + col, 2 * ks + h), vx[ks]);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // padded key lanes / padded query slots carry relwT = -inf: P = 0 there without any masking
        const float p = __builtin_amdgcn_exp2f(fmaf(st[r], c2, rhl4[r >> 2][r & 3]));
        st[r] = p;                                   // P
        dp[r] = p * dp[r];                           // dS = P (dP - delta)
      }
      if constexpr (X3f) {
#pragma unroll
        for (int j = 0; j < C::KS_B / 2; ++j) {
          const X3Frag8 pb = x3_frag8(acc_chunk(st, 2 * j, T()), acc_chunk(st, 2 * j + 1, T()));
          const X3Frag8 dsb = x3_frag8(acc_chunk(dp, 2 * j, T()), acc_chunk(dp, 2 * j + 1, T()));
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            mma32_x3_pair(dvt[d], x3_frag8(lds_perm_chunk(dot_l, 32 * d + col, qa, 2 * j, h, T()),
                                           lds_perm_chunk(dot_l, 32 * d + col, qa, 2 * j + 1, h, T())), pb);
            mma32_x3_pair(dkt[d], x3_frag8_pk(lds_perm_chunk(qt_l, 32 * d + col, qa, 2 * j, h, T()),
                                              lds_perm_chunk(qt_l, 32 * d + col, qa, 2 * j + 1, h, T())), dsb);
          }
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < C::KS_B; ++ks) {
        const Chunk pb = acc_chunk(st, ks, T());
        const Chunk dsb = acc_chunk(dp, ks, T());
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          if constexpr (TR) {
            mm32<false>(dvt[d], lds_tr_chunk<T>(do_l, d, qa, ks, lane), pb);
            mm32<false>(dkt[d], lds_tr_chunk<T>(q_l, d, qa, ks, lane), dsb);
          } else {
            mm32<false>(dvt[d], lds_perm_chunk(dot_l, 32 * d + col, qa, ks, h, T()), pb);
            mm32<false>(dkt[d], lds_perm_chunk(qt_l, 32 * d + col, qa, ks, h, T()), dsb);
          }
        }
      }
      }
    }
  }
  if (wave_valid && key_valid) {
    const long tok = (long)s * a.N + kr * a.wp + col;
    T* dkrow = (T*)a.dk + tok * a.ld + head * 64;
    T* dvrow = (T*)a.dv + tok * a.ld + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(typename Traits<T>::Vec4*)(dkrow + 32 * d + 8 * i + 4 * h) =
            pack4<T>(dkt[d][4 * i] * a.scale, dkt[d][4 * i + 1] * a.scale, dkt[d][4 * i + 2] * a.scale,
                     dkt[d][4 * i + 3] * a.scale);
        *(typename Traits<T>::Vec4*)(dvrow + 32 * d + 8 * i + 4 * h) =
            pack4<T>(dvt[d][4 * i], dvt[d][4 * i + 1], dvt[d][4 * i + 2], dvt[d][4 * i + 3]);
      }
  }
}
