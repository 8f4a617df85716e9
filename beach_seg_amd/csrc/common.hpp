// Shared device helpers for the gfx950 (CDNA4) kernels of the SegGPT hot path.
// One source serves two activation dtypes: T = __bf16 (throughput mode, bf16 MFMA) and T = float
// (parity mode, exact-f32 MFMA).  Both see the same BYTE geometry: a fragment is one 16-byte chunk
// per lane (8 bf16 or 4 f32); an LDS tile row is 128 bytes = 8 chunks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16_t;  // third activation dtype: IEEE half (11 significant bits: 8x less operand round-off than bf16 at the
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;  // same MFMA rate; the dgrad chain runs on a power-of-two
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;  // scaled gradient, seggpt_api.hip)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DEVI __device__ __forceinline__

template <typename T> struct Traits;
template <> struct Traits<bf16_t> {
  typedef bf16x8 Chunk;
  typedef bf16x4 Vec4;
  static constexpr int EPC = 8;  // elements per 16-byte chunk
};
template <> struct Traits<f16_t> {
  typedef f16x8 Chunk;
  typedef f16x4 Vec4;
  static constexpr int EPC = 8;
};
template <> struct Traits<float> {
  typedef f32x4 Chunk;
  typedef f32x4 Vec4;
  static constexpr int EPC = 4;
};

// ---- MFMA wrappers.  16x16 tile: lane l holds row/col (l & 15) and chunk (l >> 4) of a 4-chunk K step.
//      32x32 tile: lane l holds row/col (l & 31) and chunk (l >> 5) of a 2-chunk K step.
//      f32 path: element j of every lane's chunk forms one k-slice of an exact-f32 MFMA, so the fragment
//      addressing is identical to the bf16 path.
DEVI void mma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
DEVI void mma16(f32x4& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
DEVI void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
}
DEVI void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
DEVI void mma32(f32x16& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
DEVI void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

// 32x32 accumulator: register r of lane l is row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column l & 31.
DEVI int acc32_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- conversions
DEVI float to_f32(float x) { return x; }
DEVI float to_f32(bf16_t x) { return (float)x; }
DEVI float to_f32(f16_t x) { return (float)x; }
template <typename T> DEVI T from_f32(float x);
template <> DEVI float from_f32<float>(float x) { return x; }
template <> DEVI bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // v_cvt_pk_bf16_f32: RNE, NaN-safe
template <> DEVI f16_t from_f32<f16_t>(float x) { return (f16_t)x; }     // v_cvt_pk_f16_f32 (gfx950): RNE

template <typename T> DEVI typename Traits<T>::Vec4 pack4(float a, float b, float c, float d);
template <> DEVI f32x4 pack4<float>(float a, float b, float c, float d) { return f32x4{a, b, c, d}; }
template <> DEVI bf16x4 pack4<bf16_t>(float a, float b, float c, float d) {
  return bf16x4{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
}
template <> DEVI f16x4 pack4<f16_t>(float a, float b, float c, float d) {
  return f16x4{(f16_t)a, (f16_t)b, (f16_t)c, (f16_t)d};
}

// two floats -> one 32-bit word of two 16-bit values (16-bit dtypes only)
template <typename T> DEVI unsigned pack2(float a, float b);
template <> DEVI unsigned pack2<bf16_t>(float a, float b) {
  const bf16x2 v = bf16x2{(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(unsigned, v);
}
template <> DEVI unsigned pack2<f16_t>(float a, float b) {
  const f16x2 v = f16x2{(f16_t)a, (f16_t)b};
  return __builtin_bit_cast(unsigned, v);
}

// chunk from 8 (bf16) or 4 (f32) floats held in two f32x4 (second ignored for f32)
DEVI void chunk_from_f32(bf16x8& c, const float* v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = (bf16_t)v[j];
}
DEVI void chunk_from_f32(f16x8& c, const float* v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = (f16_t)v[j];
}

// ---- async global -> LDS copy of one 16-byte chunk per lane (LDS-DMA).  `lds_wave_base` must be
//      wave-uniform; lane l lands at lds_wave_base + 16 * l.
DEVI void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
DEVI void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same copy as an asm statement, for loops that prefetch the NEXT tile while they read the current one (round 4).  hipcc
// tracks the builtin form as a pending LDS write and cannot tell the tile it lands in from the tile being read (one dynamic
// LDS array): it puts `s_waitcnt vmcnt(0)` in front of the first LDS read behind the copy -- in every attention kernel that
// was 60-190 instructions after the issue, i.e. each tile waited for the prefetch it had just requested (wait_any 0.31-0.41 of
// the wave cycles, profiles/r3_pmc_summary.json).  The asm form is invisible to that bookkeeping; the caller waits with
// wait_vm0() + a barrier before the tile is read, which the loops did anyway.  M0 (the LDS base of the copy) is saved and
// restored inside the statement (hipcc does not preserve it around asm; guide section 5.7).
DEVI void glds16_asm(const void* gptr, const void* lds_wave_base) {
  const unsigned lds_uniform = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_uniform) : "memory");
}

// ---- LDS reads the compiler does not count (software-pipelined fragment reads at one wave per SIMD: hipcc's own
//      waits collapse to lgkmcnt(0) right behind the prefetch).  The caller owns the wait: `lds_wait<N>()` leaves the N
//      youngest LDS operations in flight (LDS reads return in order) and fences the scheduler so that no consumer moves
//      above it (guide section 5.7 item 1, form iii).
DEVI unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p;
}
DEVI f32x4 lds_read16_nowait(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF> DEVI f32x4 lds_read16_nw(unsigned addr) {  // ds_read_b128 with an immediate byte offset
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
template <int OFF> DEVI f32x2 lds_read8tr_nw(unsigned addr) {  // transposing 4 x 16-bit read (see attention.hpp lds_tr_chunk)
  f32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
  return v;
}
// global -> register load / register -> LDS store that hipcc does not count (register-staged operand pipeline of
// gemm_nt_kernel_v4): SGPR base + 32-bit per-lane offset; the caller waits with vmcnt / lgkmcnt before use.
DEVI const char* uniform_ptr(const char* p) {  // makes wave-uniformity provable to hipcc (an "s" asm operand needs it)
  const unsigned long v = (unsigned long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long)hi << 32) | lo);
}
DEVI f32x4 global_load16_nw(const void* sbase, unsigned voff) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
// the same with a wave-uniform base, a 32-bit per-lane byte offset (saddr form: no 64-bit per-lane address arithmetic) and the
// LDS destination as a wave-uniform byte address (lds_addr() of the piece): nothing here costs a vector instruction
DEVI void glds16_asm_s(const char* sbase_uniform, unsigned voff, unsigned lds_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase_uniform), "s"(lds_uniform) : "memory");
}
template <int OFF> DEVI void lds_write16_nw(unsigned addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2\n\ts_nop 1" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- loop indices usable as asm immediates
template <int I, int N, typename F> DEVI void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int N> DEVI void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- wave-level reductions (64 lanes)
DEVI float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// erf GELU (HF ACT2FN["gelu"] == torch F.gelu default) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|abs error| <= 1.5e-7, i.e. at fp32 rounding level): one v_rcp + one v_exp + 6 FMAs instead of the ~40
// instruction ocml erff; the exp(-x^2/2) is shared between the cdf and the pdf of gelu'.
DEVI float erf_half_exp(float ax, float e) {  // erf(ax / sqrt2) given e = exp(-ax^2 / 2), ax >= 0
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return fmaf(-p * t, e, 1.0f);
}
DEVI float gelu_f(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2/2)
  const float er = copysignf(erf_half_exp(ax, e), x);
  return 0.5f * x * (1.0f + er);
}
// gelu(x) and gelu'(x) from one exp and one erf evaluation (forward epilogue that also saves the derivative)
DEVI void gelu_both_f(float x, float& y, float& dy) {
#ifdef BSG_DIAG_NOGELU  // timing-only build: prices the GELU epilogue math
  y = x; dy = 1.f; return;
#endif
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  const float cdf = 0.5f * (1.0f + copysignf(erf_half_exp(ax, e), x));
  y = x * cdf;
  dy = fmaf(0.39894228040143267794f * x, e, cdf);
}
// The same arithmetic on element PAIRS (v_pk_mul / v_pk_fma / v_pk_add: one issue slot per two elements; exp2 / rcp stay per
// element): bit-identical results.  For epilogues that run with nothing else on the SIMD -- a lone wave issues one vector
// instruction per 4 cycles whatever its width, so the packed forms double its rate (beside MFMAs they are an anti-lever).
DEVI f32x2 fma2(const f32x2& a, const f32x2& b, const f32x2& c) { return __builtin_elementwise_fma(a, b, c); }
DEVI void gelu_both_pk(const f32x2& x, f32x2& y, f32x2& dy) {
#ifdef BSG_DIAG_NOGELU
  y = x; dy = f32x2{1.f, 1.f}; return;
#endif
  const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])};
  const f32x2 u = (x * f32x2{-0.72134752044448170368f, -0.72134752044448170368f}) * x;
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
  const f32x2 d = fma2(f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f}, ax, f32x2{1.f, 1.f});
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = fma2(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  p = fma2(p, t, f32x2{1.421413741f, 1.421413741f});
  p = fma2(p, t, f32x2{-0.284496736f, -0.284496736f});
  p = fma2(p, t, f32x2{0.254829592f, 0.254829592f});
  const f32x2 er0 = fma2(-p * t, e, f32x2{1.f, 1.f});
  const f32x2 er = f32x2{copysignf(er0[0], x[0]), copysignf(er0[1], x[1])};
  const f32x2 cdf = f32x2{0.5f, 0.5f} * (f32x2{1.f, 1.f} + er);
  y = x * cdf;
  dy = fma2(f32x2{0.39894228040143267794f, 0.39894228040143267794f} * x, e, cdf);
}
// gelu alone on an element pair: the arithmetic of gelu_f on v_pk_* instructions (x * (0.5 (1 + erf)) where gelu_f has
// (0.5 x) (1 + erf): the same bits, a power-of-two factor commutes with the rounding)
DEVI f32x2 gelu_pk(const f32x2& x) {
  const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])};
  const f32x2 u = (x * f32x2{-0.72134752044448170368f, -0.72134752044448170368f}) * x;
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
  const f32x2 d = fma2(f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f}, ax, f32x2{1.f, 1.f});
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = fma2(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  p = fma2(p, t, f32x2{1.421413741f, 1.421413741f});
  p = fma2(p, t, f32x2{-0.284496736f, -0.284496736f});
  p = fma2(p, t, f32x2{0.254829592f, 0.254829592f});
  const f32x2 er0 = fma2(-p * t, e, f32x2{1.f, 1.f});
  const f32x2 er = f32x2{copysignf(er0[0], x[0]), copysignf(er0[1], x[1])};
  return x * (f32x2{0.5f, 0.5f} * (f32x2{1.f, 1.f} + er));
}
DEVI float gelu_grad_f(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  const float er = copysignf(erf_half_exp(ax, e), x);
  return fmaf(0.39894228040143267794f * x, e, 0.5f * (1.0f + er));
}

// XCD-aware remap of a 1-D grid: the dispatcher deals consecutive block ids round-robin over the 8 XCDs,
// so give each XCD a contiguous span of logical tiles (bijective for any grid size).
DEVI int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
