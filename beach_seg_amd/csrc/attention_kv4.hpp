// dK / dV of the fused SegGPT attention, 16-bit dtypes, ONE WAVE PER SIMD (round 4).
//
// The eight-wave kernel of attention.hpp (one key grid row per wave, two waves per SIMD) runs S / dP MFMAs -> exp / dS on the
// vector ALU -> dV / dK MFMAs as one dependent chain per 32 x 32 block, re-reads the Q / dO fragments that its seven neighbours
// also read (1.75 KB of LDS reads per MFMA), and the eight chains meet at a barrier every query tile: mfma_busy 0.44, 41 % of
// the wave cycles parked at a wait (profiles/r3_pmc_summary.json).  Here a workgroup is FOUR waves, one per SIMD, each with the
// whole 512-register file:
//   * a wave owns R = 3 (2) key grid rows: K / V fragments of the rows in registers, dK^T / dV^T in 64 R accumulator registers
//     pinned in the accumulator half of the file (MFMAs through asm with an "+a" operand, as gemm_nt_kernel_v5 does), so the
//     Q / dO fragments and the transposed Q^T / dO^T fragments of a 32-query block are read from LDS ONCE and feed R rows;
//   * the rows of a wave and the query steps form ONE software pipeline of one-MFMA slots (below): the vector work of a row
//     is dealt out under the MFMAs of its neighbours, the LDS reads and the LDS-DMA pieces one to three per slot;
//   * a workgroup owns 4R key rows of one (stream, head) and streams all queries past them in 32-query steps by LDS-DMA (two
//     stages, one barrier per step); the groups of one (stream, head) run on the same XCD, so Q / dO come from its L2.
// Inputs (relwT, relhT = relh c2 - lse2, delta) are the tables the dQ kernel publishes, exactly as for the eight-wave kernel;
// the arithmetic per element and the order of the fp32 accumulation over queries are the same: dK / dV are BIT-IDENTICAL to
// the eight-wave kernel's (tools/attn_probe.py checks it).
//
// This header is compiled in its own translation unit (attention_kv4.hip) with -mllvm -amdgpu-mfma-vgpr-form: see there.
#pragma once
#include "attention.hpp"

// v_mfma_f32_32x32x16 with the accumulator pinned in the accumulator register file ("+a"): hipcc then never moves it.
// The statement is opaque to hipcc's hazard recogniser: `s_nop 1` covers a freshly VALU-written B operand (the P / dS words come
// straight from v_cvt_pk), and nothing but these statements touches an accumulator until kv4_settle().
DEVI void mfma32_agpr(f32x16& acc, const f32x4& a, const f32x4& b, bf16_t) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
DEVI void mfma32_agpr(f32x16& acc, const f32x4& a, const f32x4& b, f16_t) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// a 32x32x16 MFMA (16 passes) needs 18 wait states before its result may be read by anything but the next MFMA of the chain
template <int R> DEVI void kv4_settle(f32x16 (&dk)[R][2], f32x16 (&dv)[R][2]) {
#pragma unroll
  for (int j = 0; j < R; ++j)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dk[j][0]), "+a"(dk[j][1]), "+a"(dv[j][0]), "+a"(dv[j][1]));
}

#ifdef BSG_DIAG_KV4  // diagnostic build: where a query step spends its cycles (wave 0 of every workgroup; shares, not lengths)
__device__ unsigned long long bsg_kv4_stamps[16];
#define KV4_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define KV4_STAMP(i) do {} while (0)
#endif
template <typename T, int R> struct Kv4K {
  static constexpr int QT = 32;                       // queries per streamed step
  static constexpr int TILE = QT * 128;               // Q or dO rows of a step (128 B per row)
  static constexpr int RW = 32 * QT * 4;              // relwT rows [32 kw][QT] f32
  static constexpr int NST = 16;                      // statistics rows [NST][QT] f32: 0 = -delta, 1 + i = relhT of the group's key row i
  static constexpr int STAGE = 2 * TILE + RW + NST * QT * 4;
  static_assert(4 * R + 1 <= NST, "statistics rows");
};

template <typename T, int R>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv4_kernel(AttnBwdKvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit dtypes (transposing LDS reads)");
  static_assert(R >= 2, "the continuous pipeline needs two rows per wave");
  typedef typename Traits<T>::Chunk Chunk;
  typedef AttnK<T> C;
  typedef Kv4K<T, R> K_;
  constexpr int QT = K_::QT, QB = QT * 4, STAGE = K_::STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the host launches this instance over `a.kr_count / (4 R)` whole groups of 4R key rows starting at row a.kr_begin (every
  // row of every wave is real: no row-count branches, so the accumulators never cross a control-flow merge -- with a
  // wave-uniform `break` in the row loop hipcc copied all 192 of them between the register halves on every path)
  const int ngrp = a.kr_count / (4 * R);
  int bx, head, s;
  attn_block_ids(ngrp, a.nh, a.S, bx, head, s);
  const int nq = a.N - a.q_begin;  // queries streamed: [q_begin, N), q_begin a multiple of 64 (an even number of steps)
  const int nt_real = (nq + QT - 1) / QT;
  const int nt = (R & 1) ? (nt_real + 1) & ~1 : nt_real;  // R odd: an even number of steps (the extra one adds exact zeros, see below)
  const int kr0 = a.kr_begin + bx * 4 * R;
  const int krw = kr0 + R * wave;                   // wave w owns key rows krw .. krw + R - 1
  const long sh = (long)s * a.nh + head;
  const int npad = a.hp * 32;
  const char* qbase = (const char*)a.q + (((long)s * a.N + a.q_begin) * a.ld + head * 64) * sizeof(T);
  const char* dobase = (const char*)a.dout + (((long)s * a.N + a.q_begin) * a.ldo + head * 64) * sizeof(T);

  // loop-invariant K / V fragments of this wave's rows
  Chunk kx[R][C::KS_D], vx[R][C::KS_D];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int kr = krw + j;
    const long tok = (long)s * a.N + slot_token(kr, col, a.wp);
    const char* krow = (const char*)a.k + (tok * a.ld + head * 64) * sizeof(T);
    const char* vrow = (const char*)a.v + (tok * a.ld + head * 64) * sizeof(T);
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) {
      kx[j][ks] = *(const Chunk*)(krow + (2 * ks + h) * 16);
      vx[j][ks] = *(const Chunk*)(vrow + (2 * ks + h) * 16);
    }
  }
  // the fragments are consumed here, before the loop: hipcc then waits for their loads NOW and not at their first use inside
  // the loop, where its `s_waitcnt vmcnt(0)` would also wait for the (to hipcc invisible) LDS-DMA of the next query step
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int ks = 0; ks < C::KS_D; ++ks) { reg_consume(kx[j][ks]); reg_consume(vx[j][ks]); }
  const float c2 = a.scale * 1.44269504088896340736f;
  f32x16 dkt[R][2], dvt[R][2];
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkt[j][0][i] = dkt[j][1][i] = dvt[j][0][i] = dvt[j][1][i] = 0.f; }

  // LDS-DMA of a query step: one 1-KiB piece per wave for each of Q, dO and the relwT rows (rows 8 w + lane / 8 of the tile, the
  // source-side chunk swizzle of dma_rows), the statistics rows by waves 0 and 1.  Per-lane offsets are kernel constants; a
  // step advances the three wave-uniform bases by scalar adds.  An LDS-DMA instruction costs the issuing wave 60-100 cycles
  // (guide, cycle constants; 380 cycles per step measured for the four in a row): they go out one per MFMA slot.
  const int prow = 8 * wave + (lane >> 3), pchunk = lane & 7;
  const unsigned swzc = (unsigned)((pchunk ^ swz<128>(prow)) << 4);
  const unsigned off_q = (unsigned)(prow * a.ld * (long)sizeof(T)) + swzc, off_do = (unsigned)(prow * a.ldo * (long)sizeof(T)) + swzc;
  const unsigned off_rw = (unsigned)(prow * (long)npad * 4) + swzc;
  const char* rw_base = (const char*)(a.relwT + sh * 32 * npad + a.q_begin);
  const char* st_src = (const char*)((prow == 0 ? a.delta + sh * npad : a.relhT + (sh * a.hp + min(kr0 + max(prow - 1, 0), a.hp - 1)) * npad) + a.q_begin + 4 * pchunk);
  const long q_step = (long)QT * a.ld * sizeof(T), do_step = (long)QT * a.ldo * sizeof(T);
  const unsigned lds_w = lds_addr(smem) + wave * 1024;  // this wave's piece of stage 0's Q tile (wave-uniform)
  auto issue_piece = [&](int t, int buf, int i) {
    const unsigned lb = lds_w + buf * STAGE;
    if (i < 2) {
      const char* gb = i == 0 ? qbase : dobase;
      const long gl = i == 0 ? a.ld : a.ldo;
      if (t * QT + QT <= nq) {
        glds16_asm_s(gb + t * (i == 0 ? q_step : do_step), i == 0 ? off_q : off_do, lb + i * K_::TILE);
      } else {  // ragged last step: rows past the last query re-read it (their relwT columns hold -inf: P = 0)
        const int r = min(t * QT + prow, nq - 1);
        glds16_asm(gb + (long)r * gl * sizeof(T) + swzc, smem + buf * STAGE + i * K_::TILE + wave * 1024);
      }
    } else if (i == 2) {
      glds16_asm_s(rw_base + (long)t * QB, off_rw, lb + 2 * K_::TILE);
    } else if (wave < K_::NST / 8) {
      glds16_asm(st_src + (long)t * QB, smem + buf * STAGE + 2 * K_::TILE + K_::RW + wave * 1024);
    }
  };

  // ------------------------------------------------------------------------------------------------------------------
  // One continuous software pipeline over the "virtual rows" v = t R + j (query step t, key row j of this wave).  A lone wave
  // issues in order: the matrix pipe only works under vector instructions that stand BETWEEN the MFMAs in program order, and a
  // lone wave issues one vector instruction per 4 cycles (8 for exp): about five fit under one 32-cycle MFMA
  // (tools/probes/mfma_valu_probe.hip: 35 cycles per slot with four, 43 with five of which two exp).  A row has 16 MFMAs and
  // ~64 vector instructions (16 x fma, exp, mul + 16 conversions), dealt out over two phases of eight one-MFMA slots:
  //   A(v): S / dP MFMAs of row v + 1          | vector stream of row v, registers 8..15
  //   B(v): dV / dK MFMAs of row v (asm, "+a") | tail of row v's stream, head (registers 0..7) of row v + 1's
  // (slots are pinned in source order by sched_barrier; hipcc pads the hazards of the builtin MFMAs, the asm ones carry their
  // own s_nop).  LDS reads ride in the slots too, one to three each: the transposed fragments of a step in A(t R), the
  // accumulator seeds (relwT, -delta) and the row-bias vector of row v + 2 in B(v) behind row v's last conversion, the Q / dO
  // row fragments of step t + 1 in B(t R + R - 2).  That phase is the STEP BOUNDARY: ONE barrier at its start -- behind it
  // stage t + 1 has landed for everybody and nobody reads stage t any more -- and the four LDS-DMA pieces of step t + 2 (into
  // the buffer of stage t) in its last slots.  So neither fragment reads nor the first S / dP of a step are exposed.  The
  // pipeline runs one virtual step past the end on re-read data (clamped DMA): no "is there a next row" branches around MFMAs.
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  Chunk qc[C::KS_D], doc[C::KS_D], qtc[2][C::KS_B], dotc[2][C::KS_B];
  f32x16 st[2], dp[2];                // [v & 1]
  f32x4 rhl4[2][4];                   // relh c2 - lse2 of rows v, v + 1 per query slot: [v & 1]
  unsigned pbw[C::KS_B][4], dsw[C::KS_B][4];  // P / dS of the row being multiplied as 16-bit B operands (one word = two elements)
  auto stage_of = [&](int t) { return smem + (t & 1) * STAGE; };
  auto rows_piece = [&](const char* sl, int i) {  // Q / dO by rows (A operands of S and dP): piece i of 8
    if (i & 1) doc[i >> 1] = lds_chunk<T>(sl + K_::TILE, col, 2 * (i >> 1) + h);
    else qc[i >> 1] = lds_chunk<T>(sl, col, 2 * (i >> 1) + h);
  };
  auto tr_piece = [&](const char* sl, int i) {  // Q^T / dO^T (A operands of dK^T and dV^T): piece i of 8, two transposing reads each
    const int d = (i >> 1) & 1, ks = i >> 2;
    if (i & 1) dotc[d][ks] = lds_tr_chunk<T>(sl + K_::TILE, d, 0, ks, lane);
    else qtc[d][ks] = lds_tr_chunk<T>(sl, d, 0, ks, lane);
  };
  // accumulators start from relwT (S) and -delta (dP), plus the row-bias vector of key row j: piece i of 4 (three reads)
  auto init_piece = [&](const char* sl, int j, int b, int i) {
    const int c = 2 * i + h;  // 16-byte chunk holding query slots 8i + 4h .. + 3
    const f32x4 dl4 = *(const f32x4*)(sl + 2 * K_::TILE + K_::RW + c * 16);
    const f32x4 rw = *(const f32x4*)(sl + 2 * K_::TILE + col * QB + ((c ^ swz<QB>(col)) << 4));
    rhl4[b][i] = *(const f32x4*)(sl + 2 * K_::TILE + K_::RW + (1 + R * wave + j) * QB + c * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) { st[b][4 * i + e] = rw[e]; dp[b][4 * i + e] = dl4[e]; }
  };
  // The vector work of a row -- P = exp2(S c2 + relh c2 - lse2), dS = P (dP - delta) (0 on padded key lanes / query slots: bias
  // -inf), then the 16-bit conversions -- as a STREAM over 19 slots: slot sig runs the fma of register sig, the exp of
  // register sig - 1, the multiply of register sig - 2 and (every other slot) the conversion of the pair finished two slots ago.
  // Inside a slot all four are independent: a lone wave pays the full latency of every dependent vector instruction.  Each
  // result is pinned to its slot by an empty asm: without a use there hipcc sinks the arithmetic of a whole phase below the
  // branches of the step boundary (sched_barrier orders machine instructions inside a block, not IR that may move between blocks).
  auto w_slot = [&](int b, int sig) {
#ifdef BSG_KV4_NOVALU  // timing-only build: no vector work at all (wrong results)
    return;
#endif
    if (sig >= 0 && sig < 16) {
      st[b][sig] = fmaf(st[b][sig], c2, rhl4[b][sig >> 2][sig & 3]);
      asm volatile("" : "+v"(st[b][sig]));
    }
    if (sig >= 1 && sig < 17) {
      st[b][sig - 1] = __builtin_amdgcn_exp2f(st[b][sig - 1]);
      asm volatile("" : "+v"(st[b][sig - 1]));
    }
    if (sig >= 2 && sig < 18) {
      dp[b][sig - 2] = st[b][sig - 2] * dp[b][sig - 2];
      asm volatile("" : "+v"(dp[b][sig - 2]));
    }
    if (sig >= 4 && sig <= 18 && (sig & 1) == 0) {
      const int i = (sig - 4) >> 1;  // pair (2i, 2i + 1): word i & 3 of k-step i >> 2
      pbw[i >> 2][i & 3] = pack2<T>(st[b][2 * i], st[b][2 * i + 1]);
      dsw[i >> 2][i & 3] = pack2<T>(dp[b][2 * i], dp[b][2 * i + 1]);
      asm volatile("" : "+v"(pbw[i >> 2][i & 3]), "+v"(dsw[i >> 2][i & 3]));
    }
  };
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  auto chunk_of = [&](const unsigned (&w)[4]) { return __builtin_bit_cast(f32x4, (u32x4{w[0], w[1], w[2], w[3]})); };
#define KV4_SLOT_END __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: step 0 into stage 0 (and step 1 behind it), row 0 up to the state phase A(0) expects
#ifdef BSG_DIAG_KV4
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory");
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_piece(0, 0, i);
  wait_vm0();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_piece(min(1, nt - 1), 1, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) rows_piece(stage_of(0), i);
#pragma unroll
  for (int i = 0; i < 4; ++i) { init_piece(stage_of(0), 0, 0, i); init_piece(stage_of(0), 1, 1, i); }  // rows 0 and 1 (R >= 2: same step)
#pragma unroll
  for (int ks = 0; ks < C::KS_D; ++ks) { mma32(st[0], qc[ks], kx[0][ks]); mma32(dp[0], doc[ks], vx[0][ks]); }
#pragma unroll
  for (int sig = 0; sig < 8; ++sig) w_slot(0, sig);  // row 0's stream up to where phase A(0) takes over
  __builtin_amdgcn_sched_barrier(0);

  // (buffers indexed by the parity of the virtual row must be indexed at compile time -- a run-time index sends the arrays to
  // scratch -- so with R odd two steps are unrolled; an odd number of steps is rounded up by one step whose relwT columns are the
  // -inf padding the dQ kernel publishes behind the last query: P = dS = 0 exactly, it adds nothing)
  auto step = [&](int t, auto tpar) {
    const char* sl_t = stage_of(t);
    const char* sl_n = stage_of(t + 1);
    static_for<0, R>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      constexpr int pb_ = (decltype(tpar)::value * R + j) & 1, pn_ = pb_ ^ 1;  // buffer parity of virtual rows v, v + 1
      constexpr int jn = j == R - 1 ? 0 : j + 1;                               // key row of v + 1 (row 0 of step t + 1 behind the last)
      constexpr bool last2 = j + 2 >= R;                                       // row v + 2 belongs to step t + 1
      // ------------------------------------------------------------ phase A(v)
      KV4_STAMP(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i & 1) mma32(dp[pn_], doc[i >> 1], vx[jn][i >> 1]);
        else mma32(st[pn_], qc[i >> 1], kx[jn][i >> 1]);
        w_slot(pb_, 8 + i);
        if constexpr (j == 0) tr_piece(sl_t, i);  // this step's transposed fragments (the previous step's died with its last B)
        KV4_SLOT_END;
      }
      KV4_STAMP(1);
      // ------------------------------------------------------------ phase B(v)
      if constexpr (j == R - 2) {  // the step boundary (see above)
        wait_vm0();
        __syncthreads();
        KV4_STAMP(2);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ks = i >> 2, d = i & 1;
        if ((i >> 1) & 1) mfma32_agpr(dkt[j][d], __builtin_bit_cast(f32x4, qtc[d][ks]), chunk_of(dsw[ks]), T());
        else mfma32_agpr(dvt[j][d], __builtin_bit_cast(f32x4, dotc[d][ks]), chunk_of(pbw[ks]), T());
        w_slot(pb_, 16 + i);  // the tail of row v's stream (its last multiply and conversion)
        w_slot(pn_, i);       // the head of row v + 1's
        // row v's S / dP buffers are free behind slot 2 (converted): seeds and row bias of row v + 2, a piece per slot
        if (i >= 3 && i < 7) init_piece(last2 ? sl_n : sl_t, last2 ? j + 2 - R : j + 2, pb_, i - 3);
        if constexpr (j == R - 2) {
          rows_piece(sl_n, i);                                               // next step's Q / dO row fragments
          if (i >= 4) issue_piece(min(t + 2, nt - 1), t & 1, i - 4);  // step t + 2 (the padding step included) into the buffer stage t has left
        }
        KV4_SLOT_END;
      }
      KV4_STAMP(3);
    });
  };
  if constexpr (R & 1) {
    for (int t = 0; t < nt; t += 2) {
      step(t, I0{});
      step(t + 1, I1{});
    }
  } else {
    for (int t = 0; t < nt; ++t) step(t, I0{});
  }
#undef KV4_SLOT_END
#ifdef BSG_DIAG_KV4
  if (tid == 0 && R == 3) {
    for (int i = 0; i < 8; ++i) atomicAdd(&bsg_kv4_stamps[i], st_acc[i]);
    atomicAdd(&bsg_kv4_stamps[8], (unsigned long long)nt);
  }
#endif
  kv4_settle<R>(dkt, dvt);
  if (col < a.wp) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long tok = (long)s * a.N + (long)(krw + j) * a.wp + col;
      T* dkrow = (T*)a.dk + tok * a.ld + head * 64;
      T* dvrow = (T*)a.dv + tok * a.ld + head * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *(typename Traits<T>::Vec4*)(dkrow + 32 * d + 8 * i + 4 * h) =
              pack4<T>(dkt[j][d][4 * i] * a.scale, dkt[j][d][4 * i + 1] * a.scale, dkt[j][d][4 * i + 2] * a.scale, dkt[j][d][4 * i + 3] * a.scale);
          *(typename Traits<T>::Vec4*)(dvrow + 32 * d + 8 * i + 4 * h) =
              pack4<T>(dvt[j][d][4 * i], dvt[j][d][4 * i + 1], dvt[j][d][4 * i + 2], dvt[j][d][4 * i + 3]);
        }
    }
  }
}
