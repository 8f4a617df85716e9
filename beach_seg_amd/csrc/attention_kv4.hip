// Translation unit of the one-wave-per-SIMD attention-backward kernels (attention_kv4.hpp).  It is compiled on its own because
// it needs `-mllvm -amdgpu-mfma-vgpr-form`: in a kernel that may use the accumulator half of the register file (512 registers
// per wave) hipcc otherwise gives EVERY MFMA an accumulator-file destination and copies the S / dP results out with
// v_accvgpr_read, element by element (6.6 copies per MFMA measured).  With the flag the builtin MFMAs (S, dP: consumed by the
// vector ALU) write arch VGPRs, while dK^T / dV^T stay pinned in the accumulator file through the asm MFMAs.
#include "attention_kv4.hpp"
#include <cstdio>

template <typename T, int R> static void launch_rows(AttnBwdKvArgs k, int kr_begin, int kr_count, hipStream_t st) {
  typedef Kv4K<T, R> K_;
  constexpr int lds = 2 * K_::STAGE;
  k.kr_begin = kr_begin; k.kr_count = kr_count;
  const dim3 grid((kr_count / (4 * R)) * k.nh * k.S);
  hipLaunchKernelGGL((attn_bwd_dkv4_kernel<T, R>), grid, dim3(256), lds, st, k);
}
static bool kv4_rows_ok(int r) { return r % 4 == 0 && r >= 8 && (r % 12 != 4 || r >= 16); }
// key rows covered: the first key_rows when the caller wants no more and they split into 12- and 8-row groups, else all Hp
static int kv4_rows(const AttnBwdKvArgs& k) { return k.key_rows > 0 && k.key_rows < k.hp && kv4_rows_ok(k.key_rows) ? k.key_rows : k.hp; }
template <typename T> static void launch_all(const AttnBwdKvArgs& k, hipStream_t st) {
  // Hp = 12 a + 8 b key rows: `a` groups of 12 rows at three rows per wave, then `b` (0, 1 or 2) groups of 8 rows at two per wave
  const int rows = kv4_rows(k);
  const int b = (rows % 12) == 0 ? 0 : ((rows % 12) == 8 ? 1 : 2), main_rows = rows - 8 * b;
  if (main_rows) launch_rows<T, 3>(k, 0, main_rows, st);
  if (b) launch_rows<T, 2>(k, main_rows, 8 * b, st);
}
bool bsg_dkv4_ok(const AttnBwdKvArgs& k) { return kv4_rows_ok(k.hp) && k.q_begin % 64 == 0 && k.q_begin >= 0 && k.q_begin < k.N; }
void bsg_launch_dkv4(const AttnBwdKvArgs& k, int dtype_bf16, hipStream_t st) {
  if (dtype_bf16) launch_all<bf16_t>(k, st);
  else launch_all<f16_t>(k, st);
#ifdef BSG_DIAG_KV4
  (void)hipStreamSynchronize(st);
  unsigned long long hb[16], z[16] = {0};
  (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(bsg_kv4_stamps), sizeof(hb));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(bsg_kv4_stamps), z, sizeof(z));
  const double n = (double)(hb[8] ? hb[8] : 1);
  fprintf(stderr, "[kv4 stamps] cycles per query step (wave 0, R = 3): gaps before A %.0f | phase A x3 %.0f | DMA wait + barrier %.0f | "
          "phase B x3 %.0f\n", hb[0] / n, hb[1] / n, hb[2] / n, hb[3] / n);
#endif
}
