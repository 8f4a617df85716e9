// Does a lone wave's vector work hide under its own MFMAs?  One wave per SIMD (256 threads per workgroup, 512-register budget),
// a loop of { one v_mfma_f32_32x32x16_bf16 ; K independent vector instructions }, K = 0..8, for three forms of the MFMA:
//   0: accumulator in the accumulator file (asm, "+a")   1: accumulator in arch VGPRs (asm, "+v")
// and two kinds of filler: plain v_fma_f32 / with every third one a v_exp_f32.  Prints shader cycles per loop iteration.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_probe.hip -o /tmp/mfma_valu_probe && /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FORM, int K, int EXP>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  f32x4 a = {1.f, 2.f, 3.f, 4.f}, b = {threadIdx.x * 1e-3f, 1.f, 0.5f, 0.25f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (FORM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (EXP && (k % 3) == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(b[1]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FORM, int K, int EXP> static void run(float* out, long long* cyc) {
  const int iters = 2000, nb = 256;
  hipLaunchKernelGGL((probe<FORM, K, EXP>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((probe<FORM, K, EXP>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nb; ++i) s += h[i];
  printf("form %d (%s) K=%d %s: %.1f cycles per MFMA slot\n", FORM, FORM == 0 ? "acc in AGPR" : "acc in VGPR", K, EXP ? "fma+exp" : "fma", s / nb / iters / 4);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0, 0>(out, cyc); run<0, 2, 0>(out, cyc); run<0, 4, 0>(out, cyc); run<0, 5, 0>(out, cyc); run<0, 6, 0>(out, cyc); run<0, 8, 0>(out, cyc);
  run<0, 4, 1>(out, cyc); run<0, 5, 1>(out, cyc); run<0, 6, 1>(out, cyc);
  run<1, 0, 0>(out, cyc); run<1, 4, 0>(out, cyc); run<1, 5, 0>(out, cyc); run<1, 6, 0>(out, cyc); run<1, 5, 1>(out, cyc);
  return 0;
}
