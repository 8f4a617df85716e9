// Issue-rate probe for gfx950 (diagnostic, not product code): how do VALU / transcendental / MFMA streams of the
// waves sharing a SIMD add up?  Each case: WG = 256*W threads (W waves per SIMD), 256 WGs, every wave runs `iters`
// iterations of a fixed instruction group chosen by its role.  Prints cycles per iteration per wave (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum Role { R_FMA = 0, R_EXP = 1, R_MFMA = 2, R_MIX_M1F6 = 3, R_MIX_M1E2F4 = 4, R_IDLE = 5, R_MIX_M1F12 = 6, R_MIX_M1E4F8 = 7, R_DMA8 = 8, R_LDST8 = 9, R_DSREAD16 = 10 };

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

typedef __attribute__((ext_vector_type(4))) float f4;
__global__ __launch_bounds__(1024) void probe(int role_even, int role_odd, int iters, long long* out, float* sink, const char* gsrc) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int wave = threadIdx.x >> 6;
  const int slot = wave >> 2;  // waves w and w+4 share a SIMD
  const int role = (slot & 1) ? role_odd : role_even;
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float x8 = x0 + 8, x9 = x0 + 9, x10 = x0 + 10, x11 = x0 + 11;
  const float c = 0.999f;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f); }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (role == R_FMA) {
    for (int it = 0; it < iters; ++it) { FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5); FMA(x6); FMA(x7); }
  } else if (role == R_EXP) {
    for (int it = 0; it < iters; ++it) { EXP(x0); EXP(x1); EXP(x2); EXP(x3); EXP(x4); EXP(x5); EXP(x6); EXP(x7); }
  } else if (role == R_MFMA) {
    for (int it = 0; it < iters; ++it) { MFMA(acc0); MFMA(acc1); MFMA(acc0); MFMA(acc1); }
  } else if (role == R_DMA8) {  // 8 LDS-DMA pieces (1 KB each) per iteration, from an L2-resident 64 KB window
    const char* src = gsrc + (wave & 7) * 8192 + (threadIdx.x & 63) * 16;
    char* dst = lds + (wave & 7) * 8192;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if (role == R_LDST8) {  // the same 8 KB through registers: 8 global_load_dwordx4 + 8 ds_write_b128
    const char* src = gsrc + (wave & 7) * 8192 + (threadIdx.x & 63) * 16;
    char* dst = lds + (wave & 7) * 8192 + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
      f4 r[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = *(const volatile f4*)(src + i * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) *(volatile f4*)(dst + i * 1024) = r[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else if (role == R_DSREAD16) {
    const char* p = lds + (threadIdx.x & 63) * 16;
    f4 a = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { f4 v = *(const volatile f4*)(p + i * 1024); a += v; }
    }
    x0 += a[0];
  } else if (role == R_MIX_M1F6) {
    for (int it = 0; it < iters; ++it) {
      MFMA(acc0); FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5);
      MFMA(acc1); FMA(x6); FMA(x7); FMA(x8); FMA(x9); FMA(x10); FMA(x11);
    }
  } else if (role == R_MIX_M1F12) {
    for (int it = 0; it < iters; ++it) {
      MFMA(acc0); FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5); FMA(x6); FMA(x7); FMA(x8); FMA(x9); FMA(x10); FMA(x11);
      MFMA(acc1); FMA(x0); FMA(x1); FMA(x2); FMA(x3); FMA(x4); FMA(x5); FMA(x6); FMA(x7); FMA(x8); FMA(x9); FMA(x10); FMA(x11);
    }
  } else if (role == R_MIX_M1E2F4) {
    for (int it = 0; it < iters; ++it) {
      MFMA(acc0); EXP(x0); FMA(x1); FMA(x2); EXP(x3); FMA(x4); FMA(x5);
      MFMA(acc1); EXP(x6); FMA(x7); FMA(x8); EXP(x9); FMA(x10); FMA(x11);
    }
  } else if (role == R_MIX_M1E4F8) {
    for (int it = 0; it < iters; ++it) {
      MFMA(acc0); EXP(x0); FMA(x1); FMA(x2); EXP(x3); FMA(x4); FMA(x5); EXP(x6); FMA(x7); FMA(x8); EXP(x9); FMA(x10); FMA(x11);
      MFMA(acc1); EXP(x0); FMA(x1); FMA(x2); EXP(x3); FMA(x4); FMA(x5); EXP(x6); FMA(x7); FMA(x8); EXP(x9); FMA(x10); FMA(x11);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8 + x9 + x10 + x11 + acc0[0] + acc1[3];
  if (s == 123.456f) sink[0] = s;
}

static const char* names[] = {"fma8", "exp8", "mfma4", "m1f6 x2", "m1e2f4 x2", "idle", "m1f12 x2", "m1e4f8 x2", "dma8", "ld+dsw 8", "dsread16"};

int main() {
  long long* out; float* sink;
  hipMalloc(&out, 256 * 16 * sizeof(long long)); hipMalloc(&sink, 4);
  char* gsrc; hipMalloc(&gsrc, 65536 + 4096); hipMemset(gsrc, 1, 65536 + 4096);
  const int iters = 4000;
  struct Case { int W, re, ro; };
  std::vector<Case> cases = {
      {1, R_FMA, R_FMA}, {2, R_FMA, R_FMA}, {3, R_FMA, R_FMA}, {4, R_FMA, R_FMA},
      {1, R_EXP, R_EXP}, {2, R_EXP, R_EXP}, {4, R_EXP, R_EXP},
      {1, R_MFMA, R_MFMA}, {2, R_MFMA, R_MFMA},
      {2, R_MFMA, R_FMA}, {2, R_MFMA, R_EXP}, {2, R_FMA, R_EXP},
      {1, R_MIX_M1F6, R_MIX_M1F6}, {2, R_MIX_M1F6, R_MIX_M1F6},
      {1, R_MIX_M1F12, R_MIX_M1F12}, {2, R_MIX_M1F12, R_MIX_M1F12},
      {1, R_MIX_M1E2F4, R_MIX_M1E2F4}, {2, R_MIX_M1E2F4, R_MIX_M1E2F4},
      {1, R_MIX_M1E4F8, R_MIX_M1E4F8}, {2, R_MIX_M1E4F8, R_MIX_M1E4F8}, {3, R_MIX_M1E4F8, R_MIX_M1E4F8},
      {1, R_DMA8, R_DMA8}, {1, R_LDST8, R_LDST8}, {1, R_DSREAD16, R_DSREAD16},
      {2, R_MFMA, R_DMA8}, {2, R_MFMA, R_LDST8}, {2, R_MFMA, R_DSREAD16}, {2, R_DMA8, R_MFMA}, {2, R_LDST8, R_MFMA},
  };
  for (auto& cs : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe, dim3(256), dim3(256 * cs.W), 0, 0, cs.re, cs.ro, iters, out, sink, gsrc);
      hipDeviceSynchronize();
    }
    std::vector<long long> h(256 * 16);
    hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double se = 0, so = 0; int ne = 0, no = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 4 * cs.W; ++w) {
        if ((w >> 2) & 1) { so += h[b * 16 + w]; ++no; } else { se += h[b * 16 + w]; ++ne; }
      }
    printf("W=%d  even-slot waves: %-10s %8.1f cyc/iter", cs.W, names[cs.re], se / ne / iters);
    if (no) printf("   odd-slot waves: %-10s %8.1f cyc/iter", names[cs.ro], so / no / iters);
    printf("\n");
  }
  return 0;
}
