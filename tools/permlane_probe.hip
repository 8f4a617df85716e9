// Probe of v_permlane16_swap / v_permlane32_swap lane semantics on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned x = 100 + l, y = 200 + l;
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 new x", "p16 new y", "p32 new x", "p32 new y"};
  for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[a * 64 + i]); printf("\n"); }
  return 0;
}
