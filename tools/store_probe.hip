// store-path probe: 256 workgroups x 4 waves, each wave writes its 128 x 128 bf16 block of a 256 x 256 tile of a
// [M][4096] matrix, 24 tiles per workgroup, with P lanes per row (row segment = P x 16 B): how does the per-CU store
// rate depend on the segment length of a wave-level store instruction?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int P>
__global__ __launch_bounds__(256, 1) void k(char* out, int ldo_bytes, int tiles_n, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int lr = lane / P, lc = lane % P;       // row within the instruction's 64/P rows, 16-byte chunk within the row
  constexpr int RPI = 64 / P;                   // rows per instruction
  constexpr int CPR = 256 / (P * 16);           // instructions per row group to cover the wave's 256 bytes of columns
  u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tm = t / tiles_n, tn = t % tiles_n;
    char* base = out + (long)(tm * 256 + wm * 128) * ldo_bytes + (tn * 256 + wn * 128) * 2;
#pragma unroll 4
    for (int r = 0; r < 128; r += RPI)
#pragma unroll
      for (int c = 0; c < CPR; ++c)
        *(u32x4*)(base + (long)(r + lr) * ldo_bytes + c * P * 16 + lc * 16) = v;
  }
}
template <int P> void run(char* d, int M, int N) {
  const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, d, N * 2, tiles_n, ntiles);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<P>, dim3(256), dim3(256), 0, 0, d, N * 2, tiles_n, ntiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)M * N * 2;
  printf("P=%2d (%4d B per row segment): %.3f ms  %.2f TB/s  %.1f B/clk/CU at 2.1 GHz\n", P, P * 16, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.1e9));
}
int main() {
  const int M = 100352, N = 4096;
  char* d; hipMalloc(&d, (size_t)M * N * 2);
  run<4>(d, M, N); run<8>(d, M, N); run<16>(d, M, N); run<4>(d, M, N);
  return 0;
}
