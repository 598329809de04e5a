// What clock does the chip hold under a chip-wide f32 MFMA load, and does it depend on the operand DATA?
// One workgroup of 512 threads per CU (2 waves per SIMD), each wave a chain of independent 16x16x4 f32 MFMAs over register
// operands that are (mode 0) one constant or (mode 1) random values; s_memtime (core clock) and s_memrealtime (100 MHz) are read
// around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const float* in, float* out, long long* clk, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 8191]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 8191]; }
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[(u + q) & 7], acc[q], 0, 0, 0);
  }
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
  float *in, *out; long long* clk;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 2 * 8);
  float h[8192];
  const int iters = 40000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 8192; ++i) h[i] = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 1.0001f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, clk, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    long long hc[512]; hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
    const double mf = 256.0 * 8 * iters * 32;            // MFMAs
    printf("%s operands: %.3f ms  %.1f TFLOP/s  | WG 0: %lld core cycles in %lld x 10 ns -> %.0f MHz ; MFMA cycles per SIMD %.0f -> busy %.2f\n",
           mode ? "random  " : "constant", ms, mf * 2048 / ms * 1e-9, hc[0], hc[1], hc[0] / (hc[1] * 0.01), 2.0 * iters * 32 * 32,
           2.0 * iters * 32 * 32 / hc[0]);
  }
  return 0;
}
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_clock tools/micro/mfma_clock.hip && ./tools/micro/mfma_clock
// measured (MI355X): constant operands 34.52 ms 155.5 TFLOP/s | random operands 34.79 ms 154.3 TFLOP/s  -> 99 % of the 157.3 TFLOP/s
// f32 MFMA peak is reachable chip-wide with random data (no clock drop); the two counters above do not tick at the core / 100 MHz
// rates on this part (ratio 23.9), so the rate comes from the event time.
