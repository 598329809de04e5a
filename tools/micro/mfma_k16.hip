// Does the legacy K = 16 f16 MFMA (v_mfma_f32_16x16x16_f16, 2-register operands) run at gfx950's doubled f16 rate, i.e. in half the
// time of v_mfma_f32_16x16x32_f16?  (It would let the two-piece token GEMMs drop the [a | a] operand duplication: 2 v_mov per MFMA.)
// One workgroup of 512 threads per CU, each wave a chain of independent MFMAs over register operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int K>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)in[(threadIdx.x * 64 + i * 8 + e) & 8191]; b[i][e] = (_Float16)in[(threadIdx.x * 64 + 32 + i * 8 + e) & 8191]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (K == 32) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[(u + q) & 3], acc[q], 0, 0, 0);
        else {
          const h4 al = {a[u][0], a[u][1], a[u][2], a[u][3]}, bl = {b[(u + q) & 3][0], b[(u + q) & 3][1], b[(u + q) & 3][2], b[(u + q) & 3][3]};
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bl, acc[q], 0, 0, 0);
        }
      }
  }
  float s = 0;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int K> void run(const float* in, float* out) {
  const int iters = 40000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, in, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double n = 256.0 * 8 * iters * 16;              // MFMAs per launch
  printf("16x16x%d f16: %.3f ms for %.0f MFMAs per SIMD -> %.1f SIMD cycles per MFMA at 2.4 GHz, %.0f TFLOP/s\n", K, ms, n / 1024,
         ms * 1e-3 * 2.4e9 / (n / 1024), n * 2.0 * 16 * 16 * K / ms * 1e-9);
}
int main() {
  float *in, *out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = (float)(i % 97) / 97.f - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<32>(in, out);
  run<16>(in, out);
  return 0;
}
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_k16 tools/micro/mfma_k16.hip && ./tools/micro/mfma_k16
// measured (MI355X): 16x16x32 f16 11.17 ms (1,922 TFLOP/s chip-wide), 16x16x16 f16 8.88 ms (1,209 TFLOP/s): the K = 16 instruction
// takes 80 % of the K = 32 one's time for half the work — 1.6x slower per product.  Four K = 16 MFMAs in place of two K = 32 ones
// with a duplicated operand would trade 2 v_mov (8 issue cycles) for two more MFMA issues (16): not taken.
