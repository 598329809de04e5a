// Does the bf16 MFMA (v_mfma_f32_16x16x32_bf16) co-execute with VALU work on gfx950 — unlike the f32-input MFMA
// (tools/micro/coexec.hip)?  One workgroup of 512 threads per CU:
//   mode 0: waves 0-3 run an MFMA chain, waves 4-7 idle      mode 1: waves 4-7 run a VALU FMA chain, 0-3 idle
//   mode 2: both (different waves of the same SIMD)          mode 3: every wave interleaves MFMA and independent VALU
//   mode 4: as 3 but one wave per SIMD (256 threads)         mode 5: f32 MFMA 16x16x4 in place of bf16, one wave per SIMD, interleaved
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float a = 1.0001f, b = 0.5f;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + threadIdx.x % 7); fb[i] = (short)(0x3f00 + i); }
  const bool do_m = MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : true;
  const bool do_v = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : true;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      // interleaved in one instruction stream: 1 MFMA then 4 independent FMAs, 16 times
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (MODE == 5) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
        else acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[u & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[(u * 4 + q) & 7] = __builtin_fmaf(v[(u * 4 + q) & 7], a, b);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[q], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], a, b);
      }
    }
  }
  float s = 0;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
float run(float* out, int iters, int threads = 512) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  printf("per iteration: 16 MFMA 16x16x32 bf16 (256 cycles if back-to-back at 16 cyc), 64 v_fma_f32 (256+ cycles)\n");
  printf("mfma only                     %.3f ms\n", run<0>(out, iters));
  printf("valu only                     %.3f ms\n", run<1>(out, iters));
  printf("both, 2 waves/SIMD            %.3f ms\n", run<2>(out, iters));
  printf("interleaved, 2 waves/SIMD     %.3f ms\n", run<3>(out, iters));
  printf("interleaved, 1 wave/SIMD      %.3f ms\n", run<4>(out, iters, 256));
  printf("f32 mfma interleaved, 1 w/SIMD %.3f ms\n", run<5>(out, iters, 256));
  return 0;
}
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/coexec_bf16 tools/micro/coexec_bf16.hip && ./tools/micro/coexec_bf16
// measured (MI355X):  mfma only 2.976 ms | valu only 2.627 ms | both, 2 waves/SIMD 4.871 ms | interleaved, 2 waves/SIMD 8.149 ms |
//                      interleaved, 1 wave/SIMD 4.193 ms | f32 mfma interleaved, 1 wave/SIMD 6.455 ms
