// What slows an f32 MFMA chain on gfx950?  One 512-thread workgroup per CU; waves 0-3 (one per SIMD) run an MFMA chain,
// waves 4-7 (their SIMD neighbours) run: 0 nothing | 1 the same MFMA chain | 2 LDS reads (ds_read_b128) | 3 global loads
// (L2-resident) | 4 SALU only.  Also: mode 5 = waves 0-3 alone with a DEPENDENT chain (one accumulator).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* src, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.5f;
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 v = {0, 0, 0, 0};
  int sacc = 0;
  const float a = 1.0001f, b = 0.5f;
  if (wave < 4 || MODE == 1) {
    const long long c0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int qq = MODE == 5 ? 0 : q;
          acc[qq] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[qq], 0, 0, 0);
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = wall_clock64() - c0;     // the MFMA wave's own duration (100 MHz ticks)
  } else if (MODE == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v += *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + u * 1024 + it * 4) & 8188));
    }
  } else if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v += *reinterpret_cast<const f32x4*>(src + ((blockIdx.x * 65536 + threadIdx.x * 4 + u * 2048 + it * 4) & 1048572));
    }
  } else if (MODE == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
    }
  }
  float s = sacc;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  s += v[0] + v[1] + v[2] + v[3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
float run(float* out, const float* src, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  static long long* cyc = nullptr;
  if (!cyc) (void)hipMalloc(&cyc, 256 * 8);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, src, iters, cyc);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, src, iters, cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i];
  printf("  [MFMA wave 0: %.3f ms]  ", avg / 256 / 1e5);
  return ms;
}
int main() {
  float *out, *src;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, 1048576 * 4 + 64); (void)hipMemset(src, 0, 1048576 * 4 + 64);
  const int iters = 20000;
  printf("16 MFMA 16x16x4 f32 per iteration on waves 0-3 (one per SIMD); neighbours on the same SIMDs do:\n");
  printf("nothing                 %.3f ms\n", run<0>(out, src, iters));
  printf("the same MFMA chain     %.3f ms\n", run<1>(out, src, iters));
  printf("8 ds_read_b128 / iter   %.3f ms\n", run<2>(out, src, iters));
  printf("8 global loads / iter   %.3f ms\n", run<3>(out, src, iters));
  printf("64 s_add / iter         %.3f ms\n", run<4>(out, src, iters));
  printf("(alone, 1 accumulator)  %.3f ms\n", run<5>(out, src, iters));
  return 0;
}
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_mix tools/micro/mfma_mix.hip && ./tools/micro/mfma_mix
// measured (MI355X), [duration of MFMA wave 0] / kernel:
//   nothing 4.276 / 4.291 ms | same MFMA chain 5.045 / 8.574 | 8 ds_read_b128 4.270 / 6.580 | 8 global loads 4.261 / 8.665
//   64 s_add 4.285 / 4.308 | alone with ONE accumulator (dependent chain) 4.283 / 4.309
