// The compute core of conv_wide_kernel in isolation: 24 accumulator tiles (4 pixel blocks x 6 output blocks) per wave, per step 6 weight
// fragments read from LDS (ds_read_b128) and 96 MFMA 16x16x4 f32; 8 waves per workgroup, one workgroup per CU (2 waves per SIMD).
// Variants: PIPE = 0 (read, wait, 16 MFMAs per output block) | 1 (next fragment read while the current block's MFMAs run) ;
// ORDER = 0 (s outer, m inner) | 1 (m outer, s inner: 4 back-to-back MFMAs on one accumulator).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PIPE, int ORDER>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int steps) {
  __shared__ __attribute__((aligned(16))) float s_w[36 * 256];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 36 * 256; i += 512) s_w[i] = in[i & 8191];
  f32x4 acc[4][6], xc[4];
  for (int m = 0; m < 4; ++m) { xc[m] = *reinterpret_cast<const f32x4*>(in + ((m * 64 + lane) * 4 & 8191)); for (int n = 0; n < 6; ++n) acc[m][n] = f32x4{0, 0, 0, 0}; }
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const float* wcur = s_w + (st % 6) * 6 * 256 + lane * 4;
    if (PIPE) {
      f32x4 wfa = *reinterpret_cast<const f32x4*>(wcur);
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        f32x4 wfb = wfa;
        if (n + 1 < 6) wfb = *reinterpret_cast<const f32x4*>(wcur + (n + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfa[s], xc[m][s], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wfa = wfb;
      }
    } else {
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const f32x4 wf = *reinterpret_cast<const f32x4*>(wcur + n * 256);
        __builtin_amdgcn_sched_barrier(0);
        if (ORDER == 0) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xc[m][s], acc[m][n], 0, 0, 0);
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xc[m][s], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    xc[st & 3][0] += 1e-9f;
  }
  f32x4 tot = {0, 0, 0, 0};
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 6; ++n) tot += acc[m][n];
  if (tot[0] == 12345.f) out[blockIdx.x * 512 + threadIdx.x] = tot[1] + tot[2] + tot[3];
}
template <int PIPE, int ORDER>
void run(const float* in, float* out, int steps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PIPE, ORDER>), dim3(256), dim3(512), 0, 0, in, out, steps);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<PIPE, ORDER>), dim3(256), dim3(512), 0, 0, in, out, steps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double ideal = 2.0 * steps * 96 * 32 / 2.4e6;
  printf("PIPE %d ORDER %d : %.3f ms, MFMA-bound %.3f ms -> busy %.2f\n", PIPE, ORDER, ms, ideal, ideal / ms);
}
int main() {
  float *in, *out; hipMalloc(&in, 8192 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (i * 2654435761u >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 0>(in, out, 400);
  run<0, 1>(in, out, 400);
  run<1, 0>(in, out, 400);
  return 0;
}
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/wide_block tools/micro/wide_block.hip && ./tools/micro/wide_block
// measured (MI355X, 400 steps, launch start-up included): PIPE 0 ORDER 0: busy 0.85 | PIPE 0 ORDER 1: 0.89 | PIPE 1 ORDER 0: 0.93.
// conv_wide_kernel itself reaches 0.65-0.67; with its barriers, slab copies and activation loads removed (timing variants) 0.73, and the
// PIPE 1 form inside the kernel gained 4 %: the remainder is the loop's address / validity arithmetic and copies, paid on the MFMA port.
