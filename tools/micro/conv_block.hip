// The MFMA block of the 16 -> 16 strip convolution in isolation: per row 27 taps of {2 ds_read_b128 ; 8 MFMA 16x16x4 f32}, fragment reads
// issued PF taps ahead, 27 weight fragments in registers, 2 workgroups of 4 waves per CU (2 waves per SIMD).  No global traffic in the
// loop.  Variants: PF = 1 | 2 ; LDS = 0 (operands stay in registers) | 1 ; MT = accumulators (column blocks) per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PF, int LDS, int MT>
__global__ __launch_bounds__(256, 2) void k(const float* in, float* out, int rows) {
  __shared__ __attribute__((aligned(16))) float s[4][9 * 34 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sx = s[wave];
  for (int i = lane; i < 9 * 34 * 16; i += 64) sx[i] = in[i & 8191];
  f32x4 wr[27];
  for (int i = 0; i < 27; ++i) wr[i] = *reinterpret_cast<const f32x4*>(in + ((i * 64 + lane) * 4 & 8191));
  const int j = lane & 15, kq = lane >> 4;
  f32x4 tot = {0, 0, 0, 0};
  __syncthreads();
  for (int r = 0; r < rows; ++r) {
    f32x4 acc[MT];
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0, 0, 0, 0};
    const float* base = sx + kq * (34 * 4) + j * 4 + (r & 1) * 4;
    f32x4 xb[PF + 1][MT];
    auto rd = [&](int tp, f32x4* x) __attribute__((always_inline)) {
      const float* rp = base + (tp / 3) * 34 * 16 + (tp % 3) * 4;
#pragma unroll
      for (int m = 0; m < MT; ++m) x[m] = LDS ? *reinterpret_cast<const f32x4*>(rp + (m & 1) * 64) : f32x4{rp[0] * 0 + 1.f + m, 2.f, 3.f, 4.f};
    };
    if (LDS) {
#pragma unroll
      for (int q = 0; q < PF; ++q) rd(q, xb[q]);
    } else rd(0, xb[0]);
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) {
      if (LDS && tp + PF < 27) rd(tp + PF, xb[(tp + PF) % (PF + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 wf = wr[tp];
      const f32x4* x = LDS ? xb[tp % (PF + 1)] : xb[0];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], x[m][q], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int m = 0; m < MT; ++m) tot += acc[m];
  }
  if (tot[0] == 12345.f) out[blockIdx.x * 256 + threadIdx.x] = tot[1] + tot[2] + tot[3];
}
template <int PF, int LDS, int MT>
void run(const float* in, float* out, int rows) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PF, LDS, MT>), dim3(512), dim3(256), 0, 0, in, out, rows);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<PF, LDS, MT>), dim3(512), dim3(256), 0, 0, in, out, rows);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double ideal = 2.0 * rows * 27 * 4 * MT * 32 / 2.4e6;     // ms of MFMA time per SIMD at 2.4 GHz
  printf("PF %d LDS %d MT %d : %.3f ms, MFMA-bound %.3f ms -> busy %.2f\n", PF, LDS, MT, ms, ideal, ideal / ms);
}
int main() {
  float *in, *out; hipMalloc(&in, 8192 * 4); hipMalloc(&out, 512 * 256 * 4);
  float h[8192]; for (int i = 0; i < 8192; ++i) h[i] = (i * 2654435761u >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int rows = 400;
  run<1, 0, 2>(in, out, rows);
  run<1, 1, 2>(in, out, rows);
  run<2, 1, 2>(in, out, rows);
  run<1, 0, 4>(in, out, rows / 2);
  run<1, 1, 4>(in, out, rows / 2);
  run<2, 1, 4>(in, out, rows / 2);
  return 0;
}
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o tools/micro/conv_block tools/micro/conv_block.hip && ./tools/micro/conv_block
// measured (MI355X): PF 1 LDS 0 MT 2: busy 0.92 | PF 1 LDS 1 MT 2: 0.97 | PF 2 LDS 1 MT 2: 0.97  -> the tap loop itself (fragment reads one
// tap ahead) keeps the MFMA pipe 97 % busy; the strip kernels' losses are elsewhere (work-unit quantisation, the half-empty last strip of a
// 200-pixel row, the VALU work of staging / epilogue, which does not co-execute with the f32 MFMA).  (The MT 4 lines fold duplicate MFMAs.)
