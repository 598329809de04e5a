// Shared helpers for the idee_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/idee_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace idee {

// thread-local last-error message (include/idee_hip.h: idee_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

#define IDEE_CHECK_ARG(cond, ...)                                       \
  do {                                                                  \
    if (!(cond)) return idee::fail(IDEE_ERR_INVALID_ARG, __VA_ARGS__);  \
  } while (0)

#define IDEE_LAUNCH_CHECK(name)                                                              \
  do {                                                                                       \
    hipError_t e__ = hipGetLastError();                                                      \
    if (e__ != hipSuccess) return idee::fail(IDEE_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int imin(long a, long b) { return (int)(a < b ? a : b); }

// number of workgroups that fills the chip a few times over (256 CUs)
constexpr int kCUs = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace idee
