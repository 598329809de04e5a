// Shared helpers for the idee_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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
static inline int imax(long a, long b) { return (int)(a > b ? a : b); }

// Kernel-selection switches (IDEE_*): a process-wide option table (api.cpp) read at EVERY launch under a reader lock.  The
// environment variable of the same name is only the table's initial value; hosts change a switch between launches with
// idee_set_option (include/idee_hip.h) — no getenv() on the launch path.
bool opt_is(const char* name, const char* value);
int opt_int(const char* name);      // 0 when unset
static inline bool env_is(const char* name, const char* value) { return opt_is(name, value); }

// number of workgroups that fills the chip a few times over (256 CUs)
constexpr int kCUs = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- split-precision MFMA operands (see swin_mfma.hip "Split-bf16 GEMMs" and conv_mfma.hip conv16_split_kernel) ----
__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Two piece formats, chosen per kernel (template flag F16):
//   bf16 pieces (F16 = false): 8 + 8 mantissa bits, f32's exponent range — operand rounding 2^-18; used by the BACKWARD
//       kernels, whose gradient operands have no a-priori scale;
//   f16 pieces  (F16 = true): 11 + 11 mantissa bits — operand rounding 2^-22 (f32 level) for |x| in [2^-3, 65504], an
//       absolute floor of 2^-25 below.  Every operand class is brought into that range by an EXACT power of two: weights per
//       tensor (kernel prologues), gradients per tile (GradScale below), convolution inputs — which have no a-priori bound —
//       through the range guard (ActGuard below).  The Swin block's activation operands need none: each is the image of a
//       LayerNorm output under the block's own (prescaled) weights.
// An operand fragment is 4 registers = 8 pieces; fragments travel as u32x4 and are bit-cast at the MFMA.
// Piece mode PM of the split kernels (the former `bool F16` slot: false -> 0, true -> 1 keep their meaning):
//   PM_BF16X2 = 0  two bf16 pieces     PM_F16X2 = 1  two f16 pieces     PM_BF16X1 = 2  ONE bf16 piece (plain bf16 GEMM operands:
//   the bf16 activation path, act_dtype == IDEE_DTYPE_BF16; the lo piece does not exist and its MFMAs / LDS traffic are
//   compiled out)
enum { PM_BF16X2 = 0, PM_F16X2 = 1, PM_BF16X1 = 2 };
template <int F16>
__device__ __forceinline__ f32x4 mfma_s(u32x4 a, u32x4 b, f32x4 c) {
  if (F16 == PM_F16X2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: low half = a, high half = b
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {    // v_cvt_pk_f16_f32 (RNE)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2));
}
// ---- bf16 ACTIVATION STORAGE (idee_*_geom / idee_conv_desc .act_dtype == IDEE_DTYPE_BF16): tokens travel between the fused
// units as bf16, every kernel converts on load, accumulates in f32 and rounds to nearest-even on store ----
typedef __bf16 bf16_t;
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
// 4 consecutive channels of one token
template <typename T> __device__ __forceinline__ f32x4 ldx4(const T* p);
template <> __device__ __forceinline__ f32x4 ldx4<float>(const float* p) { return ld4(p); }
template <> __device__ __forceinline__ f32x4 ldx4<bf16_t>(const bf16_t* p) {
  const u32x2s v = *reinterpret_cast<const u32x2s*>(p);
  return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
               __uint_as_float(v[1] & 0xffff0000u)};
}
// the same load split in two, so that a kernel can issue the memory read an iteration ahead and convert at the point of use
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<bf16_t> { typedef u32x2s type; };
template <typename T> __device__ __forceinline__ typename Raw4<T>::type ldraw4(const T* p) {
  return *reinterpret_cast<const typename Raw4<T>::type*>(p);
}
__device__ __forceinline__ f32x4 cvtraw4(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 cvtraw4(u32x2s v) {
  return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
               __uint_as_float(v[1] & 0xffff0000u)};
}
template <typename T> __device__ __forceinline__ void stx4(T* p, f32x4 v);
template <> __device__ __forceinline__ void stx4<float>(float* p, f32x4 v) { st4(p, v); }
template <> __device__ __forceinline__ void stx4<bf16_t>(bf16_t* p, f32x4 v) {
  *reinterpret_cast<u32x2s*>(p) = u32x2s{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
}
template <typename T> __device__ __forceinline__ float ldx1(const T* p);
template <> __device__ __forceinline__ float ldx1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldx1<bf16_t>(const bf16_t* p) {
  return __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short*>(p)) << 16);
}

struct Split { unsigned h01, h23, l01, l23; };
template <int F16>
__device__ __forceinline__ Split split2(f32x4 x) {
  Split p;
  if (F16 == PM_BF16X1) {
    p.h01 = cvt_pk_bf16(x[0], x[1]);
    p.h23 = cvt_pk_bf16(x[2], x[3]);
    p.l01 = 0u; p.l23 = 0u;
  } else if (F16 == PM_F16X2) {
    p.h01 = cvt_pk_f16(x[0], x[1]);
    p.h23 = cvt_pk_f16(x[2], x[3]);
    // residual x - float(hi), exact, one v_fma_mix_f32 per element (the f16 half is read in place: no conversion back; hipcc does
    // not select this instruction by itself: fma((float)h, -1, x) compiles to v_cvt_f32_f16 + v_pk_add_f32, twice the count).
    // The statement computes IN PLACE ("+v": the register that holds x[i] — or the compiler's own copy of it — is source and
    // destination).  An `asm` statement is invisible to the compiler's MFMA hazard padding (tools/isa_hazards.py):
    // a fresh "=v" destination can be a register that an MFMA still in flight is going to write (the dead vDst of a chain link
    // with C != D), and the VALU result is then overwritten when the MFMA retires — the hardware needs 4 wait states there and
    // nobody inserts them.  A destination that already holds a live, compiler-visible value cannot be the vDst of an MFMA in
    // flight; if that value is itself an MFMA result, the v_cvt_pk above — which the compiler does pad — has read it first.
    float r0 = x[0], r1 = x[1], r2 = x[2], r3 = x[3];
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r0) : "v"(p.h01));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r1) : "v"(p.h01));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r2) : "v"(p.h23));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r3) : "v"(p.h23));
    p.l01 = cvt_pk_f16(r0, r1);
    p.l23 = cvt_pk_f16(r2, r3);
  } else {
    p.h01 = cvt_pk_bf16(x[0], x[1]);
    p.h23 = cvt_pk_bf16(x[2], x[3]);
    // residual x - float(hi), exact; written as x + (-hi) on float2 so that hipcc emits v_pk_add_f32 with neg_lo / neg_hi itself
    // (a float2 subtraction is scalarised; an `asm` statement would hide the instruction from the compiler's hazard padding)
    const f32x2 hf01 = {__uint_as_float(p.h01 << 16), __uint_as_float(p.h01 & 0xffff0000u)};
    const f32x2 hf23 = {__uint_as_float(p.h23 << 16), __uint_as_float(p.h23 & 0xffff0000u)};
    const f32x2 x01 = {x[0], x[1]}, x23 = {x[2], x[3]};
    const f32x2 r01 = x01 + (-hf01), r23 = x23 + (-hf23);
    p.l01 = cvt_pk_bf16(r01[0], r01[1]);
    p.l23 = cvt_pk_bf16(r23[0], r23[1]);
  }
  return p;
}

// one PAIR of elements as f16 pieces (the values split2<PM_F16X2> gives; no inline asm): hi = packed halves, lo = packed residuals
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& hi, unsigned& lo) {
  const h16x2 h = __builtin_convertvector(f32x2{a, b}, h16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = cvt_pk_f16(a - (float)h[0], b - (float)h[1]);      // (exact residuals; compiler-scheduled: v_cvt_f32_f16 x 2, v_pk_add_f32)
}

// ---- f16 pieces at f32 level: exact power-of-two scale factors (swin_mfma.hip "f16 pieces at f32 level", conv_mfma.hip) ----
__device__ __forceinline__ float pow2i(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }   // |k| <= 126
// max over the 64 lanes, every lane gets it: four DPP row rotations (16-lane rows) and gfx950's two permlane swaps — ten VALU
// instructions, no LDS round trips (six ds_bpermute steps cost ~800 cycles per call)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false));
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124 /*row_ror:4*/, 0xf, 0xf, false));
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122 /*row_ror:2*/, 0xf, 0xf, false));
  v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, false));
  typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
  u32x2w r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = mx(r[0], r[1]);
  r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return mx(r[0], r[1]);
}
__device__ __forceinline__ int imax_dev(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// am = max(am, |v0|, |v1|, |v2|, |v3|): hipcc folds each line into one v_max3_f32 with |.| source modifiers (no `asm`: the compiler
// must see every VALU instruction that may sit behind an MFMA)
__device__ __forceinline__ float amax_acc(float am, f32x4 v) {
  am = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), am);
  am = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])), am);
  return am;
}
__device__ __forceinline__ float amax4(f32x4 v) {
  return fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
}
// Power-of-two scale of a gradient tile, kept per wave: the tile's largest |dy| 2^k stays in [2^1, 2^8), re-centred on 2^4
// when a tile leaves the window; k moves up by at most 40 above the smallest k the wave has used (bounds accumulators that
// are rescaled along).
struct GradScale {
  int k = 0, kmin = 0;
  bool set = false;
  // amax: the wave's largest |dy| of this work item (any lane's copy after wave_umax).  Returns the factor the weight-gradient
  // accumulators must be multiplied with (1 = unchanged) and moves k.
  __device__ __forceinline__ float update(unsigned amax_bits) {
    constexpr int LO = 1, HI = 7, MID = 4;
    amax_bits = __builtin_amdgcn_readfirstlane(amax_bits);
    const int e = (int)(amax_bits >> 23) - 127;
    if (amax_bits == 0u || e > 127) return 1.f;     // zero tile / inf: keep the scale
    if (set && e + k >= LO && e + k <= HI) return 1.f;
    int kn = MID - e;
    kn = kn < -63 ? -63 : (kn > 63 ? 63 : kn);
    if (!set) { k = kmin = kn; set = true; return 1.f; }
    if (kn > kmin + 40) kn = kmin + 40;
    if (kn == k) return 1.f;
    const float f = pow2i(kn - k);
    k = kn;
    if (kn < kmin) kmin = kn;
    return f;
  }
  // Data-gradient staging (conv16_split_kernel): the scale of a work unit only moves DOWN.  exceeds(): this row would leave
  // the window at the top (or no scale is set yet); set_from(): scale from the largest |dy| of all rows to be staged.
  __device__ __forceinline__ bool exceeds(unsigned amax_bits) const {
    amax_bits = __builtin_amdgcn_readfirstlane(amax_bits);
    const int e = (int)(amax_bits >> 23) - 127;
    return amax_bits != 0u && e <= 127 && (!set || e + k > 7);
  }
  __device__ __forceinline__ void set_from(unsigned amax_bits) {
    amax_bits = __builtin_amdgcn_readfirstlane(amax_bits);
    const int e = (int)(amax_bits >> 23) - 127;
    if (amax_bits == 0u || e > 127) return;
    const int kn = 4 - e;
    k = kn < -63 ? -63 : (kn > 63 ? 63 : kn);
    set = true;
  }
  __device__ __forceinline__ float s() const { return pow2i(k); }
  __device__ __forceinline__ float inv() const { return pow2i(-k); }
};

// Activation guard of the f16-piece kernels whose ACTIVATION operand is not bounded by construction (convolution inputs: the
// residual stream, ReLU outputs): a power of two 2^k, k <= 0, kept per wave / workgroup and work unit, moving DOWN only inside
// a unit.  k = 0 — operands split as they are, bit-identical to an unguarded kernel — until a staged tile's largest |x| 2^k
// reaches 2^15 (an f16 hi piece overflows at 65520); then k is lowered so that this tile's largest element lands in
// [2^11, 2^12), tiles staged at the old scale are staged again and accumulators that carry the factor are multiplied along
// (exact).  Elements of a guarded tile are rounded at max(2^-22 |x|, 2^-36 of the tile's largest).  A non-finite element
// never triggers the guard: it stays non-finite through the pieces, as it does through the exact kernels.
struct ActGuard {
  int k = 0;
  float thr = 32768.f;       // 2^(15 - k)
  // am: this lane's largest |x| of the tile just staged into registers (called by the whole wave)
  __device__ __forceinline__ bool exceeds(float am) const { return __any(am >= thr && am <= 3.0e38f) != 0; }
  // lower k for a tile whose wave- (or workgroup-) wide largest |x| has the bit pattern amax_bits; returns new k - old k (<= 0)
  __device__ __forceinline__ int lower(unsigned amax_bits) {
    amax_bits = __builtin_amdgcn_readfirstlane(amax_bits);
    const int e = (int)(amax_bits >> 23) - 127;
    int kn = 11 - e;
    kn = kn > k ? k : (kn < -116 ? -116 : kn);
    const int d = kn - k;
    k = kn;
    thr = 15 - kn > 127 ? __builtin_inff() : pow2i(15 - kn);
    return d;
  }
  // finite part of a lane maximum (a non-finite element must not set the scale)
  static __device__ __forceinline__ float finite(float am) { return am <= 3.0e38f ? am : 0.f; }
  __device__ __forceinline__ float s() const { return pow2i(k); }
  __device__ __forceinline__ float inv() const { return pow2i(-k); }
};

}  // namespace idee
