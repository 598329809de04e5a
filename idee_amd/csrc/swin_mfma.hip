// Fused Video-Swin block on the f32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Replaces, for one SwinTransformerBlock3D (reference models/encoder/Swin_3D.py:224-287):
//   norm1 -> zero-pad -> roll(-shift) -> window_partition -> WindowAttention3D (:145-178)
//   -> window_reverse -> roll(+shift) -> crop -> +shortcut -> norm2 -> Mlp (:36-42) -> +residual
// with no intermediate tensor: shift / pad / partition are index arithmetic on the channels-last
// token tensor, the 0/-100 shift mask (compute_mask :340-352) comes from coordinates.
//
// "Token layout" (TL): a wave owns 64 tokens = 4 column blocks of 16; lane (j = lane&15,
// kq = lane>>4) holds, for token 16*mt + j, the 4 channels sigma(4kq..4kq+3) of every
// 16-channel chunk — exactly the B-operand layout of the 16x16x4 MFMA *and* its D layout
// (rows = channels, cols = tokens).  A linear layer Y = W X is therefore a chain of MFMAs whose
// A operands are weight fragments held in registers and whose outputs feed the next layer
// with NO data movement.  Attention scores S[key][query] = K Q^T come out with keys on rows,
// i.e. already as the B operand of P.V; only V needs a transpose through LDS.  The head split
// uses a channel permutation sigma that puts each head's channels on whole k-steps.
#include "common.h"

namespace {

using namespace idee;

struct Geom {
  int N, V, D, H, W, Dp, Hp, Wp;   // N: batch items per variable (batch index b = n * V + v)
  int wd, wh, ww, sd, sh, sw;
  int nWh, nWw, nWin;   // windows per batch item
  int items;            // window groups (64 tokens) per batch item
  int shifted;
  float scale;
  float inv_nWhW, inv_nWw, inv_whww, inv_ww;   // reciprocals for the coordinate decode (exact for operands < 2^22)
  const float* path_scale;   // (batch items, 2) DropPath factors mask / keep_prob of the attention and MLP branch, or null
};

constexpr float kLnEps = 1e-5f;
constexpr int MT = 4;   // 16-token column blocks per wave
constexpr int VS = 68;  // LDS image row stride in floats (64 tokens + pad); rows stay 16-byte aligned

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr float kLog2e = 1.4426950408889634f;

// gfx950 does not co-issue VALU work with the f32 MFMA (measured: tools/micro/coexec.hip), so every VALU instruction of
// these kernels costs MFMA time.  The element-wise math below is therefore written on f32x4 values so that the compiler
// emits PACKED f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of work per issue slot).
__device__ __forceinline__ float hsum(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// GELU(erf) with erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. at f32 rounding level):
//   Phi(a) = 1 - q (a >= 0), q (a < 0),  q = 0.5 p(t) t exp(-a^2/2),  t = 1 / (1 + 0.3275911 |a| / sqrt2)
//   gelu(a) = max(a, 0) - |a| q = 0.5 a + (0.5 |a| - |a| q)
// One v_rcp and one v_exp per element; everything else packed.
struct GeluParts { f32x4 ax, hq, E; };   // |a|, 0.5 - q, exp(-a^2/2)
__device__ __forceinline__ GeluParts gelu_parts(f32x4 a) {
  GeluParts g;
  g.ax = __builtin_elementwise_abs(a);
  const f32x4 den = fma4(g.ax, splat(0.3275911f * 0.70710678118654752f), splat(1.f));
  f32x4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = __builtin_amdgcn_rcpf(den[r]);
  // -0.5 p(t): the sign and the half are folded into the coefficients (a negation would cost a v_xor per element)
  f32x4 p = fma4(splat(-0.5f * 1.061405429f), t, splat(0.5f * 1.453152027f));
  p = fma4(p, t, splat(-0.5f * 1.421413741f));
  p = fma4(p, t, splat(0.5f * 0.284496736f));
  p = fma4(p, t, splat(-0.5f * 0.254829592f));
  const f32x4 ea = fma4(a * a, splat(-0.5f * kLog2e), splat(0.f));   // fma form keeps the constant multiply packed
#pragma unroll
  for (int r = 0; r < 4; ++r) g.E[r] = __builtin_amdgcn_exp2f(ea[r]);   // exp(-a^2/2)
  g.hq = fma4(p * t, g.E, splat(0.5f));
  return g;
}
__device__ __forceinline__ f32x4 gelu4(f32x4 a) {
  const GeluParts g = gelu_parts(a);
  return fma4(a, splat(0.5f), g.ax * g.hq);
}
// gelu(a) and d gelu / da = Phi(a) + a phi(a), sharing the exponential
__device__ __forceinline__ void gelu_both4(f32x4 a, f32x4& gl, f32x4& dgl) {
  const GeluParts g = gelu_parts(a);
  gl = fma4(a, splat(0.5f), g.ax * g.hq);
  f32x4 phi;                                       // Phi - 0.5 = +-(0.5 - q)
#pragma unroll
  for (int r = 0; r < 4; ++r) phi[r] = copysignf(g.hq[r], a[r]);
  dgl = fma4(a * splat(0.39894228040143268f), g.E, phi + splat(0.5f));
}

// All LDS in these kernels is private to a wave (indexed by the wave id), and the LDS unit executes one wave's
// DS instructions in order, so exchanging data between the lanes of a wave needs no s_barrier: only the compiler
// must be kept from moving LDS accesses across the hand-off point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// channel permutation inside a 16-chunk: position i = 4*kq + r  ->  channel.  HPC heads per chunk.
template <int HPC>
__device__ __forceinline__ int sig(int i) {
  constexpr int sp = 4 / HPC;
  const int kq = i >> 2, r = i & 3;
  return (r / sp) * (16 / HPC) + kq * sp + (r % sp);
}

struct Tok {
  long off;
  int rid;
  bool valid;
};

// token tau (0..63) of work item `item` of batch item b
template <int NW, int TOK = 64>
__device__ __forceinline__ Tok locate(const Geom& g, int b, int item, int tau, int C) {
  constexpr int G = TOK / NW;
  Tok t;
  const int gi = tau / NW, i = tau % NW;
  const int wl = item * G + gi;
  const bool active = (item < g.items) && (wl < g.nWin);
  const int wlc = active ? wl : 0;
  // x / d for small non-negative x: (x + 0.5) * (1/d) truncated is exact while x < 2^22 (grids here are < 2^21 windows)
  auto fdiv = [](int x, float inv) { return (int)(((float)x + 0.5f) * inv); };
  // every factor below is < 2^24 (window and frame extents, window indices < 2^22), so the products go through the
  // full-rate 24-bit multiplier (v_mul_u32_u24 / v_mad_u32_u24; v_mul_lo_u32 issues at a quarter of that rate)
  auto m24 = [](int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); };
  const int nWhW = m24(g.nWh, g.nWw);
  const int iwd = fdiv(wlc, g.inv_nWhW), r = wlc - m24(iwd, nWhW);
  const int iwh = fdiv(r, g.inv_nWw), iww = r - m24(iwh, g.nWw);
  const int whww = m24(g.wh, g.ww);
  const int dt = fdiv(i, g.inv_whww), r2 = i - m24(dt, whww);
  const int dh = fdiv(r2, g.inv_ww), dw = r2 - m24(dh, g.ww);
  const int pd = m24(iwd, g.wd) + dt, ph = m24(iwh, g.wh) + dh, pw = m24(iww, g.ww) + dw;   // rolled frame
  auto axis = [](int p, int P, int w, int s) { return s == 0 ? 2 : (int)(p >= P - w) + (int)(p >= P - s); };
  t.rid = axis(pd, g.Dp, g.wd, g.sd) * 9 + axis(ph, g.Hp, g.wh, g.sh) * 3 + axis(pw, g.Wp, g.ww, g.sw);
  int sd_ = pd + g.sd, sh_ = ph + g.sh, sw_ = pw + g.sw;                         // un-rolled frame
  if (sd_ >= g.Dp) sd_ -= g.Dp;
  if (sh_ >= g.Hp) sh_ -= g.Hp;
  if (sw_ >= g.Wp) sw_ -= g.Wp;
  t.valid = active && sd_ < g.D && sh_ < g.H && sw_ < g.W;
  // token index: (b D + sd) H + sh < 2^24 and the token count of the tensor < 2^32 (checked in make_geom)
  const unsigned row = (unsigned)m24(m24(b, g.D) + sd_, g.H) + (unsigned)sh_;
  const unsigned tok = __umul24(row, (unsigned)g.W) + (unsigned)sw_;
  t.off = (long)tok * C;
  return t;
}

// Weight fragment of a [rows][cols] row-major matrix W (leading dim ld) for output block at row0
// and input chunk at col0: element s = W[row0 + SM(l&15)][col0 + SK(4kq + s)]   (TRANS: W^T)
template <bool TRANS, int HM, int HK>
__device__ __forceinline__ f32x4 wfrag(const float* __restrict__ W, int ld, int row0, int col0, int lane, float mul = 1.f) {
  const int m = sig<HM>(lane & 15), kq = lane >> 4;
  f32x4 f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = sig<HK>(4 * kq + s);
    f[s] = mul * (TRANS ? W[(long)(col0 + k) * ld + row0 + m] : W[(long)(row0 + m) * ld + col0 + k]);
  }
  return f;
}
// bias / vector in TL: element r = v[base + S(4kq + r)]
template <int HS>
__device__ __forceinline__ f32x4 vfrag(const float* __restrict__ v, int base, int lane, float mul = 1.f) {
  const int kq = lane >> 4;
  return f32x4{mul * v[base + sig<HS>(4 * kq)], mul * v[base + sig<HS>(4 * kq + 1)], mul * v[base + sig<HS>(4 * kq + 2)],
               mul * v[base + sig<HS>(4 * kq + 3)]};
}

// Butterfly over the 4 lanes (kq = 0..3: lane xor 16, xor 32) that hold the channels of one token, with gfx950's
// v_permlane16_swap / v_permlane32_swap: one VALU instruction per step instead of a ds_bpermute round trip through LDS.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float chan_sum(float v) {
  u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float chan_max(float v) {
  u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int CT, int M>
__device__ __forceinline__ void layer_norm_tl(const f32x4 (&x)[CT][M], f32x4 (&xn)[CT][M], float (&rstd)[M], const bool (&keep)[M]) {
  constexpr float invC = 1.f / (16 * CT);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    f32x4 s4 = x[0][m];
#pragma unroll
    for (int c = 1; c < CT; ++c) s4 += x[c][m];
    const float mean = chan_sum(hsum(s4)) * invC;
    f32x4 d[CT], v4 = splat(0.f);
#pragma unroll
    for (int c = 0; c < CT; ++c) { d[c] = x[c][m] - splat(mean); v4 = fma4(d[c], d[c], v4); }
    rstd[m] = rsqrtf(chan_sum(hsum(v4)) * invC + kLnEps);
    const float rk = keep[m] ? rstd[m] : 0.f;      // padded tokens normalise to exactly 0
#pragma unroll
    for (int c = 0; c < CT; ++c) xn[c][m] = d[c] * splat(rk);
  }
}

// Y[OT][MT] += W X   with fragments wf[OT][KT] (4 k-steps each) and X[KT][MT] in TL
template <int OT, int KT, int M>
__device__ __forceinline__ void mm(f32x4 (&Y)[OT][M], const f32x4 (&wf)[OT][KT], const f32x4 (&X)[KT][M]) {
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < M; ++m) Y[o][m] = MFMA(wf[o][k][s], X[k][m][s], Y[o][m]);
}

// the same with the weight fragments read from LDS (lane-fragment order: fragment (o, k) at ((o * KT + k) * 64 + lane) * 4)
template <int OT, int KT, int M>
__device__ __forceinline__ void mm_lds(f32x4 (&Y)[OT][M], const float* wl, const f32x4 (&X)[KT][M], int lane) {
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const f32x4 w = ld4(wl + (o * KT + k) * 256 + lane * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < M; ++m) Y[o][m] = MFMA(w[s], X[k][m][s], Y[o][m]);
    }
}


// ---------------------------------------------------------------------------------------------
// Split-bf16 GEMMs ("bf16x4"): an f32 operand x is cut into two bf16 pieces x = hi + lo + O(2^-18 |x|)
// (hi = RNE_bf16(x), lo = RNE_bf16(x - hi)) and the product (a_hi + a_lo)(b_hi + b_lo) is formed EXACTLY (bf16 x bf16
// products are exact in f32) by TWO v_mfma_f32_16x16x32_bf16 with the pieces concatenated along K: a 16-deep f32 GEMM
// step (4 x v_mfma_f32_16x16x4_f32, 128 cycles) becomes 2 x 16 cycles on the bf16 matrix pipe, which — unlike the
// f32-input MFMA (tools/micro/coexec.hip) — overlaps with VALU work (tools/micro/coexec_bf16.hip).
// K-slot map of the 16x16x32 operands (lane = 16 g + i; slot jj = 0..7 of lane group g):
//     B (column i):  jj < 4 -> hi piece of k = 4 g + jj      jj >= 4 -> lo piece of k = 4 g + jj - 4
//     A (row i):     jj < 4 -> piece P of k = 4 g + jj       jj >= 4 -> the same piece P again      (P = hi, then lo)
// so MFMA(A_hi, B) + MFMA(A_lo, B) = sum_k (a_hi + a_lo)[k] (b_hi + b_lo)[k] with the (g, jj & 3) <-> k map of the f32
// 16x16x4 chain — every fragment layout of this file ("token layout", weight fragments, D tiles) carries over unchanged.
// Operand rounding 2^-18 relative (vs 2^-24 for f32): ~2e-6 relative error per GEMM, 100x inside the 2e-4 activation
// tolerance the parity tests state; IDEE_SWIN_MATH=f32 selects the exact-f32 kernels.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4 frag_b(const Split& p) { return u32x4{p.h01, p.h23, p.l01, p.l23}; }

// weight tile -> the two A fragments (hi | hi), (lo | lo), stored at ((tile * 2 + piece) * 64 + lane) * 4 words
// (PM_BF16X1: only the hi fragment exists; the B operands carry zeros in their upper K half)
template <int F16>
__device__ __forceinline__ void put_wfrag_b(unsigned* wl, int tile, f32x4 w, int lane) {
  const Split p = split2<F16>(w);
  *reinterpret_cast<u32x4*>(wl + ((tile * 2 + 0) * 64 + lane) * 4) = u32x4{p.h01, p.h23, p.h01, p.h23};
  if (F16 != PM_BF16X1) *reinterpret_cast<u32x4*>(wl + ((tile * 2 + 1) * 64 + lane) * 4) = u32x4{p.l01, p.l23, p.l01, p.l23};
}
__device__ __forceinline__ u32x4 get_wfrag_b(const unsigned* wl, int tile, int piece, int lane) {
  return *reinterpret_cast<const u32x4*>(wl + ((tile * 2 + piece) * 64 + lane) * 4);
}
// The same with ONE weight fragment per tile, [w_hi | w_lo] (half the LDS: what lets the C = 32 attention backward keep four
// waves per workgroup), the duplication moved to the B side: B1 = [x_hi | x_hi], B2 = [x_lo | x_lo] (register moves).
// WDUP selects the layout: tile t lives at (t * 2 + piece) * 256 words (duplicated) or at t * 256 words (single).
template <int F16>
__device__ __forceinline__ void put_wfrag_u(unsigned* wl, int tile, f32x4 w, int lane) {
  const Split p = split2<F16>(w);
  *reinterpret_cast<u32x4*>(wl + (tile * 64 + lane) * 4) = u32x4{p.h01, p.h23, p.l01, p.l23};
}
template <int F16, bool WDUP>
__device__ __forceinline__ void put_wfrag_x(unsigned* wl, int tile, f32x4 w, int lane) {
  if (WDUP) put_wfrag_b<F16>(wl, tile, w, lane); else put_wfrag_u<F16>(wl, tile, w, lane);
}
// acc += W-tile x B for M column blocks, both layouts (xb = [x_hi | x_lo] fragments)
template <int F16, bool WDUP, int M>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[M], const unsigned* wl, int tile, const u32x4 (&xb)[M], int lane) {
  if (WDUP) {
    const u32x4 ah = get_wfrag_b(wl, tile, 0, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = mfma_s<F16>(ah, xb[m], acc[m]);
    if (F16 != PM_BF16X1) {
      const u32x4 al = get_wfrag_b(wl, tile, 1, lane);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = mfma_s<F16>(al, xb[m], acc[m]);
    }
  } else {
    const u32x4 a = *reinterpret_cast<const u32x4*>(wl + (tile * 64 + lane) * 4);     // [w_hi | w_lo]  (PM_BF16X1: [w | 0])
    if (F16 == PM_BF16X1) {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = mfma_s<F16>(a, xb[m], acc[m]);             // xb = [x | 0]
    } else {
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = mfma_s<F16>(a, u32x4{xb[m][0], xb[m][1], xb[m][0], xb[m][1]}, acc[m]);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = mfma_s<F16>(a, u32x4{xb[m][2], xb[m][3], xb[m][2], xb[m][3]}, acc[m]);
    }
  }
}
template <int F16, bool WDUP, int OT, int KT, int M>
__device__ __forceinline__ void mm_x(f32x4 (&Y)[OT][M], const unsigned* wl, int tile0, const u32x4 (&XB)[KT][M], int lane) {
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int k = 0; k < KT; ++k) mma_tile<F16, WDUP, M>(Y[o], wl, tile0 + o * KT + k, XB[k], lane);
}

// Y[OT][M] += W X with the weight fragments of tiles tile0 + o * KT + k in LDS and X already split (B fragments)
template <int F16, int OT, int KT, int M>
__device__ __forceinline__ void mm_b(f32x4 (&Y)[OT][M], const unsigned* wl, int tile0, const u32x4 (&XB)[KT][M], int lane) {
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int k = 0; k < KT; ++k) mma_tile<F16, true, M>(Y[o], wl, tile0 + o * KT + k, XB[k], lane);
}

// bf16 piece images for the token-reduction GEMMs: one 64-byte row per token = 8 slots of 8 bytes (slots 0-3: the hi
// pieces of positions 4s..4s+3, slots 4-7: their lo pieces).  A lane of the token layout owns positions 4kq..4kq+3 of its
// token: one 8-byte store per piece.  Read back TRANSPOSED by ds_read_b64_tr_b16: lane (row i, group g) receives position i
// of tokens 4g..4g+3 — the K = tokens operand layout.  The slot index is XOR-swizzled with the row,
//     slot' = slot ^ (row & 7) ^ ((row >> 3) & 1),
// which makes BOTH access patterns bank-conflict-free: the 16 lanes of a ds_write_b64 group (16 consecutive rows, one slot;
// LDS writes bank on 32 words) land on 32 distinct words, and the 32 lanes of a transposed-read half (8 consecutive rows x
// 4 slots; reads bank on 64 words) on 64 distinct words.  (An un-swizzled padded 96-byte row was 4-way conflicted on the
// writes: SQ_LDS_BANK_CONFLICT was 59 % of SQ_LDS_IDX_ACTIVE in the attention backward, profiles/r02_sq_counters.txt.)
constexpr int kImgRow = 16;   // 32-bit words per token row
__device__ __forceinline__ int img_word(int row, int slot) { return row * kImgRow + 2 * (slot ^ (row & 7) ^ ((row >> 3) & 1)); }
template <int PM = PM_BF16X2>
__device__ __forceinline__ void put_img_b(unsigned* img, int row, int kq, const Split& p) {
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u32x2v*>(img + img_word(row, kq)) = u32x2v{p.h01, p.h23};
  if (PM != PM_BF16X1) *reinterpret_cast<u32x2v*>(img + img_word(row, 4 + kq)) = u32x2v{p.l01, p.l23};
}
// the 4 tokens 16 m + 4 g + {0..3} of position (lane & 15), piece `lo`
__device__ __forceinline__ unsigned long long tr_read(const unsigned* img, int m, int lo, int lane) {
  const int row = 16 * m + 4 * (lane >> 4) + ((lane & 15) >> 2);
  const unsigned* p = img + img_word(row, 4 * lo + (lane & 3));
  const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(unsigned long long, t);
}
// acc[a][b] += sum over the wave's 16 M tokens of A[a][tok] B[b][tok]   (positions a of imgA on rows, b of imgB on columns)
// PM_BF16X1: one piece — the K = 32 slots of an MFMA hold TWO 16-token blocks (M even)
template <int M, int PM = PM_BF16X2>
__device__ __forceinline__ void gemm_tok_b(f32x4& acc, const unsigned* imgA, const unsigned* imgB, int lane) {
  if (PM == PM_BF16X1) {
    static_assert(PM != PM_BF16X1 || M % 2 == 0, "one-piece token GEMM pairs the 16-token blocks");
#pragma unroll
    for (int m = 0; m + 1 < M; m += 2) {
      const unsigned long long a0 = tr_read(imgA, m, 0, lane), a1 = tr_read(imgA, m + 1, 0, lane);
      const unsigned long long b0 = tr_read(imgB, m, 0, lane), b1 = tr_read(imgB, m + 1, 0, lane);
      acc = mfma_s<PM_BF16X1>(u32x4{(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)},
                              u32x4{(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)}, acc);
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const unsigned long long ah = tr_read(imgA, m, 0, lane), al = tr_read(imgA, m, 1, lane);
    const unsigned long long bh = tr_read(imgB, m, 0, lane), bl = tr_read(imgB, m, 1, lane);
    const u32x4 B = {(unsigned)bh, (unsigned)(bh >> 32), (unsigned)bl, (unsigned)(bl >> 32)};
    acc = mfma_s<false>(u32x4{(unsigned)ah, (unsigned)(ah >> 32), (unsigned)ah, (unsigned)(ah >> 32)}, B, acc);
    acc = mfma_s<false>(u32x4{(unsigned)al, (unsigned)(al >> 32), (unsigned)al, (unsigned)(al >> 32)}, B, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int C, int NH, int NW, bool SAVE_H>
__global__ __launch_bounds__(256, C == 16 ? 2 : 1) void swin_fwd_mfma_kernel(
    const float* __restrict__ x, float* __restrict__ y, float* __restrict__ hsave, const float* __restrict__ qkv_w,
    const float* __restrict__ qkv_b, const float* __restrict__ proj_w, const float* __restrict__ proj_b,
    const float* __restrict__ fc1_w, const float* __restrict__ fc1_b, const float* __restrict__ fc2_w,
    const float* __restrict__ fc2_b, const float* __restrict__ rel_bias, Geom g) {
  constexpr int CT = C / 16, HID = 4 * C, HT = HID / 16, HD = C / NH;
  static_assert(C % 16 == 0 && HD <= 16 && 16 % HD == 0, "head_dim must divide 16");
  constexpr int HPC = 16 / HD;             // heads per 16-channel chunk
  constexpr int SP = 4 / HPC;              // k-steps per head inside a chunk
  constexpr int TPW = NW >= 16 ? NW / 16 : 1;   // column blocks per window group
  constexpr int NG = MT / TPW;             // window groups per wave
    __shared__ __attribute__((aligned(16))) float s_vt[4][C * VS];
  __shared__ __attribute__((aligned(16))) int s_rid[4][64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, v = b % g.V;
  const float ps1 = g.path_scale ? g.path_scale[2 * b] : 1.f, ps2 = g.path_scale ? g.path_scale[2 * b + 1] : 1.f;
  qkv_w += (long)v * 3 * C * C;  qkv_b += v * 3 * C;
  proj_w += (long)v * C * C;     proj_b += v * C;
  fc1_w += (long)v * HID * C;    fc1_b += v * HID;
  fc2_w += (long)v * C * HID;    fc2_b += v * C;
  rel_bias += (long)v * NH * NW * NW;

  // ---- resident weight fragments ----
  f32x4 wq[3 * CT][CT], bq[3 * CT], wp[CT][CT], bp[CT], w1[HT][CT], b1[HT], w2[CT][HT], b2[CT];
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    const float mul = o < CT ? g.scale : 1.f;   // q pre-scaled (Swin_3D.py:155)
    bq[o] = vfrag<HPC>(qkv_b, 16 * o, lane, mul);
#pragma unroll
    for (int k = 0; k < CT; ++k) wq[o][k] = wfrag<false, HPC, 1>(qkv_w, C, 16 * o, 16 * k, lane, mul);
  }
#pragma unroll
  for (int o = 0; o < CT; ++o) {
    bp[o] = vfrag<1>(proj_b, 16 * o, lane);
    b2[o] = vfrag<1>(fc2_b, 16 * o, lane);
#pragma unroll
    for (int k = 0; k < CT; ++k) wp[o][k] = wfrag<false, 1, HPC>(proj_w, C, 16 * o, 16 * k, lane);
#pragma unroll
    for (int k = 0; k < HT; ++k) w2[o][k] = wfrag<false, 1, 1>(fc2_w, HID, 16 * o, 16 * k, lane);
  }
#pragma unroll
  for (int o = 0; o < HT; ++o) {
    b1[o] = vfrag<1>(fc1_b, 16 * o, lane);
#pragma unroll
    for (int k = 0; k < CT; ++k) w1[o][k] = wfrag<false, 1, 1>(fc1_w, C, 16 * o, 16 * k, lane);
  }
  // relative-position bias per (head, key block, query block): rows = keys 4kq+r, col = query j;
  // for NW < 16 the cross-window entries of a column block are masked here once and for all
  f32x4 rb[NH][TPW][TPW];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
      for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * kq + r, qry = 16 * qt + j;
          const bool same = (key / NW) == (qry / NW);
          rb[h][kt][qt][r] = same ? rel_bias[(h * NW + qry % NW) * NW + key % NW] : -1e30f;
        }

  float* vt = s_vt[wave];
  int* rid_l = s_rid[wave];

  for (int it0 = blockIdx.x * 4; it0 < g.items; it0 += gridDim.x * 4) {
    Tok t[MT];
    bool keep[MT];
    f32x4 X[CT][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      t[m] = locate<NW>(g, b, it0 + wave, 16 * m + j, C);
      keep[m] = t[m].valid;
#pragma unroll
      for (int c = 0; c < CT; ++c) X[c][m] = t[m].valid ? ld4(x + t[m].off + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 XN[CT][MT];
    float rstd[MT];
    layer_norm_tl(X, XN, rstd, keep);   // zero padding is applied AFTER norm1 (Swin_3D.py:229-238)

    f32x4 QKV[3 * CT][MT];
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o)
#pragma unroll
      for (int m = 0; m < MT; ++m) QKV[o][m] = bq[o];
    mm(QKV, wq, XN);

    wave_sync();   // previous iteration's readers of vt / rid are done
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) vt[(16 * c + sig<HPC>(4 * kq + r)) * VS + 16 * m + j] = QKV[2 * CT + c][m][r];
    if (kq == 0)
#pragma unroll
      for (int m = 0; m < MT; ++m) rid_l[16 * m + j] = t[m].rid;
    wave_sync();

    f32x4 O[CT][MT];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = h / HPC, s0 = (h % HPC) * SP;
        f32x4 S[TPW][TPW];
        // the k-steps outermost: the TPW^2 tiles are independent accumulators, so no MFMA waits on its predecessor
#pragma unroll
        for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) S[kt][qt] = rb[h][kt][qt];
#pragma unroll
        for (int s = 0; s < SP; ++s)
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt)
              S[kt][qt] = MFMA(QKV[CT + c][grp * TPW + kt][s0 + s], QKV[c][grp * TPW + qt][s0 + s], S[kt][qt]);
        if (g.shifted) {
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const int4 rk = *reinterpret_cast<const int4*>(rid_l + 16 * (grp * TPW + kt) + 4 * kq);
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt) {
              const int rq = t[grp * TPW + qt].rid;
              S[kt][qt][0] += rk.x != rq ? -100.f : 0.f; S[kt][qt][1] += rk.y != rq ? -100.f : 0.f;
              S[kt][qt][2] += rk.z != rq ? -100.f : 0.f; S[kt][qt][3] += rk.w != rq ? -100.f : 0.f;
            }
          }
        }
        // softmax over keys (rows) for each query column
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          float mx = -3.0e38f;
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(S[kt][qt][0], S[kt][qt][1]), fmaxf(S[kt][qt][2], S[kt][qt][3])));
          mx = chan_max(mx);
          const float mxl = -mx * kLog2e;
          f32x4 l4 = splat(0.f);
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 ea = fma4(S[kt][qt], splat(kLog2e), splat(mxl));
#pragma unroll
            for (int r = 0; r < 4; ++r) S[kt][qt][r] = __builtin_amdgcn_exp2f(ea[r]);
            l4 += S[kt][qt];
          }
          const float inv = __builtin_amdgcn_rcpf(chan_sum(hsum(l4)));   // v_rcp_f32 (1 ulp), not the 10-instruction IEEE division
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) S[kt][qt] *= inv;
        }
        // O[chan][query] = sum_keys V^T[chan][key] P[key][query]; keep this head's rows
        {
          f32x4 acc[TPW];   // the query blocks interleaved: independent accumulators between dependent MFMAs
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) acc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 vf = ld4(vt + (16 * c + sig<HPC>(lane & 15)) * VS + 16 * (grp * TPW + kt) + 4 * kq);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int qt = 0; qt < TPW; ++qt) acc[qt] = MFMA(vf[s], S[kt][qt][s], acc[qt]);
          }
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
            for (int r = 0; r < SP; ++r) O[c][grp * TPW + qt][s0 + r] = acc[qt][s0 + r];
        }
      }
    }

    // proj + shortcut
    f32x4 Hs[CT][MT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) Hs[c][m] = bp[c];
    mm(Hs, wp, O);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) Hs[c][m] = fma4(Hs[c][m], splat(ps1), X[c][m]);     // shortcut + drop_path(attention branch)
    if (SAVE_H) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (t[m].valid)
#pragma unroll
          for (int c = 0; c < CT; ++c) st4(hsave + t[m].off + 16 * c + 4 * kq, Hs[c][m]);
    }
    f32x4 HN[CT][MT];
    float rstd2[MT];
    bool all[MT] = {true, true, true, true};
    layer_norm_tl(Hs, HN, rstd2, all);
    f32x4 A1[HT][MT];
#pragma unroll
    for (int o = 0; o < HT; ++o)
#pragma unroll
      for (int m = 0; m < MT; ++m) A1[o][m] = b1[o];
    mm(A1, w1, HN);
#pragma unroll
    for (int o = 0; o < HT; ++o)
#pragma unroll
      for (int m = 0; m < MT; ++m) A1[o][m] = gelu4(A1[o][m]);
    f32x4 Y[CT][MT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) Y[c][m] = b2[c];
    mm(Y, w2, A1);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m) Y[c][m] = fma4(Y[c][m], splat(ps2), Hs[c][m]);       // x + drop_path(mlp branch)
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (t[m].valid)
#pragma unroll
        for (int c = 0; c < CT; ++c) st4(y + t[m].off + 16 * c + 4 * kq, Y[c][m]);
  }
}


// ---------------------------------------------------------------------------------------------
// forward with the four linear layers (qkv, proj, fc1, fc2: 75 % of the block's FLOPs) on the split-bf16 MFMA path;
// the attention core (K Q^T, softmax, V^T P) stays on the f32 MFMA.  Same arguments and layout as swin_fwd_mfma_kernel.
// ---------------------------------------------------------------------------------------------
template <int C, int NH, int NW, bool SAVE_H, int PM /*piece mode of the linear-layer GEMMs*/, typename AT /*token storage*/>
__global__ __launch_bounds__(256, C == 16 ? 2 : 1) void swin_fwd_b_kernel(
    const AT* __restrict__ x, AT* __restrict__ y, AT* __restrict__ hsave, const float* __restrict__ qkv_w,
    const float* __restrict__ qkv_b, const float* __restrict__ proj_w, const float* __restrict__ proj_b,
    const float* __restrict__ fc1_w, const float* __restrict__ fc1_b, const float* __restrict__ fc2_w,
    const float* __restrict__ fc2_b, const float* __restrict__ rel_bias, Geom g) {
  constexpr int CT = C / 16, HID = 4 * C, HT = HID / 16, HD = C / NH;
  static_assert(C % 16 == 0 && HD <= 16 && 16 % HD == 0, "head_dim must divide 16");
  constexpr int HPC = 16 / HD;             // heads per 16-channel chunk
  constexpr int SP = 4 / HPC;              // k-steps per head inside a chunk
  constexpr int TPW = NW >= 16 ? NW / 16 : 1;   // column blocks per window group
  constexpr int NG = MT / TPW;             // window groups per wave
    __shared__ __attribute__((aligned(16))) float s_vt[4][C * VS];
  __shared__ __attribute__((aligned(16))) int s_rid[4][64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, v = b % g.V;
  const float ps1 = g.path_scale ? g.path_scale[2 * b] : 1.f, ps2 = g.path_scale ? g.path_scale[2 * b + 1] : 1.f;
  qkv_w += (long)v * 3 * C * C;  qkv_b += v * 3 * C;
  proj_w += (long)v * C * C;     proj_b += v * C;
  fc1_w += (long)v * HID * C;    fc1_b += v * HID;
  fc2_w += (long)v * C * HID;    fc2_b += v * C;
  rel_bias += (long)v * NH * NW * NW;

  // ---- weight fragments: split-bf16 A fragments in LDS (shared by the 4 waves), biases as f32 fragments ----
  constexpr int T_Q = 0, T_P = 3 * CT * CT, T_1 = T_P + CT * CT, T_2 = T_1 + HT * CT, NTILE = T_2 + CT * HT;
  constexpr int B_Q = 0, B_P = 3 * CT, B_1 = B_P + CT, B_2 = B_1 + HT, NBIAS = B_2 + CT;
  __shared__ __attribute__((aligned(16))) unsigned s_wf[NTILE * 2 * 256];
  __shared__ __attribute__((aligned(16))) float s_bias[NBIAS * 256];
  {
    int i = 0;
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o) {
      const float mul = o < CT ? g.scale : 1.f;   // q pre-scaled (Swin_3D.py:155)
      if ((i++) % 4 == wave) st4(s_bias + (B_Q + o) * 256 + lane * 4, vfrag<HPC>(qkv_b, 16 * o, lane, mul));
#pragma unroll
      for (int k = 0; k < CT; ++k)
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, T_Q + o * CT + k, wfrag<false, HPC, 1>(qkv_w, C, 16 * o, 16 * k, lane, mul), lane);
    }
#pragma unroll
    for (int o = 0; o < CT; ++o) {
      if ((i++) % 4 == wave) st4(s_bias + (B_P + o) * 256 + lane * 4, vfrag<1>(proj_b, 16 * o, lane));
      if ((i++) % 4 == wave) st4(s_bias + (B_2 + o) * 256 + lane * 4, vfrag<1>(fc2_b, 16 * o, lane));
#pragma unroll
      for (int k = 0; k < CT; ++k)
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, T_P + o * CT + k, wfrag<false, 1, HPC>(proj_w, C, 16 * o, 16 * k, lane), lane);
#pragma unroll
      for (int k = 0; k < HT; ++k)
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, T_2 + o * HT + k, wfrag<false, 1, 1>(fc2_w, HID, 16 * o, 16 * k, lane), lane);
    }
#pragma unroll
    for (int o = 0; o < HT; ++o) {
      if ((i++) % 4 == wave) st4(s_bias + (B_1 + o) * 256 + lane * 4, vfrag<1>(fc1_b, 16 * o, lane));
#pragma unroll
      for (int k = 0; k < CT; ++k)
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, T_1 + o * CT + k, wfrag<false, 1, 1>(fc1_w, C, 16 * o, 16 * k, lane), lane);
    }
    __syncthreads();
  }
  auto BIAS = [&](int i) { return ld4(s_bias + i * 256 + lane * 4); };
  // relative-position bias per (head, key block, query block): rows = keys 4kq+r, col = query j;
  // for NW < 16 the cross-window entries of a column block are masked here once and for all
  f32x4 rb[NH][TPW][TPW];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
      for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * kq + r, qry = 16 * qt + j;
          const bool same = (key / NW) == (qry / NW);
          rb[h][kt][qt][r] = same ? rel_bias[(h * NW + qry % NW) * NW + key % NW] : -1e30f;
        }

  float* vt = s_vt[wave];
  int* rid_l = s_rid[wave];

  for (int it0 = blockIdx.x * 4; it0 < g.items; it0 += gridDim.x * 4) {
    Tok t[MT];
    bool keep[MT];
    f32x4 X[CT][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      t[m] = locate<NW>(g, b, it0 + wave, 16 * m + j, C);
      keep[m] = t[m].valid;
#pragma unroll
      for (int c = 0; c < CT; ++c) X[c][m] = t[m].valid ? ldx4<AT>(x + t[m].off + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 XN[CT][MT];
    float rstd[MT];
    layer_norm_tl(X, XN, rstd, keep);   // zero padding is applied AFTER norm1 (Swin_3D.py:229-238)

    f32x4 QKV[3 * CT][MT];
    {
      u32x4 bXN[CT][MT];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) bXN[c][m] = frag_b(split2<PM>(XN[c][m]));
#pragma unroll
      for (int o = 0; o < 3 * CT; ++o) {
        const f32x4 bqo = BIAS(B_Q + o);
#pragma unroll
        for (int m = 0; m < MT; ++m) QKV[o][m] = bqo;
      }
      mm_b<PM, 3 * CT, CT, MT>(QKV, s_wf, T_Q, bXN, lane);
    }

    wave_sync();   // previous iteration's readers of vt / rid are done
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) vt[(16 * c + sig<HPC>(4 * kq + r)) * VS + 16 * m + j] = QKV[2 * CT + c][m][r];
    if (kq == 0)
#pragma unroll
      for (int m = 0; m < MT; ++m) rid_l[16 * m + j] = t[m].rid;
    wave_sync();

    f32x4 O[CT][MT];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = h / HPC, s0 = (h % HPC) * SP;
        f32x4 S[TPW][TPW];
        // the k-steps outermost: the TPW^2 tiles are independent accumulators, so no MFMA waits on its predecessor
#pragma unroll
        for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) S[kt][qt] = rb[h][kt][qt];
#pragma unroll
        for (int s = 0; s < SP; ++s)
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt)
              S[kt][qt] = MFMA(QKV[CT + c][grp * TPW + kt][s0 + s], QKV[c][grp * TPW + qt][s0 + s], S[kt][qt]);
        if (g.shifted) {
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const int4 rk = *reinterpret_cast<const int4*>(rid_l + 16 * (grp * TPW + kt) + 4 * kq);
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt) {
              const int rq = t[grp * TPW + qt].rid;
              S[kt][qt][0] += rk.x != rq ? -100.f : 0.f; S[kt][qt][1] += rk.y != rq ? -100.f : 0.f;
              S[kt][qt][2] += rk.z != rq ? -100.f : 0.f; S[kt][qt][3] += rk.w != rq ? -100.f : 0.f;
            }
          }
        }
        // softmax over keys (rows) for each query column
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          float mx = -3.0e38f;
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(S[kt][qt][0], S[kt][qt][1]), fmaxf(S[kt][qt][2], S[kt][qt][3])));
          mx = chan_max(mx);
          const float mxl = -mx * kLog2e;
          f32x4 l4 = splat(0.f);
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 ea = fma4(S[kt][qt], splat(kLog2e), splat(mxl));
#pragma unroll
            for (int r = 0; r < 4; ++r) S[kt][qt][r] = __builtin_amdgcn_exp2f(ea[r]);
            l4 += S[kt][qt];
          }
          const float inv = __builtin_amdgcn_rcpf(chan_sum(hsum(l4)));   // v_rcp_f32 (1 ulp), not the 10-instruction IEEE division
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) S[kt][qt] *= inv;
        }
        // O[chan][query] = sum_keys V^T[chan][key] P[key][query]; keep this head's rows
        {
          f32x4 acc[TPW];   // the query blocks interleaved: independent accumulators between dependent MFMAs
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) acc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 vf = ld4(vt + (16 * c + sig<HPC>(lane & 15)) * VS + 16 * (grp * TPW + kt) + 4 * kq);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int qt = 0; qt < TPW; ++qt) acc[qt] = MFMA(vf[s], S[kt][qt][s], acc[qt]);
          }
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
            for (int r = 0; r < SP; ++r) O[c][grp * TPW + qt][s0 + r] = acc[qt][s0 + r];
        }
      }
    }

    // proj + shortcut
    f32x4 Hs[CT][MT];
    {
      u32x4 bO[CT][MT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x4 bpc = BIAS(B_P + c);
#pragma unroll
        for (int m = 0; m < MT; ++m) { Hs[c][m] = bpc; bO[c][m] = frag_b(split2<PM>(O[c][m])); }
      }
      mm_b<PM, CT, CT, MT>(Hs, s_wf, T_P, bO, lane);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) Hs[c][m] = fma4(Hs[c][m], splat(ps1), X[c][m]);   // shortcut + drop_path(attention branch)
    }
    if (SAVE_H) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (t[m].valid)
#pragma unroll
          for (int c = 0; c < CT; ++c) stx4<AT>(hsave + t[m].off + 16 * c + 4 * kq, Hs[c][m]);
    }
    f32x4 HN[CT][MT];
    float rstd2[MT];
    bool all[MT] = {true, true, true, true};
    layer_norm_tl(Hs, HN, rstd2, all);
    f32x4 Y[CT][MT];
    {
      u32x4 bHN[CT][MT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x4 b2c = BIAS(B_2 + c);
#pragma unroll
        for (int m = 0; m < MT; ++m) { Y[c][m] = b2c; bHN[c][m] = frag_b(split2<PM>(HN[c][m])); }
      }
      // one hidden 16-block at a time: fc1 -> GELU -> split -> fc2 (the hidden activation never exists as a whole)
#pragma unroll
      for (int ho = 0; ho < HT; ++ho) {
        f32x4 A1[1][MT];
        const f32x4 b1o = BIAS(B_1 + ho);
#pragma unroll
        for (int m = 0; m < MT; ++m) A1[0][m] = b1o;
        mm_b<PM, 1, CT, MT>(A1, s_wf, T_1 + ho * CT, bHN, lane);
        u32x4 bG[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) bG[m] = frag_b(split2<PM>(gelu4(A1[0][m])));
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          mma_tile<PM, true, MT>(Y[c], s_wf, T_2 + c * HT + ho, bG, lane);
        }
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) Y[c][m] = fma4(Y[c][m], splat(ps2), Hs[c][m]);     // x + drop_path(mlp branch)
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (t[m].valid)
#pragma unroll
        for (int c = 0; c < CT; ++c) stx4<AT>(y + t[m].off + 16 * c + 4 * kq, Y[c][m]);
  }
}



// ---------------------------------------------------------------------------------------------
// backward helpers: token-reduction GEMMs (weight gradients) through LDS images [channel][token]
// ---------------------------------------------------------------------------------------------

// TL tile -> image rows in REAL channel order (permutation HS undone on the way out)
template <int HS, int M, int VSN = VS>
__device__ __forceinline__ void put_img(float* img, const f32x4 (&T)[M], int lane) {
  const int j = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) img[sig<HS>(4 * kq + r) * VSN + 16 * m + j] = T[m][r];
}
// acc[m][n] += sum_tok A[m][tok] B[n][tok]   over the wave's 64 tokens
template <int M = MT, int VSN = VS>
__device__ __forceinline__ void gemm_tok(f32x4& acc, const float* A, const float* B, int lane) {
  const int row = (lane & 15) * VSN + 4 * (lane >> 4);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const f32x4 fa = ld4(A + row + 16 * m), fb = ld4(B + row + 16 * m);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA(fa[s], fb[s], acc);
  }
}
// D layout: rows 4kq+r, col j -> dst[(row0 + 4kq + r) * ld + col0 + j]
__device__ __forceinline__ void flush_acc(float* dst, int ld, const f32x4& acc, int lane) {
  const int j = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(dst + (long)(r0 + r) * ld + j, acc[r]);
}
// per-lane partial sums over tokens of a TL vector: reduce over the 16 token lanes, add at channel HS(4kq+r)
template <int HS>
__device__ __forceinline__ void flush_vec(float* dst, f32x4 part, int lane) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    part[0] += __shfl_xor(part[0], o, 64); part[1] += __shfl_xor(part[1], o, 64);
    part[2] += __shfl_xor(part[2], o, 64); part[3] += __shfl_xor(part[3], o, 64);
  }
  if ((lane & 15) == 0) {
    const int kq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(dst + sig<HS>(4 * kq + r), part[r]);
  }
}

// dx = rstd * (dxn - mean(dxn) - xn * mean(dxn * xn)) + add, per token in TL
template <int CT, int M>
__device__ __forceinline__ void layer_norm_bwd_tl(const f32x4 (&dxn)[CT][M], const f32x4 (&xn)[CT][M], const float (&rstd)[M],
                                                  const f32x4 (&add)[CT][M], f32x4 (&dx)[CT][M]) {
  constexpr float invC = 1.f / (16 * CT);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    f32x4 a4 = dxn[0][m], b4 = dxn[0][m] * xn[0][m];
#pragma unroll
    for (int c = 1; c < CT; ++c) { a4 += dxn[c][m]; b4 = fma4(dxn[c][m], xn[c][m], b4); }
    const float a = chan_sum(hsum(a4)) * invC;
    const float b = chan_sum(hsum(b4)) * invC;
#pragma unroll
    for (int c = 0; c < CT; ++c) dx[c][m] = fma4(splat(rstd[m]), (dxn[c][m] - splat(a)) - xn[c][m] * splat(b), add[c][m]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1: MLP + norm2 (pure per-token; tokens in memory order)
//   in: h (block state before the MLP, saved by the forward), dy.  out: dh = dL/dh, MLP weight grads.
// ---------------------------------------------------------------------------------------------
template <int C, int M>
__global__ __launch_bounds__(256, M == 2 ? 2 : 1) void swin_mlp_bwd_kernel(
    const float* __restrict__ hs, const float* __restrict__ dy, float* __restrict__ dh, const float* __restrict__ fc1_w,
    const float* __restrict__ fc1_b, const float* __restrict__ fc2_w, float* __restrict__ g_fc1_w,
    float* __restrict__ g_fc1_b, float* __restrict__ g_fc2_w, float* __restrict__ g_fc2_b, int V, long L,
    const float* __restrict__ path_scale) {
  constexpr int CT = C / 16, HID = 4 * C, HT = HID / 16;
  constexpr int HP = 2;                 // hidden 16-blocks per pass (keeps the live set within 256 registers)
  constexpr int TOK = 16 * M;           // tokens per wave and iteration
  static_assert(HT % HP == 0, "hidden blocks per pass");
  constexpr int VSM = TOK + 4;          // image row stride (floats)
  __shared__ __attribute__((aligned(16))) float s_img[4][(2 * CT + 2) * 16 * VSM];
  // weight fragments in lane-fragment order (fragment i, lane l -> 16 bytes at (i * 64 + l) * 4), shared by the 4 waves:
  // one ds_read_b128 where used instead of four strided global loads per fragment and pass
  constexpr int F_W1 = 0, F_W2T = HT * CT, F_W1T = 2 * HT * CT, F_B1 = 3 * HT * CT, NF = 3 * HT * CT + HT;
  __shared__ __attribute__((aligned(16))) float s_wf[NF * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, v = b % V;
  fc1_w += (long)v * HID * C;  fc1_b += v * HID;  fc2_w += (long)v * C * HID;
  g_fc1_w += (long)v * HID * C;  g_fc1_b += v * HID;  g_fc2_w += (long)v * C * HID;  g_fc2_b += v * C;
  hs += (long)b * L * C;  dy += (long)b * L * C;  dh += (long)b * L * C;
  const float ps2 = path_scale ? path_scale[2 * b + 1] : 1.f;

  f32x4 aW1[HT][CT], aW2[CT][HT], pb1[HT], pb2[CT];
#pragma unroll
  for (int o = 0; o < HT; ++o) {
    pb1[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT; ++k) { aW1[o][k] = f32x4{0.f, 0.f, 0.f, 0.f}; aW2[k][o] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
#pragma unroll
  for (int k = 0; k < CT; ++k) pb2[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  {
    int i = 0;
#pragma unroll
    for (int ho = 0; ho < HT; ++ho) {
      if ((i++) % 4 == wave) st4(s_wf + (F_B1 + ho) * 256 + lane * 4, vfrag<1>(fc1_b, 16 * ho, lane));
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if ((i++) % 4 == wave) st4(s_wf + (F_W1 + ho * CT + c) * 256 + lane * 4, wfrag<false, 1, 1>(fc1_w, C, 16 * ho, 16 * c, lane));
        if ((i++) % 4 == wave) st4(s_wf + (F_W2T + ho * CT + c) * 256 + lane * 4, wfrag<true, 1, 1>(fc2_w, HID, 16 * ho, 16 * c, lane));   // [hid][c] = fc2_w[c][hid]
        if ((i++) % 4 == wave) st4(s_wf + (F_W1T + c * HT + ho) * 256 + lane * 4, wfrag<true, 1, 1>(fc1_w, C, 16 * c, 16 * ho, lane));     // [c][hid] = fc1_w[hid][c]
      }
    }
    __syncthreads();
  }
  float* img = s_img[wave];
  float* imgHN = img;                       // CT images
  float* imgDY = img + CT * 16 * VSM;        // CT images
  float* imgA = img + 2 * CT * 16 * VSM;     // dA1 tile
  float* imgB = imgA + 16 * VSM;             // gelu tile
  const long items = (L + TOK - 1) / TOK;

  for (long it0 = (long)blockIdx.x * 4; it0 < items; it0 += (long)gridDim.x * 4) {
    const long base = (it0 + wave) * TOK;
    bool valid[M];
    f32x4 HN[CT][M], DY[CT][M];
    float rstd2[M];
    {
      f32x4 Hs[CT][M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const long tok = base + 16 * m + j;
        valid[m] = tok < L;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          Hs[c][m] = valid[m] ? ld4(hs + tok * C + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
          DY[c][m] = valid[m] ? ld4(dy + tok * C + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      bool all[M];
#pragma unroll
      for (int m = 0; m < M; ++m) all[m] = true;
      layer_norm_tl(Hs, HN, rstd2, all);
    }
    f32x4 DYs[CT][M];      // gradient entering the MLP branch: drop_path factor of this batch item times dy
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < M; ++m) DYs[c][m] = DY[c][m] * ps2;
    wave_sync();   // previous iteration's image readers are done
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      put_img<1, M, VSM>(imgHN + c * 16 * VSM, HN[c], lane);
      put_img<1, M, VSM>(imgDY + c * 16 * VSM, DYs[c], lane);
#pragma unroll
      for (int m = 0; m < M; ++m) pb2[c] += DYs[c][m];
    }
    f32x4 dHN[CT][M];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < M; ++m) dHN[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int hp = 0; hp < HT / HP; ++hp) {
      // fragments of this pass come from LDS (re-read each pass instead of pinning 48 registers)
      f32x4 w1[HP][CT], w2t[HP][CT], w1t[CT][HP], A1[HP][M], dG[HP][M];
#pragma unroll
      for (int o = 0; o < HP; ++o) {
        const int ho = hp * HP + o;
        const f32x4 b1 = ld4(s_wf + (F_B1 + ho) * 256 + lane * 4);
#pragma unroll
        for (int k = 0; k < CT; ++k) {
          w1[o][k] = ld4(s_wf + (F_W1 + ho * CT + k) * 256 + lane * 4);
          w2t[o][k] = ld4(s_wf + (F_W2T + ho * CT + k) * 256 + lane * 4);
          w1t[k][o] = ld4(s_wf + (F_W1T + k * HT + ho) * 256 + lane * 4);
        }
#pragma unroll
        for (int m = 0; m < M; ++m) { A1[o][m] = b1; dG[o][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      mm(A1, w1, HN);
      mm(dG, w2t, DYs);
#pragma unroll
      for (int o = 0; o < HP; ++o)
#pragma unroll
        for (int m = 0; m < M; ++m) {
          f32x4 gl, dgl;
          gelu_both4(A1[o][m], gl, dgl);
          dG[o][m] *= dgl;                    // dG now holds dA1
          A1[o][m] = gl;                      // A1 now holds gelu(A1)
          pb1[hp * HP + o] += dG[o][m];
        }
      mm(dHN, w1t, dG);
      // weight gradients of this pass: dW1[hid][c] += dA1 . HN^T ; dW2[c][hid] += DY . G^T  (K = tokens)
#pragma unroll
      for (int o = 0; o < HP; ++o) {
        const int ho = hp * HP + o;
        if (hp + o) wave_sync();
        put_img<1, M, VSM>(imgA, dG[o], lane);
        put_img<1, M, VSM>(imgB, A1[o], lane);
        wave_sync();
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          gemm_tok<M, VSM>(aW1[ho][c], imgA, imgHN + c * 16 * VSM, lane);
          gemm_tok<M, VSM>(aW2[c][ho], imgDY + c * 16 * VSM, imgB, lane);
        }
      }
    }
    f32x4 dH[CT][M];
    layer_norm_bwd_tl(dHN, HN, rstd2, DY, dH);
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (valid[m])
#pragma unroll
        for (int c = 0; c < CT; ++c) st4(dh + (base + 16 * m + j) * C + 16 * c + 4 * kq, dH[c][m]);
  }
#pragma unroll
  for (int o = 0; o < HT; ++o) {
    flush_vec<1>(g_fc1_b + 16 * o, pb1[o], lane);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      flush_acc(g_fc1_w + (long)(16 * o) * C + 16 * c, C, aW1[o][c], lane);
      flush_acc(g_fc2_w + (long)(16 * c) * HID + 16 * o, HID, aW2[c][o], lane);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) flush_vec<1>(g_fc2_b + 16 * c, pb2[c], lane);
}


// ---------------------------------------------------------------------------------------------
// backward, part 1 on the split-bf16 MFMA path (see "Split-bf16 GEMMs" above): same algorithm, same arguments and the
// same results to ~2e-6 relative as swin_mlp_bwd_kernel.  Per 32 tokens: 80 bf16 MFMAs (16 cycles each, overlapping the
// VALU work) instead of 160 f32 MFMAs (32 cycles each, serialised with it) + 20 operand splits.
// ---------------------------------------------------------------------------------------------
template <int C, int M, int PM, typename AT>
__global__ __launch_bounds__(256, C == 16 ? 2 : 1) void swin_mlp_bwd_b_kernel(
    const AT* __restrict__ hs, const AT* __restrict__ dy, AT* __restrict__ dh, const float* __restrict__ fc1_w,
    const float* __restrict__ fc1_b, const float* __restrict__ fc2_w, float* __restrict__ g_fc1_w,
    float* __restrict__ g_fc1_b, float* __restrict__ g_fc2_w, float* __restrict__ g_fc2_b, int V, long L,
    const float* __restrict__ path_scale) {
  constexpr int CT = C / 16, HID = 4 * C, HT = HID / 16;
  constexpr int TOK = 16 * M;
  constexpr int IMGW = TOK * kImgRow;   // words per image (one 16-position chunk, TOK token rows)
  __shared__ __attribute__((aligned(16))) unsigned s_img[4][(2 * CT + 2) * IMGW];
  constexpr int F_W1 = 0, F_W2T = HT * CT, F_W1T = 2 * HT * CT, NT = 3 * HT * CT;
  __shared__ __attribute__((aligned(16))) unsigned s_wf[NT * 2 * 256];
  __shared__ __attribute__((aligned(16))) float s_b1[HT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, v = b % V;
  fc1_w += (long)v * HID * C;  fc1_b += v * HID;  fc2_w += (long)v * C * HID;
  g_fc1_w += (long)v * HID * C;  g_fc1_b += v * HID;  g_fc2_w += (long)v * C * HID;  g_fc2_b += v * C;
  hs += (long)b * L * C;  dy += (long)b * L * C;  dh += (long)b * L * C;
  const float ps2 = path_scale ? path_scale[2 * b + 1] : 1.f;

  f32x4 aW1[HT][CT], aW2[CT][HT], pb1[HT], pb2[CT];
#pragma unroll
  for (int o = 0; o < HT; ++o) {
    pb1[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT; ++k) { aW1[o][k] = f32x4{0.f, 0.f, 0.f, 0.f}; aW2[k][o] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
#pragma unroll
  for (int k = 0; k < CT; ++k) pb2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    int i = 0;
#pragma unroll
    for (int ho = 0; ho < HT; ++ho) {
      if ((i++) % 4 == wave) st4(s_b1 + ho * 256 + lane * 4, vfrag<1>(fc1_b, 16 * ho, lane));
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, F_W1 + ho * CT + c, wfrag<false, 1, 1>(fc1_w, C, 16 * ho, 16 * c, lane), lane);
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, F_W2T + ho * CT + c, wfrag<true, 1, 1>(fc2_w, HID, 16 * ho, 16 * c, lane), lane);
        if ((i++) % 4 == wave) put_wfrag_b<PM>(s_wf, F_W1T + c * HT + ho, wfrag<true, 1, 1>(fc1_w, C, 16 * c, 16 * ho, lane), lane);
      }
    }
    __syncthreads();
  }
  unsigned* img = s_img[wave];
  unsigned* imgHN = img;
  unsigned* imgDY = img + CT * IMGW;
  unsigned* imgA = img + 2 * CT * IMGW;      // dA1 tile
  unsigned* imgB = imgA + IMGW;              // gelu tile
  const long items = (L + TOK - 1) / TOK;

  for (long it0 = (long)blockIdx.x * 4; it0 < items; it0 += (long)gridDim.x * 4) {
    const long base = (it0 + wave) * TOK;
    bool valid[M];
    f32x4 HN[CT][M], DY[CT][M];
    float rstd2[M];
    {
      f32x4 Hs[CT][M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const long tok = base + 16 * m + j;
        valid[m] = tok < L;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          Hs[c][m] = valid[m] ? ldx4<AT>(hs + tok * C + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
          DY[c][m] = valid[m] ? ldx4<AT>(dy + tok * C + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      bool all[M];
#pragma unroll
      for (int m = 0; m < M; ++m) all[m] = true;
      layer_norm_tl(Hs, HN, rstd2, all);
    }
    u32x4 bHN[CT][M], bDY[CT][M];
    wave_sync();   // previous iteration's image readers are done
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const f32x4 dys = DY[c][m] * ps2;       // gradient entering the MLP branch (drop_path factor of this batch item)
        const Split sh = split2<PM>(HN[c][m]), sd = split2<PM>(dys);
        bHN[c][m] = frag_b(sh);
        bDY[c][m] = frag_b(sd);
        put_img_b<PM>(imgHN + c * IMGW, 16 * m + j, kq, sh);
        put_img_b<PM>(imgDY + c * IMGW, 16 * m + j, kq, sd);
        pb2[c] += dys;
      }
    f32x4 dHN[CT][M];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < M; ++m) dHN[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ho = 0; ho < HT; ++ho) {
      f32x4 A1[1][M], dG[1][M];
      const f32x4 b1 = ld4(s_b1 + ho * 256 + lane * 4);
#pragma unroll
      for (int m = 0; m < M; ++m) { A1[0][m] = b1; dG[0][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      mm_b<PM, 1, CT, M>(A1, s_wf, F_W1 + ho * CT, bHN, lane);
      mm_b<PM, 1, CT, M>(dG, s_wf, F_W2T + ho * CT, bDY, lane);
      Split sdg[M], sgl[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        f32x4 gl, dgl;
        gelu_both4(A1[0][m], gl, dgl);
        dG[0][m] *= dgl;                    // dA1
        pb1[ho] += dG[0][m];
        sdg[m] = split2<PM>(dG[0][m]);
        sgl[m] = split2<PM>(gl);
      }
      // dHN += W1^T dA1 (K = this hidden block)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        u32x4 bdg[M];
#pragma unroll
        for (int m = 0; m < M; ++m) bdg[m] = frag_b(sdg[m]);
        mma_tile<PM, true, M>(dHN[c], s_wf, F_W1T + c * HT + ho, bdg, lane);
      }
      // weight gradients: dW1[hid][c] += dA1 . HN^T ; dW2[c][hid] += DY . G^T  (K = tokens)
      if (ho) wave_sync();
#pragma unroll
      for (int m = 0; m < M; ++m) {
        put_img_b<PM>(imgA, 16 * m + j, kq, sdg[m]);
        put_img_b<PM>(imgB, 16 * m + j, kq, sgl[m]);
      }
      wave_sync();
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        gemm_tok_b<M, PM>(aW1[ho][c], imgA, imgHN + c * IMGW, lane);
        gemm_tok_b<M, PM>(aW2[c][ho], imgDY + c * IMGW, imgB, lane);
      }
    }
    f32x4 dH[CT][M];
    layer_norm_bwd_tl(dHN, HN, rstd2, DY, dH);
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (valid[m])
#pragma unroll
        for (int c = 0; c < CT; ++c) stx4<AT>(dh + (base + 16 * m + j) * C + 16 * c + 4 * kq, dH[c][m]);
  }
#pragma unroll
  for (int o = 0; o < HT; ++o) {
    flush_vec<1>(g_fc1_b + 16 * o, pb1[o], lane);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      flush_acc(g_fc1_w + (long)(16 * o) * C + 16 * c, C, aW1[o][c], lane);
      flush_acc(g_fc2_w + (long)(16 * c) * HID + 16 * o, HID, aW2[c][o], lane);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) flush_vec<1>(g_fc2_b + 16 * c, pb2[c], lane);
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: attention + norm1.  in: x (block input), dh (from part 1).  out: dx and the
// gradients of qkv / proj / relative-position bias.  Everything is recomputed from x.
// ---------------------------------------------------------------------------------------------
template <int C, int NH, int NW, int WPB, int MTK /*16-token column blocks per wave and work item*/>
__global__ __launch_bounds__(64 * WPB) void swin_attn_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dh, float* __restrict__ dx, const float* __restrict__ qkv_w,
    const float* __restrict__ qkv_b, const float* __restrict__ proj_w, const float* __restrict__ rel_bias,
    float* __restrict__ g_qkv_w, float* __restrict__ g_qkv_b, float* __restrict__ g_proj_w, float* __restrict__ g_proj_b,
    float* __restrict__ g_rel_bias, Geom g) {
  constexpr int CT = C / 16, HD = C / NH;
  static_assert(C % 16 == 0 && HD <= 16 && 16 % HD == 0, "head_dim must divide 16");
  constexpr int HPC = 16 / HD, SP = 4 / HPC;
  constexpr int TPW = NW >= 16 ? NW / 16 : 1, NG = MTK / TPW;
  static_assert(MTK % TPW == 0 && NG >= 1, "a work item holds whole windows");
  constexpr int VSK = 16 * MTK + 4;   // image row stride (floats)
  constexpr int IMG = 16 * VSK;
  // per wave, images [channel][token]: [0,CT) K  [CT,2CT) Q  [2CT,3CT) V  [3CT,4CT) dO  [4CT,5CT) scratch B  [5CT] scratch A
  __shared__ __attribute__((aligned(16))) float s_img[WPB][(5 * CT + 1) * IMG];
  __shared__ __attribute__((aligned(16))) int s_rid[WPB][16 * MTK];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, v = b % g.V;
  qkv_w += (long)v * 3 * C * C;  qkv_b += v * 3 * C;  proj_w += (long)v * C * C;
  rel_bias += (long)v * NH * NW * NW;
  g_qkv_w += (long)v * 3 * C * C;  g_qkv_b += v * 3 * C;  g_proj_w += (long)v * C * C;  g_proj_b += v * C;
  g_rel_bias += (long)v * NH * NW * NW;
  const float ps1 = g.path_scale ? g.path_scale[2 * b] : 1.f;

  // Weight fragments and the relative-position bias tiles live in LDS in lane-fragment order
  // (tile i, lane l -> 16 bytes at (i * 64 + l) * 4): the same for every wave of the workgroup, read with one
  // ds_read_b128 where used.  Keeping them in registers (136 per lane at C = 16) forced the accumulators of this
  // kernel through AGPR copies and scratch.
  constexpr int NRB = NH * TPW * TPW;
  constexpr int W_Q = 0, W_QT = 3 * CT * CT, W_PT = 6 * CT * CT, W_BQ = 7 * CT * CT, NWF = 7 * CT * CT + 3 * CT;
  __shared__ __attribute__((aligned(16))) float s_rb[NRB * 256];
  __shared__ __attribute__((aligned(16))) float s_wf[NWF * 256];
  {
    int i = 0;
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o) {
      const float mul = o < CT ? g.scale : 1.f;
      if ((i++) % WPB == wave) st4(s_wf + (W_BQ + o) * 256 + lane * 4, vfrag<HPC>(qkv_b, 16 * o, lane, mul));
#pragma unroll
      for (int k = 0; k < CT; ++k) {
        if ((i++) % WPB == wave) st4(s_wf + (W_Q + o * CT + k) * 256 + lane * 4, wfrag<false, HPC, 1>(qkv_w, C, 16 * o, 16 * k, lane, mul));
        // [c][o] = qkv_w[o][c] (q rows carry the scale)
        if ((i++) % WPB == wave) st4(s_wf + (W_QT + k * 3 * CT + o) * 256 + lane * 4, wfrag<true, 1, HPC>(qkv_w, C, 16 * k, 16 * o, lane, mul));
      }
    }
#pragma unroll
    for (int o = 0; o < CT; ++o)
#pragma unroll
      for (int k = 0; k < CT; ++k)   // [c'][c] = proj_w[c][c']
        if ((i++) % WPB == wave) st4(s_wf + (W_PT + o * CT + k) * 256 + lane * 4, wfrag<true, HPC, 1>(proj_w, C, 16 * o, 16 * k, lane));
    // relative-position bias tiles: rows = keys, col = query
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          if ((i++) % WPB != wave) continue;
          f32x4 f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * kq + r, qry = 16 * qt + j;
            f[r] = (key / NW) == (qry / NW) ? rel_bias[(h * NW + qry % NW) * NW + key % NW] : -1e30f;
          }
          st4(s_rb + ((h * TPW + kt) * TPW + qt) * 256 + lane * 4, f);
        }
    __syncthreads();
  }
  auto WF = [&](int i) { return ld4(s_wf + i * 256 + lane * 4); };
  auto RB = [&](int h, int kt, int qt) { return ld4(s_rb + ((h * TPW + kt) * TPW + qt) * 256 + lane * 4); };
  f32x4 dB[NH][TPW][TPW];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
      for (int qt = 0; qt < TPW; ++qt) dB[h][kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 aWq[3 * CT][CT], aWp[CT][CT], pbq[3 * CT], pbp[CT];
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    pbq[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT; ++k) aWq[o][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int o = 0; o < CT; ++o) {
    pbp[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT; ++k) aWp[o][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  float* img = s_img[wave];
  float* imgK = img;
  float* imgQ = img + CT * IMG;
  float* imgV = img + 2 * CT * IMG;
  float* imgDO = img + 3 * CT * IMG;
  float* imgB = img + 4 * CT * IMG;
  float* imgA = img + 5 * CT * IMG;
  int* rid_l = s_rid[wave];

  for (int it0 = blockIdx.x * WPB; it0 < g.items; it0 += gridDim.x * WPB) {
    Tok t[MTK];
    bool keep[MTK];
    f32x4 X[CT][MTK], DH[CT][MTK];
#pragma unroll
    for (int m = 0; m < MTK; ++m) {
      t[m] = locate<NW, 16 * MTK>(g, b, it0 + wave, 16 * m + j, C);
      keep[m] = t[m].valid;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        X[c][m] = t[m].valid ? ld4(x + t[m].off + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        DH[c][m] = t[m].valid ? ld4(dh + t[m].off + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 XN[CT][MTK];
    float rstd[MTK];
    layer_norm_tl(X, XN, rstd, keep);
    f32x4 QKV[3 * CT][MTK];
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o) {
      const f32x4 bqo = WF(W_BQ + o);
#pragma unroll
      for (int m = 0; m < MTK; ++m) QKV[o][m] = bqo;
    }
    mm_lds(QKV, s_wf + W_Q * 256, XN, lane);
    // dO = proj_w^T dH  (TL with the head permutation, like V and O)
    f32x4 DO[CT][MTK];
    f32x4 DHs[CT][MTK];    // gradient entering the attention branch: drop_path factor of this batch item times dh
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MTK; ++m) { DO[c][m] = f32x4{0.f, 0.f, 0.f, 0.f}; DHs[c][m] = DH[c][m] * ps1; pbp[c] += DHs[c][m]; }
    mm_lds(DO, s_wf + W_PT * 256, DHs, lane);

    wave_sync();   // previous iteration's image readers are done
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      put_img<HPC, MTK, VSK>(imgQ + c * IMG, QKV[c], lane);
      put_img<HPC, MTK, VSK>(imgK + c * IMG, QKV[CT + c], lane);
      put_img<HPC, MTK, VSK>(imgV + c * IMG, QKV[2 * CT + c], lane);
      put_img<HPC, MTK, VSK>(imgDO + c * IMG, DO[c], lane);
    }
    if (kq == 0)
#pragma unroll
      for (int m = 0; m < MTK; ++m) rid_l[16 * m + j] = t[m].rid;
    wave_sync();

    f32x4 O[CT][MTK], DQ[CT][MTK], DK[CT][MTK], DV[CT][MTK];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = h / HPC, s0 = (h % HPC) * SP;
        const int hrow = (16 * c + sig<HPC>(lane & 15)) * VSK + 4 * kq;   // image row of this lane's A-operand channel
        // ---- row pass: S, P, dP with keys on rows, query on the column ----
        f32x4 P[TPW][TPW], dP[TPW][TPW];
#pragma unroll
        for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) {
            f32x4 acc = RB(h, kt, qt), accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < SP; ++s) {
              acc = MFMA(QKV[CT + c][grp * TPW + kt][s0 + s], QKV[c][grp * TPW + qt][s0 + s], acc);
              accd = MFMA(QKV[2 * CT + c][grp * TPW + kt][s0 + s], DO[c][grp * TPW + qt][s0 + s], accd);
            }
            if (g.shifted) {
              const int4 rk = *reinterpret_cast<const int4*>(rid_l + 16 * (grp * TPW + kt) + 4 * kq);
              const int rq = t[grp * TPW + qt].rid;
              acc[0] += rk.x != rq ? -100.f : 0.f; acc[1] += rk.y != rq ? -100.f : 0.f;
              acc[2] += rk.z != rq ? -100.f : 0.f; acc[3] += rk.w != rq ? -100.f : 0.f;
            }
            P[kt][qt] = acc; dP[kt][qt] = accd;
          }
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          float mx = -3.0e38f;
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(P[kt][qt][0], P[kt][qt][1]), fmaxf(P[kt][qt][2], P[kt][qt][3])));
          mx = chan_max(mx);
          const float mxl = -mx * kLog2e;
          f32x4 l4 = splat(0.f);
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 ea = fma4(P[kt][qt], splat(kLog2e), splat(mxl));
#pragma unroll
            for (int r = 0; r < 4; ++r) P[kt][qt][r] = __builtin_amdgcn_exp2f(ea[r]);
            l4 += P[kt][qt];
          }
          const float inv = __builtin_amdgcn_rcpf(chan_sum(hsum(l4)));   // v_rcp_f32 (1 ulp), not the 10-instruction IEEE division
          f32x4 D4 = splat(0.f);
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            P[kt][qt] *= inv;
            D4 = fma4(P[kt][qt], dP[kt][qt], D4);
          }
          const float D = chan_sum(hsum(D4));
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            dP[kt][qt] = P[kt][qt] * (dP[kt][qt] - splat(D));   // dP now holds dS
            dB[h][kt][qt] += dP[kt][qt];
          }
        }
        // O (for dproj_w) and dQ : A operands are channel-major images of V and K
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          f32x4 acco = {0.f, 0.f, 0.f, 0.f}, accq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 vf = ld4(imgV + hrow + 16 * (grp * TPW + kt)), kf = ld4(imgK + hrow + 16 * (grp * TPW + kt));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acco = MFMA(vf[s], P[kt][qt][s], acco);
              accq = MFMA(kf[s], dP[kt][qt][s], accq);
            }
          }
#pragma unroll
          for (int r = 0; r < SP; ++r) { O[c][grp * TPW + qt][s0 + r] = acco[s0 + r]; DQ[c][grp * TPW + qt][s0 + r] = accq[s0 + r]; }
        }
        // ---- column pass: P^T and dS^T (queries on rows, key on the column) are the row-pass tiles TRANSPOSED across
        // lanes through two wave-private scratch images — 8 four-byte LDS writes and 2 sixteen-byte reads per tile
        // instead of recomputing S^T (4 MFMAs), its mask, 4 exponentials and the softmax algebra; bit-identical to P
        f32x4 PT[TPW][TPW], dST[TPW][TPW];   // [qt][kt]
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            wave_sync();   // the previous tile's readers are done
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              imgA[(4 * kq + r) * VSK + j] = P[kt][qt][r];      // [key][query]
              imgB[(4 * kq + r) * VSK + j] = dP[kt][qt][r];
            }
            wave_sync();
            PT[qt][kt] = ld4(imgA + j * VSK + 4 * kq);           // lane (key j, query group kq)
            dST[qt][kt] = ld4(imgB + j * VSK + 4 * kq);
          }
#pragma unroll
        for (int kt = 0; kt < TPW; ++kt) {
          f32x4 acck = {0.f, 0.f, 0.f, 0.f}, accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) {
            const f32x4 qf = ld4(imgQ + hrow + 16 * (grp * TPW + qt)), df = ld4(imgDO + hrow + 16 * (grp * TPW + qt));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acck = MFMA(qf[s], dST[qt][kt][s], acck);
              accv = MFMA(df[s], PT[qt][kt][s], accv);
            }
          }
#pragma unroll
          for (int r = 0; r < SP; ++r) { DK[c][grp * TPW + kt][s0 + r] = acck[s0 + r]; DV[c][grp * TPW + kt][s0 + r] = accv[s0 + r]; }
        }
      }
    }
    // ---- proj / qkv weight gradients and dxn ----
    // dproj_w[c][c'] += dH[c] O[c'] ; dqkv_w[o][c] += dQKV[o] XN[c] ; K = tokens
    f32x4 dXN[CT][MTK];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MTK; ++m) dXN[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the q gradient refers to the pre-scaled q: fold the scale into wqt (done) and into dWq / dbq here
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MTK; ++m) {
        pbq[c] += DQ[c][m] * g.scale; pbq[CT + c] += DK[c][m]; pbq[2 * CT + c] += DV[c][m];
      }
    // dxn = Wqkv^T dQKV
#pragma unroll
    for (int oc = 0; oc < CT; ++oc)
#pragma unroll
      for (int k = 0; k < CT; ++k) {
        const f32x4 wa = WF(W_QT + oc * 3 * CT + k), wb = WF(W_QT + oc * 3 * CT + CT + k), wc = WF(W_QT + oc * 3 * CT + 2 * CT + k);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < MTK; ++m) {
            dXN[oc][m] = MFMA(wa[s], DQ[k][m][s], dXN[oc][m]);
            dXN[oc][m] = MFMA(wb[s], DK[k][m][s], dXN[oc][m]);
            dXN[oc][m] = MFMA(wc[s], DV[k][m][s], dXN[oc][m]);
          }
      }
    f32x4 DX[CT][MTK];
    layer_norm_bwd_tl(dXN, XN, rstd, DH, DX);
#pragma unroll
    for (int m = 0; m < MTK; ++m)
      if (t[m].valid)
#pragma unroll
        for (int c = 0; c < CT; ++c) st4(dx + t[m].off + 16 * c + 4 * kq, DX[c][m]);

    // token-reduction GEMMs through the scratch images
#pragma unroll
    for (int ca = 0; ca < CT; ++ca) {        // dproj_w rows (c) from dH, cols (c') from O
      wave_sync();
      put_img<1, MTK, VSK>(imgA, DHs[ca], lane);
      if (ca == 0)
#pragma unroll
        for (int cb = 0; cb < CT; ++cb) put_img<HPC, MTK, VSK>(imgB + cb * IMG, O[cb], lane);
      wave_sync();
#pragma unroll
      for (int cb = 0; cb < CT; ++cb) gemm_tok<MTK, VSK>(aWp[ca][cb], imgA, imgB + cb * IMG, lane);
    }
    wave_sync();
#pragma unroll
    for (int cb = 0; cb < CT; ++cb) put_img<1, MTK, VSK>(imgB + cb * IMG, XN[cb], lane);
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o) {
      wave_sync();
      if (o < CT) {
        f32x4 tmp[MTK];
#pragma unroll
        for (int m = 0; m < MTK; ++m) tmp[m] = DQ[o][m] * g.scale;
        put_img<HPC, MTK, VSK>(imgA, tmp, lane);
      } else if (o < 2 * CT) put_img<HPC, MTK, VSK>(imgA, DK[o - CT], lane);
      else put_img<HPC, MTK, VSK>(imgA, DV[o - 2 * CT], lane);
      wave_sync();
#pragma unroll
      for (int cb = 0; cb < CT; ++cb) gemm_tok<MTK, VSK>(aWq[o][cb], imgA, imgB + cb * IMG, lane);
    }
  }

  // ---- flush ----
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    flush_vec<HPC>(g_qkv_b + 16 * o, pbq[o], lane);
#pragma unroll
    for (int k = 0; k < CT; ++k) flush_acc(g_qkv_w + (long)(16 * o) * C + 16 * k, C, aWq[o][k], lane);
  }
#pragma unroll
  for (int o = 0; o < CT; ++o) {
    flush_vec<1>(g_proj_b + 16 * o, pbp[o], lane);
#pragma unroll
    for (int k = 0; k < CT; ++k) flush_acc(g_proj_w + (long)(16 * o) * C + 16 * k, C, aWp[o][k], lane);
  }
  // d rel-bias: dB[h][kt][qt][r] is d/d bias[h][query = 16qt + j][key = 16kt + 4kq + r] (summed over groups)
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
      for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * kq + r, qry = 16 * qt + j;
          if ((key / NW) == (qry / NW)) atomicAdd(g_rel_bias + (h * NW + qry % NW) * NW + key % NW, dB[h][kt][qt][r]);
        }
}


// ---------------------------------------------------------------------------------------------
// backward, part 2 on the split-bf16 MFMA path: same algorithm, arguments and results (to ~1e-5 relative, gradients are
// compared at 1e-3) as swin_attn_bwd_kernel.  Only the K = 8 score products S = K Q^T and dP = V dO^T stay on the f32 MFMA
// (operands already in registers); the linear layers, O / dQ / dK / dV and the five token-reduction GEMMs run as piece
// products.  Q, K, V, dO (and later dH, O, XN, dQ, dK, dV) live in LDS as token-major bf16 piece images; the
// [channel][token] and the transposed P^T / dS^T operands come out of them with ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------------
// D tile (rows 4kq+r, col j) -> dst[(row0 + HR(4kq + r)) * ld + col0 + HC(j)]: the piece images keep TL positions, so the head
// permutation of q/k/v/O channels is undone here
template <int HR, int HC>
__device__ __forceinline__ void flush_acc_perm(float* dst, int ld, const f32x4& acc, int lane, float mul = 1.f) {
  const int j = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(dst + (long)sig<HR>(r0 + r) * ld + sig<HC>(j), mul * acc[r]);
}
__device__ __forceinline__ u32x4 dup2(unsigned long long a) { return u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)a, (unsigned)(a >> 32)}; }

template <int C, int NH, int NW, int WPB, int MTK, bool WDUP, int PM, typename AT>
__global__ __launch_bounds__(64 * WPB) void swin_attn_bwd_b_kernel(
    const AT* __restrict__ x, const AT* __restrict__ dh, AT* __restrict__ dx, const float* __restrict__ qkv_w,
    const float* __restrict__ qkv_b, const float* __restrict__ proj_w, const float* __restrict__ rel_bias,
    float* __restrict__ g_qkv_w, float* __restrict__ g_qkv_b, float* __restrict__ g_proj_w, float* __restrict__ g_proj_b,
    float* __restrict__ g_rel_bias, Geom g) {
  constexpr int CT = C / 16, HD = C / NH;
  static_assert(C % 16 == 0 && HD <= 16 && 16 % HD == 0, "head_dim must divide 16");
  constexpr int HPC = 16 / HD, SP = 4 / HPC;
  constexpr int TPW = NW >= 16 ? NW / 16 : 1, NG = MTK / TPW;
  static_assert(MTK % TPW == 0 && NG >= 1, "a work item holds whole windows");
  constexpr int IMGW = 16 * MTK * kImgRow;      // words per piece image of one 16-position chunk
  constexpr int TILEW = 16 * kImgRow;           // words per 16 x 16 scratch tile
  // per wave: [0,CT) K  [CT,2CT) Q  [2CT,3CT) V  [3CT,4CT) dO  [4CT,5CT) XN  [5CT,6CT) dH, then two scratch tiles
  __shared__ __attribute__((aligned(16))) unsigned s_img[WPB][6 * CT * IMGW + 2 * TILEW];
  __shared__ __attribute__((aligned(16))) int s_rid[WPB][16 * MTK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  // grid: (workgroups, V).  A workgroup stays with ONE variable's weights (its gradient accumulators live in registers across
  // the loop) and walks the window groups of all N batch items of that variable — with one grid row per batch item the
  // 256 CUs took floor(256 / (N V)) N V = 240 workgroups at the headline shape.
  const int v = blockIdx.y;
  qkv_w += (long)v * 3 * C * C;  qkv_b += v * 3 * C;  proj_w += (long)v * C * C;
  rel_bias += (long)v * NH * NW * NW;
  g_qkv_w += (long)v * 3 * C * C;  g_qkv_b += v * 3 * C;  g_proj_w += (long)v * C * C;  g_proj_b += v * C;
  g_rel_bias += (long)v * NH * NW * NW;

  constexpr int NRB = NH * TPW * TPW;
  constexpr int W_Q = 0, W_QT = 3 * CT * CT, W_PT = 6 * CT * CT, NWT = 7 * CT * CT;
  __shared__ __attribute__((aligned(16))) float s_rb[NRB * 256];
  __shared__ __attribute__((aligned(16))) unsigned s_wf[NWT * (WDUP ? 2 : 1) * 256];
  __shared__ __attribute__((aligned(16))) float s_bq[3 * CT * 256];
  {
    int i = 0;
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o) {
      const float mul = o < CT ? g.scale : 1.f;
      if ((i++) % WPB == wave) st4(s_bq + o * 256 + lane * 4, vfrag<HPC>(qkv_b, 16 * o, lane, mul));
#pragma unroll
      for (int k = 0; k < CT; ++k) {
        if ((i++) % WPB == wave) put_wfrag_x<PM, WDUP>(s_wf, W_Q + o * CT + k, wfrag<false, HPC, 1>(qkv_w, C, 16 * o, 16 * k, lane, mul), lane);
        // [c][o] = qkv_w[o][c] (q rows carry the scale)
        if ((i++) % WPB == wave) put_wfrag_x<PM, WDUP>(s_wf, W_QT + k * 3 * CT + o, wfrag<true, 1, HPC>(qkv_w, C, 16 * k, 16 * o, lane, mul), lane);
      }
    }
#pragma unroll
    for (int o = 0; o < CT; ++o)
#pragma unroll
      for (int k = 0; k < CT; ++k)   // [c'][c] = proj_w[c][c']
        if ((i++) % WPB == wave) put_wfrag_x<PM, WDUP>(s_wf, W_PT + o * CT + k, wfrag<true, HPC, 1>(proj_w, C, 16 * o, 16 * k, lane), lane);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          if ((i++) % WPB != wave) continue;
          f32x4 f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * kq + r, qry = 16 * qt + j;
            f[r] = (key / NW) == (qry / NW) ? rel_bias[(h * NW + qry % NW) * NW + key % NW] : -1e30f;
          }
          st4(s_rb + ((h * TPW + kt) * TPW + qt) * 256 + lane * 4, f);
        }
    __syncthreads();
  }
  auto RB = [&](int h, int kt, int qt) { return ld4(s_rb + ((h * TPW + kt) * TPW + qt) * 256 + lane * 4); };
  f32x4 dB[NH][TPW][TPW];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
      for (int qt = 0; qt < TPW; ++qt) dB[h][kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 aWq[3 * CT][CT], aWp[CT][CT], pbq[3 * CT], pbp[CT];
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    pbq[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT; ++k) aWq[o][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int o = 0; o < CT; ++o) {
    pbp[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT; ++k) aWp[o][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  unsigned* img = s_img[wave];
  unsigned* imgK = img;
  unsigned* imgQ = img + CT * IMGW;
  unsigned* imgV = img + 2 * CT * IMGW;
  unsigned* imgDO = img + 3 * CT * IMGW;
  unsigned* imgXN = img + 4 * CT * IMGW;
  unsigned* imgDH = img + 5 * CT * IMGW;
  unsigned* tileP = img + 6 * CT * IMGW;
  unsigned* tileS = tileP + TILEW;
  int* rid_l = s_rid[wave];

  const int gpb = (g.items + WPB - 1) / WPB;            // window groups of WPB work items per batch item
  for (int grp = blockIdx.x; grp < g.N * gpb; grp += gridDim.x) {
    const int n = grp / gpb, it0 = (grp - n * gpb) * WPB, b = n * g.V + v;
    const float ps1 = g.path_scale ? g.path_scale[2 * b] : 1.f;
    Tok t[MTK];
    bool keep[MTK];
    f32x4 X[CT][MTK], DH[CT][MTK];
#pragma unroll
    for (int m = 0; m < MTK; ++m) {
      t[m] = locate<NW, 16 * MTK>(g, b, it0 + wave, 16 * m + j, C);
      keep[m] = t[m].valid;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        X[c][m] = t[m].valid ? ldx4<AT>(x + t[m].off + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        DH[c][m] = t[m].valid ? ldx4<AT>(dh + t[m].off + 16 * c + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 XN[CT][MTK];
    float rstd[MTK];
    layer_norm_tl(X, XN, rstd, keep);
    f32x4 QKV[3 * CT][MTK], DO[CT][MTK];
    wave_sync();   // previous iteration's image readers are done
    {
      u32x4 bXN[CT][MTK], bDH[CT][MTK];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int m = 0; m < MTK; ++m) {
          const Split sx = split2<PM>(XN[c][m]), sd = split2<PM>(DH[c][m] * ps1);   // ps1: drop_path factor of the attention branch
          bXN[c][m] = frag_b(sx);
          bDH[c][m] = frag_b(sd);
          put_img_b<PM>(imgXN + c * IMGW, 16 * m + j, kq, sx);     // kept for the weight-gradient GEMMs at the end
          put_img_b<PM>(imgDH + c * IMGW, 16 * m + j, kq, sd);
          DO[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
          pbp[c] += DH[c][m] * ps1;
        }
#pragma unroll
      for (int o = 0; o < 3 * CT; ++o) {
        const f32x4 bqo = ld4(s_bq + o * 256 + lane * 4);
#pragma unroll
        for (int m = 0; m < MTK; ++m) QKV[o][m] = bqo;
      }
      mm_x<PM, WDUP, 3 * CT, CT, MTK>(QKV, s_wf, W_Q, bXN, lane);
      mm_x<PM, WDUP, CT, CT, MTK>(DO, s_wf, W_PT, bDH, lane);       // dO = proj_w^T dH
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MTK; ++m) {
        put_img_b<PM>(imgQ + c * IMGW, 16 * m + j, kq, split2<PM>(QKV[c][m]));
        put_img_b<PM>(imgK + c * IMGW, 16 * m + j, kq, split2<PM>(QKV[CT + c][m]));
        put_img_b<PM>(imgV + c * IMGW, 16 * m + j, kq, split2<PM>(QKV[2 * CT + c][m]));
        put_img_b<PM>(imgDO + c * IMGW, 16 * m + j, kq, split2<PM>(DO[c][m]));
      }
    if (kq == 0)
#pragma unroll
      for (int m = 0; m < MTK; ++m) rid_l[16 * m + j] = t[m].rid;
    wave_sync();

    f32x4 O[CT][MTK], DQ[CT][MTK], DK[CT][MTK], DV[CT][MTK];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int c = h / HPC, s0 = (h % HPC) * SP;
        // ---- row pass: S, P, dP with keys on rows, query on the column (f32 MFMA, K = head_dim) ----
        f32x4 P[TPW][TPW], dP[TPW][TPW];
#pragma unroll
        for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) { P[kt][qt] = RB(h, kt, qt); dP[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < SP; ++s)
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt) {
              P[kt][qt] = MFMA(QKV[CT + c][grp * TPW + kt][s0 + s], QKV[c][grp * TPW + qt][s0 + s], P[kt][qt]);
              dP[kt][qt] = MFMA(QKV[2 * CT + c][grp * TPW + kt][s0 + s], DO[c][grp * TPW + qt][s0 + s], dP[kt][qt]);
            }
        if (g.shifted) {
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const int4 rk = *reinterpret_cast<const int4*>(rid_l + 16 * (grp * TPW + kt) + 4 * kq);
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt) {
              const int rq = t[grp * TPW + qt].rid;
              P[kt][qt][0] += rk.x != rq ? -100.f : 0.f; P[kt][qt][1] += rk.y != rq ? -100.f : 0.f;
              P[kt][qt][2] += rk.z != rq ? -100.f : 0.f; P[kt][qt][3] += rk.w != rq ? -100.f : 0.f;
            }
          }
        }
#pragma unroll
        for (int qt = 0; qt < TPW; ++qt) {
          float mx = -3.0e38f;
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(P[kt][qt][0], P[kt][qt][1]), fmaxf(P[kt][qt][2], P[kt][qt][3])));
          mx = chan_max(mx);
          const float mxl = -mx * kLog2e;
          f32x4 l4 = splat(0.f);
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const f32x4 ea = fma4(P[kt][qt], splat(kLog2e), splat(mxl));
#pragma unroll
            for (int r = 0; r < 4; ++r) P[kt][qt][r] = __builtin_amdgcn_exp2f(ea[r]);
            l4 += P[kt][qt];
          }
          const float inv = __builtin_amdgcn_rcpf(chan_sum(hsum(l4)));
          f32x4 D4 = splat(0.f);
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            P[kt][qt] *= inv;
            D4 = fma4(P[kt][qt], dP[kt][qt], D4);
          }
          const float D = chan_sum(hsum(D4));
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            dP[kt][qt] = P[kt][qt] * (dP[kt][qt] - splat(D));   // dS
            dB[h][kt][qt] += dP[kt][qt];
          }
        }
        Split sP[TPW][TPW], sS[TPW][TPW];
#pragma unroll
        for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) { sP[kt][qt] = split2<PM>(P[kt][qt]); sS[kt][qt] = split2<PM>(dP[kt][qt]); }
        // O[chan][query] = sum_keys V^T P ; dQ[chan][query] = sum_keys K^T dS : A from the V / K piece images (transposed read)
        {
          f32x4 acco[TPW], accq[TPW];
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) { acco[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; accq[qt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) {
            const int mk = grp * TPW + kt;
            const u32x4 vh = dup2(tr_read(imgV + c * IMGW, mk, 0, lane));
            const u32x4 kh = dup2(tr_read(imgK + c * IMGW, mk, 0, lane));
#pragma unroll
            for (int qt = 0; qt < TPW; ++qt) {
              acco[qt] = mfma_s<PM>(vh, frag_b(sP[kt][qt]), acco[qt]);
              accq[qt] = mfma_s<PM>(kh, frag_b(sS[kt][qt]), accq[qt]);
            }
            if (PM != PM_BF16X1) {
              const u32x4 vl = dup2(tr_read(imgV + c * IMGW, mk, 1, lane)), kl = dup2(tr_read(imgK + c * IMGW, mk, 1, lane));
#pragma unroll
              for (int qt = 0; qt < TPW; ++qt) {
                acco[qt] = mfma_s<PM>(vl, frag_b(sP[kt][qt]), acco[qt]);
                accq[qt] = mfma_s<PM>(kl, frag_b(sS[kt][qt]), accq[qt]);
              }
            }
          }
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
            for (int r = 0; r < SP; ++r) { O[c][grp * TPW + qt][s0 + r] = acco[qt][s0 + r]; DQ[c][grp * TPW + qt][s0 + r] = accq[qt][s0 + r]; }
        }
        // ---- column pass: dK[chan][key] = sum_queries Q^T dS^T ; dV[chan][key] = sum_queries dO^T P^T.  The transposed B
        // operands are the row-pass tiles' pieces written query-major to a scratch tile and read back with the transposing
        // LDS read: two 8-byte writes and two reads per tile and tensor
        {
          f32x4 acck[TPW], accv[TPW];
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt) { acck[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; accv[kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int qt = 0; qt < TPW; ++qt) {
            const int mq = grp * TPW + qt;
            const u32x4 qh = dup2(tr_read(imgQ + c * IMGW, mq, 0, lane)), dh_ = dup2(tr_read(imgDO + c * IMGW, mq, 0, lane));
            u32x4 ql = {0u, 0u, 0u, 0u}, dl_ = {0u, 0u, 0u, 0u};
            if (PM != PM_BF16X1) { ql = dup2(tr_read(imgQ + c * IMGW, mq, 1, lane)); dl_ = dup2(tr_read(imgDO + c * IMGW, mq, 1, lane)); }
#pragma unroll
            for (int kt = 0; kt < TPW; ++kt) {
              wave_sync();   // the previous tile's readers are done
              put_img_b<PM>(tileP, j, kq, sP[kt][qt]);      // rows = queries, positions = keys
              put_img_b<PM>(tileS, j, kq, sS[kt][qt]);
              wave_sync();
              const unsigned long long ph = tr_read(tileP, 0, 0, lane), sh = tr_read(tileS, 0, 0, lane);
              unsigned long long pl = 0ull, sl = 0ull;
              if (PM != PM_BF16X1) { pl = tr_read(tileP, 0, 1, lane); sl = tr_read(tileS, 0, 1, lane); }
              const u32x4 PT = {(unsigned)ph, (unsigned)(ph >> 32), (unsigned)pl, (unsigned)(pl >> 32)};
              const u32x4 ST = {(unsigned)sh, (unsigned)(sh >> 32), (unsigned)sl, (unsigned)(sl >> 32)};
              acck[kt] = mfma_s<PM>(qh, ST, acck[kt]);
              accv[kt] = mfma_s<PM>(dh_, PT, accv[kt]);
              if (PM != PM_BF16X1) {
                acck[kt] = mfma_s<PM>(ql, ST, acck[kt]);
                accv[kt] = mfma_s<PM>(dl_, PT, accv[kt]);
              }
            }
          }
#pragma unroll
          for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
            for (int r = 0; r < SP; ++r) { DK[c][grp * TPW + kt][s0 + r] = acck[kt][s0 + r]; DV[c][grp * TPW + kt][s0 + r] = accv[kt][s0 + r]; }
        }
      }
    }
    // ---- dxn = Wqkv^T dQKV, norm1 backward ----
    Split sDQ[CT][MTK], sDK[CT][MTK], sDV[CT][MTK], sO[CT][MTK];
    f32x4 dXN[CT][MTK];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MTK; ++m) {
        dXN[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the q gradient refers to the pre-scaled q: the scale is folded into wqt (above) and into dWq / dbq at the flush
        pbq[c] += DQ[c][m]; pbq[CT + c] += DK[c][m]; pbq[2 * CT + c] += DV[c][m];
        sDQ[c][m] = split2<PM>(DQ[c][m]); sDK[c][m] = split2<PM>(DK[c][m]); sDV[c][m] = split2<PM>(DV[c][m]);
        sO[c][m] = split2<PM>(O[c][m]);
      }
#pragma unroll
    for (int oc = 0; oc < CT; ++oc)
#pragma unroll
      for (int k = 0; k < CT; ++k) {
        u32x4 bq_[MTK], bk_[MTK], bv_[MTK];
#pragma unroll
        for (int m = 0; m < MTK; ++m) { bq_[m] = frag_b(sDQ[k][m]); bk_[m] = frag_b(sDK[k][m]); bv_[m] = frag_b(sDV[k][m]); }
        mma_tile<PM, WDUP, MTK>(dXN[oc], s_wf, W_QT + oc * 3 * CT + k, bq_, lane);
        mma_tile<PM, WDUP, MTK>(dXN[oc], s_wf, W_QT + oc * 3 * CT + CT + k, bk_, lane);
        mma_tile<PM, WDUP, MTK>(dXN[oc], s_wf, W_QT + oc * 3 * CT + 2 * CT + k, bv_, lane);
      }
    f32x4 DX[CT][MTK];
    layer_norm_bwd_tl(dXN, XN, rstd, DH, DX);
#pragma unroll
    for (int m = 0; m < MTK; ++m)
      if (t[m].valid)
#pragma unroll
        for (int c = 0; c < CT; ++c) stx4<AT>(dx + t[m].off + 16 * c + 4 * kq, DX[c][m]);

    // ---- token-reduction GEMMs (weight gradients) from piece images that reuse the Q / K / V / dO slots ----
    // dproj_w[c][c'] += dH[c] O[c'] ; dqkv_w[o][c] += dQKV[o] XN[c] ; K = tokens
    unsigned* imgO = imgK;    // CT chunks
    unsigned* imgG = imgDO;   // one dQKV chunk at a time
    wave_sync();
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int m = 0; m < MTK; ++m) put_img_b<PM>(imgO + c * IMGW, 16 * m + j, kq, sO[c][m]);
    wave_sync();
#pragma unroll
    for (int ca = 0; ca < CT; ++ca)
#pragma unroll
      for (int cb = 0; cb < CT; ++cb) gemm_tok_b<MTK, PM>(aWp[ca][cb], imgDH + ca * IMGW, imgO + cb * IMGW, lane);
#pragma unroll
    for (int o = 0; o < 3 * CT; ++o) {
      if (o) wave_sync();
#pragma unroll
      for (int m = 0; m < MTK; ++m)
        put_img_b<PM>(imgG, 16 * m + j, kq, o < CT ? sDQ[o][m] : o < 2 * CT ? sDK[o - CT][m] : sDV[o - 2 * CT][m]);
      wave_sync();
#pragma unroll
      for (int cb = 0; cb < CT; ++cb) gemm_tok_b<MTK, PM>(aWq[o][cb], imgG, imgXN + cb * IMGW, lane);
    }
  }

  // ---- flush (image positions -> channels: the q/k/v/O chunks carry the head permutation) ----
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    const float mul = o < CT ? g.scale : 1.f;
    flush_vec<HPC>(g_qkv_b + 16 * o, pbq[o] * mul, lane);
#pragma unroll
    for (int k = 0; k < CT; ++k) flush_acc_perm<HPC, 1>(g_qkv_w + (long)(16 * o) * C + 16 * k, C, aWq[o][k], lane, mul);
  }
#pragma unroll
  for (int o = 0; o < CT; ++o) {
    flush_vec<1>(g_proj_b + 16 * o, pbp[o], lane);
#pragma unroll
    for (int k = 0; k < CT; ++k) flush_acc_perm<1, HPC>(g_proj_w + (long)(16 * o) * C + 16 * k, C, aWp[o][k], lane);
  }
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int kt = 0; kt < TPW; ++kt)
#pragma unroll
      for (int qt = 0; qt < TPW; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * kq + r, qry = 16 * qt + j;
          if ((key / NW) == (qry / NW)) atomicAdd(g_rel_bias + (h * NW + qry % NW) * NW + key % NW, dB[h][kt][qt][r]);
        }
}

// ---------------------------------------------------------------------------------------------
// relative-position bias: table (V, L, NH) gathered through index (NW, NW) -> dense (V, NH, NW, NW)
// (Swin_3D.py:158-162) and its scatter-add backward.
// ---------------------------------------------------------------------------------------------
__global__ void relpos_gather_kernel(const float* __restrict__ table, const long long* __restrict__ index,
                                     float* __restrict__ dense, int V, int L, int NH, int NW) {
  const int total = V * NH * NW * NW;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int ij = e % (NW * NW), h = (e / (NW * NW)) % NH, v = e / (NW * NW * NH);
    dense[e] = table[((long)v * L + index[ij]) * NH + h];
  }
}
__global__ void relpos_scatter_kernel(const float* __restrict__ ddense, const long long* __restrict__ index,
                                      float* __restrict__ dtable, int V, int L, int NH, int NW) {
  const int total = V * NH * NW * NW;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int ij = e % (NW * NW), h = (e / (NW * NW)) % NH, v = e / (NW * NW * NH);
    atomicAdd(dtable + ((long)v * L + index[ij]) * NH + h, ddense[e]);
  }
}

int make_geom(const idee_swin_geom* s, Geom& g, int& NW) {
  IDEE_CHECK_ARG(s->N > 0 && s->V > 0 && s->D > 0 && s->H > 0 && s->W > 0, "swin: empty tensor");
  IDEE_CHECK_ARG(s->wd > 0 && s->wh > 0 && s->ww > 0, "swin: bad window");
  IDEE_CHECK_ARG(s->sd >= 0 && s->sd < s->wd && s->sh >= 0 && s->sh < s->wh && s->sw >= 0 && s->sw < s->ww,
                 "swin: shift must lie in [0, window)");
  IDEE_CHECK_ARG(s->hidden == 4 * s->C, "swin: only mlp_ratio 4 is built (hidden %d, C %d)", s->hidden, s->C);
  NW = s->wd * s->wh * s->ww;
  g.N = s->N; g.V = s->V; g.D = s->D; g.H = s->H; g.W = s->W;
  g.wd = s->wd; g.wh = s->wh; g.ww = s->ww; g.sd = s->sd; g.sh = s->sh; g.sw = s->sw;
  g.Dp = cdiv(s->D, s->wd) * s->wd; g.Hp = cdiv(s->H, s->wh) * s->wh; g.Wp = cdiv(s->W, s->ww) * s->ww;
  g.nWh = g.Hp / s->wh; g.nWw = g.Wp / s->ww;
  g.nWin = (g.Dp / s->wd) * g.nWh * g.nWw;
  g.shifted = (s->sd | s->sh | s->sw) != 0;
  g.scale = s->scale;
  g.path_scale = nullptr;
  if (NW != 8 && NW != 16 && NW != 32 && NW != 64)
    return idee::fail(IDEE_ERR_UNSUPPORTED, "swin: window volume %d not in {8,16,32,64}", NW);
  g.items = cdiv(g.nWin, 64 / NW);
  g.inv_nWhW = 1.0f / (float)(g.nWh * g.nWw); g.inv_nWw = 1.0f / (float)g.nWw;
  g.inv_whww = 1.0f / (float)(g.wh * g.ww);   g.inv_ww = 1.0f / (float)g.ww;
  if ((long)g.nWin >= (1L << 22)) return idee::fail(IDEE_ERR_UNSUPPORTED, "swin: %d windows per batch item exceeds 2^22", g.nWin);
  // the token addressing multiplies 24-bit factors (locate()): rows (N V D H) and the frame width below 2^24, tokens below 2^32
  if ((long)s->N * s->V * s->D * s->H >= (1L << 24) || s->W >= (1 << 24) || g.Dp >= (1 << 24) || g.Hp >= (1 << 24) ||
      (long)s->N * s->V * s->D * s->H * s->W >= (1L << 32))
    return idee::fail(IDEE_ERR_UNSUPPORTED, "swin: tensor of %d x %d x %d x %d x %d tokens is beyond the 24-bit index arithmetic",
                      s->N, s->V, s->D, s->H, s->W);
  return IDEE_OK;
}

// Default: the exact-f32 MFMA kernels (the reference's arithmetic: fp32 nn.Linear, Swin_3D.py:145-178, 264-265).
// IDEE_SWIN_MATH=split opts into the two-piece split-precision GEMM kernels (f16 pieces forward, bf16 pieces backward:
// operand rounding 2^-22 / 2^-18) — a separately labelled bench line (bench.py --math split), never the headline.
bool split_math() { return env_is("IDEE_SWIN_MATH", "split"); }

// One round of resident workgroups: gx * B <= CUs * (workgroups per CU), so that no CU is left with a
// lone straggler round (work items are uniform; every workgroup loops over its share).
dim3 swin_grid(int items_per_wg_loop, int B, int wg_per_cu) {
  int gx = (int)(((long)kCUs * wg_per_cu) / B);
  if (gx < 1) gx = 1;
  if (gx > items_per_wg_loop) gx = items_per_wg_loop;
  return dim3(gx, B, 1);
}

}  // namespace

#define SWIN_DISPATCH(KERNEL_CALL)                                                              \
  if (C == 16 && NH == 2 && NW == 32) { KERNEL_CALL(16, 2, 32); }                               \
  else if (C == 16 && NH == 2 && NW == 8) { KERNEL_CALL(16, 2, 8); }                            \
  else if (C == 16 && NH == 2 && NW == 16) { KERNEL_CALL(16, 2, 16); }                          \
  else if (C == 32 && NH == 2 && NW == 32) { KERNEL_CALL(32, 2, 32); }                          \
  else if (C == 32 && NH == 2 && NW == 8) { KERNEL_CALL(32, 2, 8); }                            \
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "swin: (C=%d, heads=%d, window volume=%d) is not built", C, NH, NW);

// which split kernels run: PM_BF16X1 + bf16 tokens for the bf16 activation path, the two-piece modes (f16 pieces forward,
// bf16 pieces backward) for IDEE_SWIN_MATH=split on f32 tokens
extern "C" int idee_swin_block_fwd(const void* x_, void* y_, void* h_save_, const idee_swin_params* p,
                                   const idee_swin_geom* s, idee_stream_t stream) {
  IDEE_CHECK_ARG(x_ && y_ && p && s, "swin_block_fwd: null argument");
  Geom g; int NW;
  if (int rc = make_geom(s, g, NW)) return rc;
  g.path_scale = p->path_scale;
  const int C = s->C, NH = s->heads, B = s->N * s->V;
  const bool bf = s->act_dtype == IDEE_DTYPE_BF16;
  const bool split = split_math() || bf;
  const dim3 grid = swin_grid(cdiv(g.items, 4), B, C == 16 ? (split ? 3 : 2) : 1);
  const float* x = (const float*)x_; float* y = (float*)y_; float* h_save = (float*)h_save_;
#define LFWD(KERN, XT)                                                                                                \
  hipLaunchKernelGGL(KERN, grid, dim3(256), 0, (hipStream_t)stream, (const XT*)x_, (XT*)y_, (XT*)h_save_, p->qkv_w, p->qkv_b, \
                     p->proj_w, p->proj_b, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, p->rel_bias, g)
#define CALL(C_, NH_, NW_)                                                                                   \
  if (bf) {                                                                                                  \
    if (h_save) LFWD((swin_fwd_b_kernel<C_, NH_, NW_, true, PM_BF16X1, bf16_t>), bf16_t);                    \
    else LFWD((swin_fwd_b_kernel<C_, NH_, NW_, false, PM_BF16X1, bf16_t>), bf16_t);                          \
  } else if (split) {                                                                                        \
    if (h_save) LFWD((swin_fwd_b_kernel<C_, NH_, NW_, true, PM_F16X2, float>), float);                       \
    else LFWD((swin_fwd_b_kernel<C_, NH_, NW_, false, PM_F16X2, float>), float);                             \
  } else if (h_save)                                                                                         \
    hipLaunchKernelGGL((swin_fwd_mfma_kernel<C_, NH_, NW_, true>), grid, dim3(256), 0, (hipStream_t)stream, x, y, h_save, \
                       p->qkv_w, p->qkv_b, p->proj_w, p->proj_b, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, p->rel_bias, g); \
  else                                                                                                       \
    hipLaunchKernelGGL((swin_fwd_mfma_kernel<C_, NH_, NW_, false>), grid, dim3(256), 0, (hipStream_t)stream, x, y, h_save, \
                       p->qkv_w, p->qkv_b, p->proj_w, p->proj_b, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, p->rel_bias, g)
  SWIN_DISPATCH(CALL)
#undef CALL
#undef LFWD
  IDEE_LAUNCH_CHECK("swin_block_fwd");
  return IDEE_OK;
}

extern "C" int idee_swin_block_bwd(const void* x_, const void* h_saved_, const void* dy_, void* dx_, void* dh_workspace_,
                                   const idee_swin_params* p, const idee_swin_grads* d, const idee_swin_geom* s,
                                   idee_stream_t stream) {
  IDEE_CHECK_ARG(x_ && h_saved_ && dy_ && dx_ && dh_workspace_ && p && d && s, "swin_block_bwd: null argument");
  Geom g; int NW;
  if (int rc = make_geom(s, g, NW)) return rc;
  g.path_scale = p->path_scale;
  const int C = s->C, NH = s->heads, B = s->N * s->V;
  const long L = (long)s->D * s->H * s->W;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = s->act_dtype == IDEE_DTYPE_BF16;
  const bool split = split_math() || bf;
  const float* x = (const float*)x_; const float* h_saved = (const float*)h_saved_; const float* dy = (const float*)dy_;
  float* dx = (float*)dx_; float* dh_workspace = (float*)dh_workspace_;
#define LMLP(KERN, XT)                                                                                                     \
  hipLaunchKernelGGL(KERN, grid, dim3(256), 0, st, (const XT*)h_saved_, (const XT*)dy_, (XT*)dh_workspace_, p->fc1_w, p->fc1_b, \
                     p->fc2_w, d->fc1_w, d->fc1_b, d->fc2_w, d->fc2_b, s->V, L, p->path_scale)
  {   // part 1: MLP + norm2 (token order)
    if (C == 32 && split) {      // one workgroup per CU: 96 KB of weight fragments + 48 KB of piece images
      const dim3 grid = swin_grid(cdiv(cdiv(L, 32), 4), B, 1);
      if (bf) LMLP((swin_mlp_bwd_b_kernel<32, 2, PM_BF16X1, bf16_t>), bf16_t);
      else LMLP((swin_mlp_bwd_b_kernel<32, 2, PM_BF16X2, float>), float);
    } else if (C == 16 && split) {
      const dim3 grid = swin_grid(cdiv(cdiv(L, 32), 4), B, 2);
      if (bf) LMLP((swin_mlp_bwd_b_kernel<16, 2, PM_BF16X1, bf16_t>), bf16_t);
      else LMLP((swin_mlp_bwd_b_kernel<16, 2, PM_BF16X2, float>), float);
    } else if (C == 16) {   // 32 tokens per wave: half the live registers -> two waves per SIMD overlap GELU with the MFMA chains
      const dim3 grid = swin_grid(cdiv(cdiv(L, 32), 4), B, 2);
      hipLaunchKernelGGL((swin_mlp_bwd_kernel<16, 2>), grid, dim3(256), 0, st, h_saved, dy, dh_workspace, p->fc1_w, p->fc1_b,
                         p->fc2_w, d->fc1_w, d->fc1_b, d->fc2_w, d->fc2_b, s->V, L, p->path_scale);
    } else if (C == 32) {
      const dim3 grid = swin_grid(cdiv(cdiv(L, 64), 4), B, 1);
      hipLaunchKernelGGL((swin_mlp_bwd_kernel<32, 4>), grid, dim3(256), 0, st, h_saved, dy, dh_workspace, p->fc1_w, p->fc1_b,
                         p->fc2_w, d->fc1_w, d->fc1_b, d->fc2_w, d->fc2_b, s->V, L, p->path_scale);
    }
    else return idee::fail(IDEE_ERR_UNSUPPORTED, "swin: C=%d is not built", C);
  }
#undef LMLP
  // C = 16: 32 tokens per wave and work item, 8 waves per workgroup (LDS images 13.8 KB per wave): at most 256 registers
  // per lane, so two waves share a SIMD and one's LDS / global latency hides under the other's MFMA chain.
  // C = 32: 64 tokens per wave, 2 waves (48 KB of images each).
  const int WPB = C == 16 ? 8 : (split ? 4 : 2), MTK = C == 16 ? 2 : (split ? 2 : 4);
  Geom ga = g;
  ga.items = cdiv(g.nWin, (16 * MTK) / NW);
  const dim3 grid = swin_grid(cdiv(ga.items, WPB), B, 1);
  const dim3 grid_v = swin_grid(cdiv(ga.items, WPB) * s->N, s->V, 1);      // split kernels: one grid row per variable
#define LATT(KERN, XT, BLK)                                                                                          \
  hipLaunchKernelGGL(KERN, grid_v, dim3(BLK), 0, st, (const XT*)x_, (const XT*)dh_workspace_, (XT*)dx_, p->qkv_w, p->qkv_b, \
                     p->proj_w, p->rel_bias, d->qkv_w, d->qkv_b, d->proj_w, d->proj_b, d->rel_bias, ga)
#define CALL(C_, NH_, NW_)                                                                                           \
  if (C_ == 32 && bf) LATT((swin_attn_bwd_b_kernel<32, NH_, NW_, 4, 2, false, PM_BF16X1, bf16_t>), bf16_t, 256);     \
  else if (C_ == 16 && bf) LATT((swin_attn_bwd_b_kernel<16, NH_, NW_, 8, 2, true, PM_BF16X1, bf16_t>), bf16_t, 512); \
  else if (C_ == 32 && split) LATT((swin_attn_bwd_b_kernel<32, NH_, NW_, 4, 2, false, PM_BF16X2, float>), float, 256); \
  else if (C_ == 16 && split) LATT((swin_attn_bwd_b_kernel<16, NH_, NW_, 8, 2, true, PM_BF16X2, float>), float, 512); \
  else                                                                                                               \
  hipLaunchKernelGGL((swin_attn_bwd_kernel<C_, NH_, NW_, (C_ == 16 ? 8 : 2), (C_ == 16 ? 2 : 4)>), grid,             \
                     dim3(64 * (C_ == 16 ? 8 : 2)), 0, st, x,                                                        \
                     dh_workspace, dx, p->qkv_w, p->qkv_b, p->proj_w, p->rel_bias, d->qkv_w, d->qkv_b, d->proj_w,    \
                     d->proj_b, d->rel_bias, ga)
  SWIN_DISPATCH(CALL)
#undef CALL
#undef LATT
  IDEE_LAUNCH_CHECK("swin_block_bwd");
  return IDEE_OK;
}

extern "C" int idee_relpos_bias_gather(const float* table, const int64_t* index, float* dense, int V, int table_len,
                                       int heads, int Nw, idee_stream_t stream) {
  IDEE_CHECK_ARG(table && index && dense && V > 0 && table_len > 0 && heads > 0 && Nw > 0, "relpos_gather: bad argument");
  const int total = V * heads * Nw * Nw;
  hipLaunchKernelGGL(relpos_gather_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, table,
                     (const long long*)index, dense, V, table_len, heads, Nw);
  IDEE_LAUNCH_CHECK("relpos_bias_gather");
  return IDEE_OK;
}

extern "C" int idee_relpos_bias_scatter(const float* ddense, const int64_t* index, float* dtable, int V, int table_len,
                                        int heads, int Nw, idee_stream_t stream) {
  IDEE_CHECK_ARG(ddense && index && dtable && V > 0 && table_len > 0 && heads > 0 && Nw > 0, "relpos_scatter: bad argument");
  const int total = V * heads * Nw * Nw;
  hipLaunchKernelGGL(relpos_scatter_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, ddense,
                     (const long long*)index, dtable, V, table_len, heads, Nw);
  IDEE_LAUNCH_CHECK("relpos_bias_scatter");
  return IDEE_OK;
}
