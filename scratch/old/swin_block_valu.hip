// Fused Video-Swin block for gfx950: one lane per token, one window group per wave.
//
// Replaces, for one SwinTransformerBlock3D (reference models/encoder/Swin_3D.py:224-287):
//   norm1 -> zero-pad -> roll(-shift) -> window_partition -> WindowAttention3D (:145-178)
//   -> window_reverse -> roll(+shift) -> crop -> +shortcut -> norm2 -> Mlp (:36-42) -> +residual
// with NO intermediate tensor: the cyclic shift, the padding and the window partition are index
// arithmetic on the channels-last token tensor (N*V, D, H, W, C); the 0/-100 shift mask
// (compute_mask, :340-352) is evaluated from coordinates; the V per-variable weight sets are a
// batch dimension (blockIdx.y = n*V + v).
//
// Mapping: a wave owns G = 64/NW whole windows (NW tokens each), lane = g*NW + i.  Everything a
// token needs (LN, qkv, its score row, softmax, P.V, proj, LN, MLP) stays in that lane's
// registers; K/V rows of the window are exchanged through LDS and read back as broadcasts.
// Weights are wave-uniform and come through the scalar cache (s_load), not LDS.
//
// Backward recomputes the forward from the saved block input (nothing else is saved), runs the
// attention backward as a row pass (dq, d-bias) plus a column pass (dk, dv) so no cross-lane
// reduction is needed, and forms the weight gradients as token-reduction GEMMs on the f32 MFMA
// (v_mfma_f32_16x16x4_f32) from LDS tiles, accumulated in registers over the wave's whole
// share of windows and flushed once with float atomics.
#include "common.h"

namespace {

using namespace idee;

struct Geom {
  int V, D, H, W, Dp, Hp, Wp;
  int wd, wh, ww, sd, sh, sw;
  int nWh, nWw, nWin;   // windows per batch item
  int items;            // window groups (of G windows) per batch item
  int shifted;
  float scale;
};

constexpr float kLnEps = 1e-5f;

template <int C>
__device__ __forceinline__ void layer_norm(const float (&x)[C], float (&xn)[C], float& rstd) {
  float mean = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) mean += x[c];
  mean *= (1.f / C);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float d = x[c] - mean;
    var = fmaf(d, d, var);
  }
  rstd = rsqrtf(var * (1.f / C) + kLnEps);
#pragma unroll
  for (int c = 0; c < C; ++c) xn[c] = (x[c] - mean) * rstd;
}

// dx = rstd * (dxn - mean(dxn) - xn * mean(dxn * xn))
template <int C>
__device__ __forceinline__ void layer_norm_bwd(const float (&dxn)[C], const float (&xn)[C], float rstd, float (&dx)[C]) {
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    a += dxn[c];
    b = fmaf(dxn[c], xn[c], b);
  }
  a *= (1.f / C);
  b *= (1.f / C);
#pragma unroll
  for (int c = 0; c < C; ++c) dx[c] = rstd * (dxn[c] - a - xn[c] * b);
}

__device__ __forceinline__ float gelu(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float a) {
  return 0.5f * (1.f + erff(a * 0.70710678118654752f)) + a * 0.39894228040143268f * __expf(-0.5f * a * a);
}

// Token coordinates of lane (g, i) of work item `item` for batch item b.
struct Tok {
  long off;      // element offset of the token's channel vector, valid only if `valid`
  int rid;       // shift-mask region id (position in the rolled frame)
  bool valid;    // token lies inside the un-padded grid
  bool active;   // the window exists (item not past the end)
};

template <int NW>
__device__ __forceinline__ Tok locate(const Geom& g, int b, int item, int lane, int C) {
  constexpr int G = 64 / NW;
  Tok t;
  const int gi = lane / NW, i = lane % NW;
  const int wl = item * G + gi;
  t.active = (item < g.items) && (wl < g.nWin);
  const int wlc = t.active ? wl : 0;
  const int iwd = wlc / (g.nWh * g.nWw), r = wlc % (g.nWh * g.nWw);
  const int iwh = r / g.nWw, iww = r % g.nWw;
  const int dt = i / (g.wh * g.ww), r2 = i % (g.wh * g.ww);
  const int dh = r2 / g.ww, dw = r2 % g.ww;
  const int pd = iwd * g.wd + dt, ph = iwh * g.wh + dh, pw = iww * g.ww + dw;   // rolled frame
  auto axis = [](int p, int P, int w, int s) { return s == 0 ? 2 : (int)(p >= P - w) + (int)(p >= P - s); };
  t.rid = axis(pd, g.Dp, g.wd, g.sd) * 9 + axis(ph, g.Hp, g.wh, g.sh) * 3 + axis(pw, g.Wp, g.ww, g.sw);
  int sd_ = pd + g.sd, sh_ = ph + g.sh, sw_ = pw + g.sw;                         // un-rolled frame
  if (sd_ >= g.Dp) sd_ -= g.Dp;
  if (sh_ >= g.Hp) sh_ -= g.Hp;
  if (sw_ >= g.Wp) sw_ -= g.Wp;
  t.valid = t.active && sd_ < g.D && sh_ < g.H && sw_ < g.W;
  t.off = ((((long)b * g.D + sd_) * g.H + sh_) * g.W + sw_) * C;
  return t;
}

template <int C>
__device__ __forceinline__ void load_row(const float* p, float (&x)[C]) {
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    f32x4 v = ld4(p + c);
    x[c] = v[0]; x[c + 1] = v[1]; x[c + 2] = v[2]; x[c + 3] = v[3];
  }
}
template <int C>
__device__ __forceinline__ void store_row(float* p, const float (&x)[C]) {
#pragma unroll
  for (int c = 0; c < C; c += 4) st4(p + c, f32x4{x[c], x[c + 1], x[c + 2], x[c + 3]});
}

// y[o] = b[o] + sum_c w[o*CI + c] * x[c]   (w, b wave-uniform -> scalar loads)
template <int CO, int CI>
__device__ __forceinline__ void linear(const float* __restrict__ w, const float* __restrict__ b, const float (&x)[CI],
                                       float (&y)[CO]) {
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    float acc = b[o];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc = fmaf(w[o * CI + c], x[c], acc);
    y[o] = acc;
  }
}
// y[c] = sum_o w[o*CI + c] * x[o]   (transposed use of the same matrix)
template <int CO, int CI>
__device__ __forceinline__ void linear_t(const float* __restrict__ w, const float (&x)[CO], float (&y)[CI]) {
#pragma unroll
  for (int c = 0; c < CI; ++c) y[c] = 0.f;
#pragma unroll
  for (int o = 0; o < CO; ++o) {
#pragma unroll
    for (int c = 0; c < CI; ++c) y[c] = fmaf(w[o * CI + c], x[o], y[c]);
  }
}

// Score row of token i (this lane) against the NW keys of its window, head h.
// rows: LDS window base, row stride RS floats, key at [0,C) ; rid (int bits) at ridpos.
template <int HD, int NW, int RS>
__device__ __forceinline__ void score_row(float (&s)[NW], const float* qh, const float* rows, int koff, int ridpos,
                                          const float* bias_row, int rid_i, bool shifted) {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const float* kj = rows + j * RS + koff;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      f32x4 k4 = ld4(kj + d);
      acc = fmaf(qh[d], k4[0], acc);
      acc = fmaf(qh[d + 1], k4[1], acc);
      acc = fmaf(qh[d + 2], k4[2], acc);
      acc = fmaf(qh[d + 3], k4[3], acc);
    }
    acc += bias_row[j];
    if (shifted) {
      const int rid_j = __float_as_int(rows[j * RS + ridpos]);
      acc += (rid_j != rid_i) ? -100.f : 0.f;
    }
    s[j] = acc;
  }
}

template <int NW>
__device__ __forceinline__ void softmax_row(float (&s)[NW], float& mx, float& inv_l) {
  mx = s[0];
#pragma unroll
  for (int j = 1; j < NW; ++j) mx = fmaxf(mx, s[j]);
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    s[j] = __expf(s[j] - mx);
    l += s[j];
  }
  inv_l = 1.f / l;
#pragma unroll
  for (int j = 0; j < NW; ++j) s[j] *= inv_l;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int C, int NH, int NW>
__global__ __launch_bounds__(256) void swin_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ qkv_w,
    const float* __restrict__ qkv_b, const float* __restrict__ proj_w, const float* __restrict__ proj_b,
    const float* __restrict__ fc1_w, const float* __restrict__ fc1_b, const float* __restrict__ fc2_w,
    const float* __restrict__ fc2_b, const float* __restrict__ rel_bias, Geom g) {
  constexpr int HD = C / NH, HID = 4 * C, RS = 2 * C + 4, BS = NW + 1;
  __shared__ __attribute__((aligned(16))) float s_rows[4][64 * RS];
  __shared__ float s_bias[NH * NW * BS];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, v = b % g.V;
  qkv_w += (long)v * 3 * C * C;  qkv_b += v * 3 * C;
  proj_w += (long)v * C * C;     proj_b += v * C;
  fc1_w += (long)v * HID * C;    fc1_b += v * HID;
  fc2_w += (long)v * C * HID;    fc2_b += v * C;
  rel_bias += (long)v * NH * NW * NW;

  for (int e = threadIdx.x; e < NH * NW * NW; e += 256) s_bias[(e / NW) * BS + (e % NW)] = rel_bias[e];

  float* rows = s_rows[wave];
  float* myrow = rows + lane * RS;
  const float* wrows = rows + (lane / NW) * NW * RS;   // my window's rows
  const int i = lane % NW;

  for (int it0 = blockIdx.x * 4; it0 < g.items; it0 += gridDim.x * 4) {
    const Tok t = locate<NW>(g, b, it0 + wave, lane, C);
    // Opaque zero: keeps the (wave-uniform, loop-invariant) weight loads inside the loop body so
    // they stream through the scalar cache instead of being hoisted and spilled.
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const float* qkv_w_ = qkv_w + oz; const float* qkv_b_ = qkv_b + oz;
    const float* proj_w_ = proj_w + oz; const float* proj_b_ = proj_b + oz;
    const float* fc1_w_ = fc1_w + oz; const float* fc1_b_ = fc1_b + oz;
    const float* fc2_w_ = fc2_w + oz; const float* fc2_b_ = fc2_b + oz;
    float xr[C], xn[C], rstd;
    if (t.valid) load_row<C>(x + t.off, xr);
    else {
#pragma unroll
      for (int c = 0; c < C; ++c) xr[c] = 0.f;
    }
    layer_norm<C>(xr, xn, rstd);
    if (!t.valid) {
#pragma unroll
      for (int c = 0; c < C; ++c) xn[c] = 0.f;   // zero padding is applied AFTER norm1 (Swin_3D.py:229-238)
    }
    float q[C];
    {
      float qkv[3 * C];
      linear<3 * C, C>(qkv_w_, qkv_b_, xn, qkv);
#pragma unroll
      for (int c = 0; c < C; ++c) q[c] = qkv[c] * g.scale;
      __syncthreads();   // previous iteration's readers are done with the rows
#pragma unroll
      for (int c = 0; c < 2 * C; c += 4) st4(myrow + c, f32x4{qkv[C + c], qkv[C + c + 1], qkv[C + c + 2], qkv[C + c + 3]});
      myrow[2 * C] = __int_as_float(t.rid);
    }
    __syncthreads();

    float o[C];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float s[NW], mx, inv_l;
      score_row<HD, NW, RS>(s, q + h * HD, wrows, h * HD, 2 * C, s_bias + (h * NW + i) * BS, t.rid, g.shifted != 0);
      softmax_row<NW>(s, mx, inv_l);
#pragma unroll
      for (int d = 0; d < HD; ++d) o[h * HD + d] = 0.f;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const float* vj = wrows + j * RS + C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          f32x4 v4 = ld4(vj + d);
          o[h * HD + d] = fmaf(s[j], v4[0], o[h * HD + d]);
          o[h * HD + d + 1] = fmaf(s[j], v4[1], o[h * HD + d + 1]);
          o[h * HD + d + 2] = fmaf(s[j], v4[2], o[h * HD + d + 2]);
          o[h * HD + d + 3] = fmaf(s[j], v4[3], o[h * HD + d + 3]);
        }
      }
    }

    float hres[C];
    linear<C, C>(proj_w_, proj_b_, o, hres);
#pragma unroll
    for (int c = 0; c < C; ++c) hres[c] += xr[c];
    float hn[C], rstd2;
    layer_norm<C>(hres, hn, rstd2);
    float out[C];
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = hres[c] + fc2_b_[c];
#pragma unroll 4
    for (int m = 0; m < HID; ++m) {
      float a = fc1_b_[m];
#pragma unroll
      for (int c = 0; c < C; ++c) a = fmaf(fc1_w_[m * C + c], hn[c], a);
      const float ga = gelu(a);
#pragma unroll
      for (int c = 0; c < C; ++c) out[c] = fmaf(fc2_w_[c * HID + m], ga, out[c]);
    }
    if (t.valid) store_row<C>(y + t.off, out);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (recompute from the block input)
// ---------------------------------------------------------------------------------------------

// acc(16x16) += A^T B over the wave's 64 tokens: A, B are LDS tiles [64 tok][16], acc[m][n] =
// sum_tok A[tok][m] B[tok][n].  Lane (r = lane&15, kq = lane>>4) feeds A[4s+kq][r], B[4s+kq][r].
// asum accumulates this lane's A operands (column sums of A after a 4-lane reduce).
__device__ __forceinline__ void tile_gemm_tn(f32x4& acc, float& asum, const float* A, const float* B, int lane) {
  const int r = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float a = A[(4 * s + kq) * 16 + r];
    const float bb = B[(4 * s + kq) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
    asum += a;
  }
}

// write this lane's 16 values as row `lane` of a [64][16] tile
__device__ __forceinline__ void tile_put(float* T, int lane, const float* v16) {
#pragma unroll
  for (int c = 0; c < 16; c += 4) st4(T + lane * 16 + c, f32x4{v16[c], v16[c + 1], v16[c + 2], v16[c + 3]});
}

// flush a 16x16 accumulator (D layout: col = lane&15, row = (lane>>4)*4 + reg) into dst[row*ld + col]
__device__ __forceinline__ void flush_acc(float* dst, int ld, const f32x4& acc, int lane) {
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(dst + (long)(r0 + r) * ld + col, acc[r]);
}
// flush column sums held as per-lane partials (lane r + 16*kq): reduce over kq, lanes 0..15 add
__device__ __forceinline__ void flush_colsum(float* dst, float part, int lane) {
  part += __shfl_xor(part, 16, 64);
  part += __shfl_xor(part, 32, 64);
  if (lane < 16) atomicAdd(dst + lane, part);
}

template <int C, int NH, int NW>
__global__ __launch_bounds__(256) void swin_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
    const float* __restrict__ qkv_w, const float* __restrict__ qkv_b, const float* __restrict__ proj_w,
    const float* __restrict__ proj_b, const float* __restrict__ fc1_w, const float* __restrict__ fc1_b,
    const float* __restrict__ fc2_w, const float* __restrict__ fc2_b, const float* __restrict__ rel_bias,
    float* __restrict__ g_qkv_w, float* __restrict__ g_qkv_b, float* __restrict__ g_proj_w,
    float* __restrict__ g_proj_b, float* __restrict__ g_fc1_w, float* __restrict__ g_fc1_b,
    float* __restrict__ g_fc2_w, float* __restrict__ g_fc2_b, float* __restrict__ g_rel_bias, Geom g) {
  static_assert(C % 16 == 0, "C must be a multiple of 16");
  constexpr int HD = C / NH, HID = 4 * C, BS = NW + 1;
  constexpr int CT = C / 16;          // 16-wide channel tiles
  constexpr int ST = 2 * NH + 2;      // per-row stats: rid, then (max, 1/l, D) per head, padded
  constexpr int RS = 2 * C + ((1 + 3 * NH + 3) / 4) * 4;
  constexpr int MX = 2 * C + 1, IL = MX + NH, DD = IL + NH;   // stat positions within a row
  (void)ST;
  __shared__ __attribute__((aligned(16))) float s_rows[4][64 * RS];
  __shared__ __attribute__((aligned(16))) float s_tile[4][4][64 * 16];   // [wave][P,Q,R,S]
  __shared__ float s_bias[NH * NW * BS];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, v = b % g.V;
  qkv_w += (long)v * 3 * C * C;  qkv_b += v * 3 * C;
  proj_w += (long)v * C * C;     proj_b += v * C;
  fc1_w += (long)v * HID * C;    fc1_b += v * HID;
  fc2_w += (long)v * C * HID;    fc2_b += v * C;
  rel_bias += (long)v * NH * NW * NW;
  g_qkv_w += (long)v * 3 * C * C;  g_qkv_b += v * 3 * C;
  g_proj_w += (long)v * C * C;     g_proj_b += v * C;
  g_fc1_w += (long)v * HID * C;    g_fc1_b += v * HID;
  g_fc2_w += (long)v * C * HID;    g_fc2_b += v * C;
  g_rel_bias += (long)v * NH * NW * NW;

  for (int e = threadIdx.x; e < NH * NW * NW; e += 256) s_bias[(e / NW) * BS + (e % NW)] = rel_bias[e];

  float* rows = s_rows[wave];
  float* myrow = rows + lane * RS;
  const float* wrows = rows + (lane / NW) * NW * RS;
  float* tP = s_tile[wave][0];
  float* tQ = s_tile[wave][1];
  float* tR = s_tile[wave][2];
  float* tS = s_tile[wave][3];
  const int i = lane % NW;

  // weight-gradient accumulators, live across the whole item loop
  f32x4 a_fc1[HID / 16][CT], a_fc2[CT][HID / 16], a_qkv[3 * CT][CT], a_proj[CT][CT];
  float b_fc1[HID / 16], b_fc2[CT], b_qkv[3 * CT], b_proj[CT];
  float a_bias[NH][NW];
#pragma unroll
  for (int m = 0; m < HID / 16; ++m) {
    b_fc1[m] = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) { a_fc1[m][c] = f32x4{0, 0, 0, 0}; a_fc2[c][m] = f32x4{0, 0, 0, 0}; }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    b_fc2[c] = 0.f; b_proj[c] = 0.f;
#pragma unroll
    for (int d = 0; d < CT; ++d) a_proj[c][d] = f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    b_qkv[o] = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) a_qkv[o][c] = f32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int j = 0; j < NW; ++j) a_bias[h][j] = 0.f;

  for (int it0 = blockIdx.x * 4; it0 < g.items; it0 += gridDim.x * 4) {
    const Tok t = locate<NW>(g, b, it0 + wave, lane, C);
    int oz;   // opaque zero, see the forward kernel
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const float* qkv_w_ = qkv_w + oz; const float* qkv_b_ = qkv_b + oz;
    const float* proj_w_ = proj_w + oz; const float* proj_b_ = proj_b + oz;
    const float* fc1_w_ = fc1_w + oz; const float* fc1_b_ = fc1_b + oz;
    const float* fc2_w_ = fc2_w + oz;
    // ---- recompute forward ----
    float xr[C], xn[C], rstd1, dyr[C];
    if (t.valid) { load_row<C>(x + t.off, xr); load_row<C>(dy + t.off, dyr); }
    else {
#pragma unroll
      for (int c = 0; c < C; ++c) { xr[c] = 0.f; dyr[c] = 0.f; }
    }
    layer_norm<C>(xr, xn, rstd1);
    if (!t.valid) {
#pragma unroll
      for (int c = 0; c < C; ++c) xn[c] = 0.f;
    }
    float q[C], kk[C], vv[C];
    {
      float qkv[3 * C];
      linear<3 * C, C>(qkv_w_, qkv_b_, xn, qkv);
#pragma unroll
      for (int c = 0; c < C; ++c) { q[c] = qkv[c] * g.scale; kk[c] = qkv[C + c]; vv[c] = qkv[2 * C + c]; }
    }
    __syncthreads();   // (1) all readers of the previous iteration's rows/tiles are done
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      st4(myrow + c, f32x4{kk[c], kk[c + 1], kk[c + 2], kk[c + 3]});
      st4(myrow + C + c, f32x4{vv[c], vv[c + 1], vv[c + 2], vv[c + 3]});
    }
    myrow[2 * C] = __int_as_float(t.rid);
    __syncthreads();   // (2) K/V rows visible

    float o[C], mxs[NH], ils[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float s[NW];
      score_row<HD, NW, RS>(s, q + h * HD, wrows, h * HD, 2 * C, s_bias + (h * NW + i) * BS, t.rid, g.shifted != 0);
      softmax_row<NW>(s, mxs[h], ils[h]);
#pragma unroll
      for (int d = 0; d < HD; ++d) o[h * HD + d] = 0.f;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const float* vj = wrows + j * RS + C + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          f32x4 v4 = ld4(vj + d);
          o[h * HD + d] = fmaf(s[j], v4[0], o[h * HD + d]);
          o[h * HD + d + 1] = fmaf(s[j], v4[1], o[h * HD + d + 1]);
          o[h * HD + d + 2] = fmaf(s[j], v4[2], o[h * HD + d + 2]);
          o[h * HD + d + 3] = fmaf(s[j], v4[3], o[h * HD + d + 3]);
        }
      }
    }
    float hres[C], hn[C], rstd2;
    linear<C, C>(proj_w_, proj_b_, o, hres);
#pragma unroll
    for (int c = 0; c < C; ++c) hres[c] += xr[c];
    layer_norm<C>(hres, hn, rstd2);

    // ---- MLP backward; weight grads by 16-wide hidden chunks ----
    // tiles: P = hn (B operand of dfc1), Q = dy (A operand of dfc2), R = da chunk, S = gelu chunk
    float dhn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dhn[c] = 0.f;
#pragma unroll
    for (int mc = 0; mc < HID / 16; ++mc) {
      float da[16], ga[16];
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) {
        const int m = mc * 16 + mm;
        float a = fc1_b_[m];
#pragma unroll
        for (int c = 0; c < C; ++c) a = fmaf(fc1_w_[m * C + c], hn[c], a);
        ga[mm] = gelu(a);
        float dg = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) dg = fmaf(fc2_w_[c * HID + m], dyr[c], dg);
        da[mm] = dg * gelu_grad(a);
#pragma unroll
        for (int c = 0; c < C; ++c) dhn[c] = fmaf(da[mm], fc1_w_[m * C + c], dhn[c]);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        __syncthreads();   // previous tile readers done
        tile_put(tR, lane, da);
        tile_put(tS, lane, ga);
        tile_put(tP, lane, hn + ct * 16);
        tile_put(tQ, lane, dyr + ct * 16);
        __syncthreads();
        float dummy = 0.f;
        tile_gemm_tn(a_fc1[mc][ct], ct == 0 ? b_fc1[mc] : dummy, tR, tP, lane);   // dfc1[m][c] += da^T hn
        tile_gemm_tn(a_fc2[ct][mc], mc == 0 ? b_fc2[ct] : dummy, tQ, tS, lane);   // dfc2[c][m] += dy^T g
      }
    }
    // ---- through norm2 and the residual ----
    float dh[C];
    layer_norm_bwd<C>(dhn, hn, rstd2, dh);
#pragma unroll
    for (int c = 0; c < C; ++c) dh[c] += dyr[c];
    // proj backward: do = W^T dh ; dproj_w[c][c'] += dh[c] o[c'] ; dproj_b += dh
    float dov[C];
    linear_t<C, C>(proj_w_, dh, dov);
#pragma unroll
    for (int ca = 0; ca < CT; ++ca) {
#pragma unroll
      for (int cb = 0; cb < CT; ++cb) {
        __syncthreads();
        tile_put(tP, lane, dh + ca * 16);
        tile_put(tQ, lane, o + cb * 16);
        __syncthreads();
        float dummy = 0.f;
        tile_gemm_tn(a_proj[ca][cb], cb == 0 ? b_proj[ca] : dummy, tP, tQ, lane);
      }
    }

    // ---- attention backward: row pass (dq, d-bias, row stats) ----
    float dq[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dq[c] = 0.f;
    float Dh[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float s[NW];
      score_row<HD, NW, RS>(s, q + h * HD, wrows, h * HD, 2 * C, s_bias + (h * NW + i) * BS, t.rid, g.shifted != 0);
#pragma unroll
      for (int j = 0; j < NW; ++j) s[j] = __expf(s[j] - mxs[h]) * ils[h];
      float dp[NW], D = 0.f;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const float* vj = wrows + j * RS + C + h * HD;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          f32x4 v4 = ld4(vj + d);
          acc = fmaf(dov[h * HD + d], v4[0], acc);
          acc = fmaf(dov[h * HD + d + 1], v4[1], acc);
          acc = fmaf(dov[h * HD + d + 2], v4[2], acc);
          acc = fmaf(dov[h * HD + d + 3], v4[3], acc);
        }
        dp[j] = acc;
        D = fmaf(s[j], acc, D);
      }
      Dh[h] = D;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const float ds = s[j] * (dp[j] - D);
        a_bias[h][j] += ds;
        const float* kj = wrows + j * RS + h * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          f32x4 k4 = ld4(kj + d);
          dq[h * HD + d] = fmaf(ds, k4[0], dq[h * HD + d]);
          dq[h * HD + d + 1] = fmaf(ds, k4[1], dq[h * HD + d + 1]);
          dq[h * HD + d + 2] = fmaf(ds, k4[2], dq[h * HD + d + 2]);
          dq[h * HD + d + 3] = fmaf(ds, k4[3], dq[h * HD + d + 3]);
        }
      }
    }
    __syncthreads();   // (3) everyone finished reading K/V rows: overwrite them with q / do / stats
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      st4(myrow + c, f32x4{q[c], q[c + 1], q[c + 2], q[c + 3]});
      st4(myrow + C + c, f32x4{dov[c], dov[c + 1], dov[c + 2], dov[c + 3]});
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) { myrow[MX + h] = mxs[h]; myrow[IL + h] = ils[h]; myrow[DD + h] = Dh[h]; }
    __syncthreads();   // (4)

    // ---- column pass: this lane is key j = i ; loops over the queries of its window ----
    float dk[C], dv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll 4
      for (int qi = 0; qi < NW; ++qi) {
        const float* r = wrows + qi * RS;
        float sc = 0.f, dp = 0.f;
        float qv[HD], dv_[HD];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          f32x4 q4 = ld4(r + h * HD + d);
          f32x4 d4 = ld4(r + C + h * HD + d);
          qv[d] = q4[0]; qv[d + 1] = q4[1]; qv[d + 2] = q4[2]; qv[d + 3] = q4[3];
          dv_[d] = d4[0]; dv_[d + 1] = d4[1]; dv_[d + 2] = d4[2]; dv_[d + 3] = d4[3];
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) {
          sc = fmaf(qv[d], kk[h * HD + d], sc);
          dp = fmaf(dv_[d], vv[h * HD + d], dp);
        }
        sc += s_bias[(h * NW + qi) * BS + i];
        if (g.shifted) sc += (__float_as_int(r[2 * C]) != t.rid) ? -100.f : 0.f;
        const float p = __expf(sc - r[MX + h]) * r[IL + h];
        const float ds = p * (dp - r[DD + h]);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
          dk[h * HD + d] = fmaf(ds, qv[d], dk[h * HD + d]);
          dv[h * HD + d] = fmaf(p, dv_[d], dv[h * HD + d]);
        }
      }
    }
    // ---- qkv backward ----
    float dqkv[3 * C];
#pragma unroll
    for (int c = 0; c < C; ++c) { dqkv[c] = dq[c] * g.scale; dqkv[C + c] = dk[c]; dqkv[2 * C + c] = dv[c]; }
    if (!t.active) {
#pragma unroll
      for (int c = 0; c < 3 * C; ++c) dqkv[c] = 0.f;
    }
    float dxn[C];
    linear_t<3 * C, C>(qkv_w_, dqkv, dxn);
#pragma unroll
    for (int oc = 0; oc < 3 * CT; ++oc) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        __syncthreads();
        tile_put(tP, lane, dqkv + oc * 16);
        tile_put(tQ, lane, xn + ct * 16);
        __syncthreads();
        float dummy = 0.f;
        tile_gemm_tn(a_qkv[oc][ct], ct == 0 ? b_qkv[oc] : dummy, tP, tQ, lane);
      }
    }
    // ---- through norm1 and the shortcut ----
    float dxr[C];
    layer_norm_bwd<C>(dxn, xn, rstd1, dxr);
#pragma unroll
    for (int c = 0; c < C; ++c) dxr[c] += dh[c];
    if (t.valid) store_row<C>(dx + t.off, dxr);
  }

  // ---- flush weight gradients ----
#pragma unroll
  for (int m = 0; m < HID / 16; ++m) {
    flush_colsum(g_fc1_b + m * 16, b_fc1[m], lane);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      flush_acc(g_fc1_w + (long)(m * 16) * C + c * 16, C, a_fc1[m][c], lane);
      flush_acc(g_fc2_w + (long)(c * 16) * HID + m * 16, HID, a_fc2[c][m], lane);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    flush_colsum(g_fc2_b + c * 16, b_fc2[c], lane);
    flush_colsum(g_proj_b + c * 16, b_proj[c], lane);
#pragma unroll
    for (int d = 0; d < CT; ++d) flush_acc(g_proj_w + (long)(c * 16) * C + d * 16, C, a_proj[c][d], lane);
  }
#pragma unroll
  for (int o = 0; o < 3 * CT; ++o) {
    flush_colsum(g_qkv_b + o * 16, b_qkv[o], lane);
#pragma unroll
    for (int c = 0; c < CT; ++c) flush_acc(g_qkv_w + (long)(o * 16) * C + c * 16, C, a_qkv[o][c], lane);
  }
  // d-bias: lanes of the G windows of a wave hold partials of the same (h, i, j): reduce over g first
#pragma unroll
  for (int h = 0; h < NH; ++h) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      float vsum = a_bias[h][j];
#pragma unroll
      for (int off = NW; off < 64; off <<= 1) vsum += __shfl_xor(vsum, off, 64);
      if (lane < NW) atomicAdd(g_rel_bias + (h * NW + lane) * NW + j, vsum);
    }
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
  g.V = s->V; g.D = s->D; g.H = s->H; g.W = s->W;
  g.wd = s->wd; g.wh = s->wh; g.ww = s->ww; g.sd = s->sd; g.sh = s->sh; g.sw = s->sw;
  g.Dp = cdiv(s->D, s->wd) * s->wd; g.Hp = cdiv(s->H, s->wh) * s->wh; g.Wp = cdiv(s->W, s->ww) * s->ww;
  g.nWh = g.Hp / s->wh; g.nWw = g.Wp / s->ww;
  g.nWin = (g.Dp / s->wd) * g.nWh * g.nWw;
  g.shifted = (s->sd | s->sh | s->sw) != 0;
  g.scale = s->scale;
  if (NW != 8 && NW != 16 && NW != 32 && NW != 64)
    return idee::fail(IDEE_ERR_UNSUPPORTED, "swin: window volume %d not in {8,16,32,64}", NW);
  g.items = cdiv(g.nWin, 64 / NW);
  return IDEE_OK;
}

dim3 swin_grid(const Geom& g, int B, int wg_per_cu) {
  int gx = imin(cdiv(g.items, 4), cdiv((long)kCUs * wg_per_cu, B));
  if (gx < 1) gx = 1;
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

extern "C" int idee_swin_block_fwd(const float* x, float* y, const idee_swin_params* p, const idee_swin_geom* s,
                                   idee_stream_t stream) {
  IDEE_CHECK_ARG(x && y && p && s, "swin_block_fwd: null argument");
  Geom g; int NW;
  if (int rc = make_geom(s, g, NW)) return rc;
  const int C = s->C, NH = s->heads, B = s->N * s->V;
  const dim3 grid = swin_grid(g, B, 8);
#define CALL(C_, NH_, NW_)                                                                             \
  hipLaunchKernelGGL((swin_fwd_kernel<C_, NH_, NW_>), grid, dim3(256), 0, (hipStream_t)stream, x, y,   \
                     p->qkv_w, p->qkv_b, p->proj_w, p->proj_b, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, \
                     p->rel_bias, g)
  SWIN_DISPATCH(CALL)
#undef CALL
  IDEE_LAUNCH_CHECK("swin_block_fwd");
  return IDEE_OK;
}

extern "C" int idee_swin_block_bwd(const float* x, const float* dy, float* dx, const idee_swin_params* p,
                                   const idee_swin_grads* d, const idee_swin_geom* s, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && dy && dx && p && d && s, "swin_block_bwd: null argument");
  Geom g; int NW;
  if (int rc = make_geom(s, g, NW)) return rc;
  const int C = s->C, NH = s->heads, B = s->N * s->V;
  const dim3 grid = swin_grid(g, B, 2);
#define CALL(C_, NH_, NW_)                                                                              \
  hipLaunchKernelGGL((swin_bwd_kernel<C_, NH_, NW_>), grid, dim3(256), 0, (hipStream_t)stream, x, dy,   \
                     dx, p->qkv_w, p->qkv_b, p->proj_w, p->proj_b, p->fc1_w, p->fc1_b, p->fc2_w,        \
                     p->fc2_b, p->rel_bias, d->qkv_w, d->qkv_b, d->proj_w, d->proj_b, d->fc1_w,         \
                     d->fc1_b, d->fc2_w, d->fc2_b, d->rel_bias, g)
  SWIN_DISPATCH(CALL)
#undef CALL
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
