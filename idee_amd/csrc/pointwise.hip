// Bandwidth-bound kernels of the path: PatchEmbed3D(k=1)+LayerNorm, LFQ, the losses, Adam.
// All are one-pass streaming kernels with 16-byte accesses; batch-wide sums are reduced
// per block into a workspace and finished by a single block in fixed order (deterministic).
#include "common.h"

namespace {

using namespace idee;

constexpr float kLnEps = 1e-5f;

// block-wide sum of R values per thread -> thread 0 holds the totals in v[]
template <int R>
__device__ __forceinline__ void block_sum(float (&v)[R], float* smem /* >= R * 4 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = wave_sum(v[r]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int r = 0; r < R; ++r) smem[wave * R + r] = v[r];
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += smem[w * R + r];
      v[r] = s;
    }
  }
}

// -------------------------------------------------------------------------------------------
// PatchEmbed3D with patch (1,1,1): 1x1x1 conv (optional bias) + non-affine LN over channels.
// Swin_3D.py:473-491 ; encoder/CNN_3D.py:44-71.   x (N,V,CIN,L) -> y (N,V,L,E)
// -------------------------------------------------------------------------------------------
// sum over the QE = E/4 adjacent lanes that share a token
template <int QE>
__device__ __forceinline__ float tok_sum(float v) {
  // lane xor 1 / xor 2 inside a quad are DPP quad permutations (one VALU slot each) — __shfl_xor would go through
  // ds_bpermute and the LDS pipe
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
#pragma unroll
  for (int o = 4; o < QE; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int E, int CIN, typename AT>
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, AT* __restrict__ y,
                                                              int V, long L) {
  // E/4 adjacent lanes per token, 4 channels each: the output is one contiguous stream of 16-byte pieces
  constexpr int QE = E / 4;
  const int b = blockIdx.y, v = b % V;
  const float* wv = w + (long)v * E * CIN;
  const float* xb = x + (long)b * CIN * L;
  AT* yb = y + (long)b * L * E;
  const int q = threadIdx.x % QE;
  float wl[4][CIN], bl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bl[k] = bias ? bias[v * E + 4 * q + k] : 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) wl[k][c] = wv[(4 * q + k) * CIN + c];
  }
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < L * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    float xi[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) xi[c] = xb[c * L + tok];
    float e[4], sm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = bl[k];
#pragma unroll
      for (int c = 0; c < CIN; ++c) a = fmaf(wl[k][c], xi[c], a);
      e[k] = a; sm += a;
    }
    const float mean = tok_sum<QE>(sm) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = e[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    stx4<AT>(yb + tok * E + 4 * q, f32x4{(e[0] - mean) * rstd, (e[1] - mean) * rstd, (e[2] - mean) * rstd, (e[3] - mean) * rstd});
  }
}

template <int E, int CIN, typename AT>
__global__ __launch_bounds__(256) void patch_embed_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const AT* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int V, long L) {
  // E/4 adjacent lanes per token, 4 channels each: dy is read as one contiguous stream of 16-byte pieces
  constexpr int QE = E / 4;
  const int b = blockIdx.y, v = b % V;
  const float* wv = w + (long)v * E * CIN;
  const float* xb = x + (long)b * CIN * L;
  const AT* dyb = dy + (long)b * L * E;
  const int q = threadIdx.x % QE;
  float wl[4][CIN], bl[4], gw[4][CIN], gb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bl[k] = bias ? bias[v * E + 4 * q + k] : 0.f;
    gb[k] = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) { wl[k][c] = wv[(4 * q + k) * CIN + c]; gw[k][c] = 0.f; }
  }
  // four tokens per lane and trip: their loads are all in flight before the first dependent instruction
  constexpr int UN = 8;
  const long total = L * QE, stride = (long)gridDim.x * 256;
  for (long u0 = (long)blockIdx.x * 256 + threadIdx.x; u0 < total; u0 += UN * stride) {
    f32x4 gq[UN];
    float xq[UN][CIN];
    bool on[UN];
#pragma unroll
    for (int r = 0; r < UN; ++r) {
      const long u = u0 + r * stride;
      on[r] = u < total;
      const long tok = (on[r] ? u : u0) / QE;
      gq[r] = ldx4<AT>(dyb + tok * E + 4 * q);
#pragma unroll
      for (int c = 0; c < CIN; ++c) xq[r][c] = xb[c * L + tok];
    }
#pragma unroll
    for (int r = 0; r < UN; ++r) {
      const f32x4 g = on[r] ? gq[r] : f32x4{0.f, 0.f, 0.f, 0.f};     // a zero gradient contributes nothing
      float e[4], sm = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = bl[k];
#pragma unroll
        for (int c = 0; c < CIN; ++c) a = fmaf(wl[k][c], xq[r][c], a);
        e[k] = a; sm += a;
      }
      const float mean = tok_sum<QE>(sm) * (1.f / E);
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = e[k] - mean; var = fmaf(d, d, var); }
      const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { e[k] = (e[k] - mean) * rstd; s1 += g[k]; s2 = fmaf(g[k], e[k], s2); }
      s1 = tok_sum<QE>(s1) * (1.f / E); s2 = tok_sum<QE>(s2) * (1.f / E);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float de = rstd * (g[k] - s1 - e[k] * s2);
        gb[k] += de;
#pragma unroll
        for (int c = 0; c < CIN; ++c) gw[k][c] = fmaf(de, xq[r][c], gw[k][c]);
      }
    }
  }
  // lanes with equal q hold partials of the same 4 channels
  auto same_q_sum = [](float val) {
#pragma unroll
    for (int o = QE; o < 64; o <<= 1) val += __shfl_xor(val, o, 64);
    return val;
  };
  // the workgroup's 4 waves meet in LDS, one global atomic per value and workgroup (see lfq_bwd_kernel: same-address float
  // atomics retire one every ~17 ns)
  constexpr int NV = E * (CIN + 1);              // [db: E][dw: E x CIN]
  __shared__ float s_fl[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sb = same_q_sum(gb[k]);
    if (lane < QE) s_fl[wave][4 * q + k] = sb;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float sw = same_q_sum(gw[k][c]);
      if (lane < QE) s_fl[wave][E + (4 * q + k) * CIN + c] = sw;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += 256) {
    const float val = (s_fl[0][i] + s_fl[1][i]) + (s_fl[2][i] + s_fl[3][i]);
    if (i < E) { if (db) atomicAdd(db + v * E + i, val); }
    else atomicAdd(dw + (long)v * E * CIN + (i - E), val);
  }
}


// -------------------------------------------------------------------------------------------
// PatchEmbed3D with a patch (pt, ph, pw) != (1,1,1) (Swin_3D.py:473-491: Conv3d(kernel = stride = patch) after zero padding to
// a multiple of the patch, then the non-affine LN).  The caller presents the input already gathered per patch,
// x (B, K, L') with K = Cin pt ph pw in the weight's own (Cin, pt, ph, pw) order and L' output tokens, so this is the (1,1,1)
// kernel above with a RUN-TIME number of input values per token (K <= 128); the weights sit in LDS.
// -------------------------------------------------------------------------------------------
constexpr int kPatchKMax = 128;
template <int E, typename AT>
__global__ __launch_bounds__(256) void patch_embed_k_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, AT* __restrict__ y, int V, long L,
                                                                int K) {
  constexpr int QE = E / 4;
  __shared__ float s_w[E * kPatchKMax];
  const int b = blockIdx.y, v = b % V;
  for (int i = threadIdx.x; i < E * K; i += 256) s_w[i] = w[(long)v * E * K + i];
  __syncthreads();
  const float* xb = x + (long)b * K * L;
  AT* yb = y + (long)b * L * E;
  const int q = threadIdx.x % QE;
  float bl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bl[k] = bias ? bias[v * E + 4 * q + k] : 0.f;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < L * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    float e[4] = {bl[0], bl[1], bl[2], bl[3]};
    for (int c = 0; c < K; ++c) {
      const float xi = xb[c * L + tok];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = fmaf(s_w[(4 * q + k) * K + c], xi, e[k]);
    }
    float sm = (e[0] + e[1]) + (e[2] + e[3]);
    const float mean = tok_sum<QE>(sm) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = e[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    stx4<AT>(yb + tok * E + 4 * q, f32x4{(e[0] - mean) * rstd, (e[1] - mean) * rstd, (e[2] - mean) * rstd, (e[3] - mean) * rstd});
  }
}
// backward: blockIdx.z owns a chunk of KC input values (its weight-gradient accumulators stay in registers); every block
// recomputes the token's convolution over all K (the LN statistics need it) — K / KC-fold redundant FMAs on a streaming kernel
template <int E, typename AT>
__global__ __launch_bounds__(256) void patch_embed_k_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const AT* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ db, int V, long L, int K) {
  constexpr int QE = E / 4, KC = 8;
  __shared__ float s_w[E * kPatchKMax];
  const int b = blockIdx.y, v = b % V, c0 = blockIdx.z * KC;
  for (int i = threadIdx.x; i < E * K; i += 256) s_w[i] = w[(long)v * E * K + i];
  __syncthreads();
  const float* xb = x + (long)b * K * L;
  const AT* dyb = dy + (long)b * L * E;
  const int q = threadIdx.x % QE;
  float bl[4], gw[4][KC], gb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bl[k] = bias ? bias[v * E + 4 * q + k] : 0.f;
    gb[k] = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) gw[k][c] = 0.f;
  }
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < L * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const f32x4 g = ldx4<AT>(dyb + tok * E + 4 * q);
    float e[4] = {bl[0], bl[1], bl[2], bl[3]};
    for (int c = 0; c < K; ++c) {
      const float xi = xb[c * L + tok];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = fmaf(s_w[(4 * q + k) * K + c], xi, e[k]);
    }
    const float mean = tok_sum<QE>((e[0] + e[1]) + (e[2] + e[3])) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = e[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = (e[k] - mean) * rstd; s1 += g[k]; s2 = fmaf(g[k], e[k], s2); }
    s1 = tok_sum<QE>(s1) * (1.f / E); s2 = tok_sum<QE>(s2) * (1.f / E);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float de = rstd * (g[k] - s1 - e[k] * s2);
      gb[k] += de;
#pragma unroll
      for (int c = 0; c < KC; ++c) gw[k][c] = fmaf(de, c0 + c < K ? xb[(long)(c0 + c) * L + tok] : 0.f, gw[k][c]);
    }
  }
  auto same_q_sum = [](float val) {
#pragma unroll
    for (int o = QE; o < 64; o <<= 1) val += __shfl_xor(val, o, 64);
    return val;
  };
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sb = same_q_sum(gb[k]);
    if (lane < QE && db && blockIdx.z == 0) atomicAdd(db + v * E + 4 * q + k, sb);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const float sw = same_q_sum(gw[k][c]);
      if (lane < QE && c0 + c < K) atomicAdd(dw + ((long)v * E + 4 * q + k) * K + c0 + c, sw);
    }
  }
}

// -------------------------------------------------------------------------------------------
// Stage embedding of a hierarchical Swin encoder: PatchEmbed3D with patch (1,1,1) between two stages of different width
// (BasicLayer builds it when in_dim != dim, Swin_3D.py:414-419; forward :473-491): y = LN_noaffine(W x + b) per token,
// input ALREADY channels-last tokens (B, L, CIN), CIN a multiple of 16.  Backward also returns dx.
// E/4 adjacent lanes per token, 4 output channels each; the token's CIN inputs are read by all of them (broadcast loads).
// -------------------------------------------------------------------------------------------
template <int E, int CIN>
__global__ __launch_bounds__(256) void stage_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int V, long L) {
  constexpr int QE = E / 4;
  const int b = blockIdx.y, v = b % V;
  const float* wv = w + (long)v * E * CIN;
  const float* xb = x + (long)b * L * CIN;
  float* yb = y + (long)b * L * E;
  const int q = threadIdx.x % QE;
  f32x4 wl[4][CIN / 4];
  float bl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bl[k] = bias ? bias[v * E + 4 * q + k] : 0.f;
#pragma unroll
    for (int c = 0; c < CIN / 4; ++c) wl[k][c] = ld4(wv + (4 * q + k) * CIN + 4 * c);
  }
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < L * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    float e[4] = {bl[0], bl[1], bl[2], bl[3]};
#pragma unroll
    for (int c = 0; c < CIN / 4; ++c) {
      const f32x4 xi = ld4(xb + tok * CIN + 4 * c);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] += (wl[k][c][0] * xi[0] + wl[k][c][1] * xi[1]) + (wl[k][c][2] * xi[2] + wl[k][c][3] * xi[3]);
    }
    const float mean = tok_sum<QE>(e[0] + e[1] + e[2] + e[3]) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = e[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    st4(yb + tok * E + 4 * q, f32x4{(e[0] - mean) * rstd, (e[1] - mean) * rstd, (e[2] - mean) * rstd, (e[3] - mean) * rstd});
  }
}

template <int E, int CIN>
__global__ __launch_bounds__(256) void stage_embed_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ dy,
                                                              float* __restrict__ dx, float* __restrict__ dw,
                                                              float* __restrict__ db, int V, long L) {
  constexpr int QE = E / 4, CQ = CIN / 4;
  static_assert(CQ % QE == 0 || QE % CQ == 0, "lane <-> input-channel-quad map");
  const int b = blockIdx.y, v = b % V;
  const float* wv = w + (long)v * E * CIN;
  const float* xb = x + (long)b * L * CIN;
  const float* dyb = dy + (long)b * L * E;
  float* dxb = dx + (long)b * L * CIN;
  const int q = threadIdx.x % QE;
  f32x4 wl[4][CQ], gw[4][CQ];
  float bl[4], gb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bl[k] = bias ? bias[v * E + 4 * q + k] : 0.f;
    gb[k] = 0.f;
#pragma unroll
    for (int c = 0; c < CQ; ++c) { wl[k][c] = ld4(wv + (4 * q + k) * CIN + 4 * c); gw[k][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  const long total = L * QE;
  // every lane of a token group runs the same trip count (the group reductions below need all QE lanes)
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const f32x4 g = ld4(dyb + tok * E + 4 * q);
    f32x4 xi[CQ];
    float e[4] = {bl[0], bl[1], bl[2], bl[3]};
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      xi[c] = ld4(xb + tok * CIN + 4 * c);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] += (wl[k][c][0] * xi[c][0] + wl[k][c][1] * xi[c][1]) + (wl[k][c][2] * xi[c][2] + wl[k][c][3] * xi[c][3]);
    }
    const float mean = tok_sum<QE>(e[0] + e[1] + e[2] + e[3]) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = e[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = (e[k] - mean) * rstd; s1 += g[k]; s2 = fmaf(g[k], e[k], s2); }
    s1 = tok_sum<QE>(s1) * (1.f / E); s2 = tok_sum<QE>(s2) * (1.f / E);
    float de[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      de[k] = rstd * (g[k] - s1 - e[k] * s2);
      gb[k] += de[k];
#pragma unroll
      for (int c = 0; c < CQ; ++c) gw[k][c] += xi[c] * de[k];
    }
    // dx[c] = sum_e W[e][c] de[e]: this lane's 4 rows first, then across the QE lanes of the token; lane q keeps quads
    // c with c % QE == q (CQ >= QE) or lane q < CQ keeps quad q (CQ < QE)
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      f32x4 pc = wl[0][c] * de[0] + wl[1][c] * de[1] + wl[2][c] * de[2] + wl[3][c] * de[3];
      pc[0] = tok_sum<QE>(pc[0]); pc[1] = tok_sum<QE>(pc[1]); pc[2] = tok_sum<QE>(pc[2]); pc[3] = tok_sum<QE>(pc[3]);
      if (c % QE == q) st4(dxb + tok * CIN + 4 * c, pc);
    }
  }
  auto same_q_sum = [](float val) {
#pragma unroll
    for (int o = QE; o < 64; o <<= 1) val += __shfl_xor(val, o, 64);
    return val;
  };
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sb = same_q_sum(gb[k]);
    if (lane < QE && db) atomicAdd(db + v * E + 4 * q + k, sb);
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sw = same_q_sum(gw[k][c][r]);
        if (lane < QE) atomicAdd(dw + ((long)v * E + 4 * q + k) * CIN + 4 * c + r, sw);
      }
  }
}

// -------------------------------------------------------------------------------------------
// LFQ (models/codebook/LFQ.py:183-307), projections on, D = log2(codebook_size) code bits.
// tokens: z (M, E) -> z_q (M, E), indices int64 (M).  Training adds the aux-loss sums.
// sums per block: [0] sum_i H_i  [1] sum (s-q)^2  [2..2+K) sum_i p_ik
// -------------------------------------------------------------------------------------------
constexpr float kLogEps = 1e-5f;   // LFQ.py:52
constexpr float kInvTemp2 = 200.f; // 2 * inv_temperature(100): logits = -(-2 x.c) * 100   LFQ.py:238-240

template <int D>
__device__ __forceinline__ void lfq_probs(const float (&s)[D], float (&p)[1 << D]) {
  constexpr int K = 1 << D;
  float mx = -3.4e38f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) l += ((k >> (D - 1 - i)) & 1) ? s[i] : -s[i];
    p[k] = l * kInvTemp2;
    mx = fmaxf(mx, p[k]);
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { p[k] = __expf(p[k] - mx); sum += p[k]; }
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < K; ++k) p[k] *= inv;
}


// E/4 adjacent lanes per token, each owning 4 channels: every access is a 16-byte piece of a contiguous stream.
// NCB codebooks of D bits each (LFQ.py num_codebooks: the projected token is split 'b n (c d) -> b n c d', :215): DT = NCB D code
// bits per token; every codebook has its own softmax over its 2^D codes, the quantisation, straight-through path and commitment
// term are per bit.  sums per block: [0] sum over tokens and codebooks of H  [1] sum (s-q)^2  [2 + c K + k] sum_tokens p_c[k]
template <int E, int D, int NCB, typename AT>
__global__ __launch_bounds__(256) void lfq_fwd_kernel(const AT* __restrict__ z, const float* __restrict__ w_in,
                                                      const float* __restrict__ b_in, const float* __restrict__ w_out,
                                                      const float* __restrict__ b_out, AT* __restrict__ zq,
                                                      long long* __restrict__ indices, float* __restrict__ partials,
                                                      float* __restrict__ qs, long M, int training) {
  constexpr int K = 1 << D, DT = D * NCB, R = 2 + NCB * K, QE = E / 4;
  __shared__ float smem[R * 4];
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  const int q = threadIdx.x % QE;
  f32x4 wi[DT], wo[DT], bo;
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    wi[i] = ld4(w_in + i * E + 4 * q);
    wo[i] = f32x4{w_out[(4 * q) * DT + i], w_out[(4 * q + 1) * DT + i], w_out[(4 * q + 2) * DT + i], w_out[(4 * q + 3) * DT + i]};
  }
  bo = ld4(b_out + 4 * q);
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const f32x4 zi = ldx4<AT>(z + tok * E + 4 * q);
    float s[DT], qv[DT];
    long long idx = 0;
    f32x4 o = bo;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const float a = tok_sum<QE>(zi[0] * wi[i][0] + zi[1] * wi[i][1] + zi[2] * wi[i][2] + zi[3] * wi[i][3]) + b_in[i];
      s[i] = a;
      qv[i] = a > 0.f ? 1.f : -1.f;                // strict > (LFQ.py:222,234)
      idx |= (long long)(a > 0.f) << (DT - 1 - i);  // (the NCB per-codebook indices, most significant codebook first: D bits each)
      o += wo[i] * qv[i];
    }
    stx4<AT>(zq + tok * E + 4 * q, o);
    if (q == 0) {
      indices[tok] = idx;
      if (qs)
#pragma unroll
        for (int i = 0; i < DT; ++i) qs[tok * DT + i] = qv[i];
      if (training) {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
          float sc[D], p[K];
#pragma unroll
          for (int i = 0; i < D; ++i) sc[i] = s[c * D + i];
          lfq_probs<D>(sc, p);
#pragma unroll
          for (int k = 0; k < K; ++k) { acc[0] -= p[k] * __logf(fmaxf(p[k], kLogEps)); acc[2 + c * K + k] += p[k]; }
        }
#pragma unroll
        for (int i = 0; i < DT; ++i) { const float d = s[i] - qv[i]; acc[1] = fmaf(d, d, acc[1]); }
      }
    }
  }
  if (training) {
    block_sum<R>(acc, smem);
    if (threadIdx.x == 0)
#pragma unroll
      for (int r = 0; r < R; ++r) partials[(long)blockIdx.x * R + r] = acc[r];
  }
}

// stats out: [0] aux loss, [1] per-sample entropy (mean over tokens and codebooks), [2] codebook entropy (mean over codebooks of
// H(avg prob)), [3] commit, [4 + c K + k] avg prob of code k of codebook c   (LFQ.py:258-262)
template <int D, int NCB>
__global__ void lfq_finalize_kernel(const float* __restrict__ partials, int nblocks, float* __restrict__ stats, long M,
                                    float w_ent, float gamma, float w_commit) {
  constexpr int K = 1 << D, R = 2 + NCB * K;
  __shared__ double sm[R][64];
  const int t = threadIdx.x;   // 64 threads
  for (int r = 0; r < R; ++r) {
    double s = 0.0;
    #pragma unroll 8      // (independent loads in flight)
  for (int b = t; b < nblocks; b += 64) s += (double)partials[(long)b * R + r];
    sm[r][t] = s;
  }
  __syncthreads();
  if (t == 0) {
    double tot[R];
    for (int r = 0; r < R; ++r) { double s = 0.0; for (int i = 0; i < 64; ++i) s += sm[r][i]; tot[r] = s; }
    const float per_sample = (float)(tot[0] / ((double)M * NCB));
    const float commit = (float)(tot[1] / ((double)M * D * NCB));
    float hbs = 0.f;
    for (int c = 0; c < NCB; ++c) {
      float hb = 0.f;
      for (int k = 0; k < K; ++k) {
        const float a = (float)(tot[2 + c * K + k] / (double)M);
        stats[4 + c * K + k] = a;
        hb -= a * logf(fmaxf(a, kLogEps));
      }
      hbs += hb;
    }
    hbs *= 1.f / NCB;
    stats[1] = per_sample; stats[2] = hbs; stats[3] = commit;
    stats[0] = commit * w_commit + w_ent * per_sample - gamma * hbs;
  }
}

// backward.  gscal[0] = upstream gradient of the aux loss.  grads accumulate with atomics.
template <int E, int D, int NCB, bool HASG /*a dense d z_q exists*/, typename AT>
__global__ __launch_bounds__(256) void lfq_bwd_kernel(const AT* __restrict__ z, const AT* __restrict__ dzq,
                                                      const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                      const float* __restrict__ w_out, const float* __restrict__ stats,
                                                      const float* __restrict__ gscal, const float* __restrict__ dqs,
                                                      AT* __restrict__ dz, float* __restrict__ dw_in,
                                                      float* __restrict__ db_in, float* __restrict__ dw_out,
                                                      float* __restrict__ db_out, long M, float w_ent, float gamma,
                                                      float w_commit) {
  constexpr int K = 1 << D, DT = D * NCB, QE = E / 4;
  const float g = gscal[0];
  const float invM = 1.f / (float)M;
  float ub[NCB][K];   // d Hcodebook_c / d a_ck
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int k = 0; k < K; ++k) { const float a = stats[4 + c * K + k]; ub[c][k] = -(logf(fmaxf(a, kLogEps)) + (a >= kLogEps ? 1.f : 0.f)); }
  const int q = threadIdx.x % QE;
  f32x4 wi[DT], wo[DT], gwi[DT], gwo[DT], gbo = {0.f, 0.f, 0.f, 0.f};
  float gbi[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    wi[i] = ld4(w_in + i * E + 4 * q);
    wo[i] = f32x4{w_out[(4 * q) * DT + i], w_out[(4 * q + 1) * DT + i], w_out[(4 * q + 2) * DT + i], w_out[(4 * q + 3) * DT + i]};
    gwi[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gwo[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gbi[i] = 0.f;
  }
  // UN tokens per lane and trip: 2 UN sixteen-byte loads in flight before the first dependent instruction
  constexpr int UN = 8;
  const long total = M * QE, stride = (long)gridDim.x * 256;
  for (long u0 = (long)blockIdx.x * 256 + threadIdx.x; u0 < total; u0 += UN * stride) {
   long tokq[UN];
   f32x4 ziq[UN], goq[UN];
   float dqq[UN][DT];
#pragma unroll
   for (int r = 0; r < UN; ++r) {
     const long u = u0 + r * stride;
     tokq[r] = (u < total ? u : u0) / QE;
     ziq[r] = ldx4<AT>(z + tokq[r] * E + 4 * q);
     goq[r] = HASG ? ldx4<AT>(dzq + tokq[r] * E + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};   // no dense consumer of z_q: no read
#pragma unroll
     for (int i = 0; i < DT; ++i) dqq[r][i] = dqs ? dqs[tokq[r] * DT + i] : 0.f;
   }
#pragma unroll
   for (int half = 0; half < UN; ++half) {
    if (u0 + half * stride >= total) break;
    const long tok = tokq[half];
    const f32x4 zi = ziq[half], go = goq[half];
    float s[DT], qv[DT], ds[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const float a = tok_sum<QE>(zi[0] * wi[i][0] + zi[1] * wi[i][1] + zi[2] * wi[i][2] + zi[3] * wi[i][3]) + b_in[i];
      s[i] = a; qv[i] = a > 0.f ? 1.f : -1.f;
      // straight-through: d x_ste = W_out^T dz_q
      const float d = HASG ? tok_sum<QE>(go[0] * wo[i][0] + go[1] * wo[i][1] + go[2] * wo[i][2] + go[3] * wo[i][3]) : 0.f;
      ds[i] = d + dqq[half][i] + g * w_commit * 2.f * (s[i] - qv[i]) * invM * (1.f / DT);
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      float sc[D], p[K];
#pragma unroll
      for (int i = 0; i < D; ++i) sc[i] = s[c * D + i];
      lfq_probs<D>(sc, p);
      float us = 0.f, ubs = 0.f, uu[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        uu[k] = -(__logf(fmaxf(p[k], kLogEps)) + (p[k] >= kLogEps ? 1.f : 0.f));
        us = fmaf(uu[k], p[k], us);
        ubs = fmaf(ub[c][k], p[k], ubs);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        // d/dlogit_k of (w_ent * H_c(token) / (M NCB) - gamma * Hcodebook_c / NCB)
        const float dl = p[k] * (w_ent * (uu[k] - us) - gamma * (ub[c][k] - ubs)) * invM * (1.f / NCB) * g * kInvTemp2;
#pragma unroll
        for (int i = 0; i < D; ++i) ds[c * D + i] += ((k >> (D - 1 - i)) & 1) ? dl : -dl;
      }
    }
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      o += wi[i] * ds[i];
      gwi[i] += zi * ds[i];
      if (HASG) gwo[i] += go * qv[i];
      if (q == 0) gbi[i] += ds[i];
    }
    if (HASG) gbo += go;
    stx4<AT>(dz + tok * E + 4 * q, o);
   }
  }
  // lanes with equal q hold partials of the same 4 channels: reduce over them; the workgroup's 4 waves then meet in LDS and ONE
  // thread per value issues the global atomic.  (Per-wave atomics — 8,192 waves onto each of ~50 addresses — were 0.138 of this
  // kernel's 0.324 ms: same-address float atomics retire one every ~17 ns, whatever the line layout.)
  auto same_q_sum = [](float v) {
#pragma unroll
    for (int o = QE; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  constexpr int NV = DT + 2 * DT * E + E;          // [db_in: DT][dw_in: DT x E][dw_out: E x DT][db_out: E]
  __shared__ float s_fl[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* fl = s_fl[wave];
#pragma unroll
  for (int i = 0; i < DT; ++i) {
    const float bi = same_q_sum(gbi[i]);
    if (lane == 0) fl[i] = bi;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float a = same_q_sum(gwi[i][c]), b2 = same_q_sum(gwo[i][c]);
      if (lane < QE) { fl[DT + i * E + 4 * q + c] = a; fl[DT + DT * E + (4 * q + c) * DT + i] = b2; }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float a = same_q_sum(gbo[c]);
    if (lane < QE) fl[DT + 2 * DT * E + 4 * q + c] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += 256) {
    const float v = (s_fl[0][i] + s_fl[1][i]) + (s_fl[2][i] + s_fl[3][i]);
    float* dst = i < DT ? db_in + i : i < DT + DT * E ? dw_in + (i - DT) : i < DT + 2 * DT * E ? dw_out + (i - DT - DT * E) : db_out + (i - DT - 2 * DT * E);
    atomicAdd(dst, v);
  }
}

// -------------------------------------------------------------------------------------------
// LFQ WITHOUT projections (LFQ.py:98-101: dim == num_codebooks * log2(codebook_size) -> project_in / project_out are nn.Identity):
// the token's E channels ARE its code bits, D consecutive channels per codebook (D = 1, 2 or 4: a codebook lies inside one lane's 4
// channels), NCB = E / D codebooks.  z_q = sign(z) with the straight-through gradient; entropy / commitment terms as above, per
// codebook.  Block sums per lane quad q: [0] sum H, [1] sum (z - q)^2, [2 + (q (4 / D) + c) K + k] sum_tokens p.
// -------------------------------------------------------------------------------------------
template <int E, int D, typename AT>
__global__ __launch_bounds__(256) void lfq_np_fwd_kernel(const AT* __restrict__ z, AT* __restrict__ zq, long long* __restrict__ indices,
                                                         float* __restrict__ partials, long M, int training) {
  constexpr int K = 1 << D, QE = E / 4, CL = 4 / D, NCB = E / D, R = 2 + NCB * K, RL = 2 + CL * K;
  __shared__ float smem[4][QE][RL];
  float acc[RL];
#pragma unroll
  for (int r = 0; r < RL; ++r) acc[r] = 0.f;
  const int q = threadIdx.x % QE;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const f32x4 zi = ldx4<AT>(z + tok * E + 4 * q);
    f32x4 o;
    unsigned nib = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = zi[i] > 0.f ? 1.f : -1.f; nib |= (unsigned)(zi[i] > 0.f) << (3 - i); }   // strict > (LFQ.py:222,234)
    stx4<AT>(zq + tok * E + 4 * q, o);
    // the token's E bits, channel 0 most significant: every lane contributes its nibble
    unsigned long long idx = (unsigned long long)nib << (E - 4 - 4 * q);
#pragma unroll
    for (int off = 1; off < QE; off <<= 1) idx |= __shfl_xor((unsigned long long)idx, off, 64);
    if (q == 0) indices[tok] = (long long)idx;
    if (training) {
#pragma unroll
      for (int c = 0; c < CL; ++c) {
        float sc[D], p[K];
#pragma unroll
        for (int i = 0; i < D; ++i) sc[i] = zi[c * D + i];
        lfq_probs<D>(sc, p);
#pragma unroll
        for (int k = 0; k < K; ++k) { acc[0] -= p[k] * __logf(fmaxf(p[k], kLogEps)); acc[2 + c * K + k] += p[k]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d = zi[i] - o[i]; acc[1] = fmaf(d, d, acc[1]); }
    }
  }
  if (training) {
    // lanes with equal q hold the same codebooks: reduce over them inside the wave, then over the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      float v = acc[r];
#pragma unroll
      for (int o = QE; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane < QE) smem[wave][q][r] = v;
    }
    __syncthreads();
    float* out = partials + (long)blockIdx.x * R;
    for (int i = threadIdx.x; i < R; i += 256) {
      float v = 0.f;
      if (i < 2) {
        for (int w = 0; w < 4; ++w) for (int qq = 0; qq < QE; ++qq) v += smem[w][qq][i];
      } else {
        const int bin = i - 2, qq = bin / (CL * K), r = 2 + bin % (CL * K);
        for (int w = 0; w < 4; ++w) v += smem[w][qq][r];
      }
      out[i] = v;
    }
  }
}

template <int E, int D, bool HASG, typename AT>
__global__ __launch_bounds__(256) void lfq_np_bwd_kernel(const AT* __restrict__ z, const AT* __restrict__ dzq, const float* __restrict__ stats,
                                                         const float* __restrict__ gscal, const float* __restrict__ dqs,
                                                         AT* __restrict__ dz, long M, float w_ent, float gamma, float w_commit) {
  constexpr int K = 1 << D, QE = E / 4, CL = 4 / D, NCB = E / D;
  const float g = gscal[0];
  const float invM = 1.f / (float)M;
  const int q = threadIdx.x % QE;
  float ub[CL][K];
#pragma unroll
  for (int c = 0; c < CL; ++c)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float a = stats[4 + (q * CL + c) * K + k];
      ub[c][k] = -(logf(fmaxf(a, kLogEps)) + (a >= kLogEps ? 1.f : 0.f));
    }
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const f32x4 zi = ldx4<AT>(z + tok * E + 4 * q);
    f32x4 ds = HASG ? ldx4<AT>(dzq + tok * E + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};      // straight-through: d z_q / d z = 1
    if (dqs) ds += ld4(dqs + tok * E + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float qv = zi[i] > 0.f ? 1.f : -1.f;
      ds[i] += g * w_commit * 2.f * (zi[i] - qv) * invM * (1.f / E);
    }
#pragma unroll
    for (int c = 0; c < CL; ++c) {
      float sc[D], p[K];
#pragma unroll
      for (int i = 0; i < D; ++i) sc[i] = zi[c * D + i];
      lfq_probs<D>(sc, p);
      float us = 0.f, ubs = 0.f, uu[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        uu[k] = -(__logf(fmaxf(p[k], kLogEps)) + (p[k] >= kLogEps ? 1.f : 0.f));
        us = fmaf(uu[k], p[k], us);
        ubs = fmaf(ub[c][k], p[k], ubs);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float dl = p[k] * (w_ent * (uu[k] - us) - gamma * (ub[c][k] - ubs)) * invM * (1.f / NCB) * g * kInvTemp2;
#pragma unroll
        for (int i = 0; i < D; ++i) ds[c * D + i] += ((k >> (D - 1 - i)) & 1) ? dl : -dl;
      }
    }
    stx4<AT>(dz + tok * E + 4 * q, ds);
  }
}

// -------------------------------------------------------------------------------------------
// class-frequency-weighted BCE with logits (models/losses.py:64-124)
// -------------------------------------------------------------------------------------------
// counts[0..1] = torch.histc(target[valid], bins=2) with its data-derived range (min==max -> [min-1,max+1])
__global__ void bce_counts_kernel(const float* __restrict__ target, const float* __restrict__ valid, long M,
                                  float* __restrict__ counts) {
  __shared__ float smem[3 * 16];
  float v[3] = {3.4e38f, -3.4e38f, 0.f};
  // one workgroup (the two passes are dependent and M is a mask-sized array): 16-byte loads when the layout allows
  const bool vec = (M % 4 == 0) && (((size_t)target | (size_t)(valid ? valid : target)) % 16 == 0);
  if (vec) {
    // (one workgroup, two dependent passes: UN independent 16-byte loads in flight per thread — one at a time, each trip paid a
    //  full memory round trip: 50 us for 160 k elements)
    constexpr int UN = 8;
    for (long i0 = threadIdx.x; i0 < M / 4; i0 += (long)UN * blockDim.x) {
      f32x4 t4[UN], m4[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const long i = i0 + (long)u * blockDim.x;
        const bool in = i < M / 4;
        t4[u] = in ? ld4(target + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        m4[u] = in ? (valid ? ld4(valid + 4 * i) : f32x4{1.f, 1.f, 1.f, 1.f}) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (m4[u][k] != 0.f) { v[0] = fminf(v[0], t4[u][k]); v[1] = fmaxf(v[1], t4[u][k]); }
    }
  } else {
    for (long i = threadIdx.x; i < M; i += blockDim.x) {
      if (valid && valid[i] == 0.f) continue;
      v[0] = fminf(v[0], target[i]); v[1] = fmaxf(v[1], target[i]);
    }
  }
  // min / max reduce
  for (int o = 32; o > 0; o >>= 1) { v[0] = fminf(v[0], __shfl_xor(v[0], o, 64)); v[1] = fmaxf(v[1], __shfl_xor(v[1], o, 64)); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { smem[wave] = v[0]; smem[16 + wave] = v[1]; }
  __syncthreads();
  float mn = smem[0], mx = smem[16];
  for (int w = 1; w < nw; ++w) { mn = fminf(mn, smem[w]); mx = fmaxf(mx, smem[16 + w]); }
  if (mn == mx) { mn -= 1.f; mx += 1.f; }
  float c1 = 0.f, c0 = 0.f;
  auto count = [&](float t) {
    int bin = (int)((t - mn) / (mx - mn) * 2.f);
    bin = bin > 1 ? 1 : bin;
    if (bin == 1) c1 += 1.f; else c0 += 1.f;
  };
  if (vec) {
    constexpr int UN = 8;
    for (long i0 = threadIdx.x; i0 < M / 4; i0 += (long)UN * blockDim.x) {
      f32x4 t4[UN], m4[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const long i = i0 + (long)u * blockDim.x;
        const bool in = i < M / 4;
        t4[u] = in ? ld4(target + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        m4[u] = in ? (valid ? ld4(valid + 4 * i) : f32x4{1.f, 1.f, 1.f, 1.f}) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (m4[u][k] != 0.f) count(t4[u][k]);
    }
  } else {
    for (long i = threadIdx.x; i < M; i += blockDim.x) {
      if (valid && valid[i] == 0.f) continue;
      count(target[i]);
    }
  }
  c0 = wave_sum(c0); c1 = wave_sum(c1);
  __syncthreads();
  if (lane == 0) { smem[wave] = c0; smem[16 + wave] = c1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < nw; ++w) { a += smem[w]; b += smem[16 + w]; }
    counts[0] = a; counts[1] = b;
  }
}

// The same counts from many workgroups (the one-workgroup kernel above is bound by its per-element bin arithmetic on ONE CU:
// 41 us for 160 k elements): every workgroup finds the range itself — the min / max pass is two instructions per element and
// the array is L2-resident — and bins only its own slice, adding its two partial counts (integers in f32: exact and
// order-independent below 2^24, which the host checks) onto `counts`, zeroed by the host.  16-byte-loadable arrays only.
__global__ __launch_bounds__(1024) void bce_counts_mw_kernel(const float* __restrict__ target, const float* __restrict__ valid, long M,
                                                              float* __restrict__ counts) {
  __shared__ float smem[3 * 16];
  float mn = 3.4e38f, mx = -3.4e38f;
  constexpr int UN = 8;
  const long M4 = M / 4;
  for (long i0 = threadIdx.x; i0 < M4; i0 += (long)UN * blockDim.x) {
    f32x4 t4[UN], m4[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long i = i0 + (long)u * blockDim.x;
      const bool in = i < M4;
      t4[u] = ld4(target + 4 * (in ? i : i0));
      m4[u] = in ? (valid ? ld4(valid + 4 * i) : f32x4{1.f, 1.f, 1.f, 1.f}) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mn = fminf(mn, m4[u][k] != 0.f ? t4[u][k] : 3.4e38f);
        mx = fmaxf(mx, m4[u][k] != 0.f ? t4[u][k] : -3.4e38f);
      }
  }
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { smem[wave] = mn; smem[16 + wave] = mx; }
  __syncthreads();
  mn = smem[0]; mx = smem[16];
  for (int w = 1; w < nw; ++w) { mn = fminf(mn, smem[w]); mx = fmaxf(mx, smem[16 + w]); }
  if (mn == mx) { mn -= 1.f; mx += 1.f; }
  float c1 = 0.f, c0 = 0.f;
  // this workgroup's slice of the 16-byte units
  const long per = (M4 + gridDim.x - 1) / gridDim.x, lo = (long)blockIdx.x * per, hi = lo + per < M4 ? lo + per : M4;
  for (long i0 = lo + threadIdx.x; i0 < hi; i0 += (long)UN * blockDim.x) {
    f32x4 t4[UN], m4[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long i = i0 + (long)u * blockDim.x;
      const bool in = i < hi;
      t4[u] = ld4(target + 4 * (in ? i : i0));
      m4[u] = in ? (valid ? ld4(valid + 4 * i) : f32x4{1.f, 1.f, 1.f, 1.f}) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (m4[u][k] != 0.f) {
          int bin = (int)((t4[u][k] - mn) / (mx - mn) * 2.f);      // (the one-workgroup kernel's expression)
          bin = bin > 1 ? 1 : bin;
          if (bin == 1) c1 += 1.f; else c0 += 1.f;
        }
  }
  c0 = wave_sum(c0); c1 = wave_sum(c1);
  __syncthreads();
  if (lane == 0) { smem[wave] = c0; smem[16 + wave] = c1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < nw; ++w) { a += smem[w]; b += smem[16 + w]; }
    if (a != 0.f) atomicAdd(counts, a);
    if (b != 0.f) atomicAdd(counts + 1, b);
  }
}

__device__ __forceinline__ void class_weights(const float* counts, float& w0, float& w1) {
  const float tot = counts[0] + counts[1];
  w0 = logf(powf(counts[0] / tot, -0.5f) + 1.1f);   // losses.py:83-84 / :118-119
  w1 = logf(powf(counts[1] / tot, -0.5f) + 1.1f);
}

// pred (P, M), target (M): sums[p] = sum_i bce(pred[p,i], t_i) * w[t_i] (* valid_i); sums[P] = sum valid (or M)
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ valid, const float* __restrict__ counts,
                                                      float* __restrict__ partials, int P, long M) {
  __shared__ float smem[8];
  float w0, w1;
  class_weights(counts, w0, w1);
  for (int p = 0; p <= P; ++p) {
    float acc[1] = {0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256) {
      const float vm = valid ? (valid[i] != 0.f ? 1.f : 0.f) : 1.f;
      if (p == P) { acc[0] += valid ? valid[i] : 1.f; continue; }
      const float t = target[i], x = pred[(long)p * M + i];
      const float w = ((long long)t == 0 ? w0 : w1) * vm;
      const float l = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
      acc[0] += (vm != 0.f) ? l * w : 0.f;
    }
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) partials[(long)blockIdx.x * (P + 1) + p] = acc[0];
  }
}
// out[p] = sum_blocks partial[p] / sum_blocks partial[P]   (P+1 columns; 64 threads, fixed summation order)
__global__ void ratio_finalize_kernel(const float* __restrict__ partials, int nblocks, int P, float* __restrict__ out) {
  __shared__ double sm[64];
  __shared__ double den_s;
  const int t = threadIdx.x;
  for (int p = P; p >= 0; --p) {
    double s = 0.0;
    #pragma unroll 8      // (independent loads in flight)
    for (int b = t; b < nblocks; b += 64) s += (double)partials[(long)b * (P + 1) + p];
    sm[t] = s;
    __syncthreads();
    if (t == 0) {
      double tot = 0.0;
      for (int i = 0; i < 64; ++i) tot += sm[i];
      if (p == P) { den_s = tot; out[P] = (float)tot; }
      else out[p] = (float)(tot / den_s);
    }
    __syncthreads();
  }
}
// dpred[p,i] = gout[p] * w[t_i] * valid_i * (sigmoid(x) - t) / denom
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ valid, const float* __restrict__ counts,
                                                      const float* __restrict__ gout, const float* __restrict__ denom,
                                                      float* __restrict__ dpred, int P, long M) {
  float w0, w1;
  class_weights(counts, w0, w1);
  const float inv = 1.f / denom[0];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)P * M; e += (long)gridDim.x * 256) {
    const long i = e % M; const int p = (int)(e / M);
    const float t = target[i], x = pred[e];
    const bool vm = valid ? valid[i] != 0.f : true;
    const float w = ((long long)t == 0 ? w0 : w1);
    const float sg = 1.f / (1.f + __expf(-x));
    dpred[e] = vm ? gout[p] * w * (sg - t) * inv : 0.f;
  }
}

// -------------------------------------------------------------------------------------------
// masked L1 "anomaly" loss (models/losses.py:15-61, :127-168) on channels-last z_q (N,V,T,H,W,E)
// weight = 1 - min(mask_extreme[n,h,w] + mask_valid[n,t,h,w], 1); target = vq0[e]
// partial sums per block: [0] sum |z_q - vq0| * weight   [1] sum weight (over the full repeated shape)
// -------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void anomaly_fwd_kernel(const float* __restrict__ zq, const float* __restrict__ mask,
                                                          const float* __restrict__ mvalid, const float* __restrict__ vq0,
                                                          float* __restrict__ partials, int N, int V, int T, long HW) {
  constexpr int QE = E / 4;
  __shared__ float smem[8];
  const int q = threadIdx.x % QE;
  const f32x4 c = ld4(vq0 + 4 * q);
  float acc[2] = {0.f, 0.f};
  const long M = (long)N * V * T * HW;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const long hw = tok % HW; long r = tok / HW;
    const int t = (int)(r % T); r /= T;
    const int n = (int)(r / V);
    float m = mask[(long)n * HW + hw];
    if (mvalid) m = fminf(m + mvalid[((long)n * T + t) * HW + hw], 1.f);
    const float wgt = 1.f - m;
    acc[1] += wgt * 4.f;
    if (m == 1.f) continue;   // target := prediction there (losses.py:162): exactly zero contribution
    const f32x4 v = ld4(zq + tok * E + 4 * q);
    acc[0] = fmaf(fabsf(v[0] - c[0]) + fabsf(v[1] - c[1]) + fabsf(v[2] - c[2]) + fabsf(v[3] - c[3]), wgt, acc[0]);
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) { partials[(long)blockIdx.x * 2] = acc[0]; partials[(long)blockIdx.x * 2 + 1] = acc[1]; }
}
template <int E>
__global__ __launch_bounds__(256) void anomaly_bwd_kernel(const float* __restrict__ zq, const float* __restrict__ mask,
                                                          const float* __restrict__ mvalid, const float* __restrict__ vq0,
                                                          const float* __restrict__ gout, const float* __restrict__ denom,
                                                          float* __restrict__ dzq, int N, int V, int T, long HW) {
  constexpr int QE = E / 4;
  const int q = threadIdx.x % QE;
  const f32x4 c = ld4(vq0 + 4 * q);
  const float sc = gout[0] / denom[0];
  const long M = (long)N * V * T * HW;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const long hw = tok % HW; long r = tok / HW;
    const int t = (int)(r % T); r /= T;
    const int n = (int)(r / V);
    float m = mask[(long)n * HW + hw];
    if (mvalid) m = fminf(m + mvalid[((long)n * T + t) * HW + hw], 1.f);
    const float k = sc * (1.f - m);
    f32x4 v = ld4(zq + tok * E + 4 * q);
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
      const float d = v[uu] - c[uu];
      v[uu] = (m == 1.f) ? 0.f : (d > 0.f ? k : (d < 0.f ? -k : 0.f));
    }
    st4(dzq + tok * E + 4 * q, v);
  }
}


// -------------------------------------------------------------------------------------------
// The same loss on LFQ's code bits (rank-d form): z_q = b_out + sum_i q_i w_out[:, i] is recomputed per token
// from its D signs with the operations of lfq_fwd_kernel (so a code-0 token reproduces vq0 bit for bit and contributes an
// exact zero, as it does through z_q), instead of reading the 64-byte z_q row; the backward returns the gradient of the signs
// (straight-through input of lfq_bwd) and of project_out directly instead of a dense d z_q.  One thread per token.
// partials per block, forward: [0] sum |z_q - vq0| * weight  [1] sum weight * E ; backward: E * (D + 1) sums
// (d w_out[e][i], then d b_out[e]).
// -------------------------------------------------------------------------------------------
template <int E, int D>
__global__ __launch_bounds__(256) void anomaly_signs_fwd_kernel(const float* __restrict__ signs, const float* __restrict__ mask,
                                                                const float* __restrict__ mvalid, const float* __restrict__ w_out,
                                                                const float* __restrict__ b_out, const float* __restrict__ vq0,
                                                                float* __restrict__ partials, int N, int V, int T, long HW) {
  __shared__ float smem[8];
  __shared__ float s_w[E * D], s_b[E], s_c[E];
  for (int i = threadIdx.x; i < E * D; i += 256) s_w[i] = w_out[i];
  for (int i = threadIdx.x; i < E; i += 256) { s_b[i] = b_out[i]; s_c[i] = vq0[i]; }
  __syncthreads();
  float acc[2] = {0.f, 0.f};
  // 32-bit token arithmetic (the entry point checks the count): three 64-bit divisions by run-time values per token are ~450
  // instructions, more than the rest of the token's work
  const unsigned M = (unsigned)N * V * T * (unsigned)HW, HWu = (unsigned)HW;
  for (unsigned tok = blockIdx.x * 256 + threadIdx.x; tok < M; tok += gridDim.x * 256) {
    const unsigned r0 = tok / HWu, hw = tok - r0 * HWu, r1 = r0 / (unsigned)T;
    const int t = (int)(r0 - r1 * T);
    const int n = (int)(r1 / (unsigned)V);
    float m = mask[(long)n * HW + hw];
    if (mvalid) m = fminf(m + mvalid[((long)n * T + t) * HW + hw], 1.f);
    const float wgt = 1.f - m;
    acc[1] += wgt * (float)E;
    if (m == 1.f) continue;   // target := prediction there (losses.py:162): exactly zero contribution
    float q[D];
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] = signs[(long)tok * D + i];
    float l1 = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float o = s_b[e];
#pragma unroll
      for (int i = 0; i < D; ++i) o += s_w[e * D + i] * q[i];
      l1 += fabsf(o - s_c[e]);
    }
    acc[0] = fmaf(l1, wgt, acc[0]);
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) { partials[(long)blockIdx.x * 2] = acc[0]; partials[(long)blockIdx.x * 2 + 1] = acc[1]; }
}
template <int E, int D>
__global__ __launch_bounds__(256) void anomaly_signs_bwd_kernel(const float* __restrict__ signs, const float* __restrict__ mask,
                                                                const float* __restrict__ mvalid, const float* __restrict__ w_out,
                                                                const float* __restrict__ b_out, const float* __restrict__ vq0,
                                                                const float* __restrict__ gout, const float* __restrict__ denom,
                                                                float* __restrict__ dsigns, float* __restrict__ partials,
                                                                int N, int V, int T, long HW) {
  constexpr int R = E * (D + 1);
  __shared__ float s_red[4][R];
  __shared__ float s_w[E * D], s_b[E], s_c[E];
  for (int i = threadIdx.x; i < E * D; i += 256) s_w[i] = w_out[i];
  for (int i = threadIdx.x; i < E; i += 256) { s_b[i] = b_out[i]; s_c[i] = vq0[i]; }
  __syncthreads();
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  const float sc = gout[0] / denom[0];
  // 32-bit token arithmetic (the entry point checks the count): three 64-bit divisions by run-time values per token are ~450
  // instructions, more than the rest of the token's work
  const unsigned M = (unsigned)N * V * T * (unsigned)HW, HWu = (unsigned)HW;
  for (unsigned tok = blockIdx.x * 256 + threadIdx.x; tok < M; tok += gridDim.x * 256) {
    const unsigned r0 = tok / HWu, hw = tok - r0 * HWu, r1 = r0 / (unsigned)T;
    const int t = (int)(r0 - r1 * T);
    const int n = (int)(r1 / (unsigned)V);
    float m = mask[(long)n * HW + hw];
    if (mvalid) m = fminf(m + mvalid[((long)n * T + t) * HW + hw], 1.f);
    const float k = sc * (1.f - m);
    float q[D], dq[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { q[i] = signs[(long)tok * D + i]; dq[i] = 0.f; }
    if (m != 1.f) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float o = s_b[e];
#pragma unroll
        for (int i = 0; i < D; ++i) o += s_w[e * D + i] * q[i];
        const float d = o - s_c[e];
        const float g = d > 0.f ? k : (d < 0.f ? -k : 0.f);       // d |.| at 0 is 0 (torch's L1Loss)
#pragma unroll
        for (int i = 0; i < D; ++i) { dq[i] = fmaf(g, s_w[e * D + i], dq[i]); acc[e * D + i] = fmaf(g, q[i], acc[e * D + i]); }
        acc[E * D + e] += g;
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) dsigns[(long)tok * D + i] = dq[i];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float v = acc[r];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_red[wave][r] = v;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256)
    partials[(long)blockIdx.x * R + r] = (s_red[0][r] + s_red[1][r]) + (s_red[2][r] + s_red[3][r]);
}
// out[r] = sum over blocks of partials[b][r], fixed order (one 64-thread workgroup per value)
__global__ void column_sum_kernel(const float* __restrict__ partials, int nblocks, int R, float* __restrict__ out) {
  __shared__ double sm[64];
  const int r = blockIdx.x, t = threadIdx.x;
  double s = 0.0;
  #pragma unroll 8      // (independent loads in flight)
  for (int b = t; b < nblocks; b += 64) s += (double)partials[(long)b * R + r];
  sm[t] = s;
  __syncthreads();
  if (t == 0) {
    double tot = 0.0;
    for (int i = 0; i < 64; ++i) tot += sm[i];
    out[r] = (float)tot;
  }
}

// -------------------------------------------------------------------------------------------
// Adam over one flat buffer (torch.optim.Adam math: L2 decay in the gradient, eps outside the sqrt)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i] * gscale + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// -------------------------------------------------------------------------------------------
// conv_block tail (encoder/CNN_3D.py:131-142): y = shortcut + ReLU(LayerNorm_affine(c)),  c = conv output.
// tokens (B, L, E) channels-last, gamma/beta (V, E), b % V selects the set.
// -------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void ln_relu_res_fwd_kernel(const float* __restrict__ c, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ sc,
                                                              float* __restrict__ y, int V, long L,
                                                              const float* __restrict__ item_scale) {
  // E/4 adjacent lanes per token, 4 channels each (contiguous 16-byte streams in and out)
  constexpr int QE = E / 4;
  const int b = blockIdx.y, v = b % V, q = threadIdx.x % QE;
  const float ps = item_scale ? item_scale[b] : 1.f;      // DropPath factor of this batch item (encoder/CNN_3D.py:140,142)
  const f32x4 gv = ld4(gamma + v * E + 4 * q), bv = ld4(beta + v * E + 4 * q);
  const long base = (long)b * L * E;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < L * QE; u += (long)gridDim.x * 256) {
    const long off = base + (u / QE) * E + 4 * q;
    const f32x4 x = ld4(c + off), s4 = ld4(sc + off);
    const float mean = tok_sum<QE>((x[0] + x[1]) + (x[2] + x[3])) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = x[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf(fmaxf(fmaf((x[k] - mean) * rstd, gv[k], bv[k]), 0.f), ps, s4[k]);
    st4(y + off, o);
  }
}

template <int E>
__global__ __launch_bounds__(256) void ln_relu_res_bwd_kernel(const float* __restrict__ c, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ dy,
                                                              float* __restrict__ dc, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int V, long L,
                                                              const float* __restrict__ item_scale) {
  constexpr int QE = E / 4;
  const int b = blockIdx.y, v = b % V, q = threadIdx.x % QE;
  const float ps = item_scale ? item_scale[b] : 1.f;
  const f32x4 gv = ld4(gamma + v * E + 4 * q), bv = ld4(beta + v * E + 4 * q);
  const long base = (long)b * L * E;
  f32x4 gg = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < L * QE; u += (long)gridDim.x * 256) {
    const long off = base + (u / QE) * E + 4 * q;
    const f32x4 x = ld4(c + off), g = ld4(dy + off);
    const float mean = tok_sum<QE>((x[0] + x[1]) + (x[2] + x[3])) * (1.f / E);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = x[k] - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(tok_sum<QE>(var) * (1.f / E) + kLnEps);
    f32x4 xn, dxn;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xn[k] = (x[k] - mean) * rstd;
      const float du = fmaf(xn[k], gv[k], bv[k]) > 0.f ? g[k] * ps : 0.f;
      gg[k] = fmaf(du, xn[k], gg[k]);
      gb[k] += du;
      dxn[k] = du * gv[k];
      s1 += dxn[k];
      s2 = fmaf(dxn[k], xn[k], s2);
    }
    s1 = tok_sum<QE>(s1) * (1.f / E); s2 = tok_sum<QE>(s2) * (1.f / E);
    st4(dc + off, f32x4{rstd * (dxn[0] - s1 - xn[0] * s2), rstd * (dxn[1] - s1 - xn[1] * s2),
                        rstd * (dxn[2] - s1 - xn[2] * s2), rstd * (dxn[3] - s1 - xn[3] * s2)});
  }
  // lanes with equal q hold partials of the same 4 channels
  auto same_q_sum = [](float val) {
#pragma unroll
    for (int o = QE; o < 64; o <<= 1) val += __shfl_xor(val, o, 64);
    return val;
  };
  // the workgroup's 4 waves meet in LDS: one global atomic per value and workgroup (see lfq_bwd_kernel)
  __shared__ float s_fl[4][2 * E];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = same_q_sum(gg[k]), bb = same_q_sum(gb[k]);
    if (lane < QE) { s_fl[wave][4 * q + k] = a; s_fl[wave][E + 4 * q + k] = bb; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * E; i += 256) {
    const float val = (s_fl[0][i] + s_fl[1][i]) + (s_fl[2][i] + s_fl[3][i]);
    atomicAdd((i < E ? dgamma : dbeta) + v * E + (i < E ? i : i - E), val);
  }
}

// -------------------------------------------------------------------------------------------
// nn.Dropout of the classifier heads (classifier/CNN_3D.py:41, :53 — after the first ReLU): x <- x * keep-mask, IN PLACE, the
// 1 / (1 - p) factor being folded into the next convolution's weights by the caller.  Because the input is a ReLU output,
// the masked activation alone carries both masks for the backward pass (x_masked > 0 <=> ReLU active and kept).
// Counter-based generator: element i of call `seed` keeps iff hash(seed, i) >= p * 2^32 (murmur3 finaliser).
// -------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void dropout_mask_kernel(AT* __restrict__ x, long n4, unsigned thr, unsigned long long seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = ldx4<AT>(x + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned long long h = (unsigned long long)(4 * i + k) ^ (seed * 0x9E3779B97F4A7C15ull);
      h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
      if ((unsigned)h < thr) v[k] = 0.f;
    }
    stx4<AT>(x + 4 * i, v);
  }
}

// -------------------------------------------------------------------------------------------
// Euclidean nearest-neighbour vector quantisation (models/codebook/VQ.py: cdist :44-48, argmax of -dist
// in EuclideanCodebook.forward :477-555, straight-through :986, commitment mse :1056), reference defaults:
// learnable codebook, no EMA, no projections, one head.   x (M, E) tokens, embed (K, E).
// -------------------------------------------------------------------------------------------
constexpr int kVqMaxFloats = 8192;   // codebook floats kept in LDS

template <int E>
__global__ __launch_bounds__(256) void vq_nn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ embed,
                                                        float* __restrict__ q, long long* __restrict__ indices,
                                                        float* __restrict__ partials, long M, int K, int training) {
  constexpr int QE = E / 4;
  __shared__ __attribute__((aligned(16))) float s_e[kVqMaxFloats];
  __shared__ float s_e2[kVqMaxFloats / E];
  __shared__ float smem[4];
  for (int i = threadIdx.x; i < K * E; i += 256) s_e[i] = embed[i];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) {
    float a = 0.f;
    for (int e = 0; e < E; ++e) a = fmaf(s_e[k * E + e], s_e[k * E + e], a);
    s_e2[k] = a;
  }
  __syncthreads();
  const int qq = threadIdx.x % QE;
  float acc[1] = {0.f};
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const f32x4 xv = ld4(x + tok * E + 4 * qq);
    const float x2 = tok_sum<QE>(xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3]);
    float best = -3.4e38f;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      const f32x4 ev = ld4(s_e + k * E + 4 * qq);
      const float xe = tok_sum<QE>(xv[0] * ev[0] + xv[1] * ev[1] + xv[2] * ev[2] + xv[3] * ev[3]);
      const float d = -sqrtf(fmaxf(x2 + s_e2[k] - 2.f * xe, 0.f));   // -cdist
      if (d > best) { best = d; bi = k; }                               // first maximum wins (argmax)
    }
    const f32x4 ev = ld4(s_e + bi * E + 4 * qq);
    st4(q + tok * E + 4 * qq, ev);      // value of x + (q - x).detach() is q
    if (qq == 0) indices[tok] = bi;
    if (training) {
      const f32x4 d4 = ev - xv;
      acc[0] += d4[0] * d4[0] + d4[1] * d4[1] + d4[2] * d4[2] + d4[3] * d4[3];
    }
  }
  if (training) {
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) { partials[(long)blockIdx.x * 2] = acc[0]; partials[(long)blockIdx.x * 2 + 1] = 1.f; }
  }
}

// dx = dq (straight-through) + g * 2 (x - q) / (M E) ; dembed[k] += g * sum_{idx = k} 2 (q - x) / (M E)
template <int E>
__global__ __launch_bounds__(256) void vq_nn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ embed,
                                                        const long long* __restrict__ indices, const float* __restrict__ dq,
                                                        const float* __restrict__ gloss, float* __restrict__ dx,
                                                        float* __restrict__ dembed, long M, int K, float w_commit) {
  constexpr int QE = E / 4;
  __shared__ float s_g[kVqMaxFloats];
  for (int i = threadIdx.x; i < K * E; i += 256) s_g[i] = 0.f;
  __syncthreads();
  const int qq = threadIdx.x % QE;
  const float c = gloss[0] * w_commit * 2.f / ((float)M * E);
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const int k = (int)indices[tok];
    const f32x4 xv = ld4(x + tok * E + 4 * qq), ev = ld4(embed + (long)k * E + 4 * qq), gq = ld4(dq + tok * E + 4 * qq);
    const f32x4 d = (xv - ev) * c;
    st4(dx + tok * E + 4 * qq, gq + d);
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) atomicAdd(&s_g[k * E + 4 * qq + uu], -d[uu]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * E; i += 256) atomicAdd(dembed + i, s_g[i]);
}


// -------------------------------------------------------------------------------------------
// EuclideanCodebook's EMA path (models/codebook/VQ.py:519-545), k-means initialisation (:213-253, :354-377) and dead-code
// expiry (:449-475).  The per-token reductions (cluster sizes, per-code sums of the tokens) are the only heavy part:
// block-level accumulation in LDS, one atomic flush per block; everything else works on (K, E) and runs in one workgroup.
// -------------------------------------------------------------------------------------------
// counts[k] += #{tok : idx = k} ; sums[k][:] += sum_{idx = k} x[tok][:]        (embed_onehot.sum / einsum 'h n d, h n c -> h c d')
template <int E>
__global__ __launch_bounds__(256) void vq_cluster_stats_kernel(const float* __restrict__ x, const long long* __restrict__ indices,
                                                               float* __restrict__ counts, float* __restrict__ sums, long M, int K) {
  constexpr int QE = E / 4;
  __shared__ float s_s[kVqMaxFloats];
  __shared__ float s_c[kVqMaxFloats / E];
  for (int i = threadIdx.x; i < K * E; i += 256) s_s[i] = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) s_c[i] = 0.f;
  __syncthreads();
  const int qq = threadIdx.x % QE;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < M * QE; u += (long)gridDim.x * 256) {
    const long tok = u / QE;
    const int k = (int)indices[tok];
    const f32x4 xv = ld4(x + tok * E + 4 * qq);
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) atomicAdd(&s_s[k * E + 4 * qq + uu], xv[uu]);
    if (qq == 0) atomicAdd(&s_c[k], 1.f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * E; i += 256) if (s_s[i] != 0.f) atomicAdd(sums + i, s_s[i]);
  for (int i = threadIdx.x; i < K; i += 256) if (s_c[i] != 0.f) atomicAdd(counts + i, s_c[i]);
}

// ema_inplace on cluster_size and embed_avg (lerp with 1 - decay, :56-62), Laplace smoothing (:126-128), embed = embed_avg /
// smoothed size (:539-543).  One workgroup.
__global__ __launch_bounds__(256) void vq_ema_update_kernel(float* __restrict__ cluster_size, float* __restrict__ embed_avg,
                                                            float* __restrict__ embed, const float* __restrict__ new_size,
                                                            const float* __restrict__ new_sum, int K, int E, float decay, float eps) {
  __shared__ float s_red[256];
  const float w = 1.f - decay;
  float part = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float c = cluster_size[k], cs = c + w * (new_size[k] - c);        // torch.lerp: start + weight * (end - start), weight < 0.5
    cluster_size[k] = cs;
    part += cs;
  }
  s_red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o]; __syncthreads(); }
  const float total = s_red[0];
  for (int i = threadIdx.x; i < K * E; i += 256) {
    const int k = i / E;
    const float a = embed_avg[i], av = a + w * (new_sum[i] - a);
    embed_avg[i] = av;
    const float smoothed = (cluster_size[k] + eps) / (total + K * eps) * total;
    embed[i] = av / smoothed;
  }
}

// one k-means step (:231-251): means = bins == 0 ? means : sums / bins
__global__ void vq_kmeans_update_kernel(float* __restrict__ means, const float* __restrict__ bins, const float* __restrict__ sums,
                                        int K, int E) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K * E; i += gridDim.x * blockDim.x) {
    const float n = bins[i / E];
    if (n != 0.f) means[i] = sums[i] / n;
  }
}

// expire_codes_ / replace (:449-475): the r-th code whose EMA cluster size is below the threshold takes the r-th candidate
// sample; its cluster size is reset and embed_avg = sample * reset.  One workgroup (the rank is a prefix count over K).
__global__ __launch_bounds__(256) void vq_expire_kernel(float* __restrict__ embed, float* __restrict__ cluster_size,
                                                        float* __restrict__ embed_avg, const float* __restrict__ x,
                                                        const long long* __restrict__ cand, int K, int E, float threshold,
                                                        float reset, int* __restrict__ n_expired) {
  __shared__ int s_rank[kVqMaxFloats / 16];
  if (threadIdx.x == 0) {
    int r = 0;
    for (int k = 0; k < K; ++k) { const bool dead = cluster_size[k] < threshold; s_rank[k] = dead ? r : -1; r += dead; }
    if (n_expired) *n_expired = r;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * E; i += 256) {
    const int k = i / E, r = s_rank[k];
    if (r < 0) continue;
    const float v = x[cand[r] * E + (i - k * E)];
    embed[i] = v;
    embed_avg[i] = v * reset;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) if (s_rank[k] >= 0) cluster_size[k] = reset;
}


// -------------------------------------------------------------------------------------------
// Step glue (replaces ~15 tiny torch kernels per step incl. a Tensile GEMM, a cat and an index kernel):
//   lfq_code_vectors : LFQ.indices_to_codes(indices) with project_out (LFQ.py:152-181) for a handful of codes:
//                      out[i][e] = b_out[e] + sum_d (+-1)[bit d of indices[i]] w_out[e][d]
//   loss_combine     : total = mean_t bce[0,t] + lam * l_anom + sum_{p>=1} mean_t bce[p,t] + loss_z_q
//                      (train_synthetic.py:182-201 / train_ERA5_Land.py:181-198; T3 temporal slices per prediction)
// -------------------------------------------------------------------------------------------
__global__ void lfq_code_vectors_kernel(const long long* __restrict__ indices, const float* __restrict__ w_out,
                                        const float* __restrict__ b_out, float* __restrict__ out, int n, int E, int D) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * E; i += gridDim.x * blockDim.x) {
    const int e = i % E;
    const long long idx = indices[i / E];
    float a = b_out[e];
    for (int d = 0; d < D; ++d) a += ((idx >> (D - 1 - d)) & 1) ? w_out[e * D + d] : -w_out[e * D + d];
    out[i] = a;
  }
}
// out[0] total, [1] extreme, [2] anomaly, [3] var, [4] z_q
__global__ void loss_combine_fwd_kernel(const float* __restrict__ bce, int P, int T3, const float* __restrict__ l_anom, float lam,
                                        const float* __restrict__ loss_zq, float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  float ext = 0.f, var = 0.f;
  for (int t = 0; t < T3; ++t) ext += bce[t];
  ext /= T3;
  for (int p = 1; p < P; ++p) {
    float m = 0.f;
    for (int t = 0; t < T3; ++t) m += bce[p * T3 + t];
    var += m / T3;
  }
  out[1] = ext; out[2] = l_anom[0]; out[3] = var; out[4] = loss_zq[0];
  out[0] = ((ext + l_anom[0] * lam) + var) + loss_zq[0];
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, int P, int T3, float lam, float* __restrict__ dbce,
                                        float* __restrict__ d_anom, float* __restrict__ d_zq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P * T3) dbce[i] = g[0] / T3;
  if (i == 0) { d_anom[0] = g[0] * lam; d_zq[0] = g[0]; }
}

// -------------------------------------------------------------------------------------------
// Device-side bookkeeping of the training drivers (utils/utils_train.py): confusion counts of the thresholded
// prediction (evaluator :175-267, evaluator_synthetic :269-348) and the majority-vote anomaly collector (:529-554),
// so that the per-iteration D2H syncs of train_synthetic.py:207-215 become one read per epoch.
// counts[label*4 + {0 correct, 1 seen, 2 union, 3 predicted}], label in {0, 1}; counts[8] = pixels seen (valid)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void confusion_counts_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                               const float* __restrict__ valid, long M, float thr,
                                                               unsigned long long* __restrict__ counts) {
  unsigned c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256) {
    if (valid && !(valid[i] > 0.f)) continue;
    const float pr = 1.f / (1.f + expf(-logits[i]));     // torch.sigmoid, then `pred > threshold` (train_synthetic.py:209-212)
    const int pc = pr > thr ? 1 : 0;
    const float g = gt[i];
    c[8] += 1;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const bool p = pc == l, t = g == (float)l;
      c[l * 4 + 0] += p && t; c[l * 4 + 1] += t; c[l * 4 + 2] += p || t; c[l * 4 + 3] += p;
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + k, (unsigned long long)v);
  }
}

// acc[v][idx_n - dt + 1 + t][hw] += anomaly[n][v][dt - 1 - t][hw] ; cnt likewise += 1      (np.flip over time)
__global__ __launch_bounds__(256) void anomaly_vote_kernel(const long long* __restrict__ anomaly, const int* __restrict__ idx_end,
                                                           float* __restrict__ acc, float* __restrict__ cnt, int n_batch, int V,
                                                           int dt, long HW, int TT) {
  const long total = (long)n_batch * V * dt * HW;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long hw = e % HW; long r = e / HW;
    const int t = (int)(r % dt); r /= dt;
    const int v = (int)(r % V); const int n = (int)(r / V);
    const int tt = idx_end[n] - dt + 1 + t;
    if (tt < 0 || tt >= TT) continue;
    const float a = (float)anomaly[(((long)n * V + v) * dt + (dt - 1 - t)) * HW + hw];
    atomicAdd(acc + ((long)v * TT + tt) * HW + hw, a);
    atomicAdd(cnt + ((long)v * TT + tt) * HW + hw, 1.f);
  }
}
__global__ __launch_bounds__(256) void anomaly_vote_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                                                    float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float r = acc[i] / cnt[i];                       // 0/0 -> NaN stays NaN, as in the reference
    out[i] = r >= 0.5f ? 1.f : (r < 0.5f ? 0.f : r);
  }
}

// -------------------------------------------------------------------------------------------
// Rank-d classifier input (disclosed in DESIGN.md): LFQ's output is z_q = b_out + sum_i q_i w_out[:, i]
// with q in {-1,+1}^D per token, so the classifier's first (2,3,3)-stride-(2,1,1) convolutions on the V*C channels of
// z_q equal (1,3,3) convolutions on the PAIRED SIGN FIELD S (N,T/2,H,W,CP): with R = V*D + 1,
//   channel kt*R + v*D + i = q_i of variable v in frame 2*t1 + kt,   channel kt*R + V*D = 1
// (the constant's zero padding reproduces the border behaviour of the zero-padded z_q exactly), other channels 0, and
//   W_eff[co][kt*R + v*D+i][kh,kw] = sum_c W[co][v*C+c][kt,kh,kw] w_out[c][i]
//   W_eff[co][kt*R + V*D][kh,kw]   = sum_{v,c} W[co][v*C+c][kt,kh,kw] b_out[c].
// -------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void sign_field_fwd_kernel(const float* __restrict__ qs, AT* __restrict__ sp, int N, int V,
                                                             int T, long HW, int D, int CP) {
  const int T1 = T / 2, R = V * D + 1, Q = CP / 4;
  const unsigned npix = (unsigned)N * T1 * (unsigned)HW, HWu = (unsigned)HW;      // (32-bit index arithmetic; the entry point checks the count)
  for (unsigned u = blockIdx.x * 256 + threadIdx.x; u < npix * Q; u += gridDim.x * 256) {
    const unsigned pix = u / (unsigned)Q; const int q = (int)(u - pix * Q);
    const unsigned r = pix / HWu, p = pix - r * HWu;
    const long n = r / (unsigned)T1; const int t1 = (int)(r - (unsigned)n * T1);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int chp = 4 * q + k;
      const int kt = chp / R, ch = chp % R;
      float val = 0.f;
      if (kt < 2) val = ch < V * D ? qs[(((n * V + ch / D) * T + 2 * t1 + kt) * HW + p) * D + ch % D] : 1.f;
      o[k] = val;
    }
    stx4<AT>(sp + (long)pix * CP + 4 * q, o);
  }
}
// One thread per (pixel of the paired field, frame of the pair): its R - 1 = V D gradient channels are read as consecutive words
// of one pixel row (the pixel's 64 bytes are shared by the two frames' threads) and written to V D planes, consecutive pixels
// to consecutive addresses (the element-per-thread form read one word per 64-byte line: 67 us for 70 MB).
template <typename AT>
__global__ __launch_bounds__(256) void sign_field_bwd_kernel(const AT* __restrict__ dsp, float* __restrict__ dqs, int N, int V,
                                                             int T, long HW, int D, int CP) {
  const int T1 = T / 2, R = V * D + 1;
  const unsigned total = (unsigned)N * T * (unsigned)HW, HWu = (unsigned)HW;               // (n, t, pixel); 32-bit (the entry point checks the count)
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const unsigned r = e / HWu;
    const long p = e - r * HWu;
    const long n = r / (unsigned)T; const int t = (int)(r - (unsigned)n * T);
    const AT* src = dsp + ((n * T1 + t / 2) * HW + p) * CP + (t & 1) * R;
    for (int v = 0; v < V; ++v)
      for (int i = 0; i < D; ++i)
        dqs[(((n * V + v) * T + t) * HW + p) * D + i] = t < 2 * T1 ? ldx1<AT>(src + v * D + i) : 0.f;   // an odd last frame is never read
  }
}
// w (Co, V*C, 2, 9) joint | (V, Co, C, 2, 9) per_var -> weff (rows, CP, 9), rows = Co | V*Co (row v*Co+co sees only variable v)
__global__ void fold_fwd_kernel(const float* __restrict__ w, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                float* __restrict__ weff, int Co, int V, int C, int D, int CP, int per_var) {
  const int rows = per_var ? V * Co : Co, R = V * D + 1;
  const long total = (long)rows * CP * 9;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap9 = (int)(e % 9); long r = e / 9;
    const int chp = (int)(r % CP); const int row = (int)(r / CP);
    const int kt = chp / R, ch = chp % R;
    float acc = 0.f;
    if (kt < 2) {
      const int tap = kt * 9 + tap9;
      const int vr = per_var ? row / Co : -1;
      for (int v = 0; v < V; ++v) {
        if (ch < V * D && v != ch / D) continue;
        if (per_var && v != vr) continue;
        const float* wrow = per_var ? w + ((long)row * C) * 18 : w + ((long)row * V * C + (long)v * C) * 18;
#pragma unroll 8                  // (independent loads in flight: one load and its wait per trip made this a chain of C round trips)
        for (int c = 0; c < C; ++c)
          acc = fmaf(wrow[(long)c * 18 + tap], ch < V * D ? w_out[c * D + ch % D] : b_out[c], acc);
      }
    }
    weff[e] = acc;
  }
}
// dw[row][(v) c][kt, tap9] = sum_i dweff[row][kt*R + v*D+i][tap9] w_out[c][i] + dweff[row][kt*R + V*D][tap9] b_out[c]
__global__ void fold_bwd_w_kernel(const float* __restrict__ dweff, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                  float* __restrict__ dw, int Co, int V, int C, int D, int CP, int per_var) {
  const int rows = per_var ? V * Co : Co, R = V * D + 1;
  const int G = per_var ? 1 : V;
  const long total = (long)rows * G * C * 18;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % 18); long r = e / 18;
    const int c = (int)(r % C); r /= C;
    const int g = (int)(r % G); const int row = (int)(r / G);
    const int v = per_var ? row / Co : g;
    const int kt = tap / 9, tap9 = tap % 9;
    const float* dwe = dweff + ((long)row * CP + kt * R) * 9 + tap9;
    float acc = dwe[(long)(V * D) * 9] * b_out[c];
#pragma unroll 4
    for (int i = 0; i < D; ++i) acc = fmaf(dwe[(long)(v * D + i) * 9], w_out[c * D + i], acc);
    dw[e] = acc;
  }
}
// dw_out[c][i] += sum_{row, v, tap} w[row][(v) c][tap] dweff[row][kt*R + v*D+i][tap9] ; db_out[c] likewise with the constant channel
__global__ void fold_bwd_p_kernel(const float* __restrict__ dweff, const float* __restrict__ w, float* __restrict__ dw_out,
                                  float* __restrict__ db_out, int Co, int V, int C, int D, int CP, int per_var) {
  // one block per (c, i) with i == D meaning b_out
  const int c = blockIdx.x / (D + 1), i = blockIdx.x % (D + 1);
  const int rows = per_var ? V * Co : Co, R = V * D + 1;
  const int G = per_var ? 1 : V;
  float acc = 0.f;
  const long total = (long)rows * G * 18;
  // eight (w, dweff) pairs requested per trip before the first is used (one pair and its wait per trip: 40 serialized memory
  // round trips per thread at the joint head's size — `#pragma unroll` alone left the loads in program order behind their fmas)
  constexpr int UN = 8;
  for (long e0 = threadIdx.x; e0 < total; e0 += (long)UN * blockDim.x) {
    float wv[UN], dv[UN];
    bool on_[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = e0 + (long)u * blockDim.x;
      const bool on = e < total;
      on_[u] = on;
      const long ec = on ? e : e0;
      const int tap = (int)(ec % 18); long r = ec / 18;
      const int g = (int)(r % G); const int row = (int)(r / G);
      const int v = per_var ? row / Co : g;
      const int ch = i < D ? v * D + i : V * D;
      wv[u] = w[(((long)row * G + g) * C + c) * 18 + tap];
      dv[u] = dweff[((long)row * CP + (tap / 9) * R + ch) * 9 + tap % 9];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc = fmaf(on_[u] ? wv[u] : 0.f, dv[u], acc);      // (masked AFTER all loads are out)
  }
  __shared__ float sm[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = sm[0] + sm[1] + sm[2] + sm[3];
    if (i < D) atomicAdd(dw_out + c * D + i, t); else atomicAdd(db_out + c, t);
  }
}

// -------------------------------------------------------------------------------------------
// PatchMerging (Swin_3D.py:290-335; never instantiated by the reference's own configs, built for API parity):
// gather 4 neighbours -> LayerNorm(4C, affine) -> Linear(4C -> 2C, no bias).  Sampling as the reference writes it:
// x_q at (d, h, w) = (2 d2 + (q & 1), 2 h2 + (q & 1), 2 w2 + (q >> 1)) for D > 1, and with d = 0 for D == 1;
// odd extents are zero-padded.  16 lanes per output token, C/4 input channels per lane.
// -------------------------------------------------------------------------------------------
template <int C>
struct PMTok {
  float v[C / 4];     // gathered input values of this lane (zero when padded)
  long src;           // offset of the lane's first value in x, or -1
  float mean, rstd;
};
template <int C>
__device__ __forceinline__ PMTok<C> pm_gather(const float* __restrict__ x, long tok, long ntok, int D, int H, int W, int D2,
                                               int H2, int W2, int l16) {
  PMTok<C> t;
  const int q = l16 >> 2, part = l16 & 3;
  t.src = -1;
#pragma unroll
  for (int k = 0; k < C / 4; ++k) t.v[k] = 0.f;
  if (tok < ntok) {
    long r = tok;
    const int w2 = (int)(r % W2); r /= W2;
    const int h2 = (int)(r % H2); r /= H2;
    const int d2 = (int)(r % D2); const long b = r / D2;
    const int d = D == 1 ? 0 : 2 * d2 + (q & 1), h = 2 * h2 + (q & 1), w = 2 * w2 + (q >> 1);
    if (d < D && h < H && w < W) {
      t.src = ((((long)b * D + d) * H + h) * W + w) * C + part * (C / 4);
#pragma unroll
      for (int k = 0; k < C / 16; ++k) {
        const f32x4 f = ld4(x + t.src + 4 * k);
        t.v[4 * k] = f[0]; t.v[4 * k + 1] = f[1]; t.v[4 * k + 2] = f[2]; t.v[4 * k + 3] = f[3];
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < C / 4; ++k) s += t.v[k];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
  t.mean = s * (1.f / (4 * C));
  float vs = 0.f;
#pragma unroll
  for (int k = 0; k < C / 4; ++k) { const float dlt = t.v[k] - t.mean; vs = fmaf(dlt, dlt, vs); }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) vs += __shfl_xor(vs, o, 64);
  t.rstd = rsqrtf(vs * (1.f / (4 * C)) + 1e-5f);
  return t;
}

template <int C>
__global__ __launch_bounds__(256) void patch_merging_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ w,
                                                                float* __restrict__ y, long ntok, int D, int H, int W, int D2,
                                                                int H2, int W2) {
  constexpr int K = 4 * C, O = 2 * C;
  __shared__ float s_w[O][K + 1];
  __shared__ float s_xn[16][K];
  const int tid = threadIdx.x, tl = tid >> 4, l16 = tid & 15;
  for (int e = tid; e < O * K; e += 256) s_w[e / K][e % K] = w[e];
  const int i0 = (l16 >> 2) * C + (l16 & 3) * (C / 4);     // this lane's first index in the 4C vector
  for (long base = (long)blockIdx.x * 16; base < ntok; base += (long)gridDim.x * 16) {
    const long tok = base + tl;
    const PMTok<C> t = pm_gather<C>(x, tok, ntok, D, H, W, D2, H2, W2, l16);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < C / 4; ++k) s_xn[tl][i0 + k] = (t.v[k] - t.mean) * t.rstd * gamma[i0 + k] + beta[i0 + k];
    __syncthreads();
    if (tok < ntok) {
#pragma unroll
      for (int oo = 0; oo < O / 16; ++oo) {
        const int o = l16 * (O / 16) + oo;
        float acc = 0.f;
        for (int i = 0; i < K; ++i) acc = fmaf(s_xn[tl][i], s_w[o][i], acc);
        y[tok * O + o] = acc;
      }
    }
  }
}

template <int C>
__global__ __launch_bounds__(256) void patch_merging_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ w, float* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ dw, long ntok, int D, int H, int W, int D2,
                                                                int H2, int W2) {
  constexpr int K = 4 * C, O = 2 * C, WPT = O * K / 256;   // dW entries per thread
  __shared__ float s_w[O][K + 1];
  __shared__ float s_xa[16][K];     // affine-normalised inputs (the Linear's input)
  __shared__ float s_dy[16][O];
  const int tid = threadIdx.x, tl = tid >> 4, l16 = tid & 15;
  for (int e = tid; e < O * K; e += 256) s_w[e / K][e % K] = w[e];
  const int i0 = (l16 >> 2) * C + (l16 & 3) * (C / 4);
  float aw[WPT], ag[C / 4], ab[C / 4];
#pragma unroll
  for (int k = 0; k < WPT; ++k) aw[k] = 0.f;
#pragma unroll
  for (int k = 0; k < C / 4; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
  for (long base = (long)blockIdx.x * 16; base < ntok; base += (long)gridDim.x * 16) {
    const long tok = base + tl;
    const PMTok<C> t = pm_gather<C>(x, tok, ntok, D, H, W, D2, H2, W2, l16);
    __syncthreads();
    float xh[C / 4];
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
      xh[k] = (t.v[k] - t.mean) * t.rstd;
      s_xa[tl][i0 + k] = tok < ntok ? xh[k] * gamma[i0 + k] + beta[i0 + k] : 0.f;
    }
#pragma unroll
    for (int oo = 0; oo < O / 16; ++oo) s_dy[tl][l16 * (O / 16) + oo] = tok < ntok ? dy[tok * O + l16 * (O / 16) + oo] : 0.f;
    __syncthreads();
    // d(affine output)[i] = sum_o dy[o] w[o][i]
    float da[C / 4], sa = 0.f, sb = 0.f, dxh[C / 4];
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
      float acc = 0.f;
      for (int o = 0; o < O; ++o) acc = fmaf(s_dy[tl][o], s_w[o][i0 + k], acc);
      da[k] = acc;
      ag[k] = fmaf(acc, xh[k], ag[k]);
      ab[k] += acc;
      dxh[k] = acc * gamma[i0 + k];
      sa += dxh[k];
      sb = fmaf(dxh[k], xh[k], sb);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
    sa *= 1.f / K; sb *= 1.f / K;
    if (t.src >= 0)
#pragma unroll
      for (int k = 0; k < C / 4; ++k) dx[t.src + k] = t.rstd * (dxh[k] - sa - xh[k] * sb);
    // dW[o][i] += sum over the 16 tokens of dy[o] xa[i]
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int e = tid + 256 * k, o = e / K, i = e % K;
      float acc = aw[k];
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) acc = fmaf(s_dy[tt][o], s_xa[tt][i], acc);
      aw[k] = acc;
    }
  }
#pragma unroll
  for (int k = 0; k < WPT; ++k) atomicAdd(dw + tid + 256 * k, aw[k]);
#pragma unroll
  for (int k = 0; k < C / 4; ++k) { atomicAdd(dgamma + i0 + k, ag[k]); atomicAdd(dbeta + i0 + k, ab[k]); }
}

int stream_grid(long items) { return imin(cdiv(items, 256), kCUs * 8); }
// resident 256-thread workgroups per CU of a kernel (occupancy API): a grid of kCUs * that many workgroups runs in ONE
// round — a fixed "8 per CU" leaves a second, mostly empty round behind for kernels that fit fewer
template <class K>
int wg_per_cu(K kernel) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess || n < 1) n = 8;
  return n > 8 ? 8 : n;
}

}  // namespace

// ---- C-ABI ----------------------------------------------------------------------------------
extern "C" int idee_patch_embed_ln_fwd(const float* x, const float* w, const float* b, void* y, int N, int V, int Cin,
                                       long L, int E, int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && w && y && N > 0 && V > 0 && L > 0, "patch_embed_ln_fwd: bad argument");
  const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * 8, (long)N * V)), N * V);
#define PE(E_, C_)                                                                                                              \
  do {                                                                                                                          \
    if (act_dtype == IDEE_DTYPE_BF16)                                                                                           \
      hipLaunchKernelGGL((patch_embed_fwd_kernel<E_, C_, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (bf16_t*)y, V, L); \
    else hipLaunchKernelGGL((patch_embed_fwd_kernel<E_, C_, float>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (float*)y, V, L); \
  } while (0)
  if (E == 16 && Cin == 1) PE(16, 1); else if (E == 16 && Cin == 2) PE(16, 2);
  else if (E == 32 && Cin == 1) PE(32, 1); else if (E == 32 && Cin == 2) PE(32, 2);
  else if ((E == 16 || E == 32) && Cin <= kPatchKMax) {      // Cin = Cin pt ph pw values per token of a gathered patch input
#define PEK(E_, AT_) hipLaunchKernelGGL((patch_embed_k_fwd_kernel<E_, AT_>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (AT_*)y, V, L, Cin)
    if (act_dtype == IDEE_DTYPE_BF16) { if (E == 16) PEK(16, bf16_t); else PEK(32, bf16_t); }
    else { if (E == 16) PEK(16, float); else PEK(32, float); }
#undef PEK
  }
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "patch_embed_ln: (E=%d, Cin=%d) not built", E, Cin);
#undef PE
  IDEE_LAUNCH_CHECK("patch_embed_ln_fwd");
  return IDEE_OK;
}

extern "C" int idee_patch_embed_ln_bwd(const float* x, const float* w, const float* b, const void* dy, float* dw, float* db,
                                       int N, int V, int Cin, long L, int E, int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && w && dy && dw && N > 0 && V > 0 && L > 0, "patch_embed_ln_bwd: bad argument");
#define PE(E_, C_)                                                                                                  \
  do {                                                                                                              \
    const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * wg_per_cu(patch_embed_bwd_kernel<E_, C_, float>), (long)N * V)), N * V); \
    if (act_dtype == IDEE_DTYPE_BF16)                                                                               \
      hipLaunchKernelGGL((patch_embed_bwd_kernel<E_, C_, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (const bf16_t*)dy, dw, db, V, L); \
    else hipLaunchKernelGGL((patch_embed_bwd_kernel<E_, C_, float>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (const float*)dy, dw, db, V, L); \
  } while (0)
  if (E == 16 && Cin == 1) PE(16, 1); else if (E == 16 && Cin == 2) PE(16, 2);
  else if (E == 32 && Cin == 1) PE(32, 1); else if (E == 32 && Cin == 2) PE(32, 2);
  else if ((E == 16 || E == 32) && Cin <= kPatchKMax) {
    const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * 4, (long)N * V)), N * V, cdiv(Cin, 8));
#define PEK(E_, AT_) hipLaunchKernelGGL((patch_embed_k_bwd_kernel<E_, AT_>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, (const AT_*)dy, dw, db, V, L, Cin)
    if (act_dtype == IDEE_DTYPE_BF16) { if (E == 16) PEK(16, bf16_t); else PEK(32, bf16_t); }
    else { if (E == 16) PEK(16, float); else PEK(32, float); }
#undef PEK
  }
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "patch_embed_ln: (E=%d, Cin=%d) not built", E, Cin);
#undef PE
  IDEE_LAUNCH_CHECK("patch_embed_ln_bwd");
  return IDEE_OK;
}


#define SE_DISPATCH(CALL)                                                                      \
  if (E == 32 && Cin == 16) { CALL(32, 16); } else if (E == 16 && Cin == 32) { CALL(16, 32); } \
  else if (E == 16 && Cin == 16) { CALL(16, 16); } else if (E == 32 && Cin == 32) { CALL(32, 32); } \
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "stage_embed_ln: (E=%d, Cin=%d) not built (widths 16 and 32)", E, Cin);

extern "C" int idee_stage_embed_ln_fwd(const float* x, const float* w, const float* b, float* y, int B, int V, int Cin, long L,
                                       int E, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && w && y && B > 0 && V > 0 && L > 0 && B % V == 0, "stage_embed_ln_fwd: bad argument");
  const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * 8, (long)B)), B);
#define SE(E_, C_) hipLaunchKernelGGL((stage_embed_fwd_kernel<E_, C_>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, y, V, L)
  SE_DISPATCH(SE)
#undef SE
  IDEE_LAUNCH_CHECK("stage_embed_ln_fwd");
  return IDEE_OK;
}

extern "C" int idee_stage_embed_ln_bwd(const float* x, const float* w, const float* b, const float* dy, float* dx, float* dw,
                                       float* db, int B, int V, int Cin, long L, int E, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && w && dy && dx && dw && B > 0 && V > 0 && L > 0 && B % V == 0, "stage_embed_ln_bwd: bad argument");
  // whole token groups per trip: the grid stride (gridDim.x * 256 lanes) is a multiple of E/4
  const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * 4, (long)B)), B);
#define SE(E_, C_) hipLaunchKernelGGL((stage_embed_bwd_kernel<E_, C_>), grid, dim3(256), 0, (hipStream_t)stream, x, w, b, dy, dx, dw, db, V, L)
  SE_DISPATCH(SE)
#undef SE
  IDEE_LAUNCH_CHECK("stage_embed_ln_bwd");
  return IDEE_OK;
}

static int log2i(int k) { int d = 0; while ((1 << d) < k) ++d; return d; }

extern "C" long idee_lfq_workspace_floats(long M, int codebook_size, int num_codebooks) {
  return (long)stream_grid(M * 4) * (2 + (long)(num_codebooks > 0 ? num_codebooks : 1) * codebook_size);
}

// (dim E, D bits per codebook, NCB codebooks): what models/build.py constructs (one codebook of 2 .. 8 codes) and the
// several-codebook forms of LFQ.py:92-101 with up to 4 code bits per token
#define LFQ_DISPATCH(CALL)                                                                             \
  if (E == 16 && D == 1 && NCB == 1) { CALL(16, 1, 1); } else if (E == 16 && D == 2 && NCB == 1) { CALL(16, 2, 1); }   \
  else if (E == 16 && D == 3 && NCB == 1) { CALL(16, 3, 1); } else if (E == 32 && D == 1 && NCB == 1) { CALL(32, 1, 1); } \
  else if (E == 32 && D == 2 && NCB == 1) { CALL(32, 2, 1); }                                          \
  else if (E == 16 && D == 1 && NCB == 2) { CALL(16, 1, 2); } else if (E == 16 && D == 2 && NCB == 2) { CALL(16, 2, 2); } \
  else if (E == 16 && D == 1 && NCB == 4) { CALL(16, 1, 4); } else if (E == 32 && D == 1 && NCB == 2) { CALL(32, 1, 2); } \
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "lfq: (dim=%d, codebook_size=%d, num_codebooks=%d) not built", E, 1 << D, NCB);

extern "C" int idee_lfq_fwd(const void* z, const float* w_in, const float* b_in, const float* w_out, const float* b_out,
                            void* zq, int64_t* indices, float* stats, float* workspace, float* signs, long M, int E,
                            int codebook_size, int num_codebooks, int training, float w_entropy, float gamma, float w_commit,
                            int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(z && w_in && b_in && w_out && b_out && zq && indices && M > 0, "lfq_fwd: bad argument");
  IDEE_CHECK_ARG(!training || (stats && workspace), "lfq_fwd: training needs stats and workspace");
  const int D = log2i(codebook_size), NCB = num_codebooks;
  IDEE_CHECK_ARG((1 << D) == codebook_size && D >= 1, "lfq: codebook size %d is not a power of two", codebook_size);
  IDEE_CHECK_ARG(NCB >= 1, "lfq: num_codebooks %d", NCB);
  const int nb = stream_grid(M * 4);
  hipStream_t st = (hipStream_t)stream;
#define CALL(E_, D_, C_)                                                                                       \
  if (act_dtype == IDEE_DTYPE_BF16)                                                                            \
    hipLaunchKernelGGL((lfq_fwd_kernel<E_, D_, C_, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)z, w_in, b_in, w_out, b_out, \
                       (bf16_t*)zq, (long long*)indices, workspace, signs, M, training);                       \
  else hipLaunchKernelGGL((lfq_fwd_kernel<E_, D_, C_, float>), dim3(nb), dim3(256), 0, st, (const float*)z, w_in, b_in, w_out, b_out, \
                     (float*)zq, (long long*)indices, workspace, signs, M, training);                          \
  if (training) hipLaunchKernelGGL((lfq_finalize_kernel<D_, C_>), dim3(1), dim3(64), 0, st, workspace, nb, stats, M, \
                                   w_entropy, gamma, w_commit)
  LFQ_DISPATCH(CALL)
#undef CALL
  IDEE_LAUNCH_CHECK("lfq_fwd");
  return IDEE_OK;
}

extern "C" int idee_lfq_bwd(const void* z, const void* dzq, const float* w_in, const float* b_in, const float* w_out,
                            const float* stats, const float* gloss, const float* dsigns, void* dz, float* dw_in, float* db_in,
                            float* dw_out, float* db_out, long M, int E, int codebook_size, int num_codebooks, float w_entropy,
                            float gamma, float w_commit, int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(z && w_in && b_in && w_out && stats && gloss && dz && dw_in && db_in && dw_out && db_out && M > 0,
                 "lfq_bwd: bad argument");     // dzq may be null: z_q had no dense consumer (gradients arrive through dsigns)
  const int D = log2i(codebook_size), NCB = num_codebooks;
  IDEE_CHECK_ARG((1 << D) == codebook_size && D >= 1, "lfq: codebook size %d is not a power of two", codebook_size);
  IDEE_CHECK_ARG(NCB >= 1, "lfq: num_codebooks %d", NCB);
#define CALL2(E_, D_, C_, AT_)                                                                                     \
  if (dzq) hipLaunchKernelGGL((lfq_bwd_kernel<E_, D_, C_, true, AT_>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const AT_*)z, (const AT_*)dzq, w_in, b_in, \
                     w_out, stats, gloss, dsigns, (AT_*)dz, dw_in, db_in, dw_out, db_out, M, w_entropy, gamma, w_commit);  \
  else hipLaunchKernelGGL((lfq_bwd_kernel<E_, D_, C_, false, AT_>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const AT_*)z, (const AT_*)dzq, w_in, b_in,    \
                     w_out, stats, gloss, dsigns, (AT_*)dz, dw_in, db_in, dw_out, db_out, M, w_entropy, gamma, w_commit)
#define CALL(E_, D_, C_)                                                                                           \
  const int nb = imin(cdiv(M * 4, 256), kCUs * wg_per_cu(lfq_bwd_kernel<E_, D_, C_, true, float>));                \
  if (act_dtype == IDEE_DTYPE_BF16) { CALL2(E_, D_, C_, bf16_t); } else { CALL2(E_, D_, C_, float); }
  LFQ_DISPATCH(CALL)
#undef CALL
#undef CALL2
  IDEE_LAUNCH_CHECK("lfq_bwd");
  return IDEE_OK;
}

// ---- LFQ without projections ----
#define LFQNP_DISPATCH(CALL)                                                                           \
  if (E == 16 && D == 1) { CALL(16, 1); } else if (E == 16 && D == 2) { CALL(16, 2); }                   \
  else if (E == 16 && D == 4) { CALL(16, 4); } else if (E == 32 && D == 1) { CALL(32, 1); }              \
  else if (E == 32 && D == 2) { CALL(32, 2); } else if (E == 32 && D == 4) { CALL(32, 4); }              \
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "lfq (no projections): (dim=%d, codebook_size=%d) not built", E, 1 << D);

extern "C" long idee_lfq_noproj_workspace_floats(long M, int E, int codebook_size) {
  const int D = log2i(codebook_size);
  return (long)stream_grid(M * 4) * (2 + (long)(E / (D > 0 ? D : 1)) * codebook_size);
}

extern "C" int idee_lfq_noproj_fwd(const void* z, void* zq, int64_t* indices, float* stats, float* workspace, long M, int E,
                                   int codebook_size, int training, float w_entropy, float gamma, float w_commit, int act_dtype,
                                   idee_stream_t stream) {
  IDEE_CHECK_ARG(z && zq && indices && M > 0, "lfq_noproj_fwd: bad argument");
  IDEE_CHECK_ARG(!training || (stats && workspace), "lfq_noproj_fwd: training needs stats and workspace");
  const int D = log2i(codebook_size);
  IDEE_CHECK_ARG((1 << D) == codebook_size && D >= 1, "lfq: codebook size %d is not a power of two", codebook_size);
  const int nb = stream_grid(M * 4);
  hipStream_t st = (hipStream_t)stream;
#define CALL(E_, D_)                                                                                           \
  if (act_dtype == IDEE_DTYPE_BF16)                                                                            \
    hipLaunchKernelGGL((lfq_np_fwd_kernel<E_, D_, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)z, (bf16_t*)zq, \
                       (long long*)indices, workspace, M, training);                                           \
  else hipLaunchKernelGGL((lfq_np_fwd_kernel<E_, D_, float>), dim3(nb), dim3(256), 0, st, (const float*)z, (float*)zq, \
                          (long long*)indices, workspace, M, training);                                        \
  if (training) hipLaunchKernelGGL((lfq_finalize_kernel<D_, E_ / D_>), dim3(1), dim3(64), 0, st, workspace, nb, stats, M, \
                                   w_entropy, gamma, w_commit)
  LFQNP_DISPATCH(CALL)
#undef CALL
  IDEE_LAUNCH_CHECK("lfq_noproj_fwd");
  return IDEE_OK;
}

extern "C" int idee_lfq_noproj_bwd(const void* z, const void* dzq, const float* stats, const float* gloss, const float* dsigns,
                                   void* dz, long M, int E, int codebook_size, float w_entropy, float gamma, float w_commit,
                                   int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(z && stats && gloss && dz && M > 0, "lfq_noproj_bwd: bad argument");
  const int D = log2i(codebook_size);
  IDEE_CHECK_ARG((1 << D) == codebook_size && D >= 1, "lfq: codebook size %d is not a power of two", codebook_size);
  const int nb = stream_grid(M * 4);
  hipStream_t st = (hipStream_t)stream;
#define CALL2(E_, D_, AT_)                                                                                     \
  if (dzq) hipLaunchKernelGGL((lfq_np_bwd_kernel<E_, D_, true, AT_>), dim3(nb), dim3(256), 0, st, (const AT_*)z, (const AT_*)dzq, stats, \
                              gloss, dsigns, (AT_*)dz, M, w_entropy, gamma, w_commit);                         \
  else hipLaunchKernelGGL((lfq_np_bwd_kernel<E_, D_, false, AT_>), dim3(nb), dim3(256), 0, st, (const AT_*)z, (const AT_*)dzq, stats, \
                          gloss, dsigns, (AT_*)dz, M, w_entropy, gamma, w_commit)
#define CALL(E_, D_) if (act_dtype == IDEE_DTYPE_BF16) { CALL2(E_, D_, bf16_t); } else { CALL2(E_, D_, float); }
  LFQNP_DISPATCH(CALL)
#undef CALL
#undef CALL2
  IDEE_LAUNCH_CHECK("lfq_noproj_bwd");
  return IDEE_OK;
}

extern "C" int idee_bce_class_counts(const float* target, const float* valid, long M, float* counts, idee_stream_t stream) {
  IDEE_CHECK_ARG(target && counts && M > 0, "bce_class_counts: bad argument");
  const bool vec = (M % 4 == 0) && (((size_t)target | (size_t)(valid ? valid : target)) % 16 == 0);
  if (vec && M >= 32768 && M < (1L << 24)) {      // (counts stay exact integers in f32, whatever the order of the partial sums)
    const int nwg = imin(64, (int)cdiv(M / 4, 2048));
    hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return idee::fail(IDEE_ERR_LAUNCH, "bce_class_counts: hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(bce_counts_mw_kernel, dim3(nwg), dim3(1024), 0, (hipStream_t)stream, target, valid, M, counts);
  } else {
    hipLaunchKernelGGL(bce_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, target, valid, M, counts);
  }
  IDEE_LAUNCH_CHECK("bce_class_counts");
  return IDEE_OK;
}

extern "C" long idee_bce_workspace_floats(int P, long M) { return (long)imin(cdiv(M, 256), 256) * (P + 1); }

extern "C" int idee_bce_weighted_fwd(const float* pred, const float* target, const float* valid, const float* counts,
                                     float* loss, float* workspace, int P, long M, idee_stream_t stream) {
  IDEE_CHECK_ARG(pred && target && counts && loss && workspace && P > 0 && M > 0, "bce_weighted_fwd: bad argument");
  const int nb = imin(cdiv(M, 256), 256);
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, valid, counts, workspace, P, M);
  hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, nb, P, loss);
  IDEE_LAUNCH_CHECK("bce_weighted_fwd");
  return IDEE_OK;
}

extern "C" int idee_bce_weighted_bwd(const float* pred, const float* target, const float* valid, const float* counts,
                                     const float* gout, const float* denom, float* dpred, int P, long M, idee_stream_t stream) {
  IDEE_CHECK_ARG(pred && target && counts && gout && denom && dpred && P > 0 && M > 0, "bce_weighted_bwd: bad argument");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(stream_grid((long)P * M)), dim3(256), 0, (hipStream_t)stream, pred, target, valid,
                     counts, gout, denom, dpred, P, M);
  IDEE_LAUNCH_CHECK("bce_weighted_bwd");
  return IDEE_OK;
}

extern "C" long idee_anomaly_workspace_floats(long tokens) { return (long)stream_grid(tokens * 4) * 2; }

extern "C" int idee_anomaly_l1_fwd(const float* zq, const float* mask_extreme, const float* mask_valid, const float* vq0,
                                   float* out /* [loss, denom] */, float* workspace, int N, int V, int T, long HW, int E,
                                   idee_stream_t stream) {
  IDEE_CHECK_ARG(zq && mask_extreme && vq0 && out && workspace && N > 0 && V > 0 && T > 0 && HW > 0, "anomaly_l1_fwd: bad argument");
  const long M = (long)N * V * T * HW;
  const int nb = stream_grid(M * 4);
  hipStream_t st = (hipStream_t)stream;
  if (E == 16) hipLaunchKernelGGL((anomaly_fwd_kernel<16>), dim3(nb), dim3(256), 0, st, zq, mask_extreme, mask_valid, vq0, workspace, N, V, T, HW);
  else if (E == 32) hipLaunchKernelGGL((anomaly_fwd_kernel<32>), dim3(nb), dim3(256), 0, st, zq, mask_extreme, mask_valid, vq0, workspace, N, V, T, HW);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "anomaly_l1: E=%d not built", E);
  hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, nb, 1, out);
  IDEE_LAUNCH_CHECK("anomaly_l1_fwd");
  return IDEE_OK;
}

extern "C" int idee_anomaly_l1_bwd(const float* zq, const float* mask_extreme, const float* mask_valid, const float* vq0,
                                   const float* gout, const float* denom, float* dzq, int N, int V, int T, long HW, int E,
                                   idee_stream_t stream) {
  IDEE_CHECK_ARG(zq && mask_extreme && vq0 && gout && denom && dzq && N > 0 && V > 0 && T > 0 && HW > 0, "anomaly_l1_bwd: bad argument");
  const long M = (long)N * V * T * HW;
  const int nb = stream_grid(M * 4);
  hipStream_t st = (hipStream_t)stream;
  if (E == 16) hipLaunchKernelGGL((anomaly_bwd_kernel<16>), dim3(nb), dim3(256), 0, st, zq, mask_extreme, mask_valid, vq0, gout, denom, dzq, N, V, T, HW);
  else if (E == 32) hipLaunchKernelGGL((anomaly_bwd_kernel<32>), dim3(nb), dim3(256), 0, st, zq, mask_extreme, mask_valid, vq0, gout, denom, dzq, N, V, T, HW);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "anomaly_l1: E=%d not built", E);
  IDEE_LAUNCH_CHECK("anomaly_l1_bwd");
  return IDEE_OK;
}


// rank-d form of the anomaly loss: reads the LFQ code bits (signs (N,V,T,H,W,D), as idee_lfq_fwd writes them) instead of z_q
static int anomaly_signs_blocks(long tokens) { return imin(cdiv(tokens, 256), kCUs * 8); }
extern "C" long idee_anomaly_signs_workspace_floats(long tokens, int E, int codebook_size) {
  return (long)anomaly_signs_blocks(tokens) * imax(2, E * (log2i(codebook_size) + 1));
}
#define ANOM_SIGNS_DISPATCH(CALL)                                                                        \
  if (E == 16 && D == 1) { CALL(16, 1); } else if (E == 16 && D == 2) { CALL(16, 2); }                   \
  else if (E == 16 && D == 3) { CALL(16, 3); } else if (E == 32 && D == 1) { CALL(32, 1); }              \
  else if (E == 32 && D == 2) { CALL(32, 2); }                                                           \
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "anomaly_l1_signs: (dim=%d, codebook_size=%d) not built", E, 1 << D);

extern "C" int idee_anomaly_l1_signs_fwd(const float* signs, const float* mask_extreme, const float* mask_valid,
                                         const float* w_out, const float* b_out, const float* vq0, float* out /* [loss, denom] */,
                                         float* workspace, int N, int V, int T, long HW, int E, int codebook_size,
                                         idee_stream_t stream) {
  IDEE_CHECK_ARG(signs && mask_extreme && w_out && b_out && vq0 && out && workspace && N > 0 && V > 0 && T > 0 && HW > 0,
                 "anomaly_l1_signs_fwd: bad argument");
  const int D = log2i(codebook_size);
  IDEE_CHECK_ARG((1 << D) == codebook_size && D >= 1, "anomaly_l1_signs: codebook size %d is not a power of two", codebook_size);
  const long M = (long)N * V * T * HW;
  IDEE_CHECK_ARG(M + 256L * 4096 < (1L << 31), "anomaly_l1_signs_fwd: more than 2^31 tokens");      // 32-bit token index in the kernel
  const int nb = anomaly_signs_blocks(M);
  hipStream_t st = (hipStream_t)stream;
#define CALL(E_, D_)                                                                                            \
  hipLaunchKernelGGL((anomaly_signs_fwd_kernel<E_, D_>), dim3(nb), dim3(256), 0, st, signs, mask_extreme, mask_valid, w_out, \
                     b_out, vq0, workspace, N, V, T, HW)
  ANOM_SIGNS_DISPATCH(CALL)
#undef CALL
  hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, nb, 1, out);
  IDEE_LAUNCH_CHECK("anomaly_l1_signs_fwd");
  return IDEE_OK;
}

extern "C" int idee_anomaly_l1_signs_bwd(const float* signs, const float* mask_extreme, const float* mask_valid,
                                         const float* w_out, const float* b_out, const float* vq0, const float* gout,
                                         const float* denom, float* dsigns, float* dw_out, float* db_out, float* workspace,
                                         int N, int V, int T, long HW, int E, int codebook_size, idee_stream_t stream) {
  IDEE_CHECK_ARG(signs && mask_extreme && w_out && b_out && vq0 && gout && denom && dsigns && dw_out && db_out && workspace &&
                 N > 0 && V > 0 && T > 0 && HW > 0, "anomaly_l1_signs_bwd: bad argument");
  IDEE_CHECK_ARG(db_out == dw_out + (long)E * log2i(codebook_size), "anomaly_l1_signs_bwd: db_out must follow dw_out (one buffer of E (D + 1) floats)");
  const int D = log2i(codebook_size);
  IDEE_CHECK_ARG((1 << D) == codebook_size && D >= 1, "anomaly_l1_signs: codebook size %d is not a power of two", codebook_size);
  const long M = (long)N * V * T * HW;
  IDEE_CHECK_ARG(M + 256L * 4096 < (1L << 31), "anomaly_l1_signs_bwd: more than 2^31 tokens");      // 32-bit token index in the kernel
  const int nb = anomaly_signs_blocks(M);
  hipStream_t st = (hipStream_t)stream;
#define CALL(E_, D_)                                                                                            \
  hipLaunchKernelGGL((anomaly_signs_bwd_kernel<E_, D_>), dim3(nb), dim3(256), 0, st, signs, mask_extreme, mask_valid, w_out, \
                     b_out, vq0, gout, denom, dsigns, workspace, N, V, T, HW)
  ANOM_SIGNS_DISPATCH(CALL)
#undef CALL
  hipLaunchKernelGGL(column_sum_kernel, dim3(E * (D + 1)), dim3(64), 0, st, workspace, nb, E * (D + 1), dw_out);
  IDEE_LAUNCH_CHECK("anomaly_l1_signs_bwd");
  return IDEE_OK;
}

extern "C" int idee_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, float grad_scale, idee_stream_t stream) {
  IDEE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad argument");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2s, grad_scale);
  IDEE_LAUNCH_CHECK("adam_step");
  return IDEE_OK;
}

extern "C" int idee_ln_relu_res_fwd(const float* c, const float* gamma, const float* beta, const float* shortcut, float* y,
                                    int B, int V, long L, int E, const float* item_scale, idee_stream_t stream) {
  IDEE_CHECK_ARG(c && gamma && beta && shortcut && y && B > 0 && V > 0 && L > 0, "ln_relu_res_fwd: bad argument");
  const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * 8, B)), B);
  if (E == 16) hipLaunchKernelGGL((ln_relu_res_fwd_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, c, gamma, beta, shortcut, y, V, L, item_scale);
  else if (E == 32) hipLaunchKernelGGL((ln_relu_res_fwd_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, c, gamma, beta, shortcut, y, V, L, item_scale);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "ln_relu_res: E=%d not built", E);
  IDEE_LAUNCH_CHECK("ln_relu_res_fwd");
  return IDEE_OK;
}

extern "C" int idee_ln_relu_res_bwd(const float* c, const float* gamma, const float* beta, const float* dy, float* dc,
                                    float* dgamma, float* dbeta, int B, int V, long L, int E, const float* item_scale,
                                    idee_stream_t stream) {
  IDEE_CHECK_ARG(c && gamma && beta && dy && dc && dgamma && dbeta && B > 0 && V > 0 && L > 0, "ln_relu_res_bwd: bad argument");
  const int per_cu = E == 16 ? wg_per_cu(ln_relu_res_bwd_kernel<16>) : wg_per_cu(ln_relu_res_bwd_kernel<32>);
  const dim3 grid(imin(cdiv(L * (E / 4), 256), cdiv((long)kCUs * per_cu, B)), B);
  if (E == 16) hipLaunchKernelGGL((ln_relu_res_bwd_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, c, gamma, beta, dy, dc, dgamma, dbeta, V, L, item_scale);
  else if (E == 32) hipLaunchKernelGGL((ln_relu_res_bwd_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, c, gamma, beta, dy, dc, dgamma, dbeta, V, L, item_scale);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "ln_relu_res: E=%d not built", E);
  IDEE_LAUNCH_CHECK("ln_relu_res_bwd");
  return IDEE_OK;
}


extern "C" int idee_dropout_mask_(void* x, long n, float p, uint64_t seed, int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && n > 0 && n % 4 == 0 && p >= 0.f && p < 1.f, "dropout_mask_: bad argument (n must be a multiple of 4, 0 <= p < 1)");
  const unsigned thr = (unsigned)((double)p * 4294967296.0);
  if (act_dtype == IDEE_DTYPE_BF16)
    hipLaunchKernelGGL(dropout_mask_kernel<bf16_t>, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n / 4, thr,
                       (unsigned long long)seed);
  else
    hipLaunchKernelGGL(dropout_mask_kernel<float>, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (float*)x, n / 4, thr,
                     (unsigned long long)seed);
  IDEE_LAUNCH_CHECK("dropout_mask_");
  return IDEE_OK;
}

extern "C" long idee_vq_workspace_floats(long M) { return (long)stream_grid(M * 4) * 2; }

extern "C" int idee_vq_nn_fwd(const float* x, const float* embed, float* q, int64_t* indices, float* loss, float* workspace,
                              long M, int E, int K, int training, float w_commit, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && embed && q && indices && M > 0 && K > 0, "vq_nn_fwd: bad argument");
  IDEE_CHECK_ARG(!training || (loss && workspace), "vq_nn_fwd: training needs loss and workspace");
  if ((long)K * E > kVqMaxFloats) return idee::fail(IDEE_ERR_UNSUPPORTED, "vq_nn: codebook %d x %d exceeds %d floats", K, E, kVqMaxFloats);
  const int nb = stream_grid(M * 4);
  hipStream_t st = (hipStream_t)stream;
  if (E == 16) hipLaunchKernelGGL((vq_nn_fwd_kernel<16>), dim3(nb), dim3(256), 0, st, x, embed, q, (long long*)indices, workspace, M, K, training);
  else if (E == 32) hipLaunchKernelGGL((vq_nn_fwd_kernel<32>), dim3(nb), dim3(256), 0, st, x, embed, q, (long long*)indices, workspace, M, K, training);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "vq_nn: dim %d not built", E);
  if (training) {
    // loss[0] = w_commit * sum / (M * E): reuse the ratio finaliser with denominator column = 1 per block, then scale
    hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, nb, 1, loss);
  }
  IDEE_LAUNCH_CHECK("vq_nn_fwd");
  return IDEE_OK;
}

extern "C" int idee_vq_nn_bwd(const float* x, const float* embed, const int64_t* indices, const float* dq, const float* gloss,
                              float* dx, float* dembed, long M, int E, int K, float w_commit, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && embed && indices && dq && gloss && dx && dembed && M > 0 && K > 0, "vq_nn_bwd: bad argument");
  if ((long)K * E > kVqMaxFloats) return idee::fail(IDEE_ERR_UNSUPPORTED, "vq_nn: codebook %d x %d exceeds %d floats", K, E, kVqMaxFloats);
  const int nb = imin(cdiv(M * 4, 256), kCUs * 4);
  hipStream_t st = (hipStream_t)stream;
  if (E == 16) hipLaunchKernelGGL((vq_nn_bwd_kernel<16>), dim3(nb), dim3(256), 0, st, x, embed, (const long long*)indices, dq, gloss, dx, dembed, M, K, w_commit);
  else if (E == 32) hipLaunchKernelGGL((vq_nn_bwd_kernel<32>), dim3(nb), dim3(256), 0, st, x, embed, (const long long*)indices, dq, gloss, dx, dembed, M, K, w_commit);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "vq_nn: dim %d not built", E);
  IDEE_LAUNCH_CHECK("vq_nn_bwd");
  return IDEE_OK;
}


extern "C" int idee_vq_cluster_stats(const float* x, const int64_t* indices, float* counts, float* sums, long M, int E, int K,
                                     idee_stream_t stream) {
  IDEE_CHECK_ARG(x && indices && counts && sums && M > 0 && K > 0, "vq_cluster_stats: bad argument");
  if ((long)K * E > kVqMaxFloats) return idee::fail(IDEE_ERR_UNSUPPORTED, "vq: codebook %d x %d exceeds %d floats", K, E, kVqMaxFloats);
  const int nb = imin(cdiv(M * (E / 4), 256), kCUs * 4);
  hipStream_t st = (hipStream_t)stream;
  if (E == 16) hipLaunchKernelGGL((vq_cluster_stats_kernel<16>), dim3(nb), dim3(256), 0, st, x, (const long long*)indices, counts, sums, M, K);
  else if (E == 32) hipLaunchKernelGGL((vq_cluster_stats_kernel<32>), dim3(nb), dim3(256), 0, st, x, (const long long*)indices, counts, sums, M, K);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "vq: dim %d not built", E);
  IDEE_LAUNCH_CHECK("vq_cluster_stats");
  return IDEE_OK;
}

extern "C" int idee_vq_ema_update(float* cluster_size, float* embed_avg, float* embed, const float* new_size, const float* new_sum,
                                  int K, int E, float decay, float eps, idee_stream_t stream) {
  IDEE_CHECK_ARG(cluster_size && embed_avg && embed && new_size && new_sum && K > 0 && E > 0, "vq_ema_update: bad argument");
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cluster_size, embed_avg, embed, new_size,
                     new_sum, K, E, decay, eps);
  IDEE_LAUNCH_CHECK("vq_ema_update");
  return IDEE_OK;
}

extern "C" int idee_vq_kmeans_update(float* means, const float* bins, const float* sums, int K, int E, idee_stream_t stream) {
  IDEE_CHECK_ARG(means && bins && sums && K > 0 && E > 0, "vq_kmeans_update: bad argument");
  hipLaunchKernelGGL(vq_kmeans_update_kernel, dim3(cdiv((long)K * E, 256)), dim3(256), 0, (hipStream_t)stream, means, bins, sums, K, E);
  IDEE_LAUNCH_CHECK("vq_kmeans_update");
  return IDEE_OK;
}

extern "C" int idee_vq_expire_codes(float* embed, float* cluster_size, float* embed_avg, const float* x, const int64_t* candidates,
                                    int K, int E, float threshold, float reset, int32_t* n_expired, idee_stream_t stream) {
  IDEE_CHECK_ARG(embed && cluster_size && embed_avg && x && candidates && K > 0 && E > 0, "vq_expire_codes: bad argument");
  if (K > kVqMaxFloats / 16) return idee::fail(IDEE_ERR_UNSUPPORTED, "vq_expire_codes: %d codes exceed %d", K, kVqMaxFloats / 16);
  hipLaunchKernelGGL(vq_expire_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, embed, cluster_size, embed_avg, x,
                     (const long long*)candidates, K, E, threshold, reset, (int*)n_expired);
  IDEE_LAUNCH_CHECK("vq_expire_codes");
  return IDEE_OK;
}


extern "C" int idee_lfq_code_vectors(const int64_t* indices, const float* w_out, const float* b_out, float* out, int n, int E,
                                     int codebook_size, idee_stream_t stream) {
  IDEE_CHECK_ARG(indices && w_out && b_out && out && n > 0 && E > 0 && codebook_size >= 2, "lfq_code_vectors: bad argument");
  hipLaunchKernelGGL(lfq_code_vectors_kernel, dim3(cdiv((long)n * E, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)indices, w_out, b_out, out, n, E, log2i(codebook_size));
  IDEE_LAUNCH_CHECK("lfq_code_vectors");
  return IDEE_OK;
}

extern "C" int idee_loss_combine_fwd(const float* bce, int P, int T3, const float* l_anom, float lambda_anomaly,
                                     const float* loss_zq, float* out5, idee_stream_t stream) {
  IDEE_CHECK_ARG(bce && l_anom && loss_zq && out5 && P >= 1 && T3 >= 1, "loss_combine_fwd: bad argument");
  hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bce, P, T3, l_anom, lambda_anomaly, loss_zq, out5);
  IDEE_LAUNCH_CHECK("loss_combine_fwd");
  return IDEE_OK;
}

extern "C" int idee_loss_combine_bwd(const float* gtotal, int P, int T3, float lambda_anomaly, float* dbce, float* d_anom,
                                     float* d_zq, idee_stream_t stream) {
  IDEE_CHECK_ARG(gtotal && dbce && d_anom && d_zq && P >= 1 && T3 >= 1, "loss_combine_bwd: bad argument");
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(cdiv((long)P * T3, 256)), dim3(256), 0, (hipStream_t)stream, gtotal, P, T3,
                     lambda_anomaly, dbce, d_anom, d_zq);
  IDEE_LAUNCH_CHECK("loss_combine_bwd");
  return IDEE_OK;
}

extern "C" int idee_confusion_counts(const float* logits, const float* gt, const float* valid, long M, float threshold,
                                     uint64_t* counts, idee_stream_t stream) {
  IDEE_CHECK_ARG(logits && gt && counts && M > 0, "confusion_counts: bad argument");
  hipLaunchKernelGGL(confusion_counts_kernel, dim3(imin(cdiv(M, 256), 512)), dim3(256), 0, (hipStream_t)stream, logits, gt, valid,
                     M, threshold, (unsigned long long*)counts);
  IDEE_LAUNCH_CHECK("confusion_counts");
  return IDEE_OK;
}

extern "C" int idee_anomaly_vote_accumulate(const int64_t* anomaly, const int32_t* idx_end, float* acc, float* cnt, int n_batch,
                                            int V, int dt, long HW, int TT, idee_stream_t stream) {
  IDEE_CHECK_ARG(anomaly && idx_end && acc && cnt && n_batch > 0 && V > 0 && dt > 0 && HW > 0 && TT > 0, "anomaly_vote_accumulate: bad argument");
  hipLaunchKernelGGL(anomaly_vote_kernel, dim3(stream_grid((long)n_batch * V * dt * HW)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)anomaly, idx_end, acc, cnt, n_batch, V, dt, HW, TT);
  IDEE_LAUNCH_CHECK("anomaly_vote_accumulate");
  return IDEE_OK;
}

extern "C" int idee_anomaly_vote_finalize(const float* acc, const float* cnt, float* out, long n, idee_stream_t stream) {
  IDEE_CHECK_ARG(acc && cnt && out && n > 0, "anomaly_vote_finalize: bad argument");
  hipLaunchKernelGGL(anomaly_vote_finalize_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, acc, cnt, out, n);
  IDEE_LAUNCH_CHECK("anomaly_vote_finalize");
  return IDEE_OK;
}

extern "C" int idee_lfq_sign_field_fwd(const float* signs, void* sp, int N, int V, int T, long HW, int D, int CP, int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(signs && sp && N > 0 && V > 0 && T >= 2 && HW > 0 && D > 0 && CP % 16 == 0 && 2 * (V * D + 1) <= CP, "lfq_sign_field_fwd: bad argument");
  IDEE_CHECK_ARG((long)N * (T / 2) * HW * (CP / 4) + 256L * 65536 < (1L << 32), "lfq_sign_field_fwd: more than 2^32 output quads");   // 32-bit index in the kernel
  const dim3 grid(stream_grid((long)N * (T / 2) * HW * (CP / 4)));
  if (act_dtype == IDEE_DTYPE_BF16) hipLaunchKernelGGL(sign_field_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, signs, (bf16_t*)sp, N, V, T, HW, D, CP);
  else hipLaunchKernelGGL(sign_field_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, signs, (float*)sp, N, V, T, HW, D, CP);
  IDEE_LAUNCH_CHECK("lfq_sign_field_fwd");
  return IDEE_OK;
}
extern "C" int idee_lfq_sign_field_bwd(const void* dsp, float* dsigns, int N, int V, int T, long HW, int D, int CP, int act_dtype, idee_stream_t stream) {
  IDEE_CHECK_ARG(dsp && dsigns && N > 0 && V > 0 && T >= 2 && HW > 0 && D > 0 && CP % 16 == 0 && 2 * (V * D + 1) <= CP, "lfq_sign_field_bwd: bad argument");
  IDEE_CHECK_ARG((long)N * T * HW + 256L * 4096 < (1L << 32), "lfq_sign_field_bwd: more than 2^32 pixels");   // 32-bit index in the kernel
  const dim3 grid(stream_grid((long)N * T * HW));
  if (act_dtype == IDEE_DTYPE_BF16) hipLaunchKernelGGL(sign_field_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dsp, dsigns, N, V, T, HW, D, CP);
  else hipLaunchKernelGGL(sign_field_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dsp, dsigns, N, V, T, HW, D, CP);
  IDEE_LAUNCH_CHECK("lfq_sign_field_bwd");
  return IDEE_OK;
}
extern "C" int idee_lfq_fold_weights_fwd(const float* w, const float* w_out, const float* b_out, float* weff, int Co, int V, int C,
                                         int D, int CP, int per_var, idee_stream_t stream) {
  IDEE_CHECK_ARG(w && w_out && b_out && weff && Co > 0 && V > 0 && C > 0 && D > 0 && 2 * (V * D + 1) <= CP, "lfq_fold_weights_fwd: bad argument");
  const long total = (long)(per_var ? V * Co : Co) * CP * 9;
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(imin(cdiv(total, 256), 1024)), dim3(256), 0, (hipStream_t)stream, w, w_out, b_out, weff, Co, V, C, D, CP, per_var);
  IDEE_LAUNCH_CHECK("lfq_fold_weights_fwd");
  return IDEE_OK;
}
extern "C" int idee_lfq_fold_weights_bwd(const float* dweff, const float* w, const float* w_out, const float* b_out, float* dw,
                                         float* dw_out, float* db_out, int Co, int V, int C, int D, int CP, int per_var,
                                         idee_stream_t stream) {
  IDEE_CHECK_ARG(dweff && w && w_out && b_out && dw && dw_out && db_out && Co > 0 && V > 0 && C > 0 && D > 0 && 2 * (V * D + 1) <= CP,
                 "lfq_fold_weights_bwd: bad argument");
  const long total = (long)(per_var ? V * Co : Co) * (per_var ? 1 : V) * C * 18;
  hipLaunchKernelGGL(fold_bwd_w_kernel, dim3(imin(cdiv(total, 256), 1024)), dim3(256), 0, (hipStream_t)stream, dweff, w_out, b_out, dw, Co, V, C, D, CP, per_var);
  hipLaunchKernelGGL(fold_bwd_p_kernel, dim3(C * (D + 1)), dim3(256), 0, (hipStream_t)stream, dweff, w, dw_out, db_out, Co, V, C, D, CP, per_var);
  IDEE_LAUNCH_CHECK("lfq_fold_weights_bwd");
  return IDEE_OK;
}

extern "C" int idee_patch_merging_fwd(const float* x, const float* gamma, const float* beta, const float* w, float* y, int B, int D,
                                      int H, int W, int C, idee_stream_t stream) {
  IDEE_CHECK_ARG(x && gamma && beta && w && y && B > 0 && D > 0 && H > 0 && W > 0, "patch_merging_fwd: bad argument");
  const int D2 = D == 1 ? 1 : (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long ntok = (long)B * D2 * H2 * W2;
  const dim3 grid(imin(cdiv(ntok, 16), kCUs * 8));
  if (C == 16) hipLaunchKernelGGL(patch_merging_fwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, w, y, ntok, D, H, W, D2, H2, W2);
  else if (C == 32) hipLaunchKernelGGL(patch_merging_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, w, y, ntok, D, H, W, D2, H2, W2);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "patch_merging: C=%d is not built (16, 32)", C);
  IDEE_LAUNCH_CHECK("patch_merging_fwd");
  return IDEE_OK;
}
extern "C" int idee_patch_merging_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* w,
                                      float* dx, float* dgamma, float* dbeta, float* dw, int B, int D, int H, int W, int C,
                                      idee_stream_t stream) {
  IDEE_CHECK_ARG(x && dy && gamma && beta && w && dx && dgamma && dbeta && dw && B > 0 && D > 0 && H > 0 && W > 0,
                 "patch_merging_bwd: bad argument");
  const int D2 = D == 1 ? 1 : (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long ntok = (long)B * D2 * H2 * W2;
  const dim3 grid(imin(cdiv(ntok, 16), kCUs * 2));
  if (C == 16) hipLaunchKernelGGL(patch_merging_bwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x, dy, gamma, beta, w, dx, dgamma, dbeta, dw, ntok, D, H, W, D2, H2, W2);
  else if (C == 32) hipLaunchKernelGGL(patch_merging_bwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, dy, gamma, beta, w, dx, dgamma, dbeta, dw, ntok, D, H, W, D2, H2, W2);
  else return idee::fail(IDEE_ERR_UNSUPPORTED, "patch_merging: C=%d is not built (16, 32)", C);
  IDEE_LAUNCH_CHECK("patch_merging_bwd");
  return IDEE_OK;
}
