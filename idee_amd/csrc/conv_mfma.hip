// Implicit-GEMM 3-D convolutions on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Covers every convolution of the hot path, channels-last, V per-variable weight sets batched:
//   mode K333_REPLICATE : Conv3d 3x3x3, stride 1, replicate padding 1  (proj_var, Swin_3D.py:586-592;
//                         conv_block convs, encoder/CNN_3D.py:99-106)
//   mode K233_S2_ZERO   : Conv3d (2,3,3), stride (2,1,1), zero padding (0,1,1)
//                         (classifier heads, classifier/CNN_3D.py:38-40,83-85)
// forward, data gradient and weight gradient.  Output tile of a wave = up to 64 pixels along W
// (4 MFMA column blocks of 16) x NT blocks of 16 output channels, D^T orientation
// (rows = channels, cols = pixels) so that each lane ends up with 4 consecutive channels of one
// pixel and stores them as one 16-byte word.  The GEMM K dimension runs over (tap, 16-channel
// group, 4 k-steps); each lane fetches 4 consecutive input channels of its pixel with one 16-byte
// load and the matching 4 k-steps of the weight fragment with one 16-byte load from a
// fragment-ordered ("packed") weight buffer (LDS-resident for the 16->16 convs, L2-resident
// for the 96-wide joint head).  Replicate padding is a clamp on the gather address; its data
// gradient uses the exact (q, tap) pair lists in T and H (tile-uniform) and a padded frame with
// float atomics on the two edge columns in W.
#include "common.h"
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace idee;
using namespace idee_conv;

template <int MODE, int NT, bool WLDS, bool CIN1>
__global__ __launch_bounds__(256, NT >= 6 ? 2 : 1) void conv_kernel(ConvArgs a) {
  constexpr int MT = 4;
  __shared__ __attribute__((aligned(16))) float s_w[WLDS ? 27 * 256 : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, ntg = blockIdx.z;
  const int wset = b % a.wsets;
  const long tapstride = (long)a.Gk * a.GnTot * 256;
  const float* wp = a.wp + (long)wset * a.ntaps * tapstride + (long)ntg * NT * 256;
  if (WLDS) {   // NT == GnTot == 1, Gk == 1: ntaps KB, loaded once per workgroup
    for (int e = threadIdx.x * 4; e < a.ntaps * 256; e += 1024) st4(s_w + e, ld4(wp + e));
    __syncthreads();
  }
  // width in pixels of the frame the tile columns index (padded by 2 for the replicate dgrad)
  const int Wframe = (MODE == DGRAD_REP) ? a.Wo + 2 : a.Wo;

  for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    const int wt = tile % a.wtiles, r = tile / a.wtiles;
    const int hg = r % a.hgroups, t = r / a.hgroups;
    const int h = hg * 4 + wave;
    if (h >= a.Ho) continue;   // no barriers inside the tile loop
    const int my_mt = a.sub_base + (wt < a.sub_rem ? 1 : 0);
    const int w0 = 16 * (wt * a.sub_base + min(wt, a.sub_rem));

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* in_b = a.in + (long)b * a.i_sB;
#pragma unroll
    for (int ot = 0; ot < 3; ++ot) {
      const Slot st = slot_t<MODE>(ot, t, a.Ti, a.kt);
      if (!st.ok) continue;
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const Slot sh = slot_h<MODE>(oh, h, a.Hi);
        if (!sh.ok) continue;
        const float* rowp = in_b + st.src * a.i_sT + sh.src * a.i_sH + (CIN1 ? 0 : 4 * kq);
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) {
          const int tap = (st.tap * 3 + sh.tap) * 3 + ow;
          long off[MT];
          bool okm[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int p = w0 + m * 16 + j;
            bool ok;
            const int q = src_w<MODE>(ow, p, a.Wi, ok);
            okm[m] = ok && (p < Wframe) && (m < my_mt);
            off[m] = (long)q * a.i_sW;
          }
          for (int g = 0; g < a.Gk; ++g) {
            f32x4 xf[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              if (CIN1) {
                float v0 = (okm[m] && kq == 0) ? rowp[off[m]] : 0.f;
                xf[m] = f32x4{v0, 0.f, 0.f, 0.f};
              } else {
                xf[m] = okm[m] ? ld4(rowp + off[m] + goff(g, a.i_gpu, a.i_sU, a.i_sG)) : f32x4{0.f, 0.f, 0.f, 0.f};
              }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              f32x4 wf;
              if (WLDS) wf = ld4(s_w + tap * 256 + lane * 4);
              else wf = ld4(wp + tap * tapstride + ((long)g * a.GnTot + n) * 256 + lane * 4);
#pragma unroll
              for (int s = 0; s < (CIN1 ? 1 : 4); ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xf[m][s], acc[m][n], 0, 0, 0);
            }
          }
        }
      }
    }

    // ---- epilogue: lane holds channels 4kq..4kq+3 of block n for pixel column j ----
    float* out_r = a.out + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH;
    const float* ref_r = a.relu_ref ? a.relu_ref + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH : nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m >= my_mt) continue;
      const int p = w0 + m * 16 + j;
      if (p >= Wframe) continue;
      int wdst = p;
      bool atomic = false;
      if (MODE == DGRAD_REP) {
        wdst = min(max(p - 1, 0), a.Wo - 1);
        atomic = (p <= 1) || (p >= a.Wo);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int cg = ntg * NT + n;
        f32x4 v = acc[m][n];
        if (a.bias) {
          if (a.cout_real == 1) v[0] += a.bias[wset];   // (wsets, 1) bias of the single-class heads
          else v += ld4(a.bias + (long)wset * a.GnTot * 16 + cg * 16 + 4 * kq);
        }
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const long o = (long)wdst * a.o_sW + goff(cg, a.o_gpu, a.o_sU, a.o_sG) + 4 * kq;
        if (ref_r) {
          const f32x4 rr = ld4(ref_r + o);
          v[0] = rr[0] > 0.f ? v[0] : 0.f; v[1] = rr[1] > 0.f ? v[1] : 0.f;
          v[2] = rr[2] > 0.f ? v[2] : 0.f; v[3] = rr[3] > 0.f ? v[3] : 0.f;
        }
        if (a.cout_real == 1) {
          if (kq == 0) out_r[(long)wdst * a.o_sW] = a.addend ? (a.addend + (out_r - a.out))[(long)wdst * a.o_sW] + v[0] : v[0];
        } else if (atomic) {
          atomicAdd(out_r + o, v[0]); atomicAdd(out_r + o + 1, v[1]);
          atomicAdd(out_r + o + 2, v[2]); atomicAdd(out_r + o + 3, v[3]);
        } else {
          if (a.addend) v += ld4(a.addend + (out_r - a.out) + o);
          st4(out_r + o, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dWp[wset][tap][cin][coutPad] += sum_pixels x[src(pixel, tap)][cin] * dy[pixel][cout]
// One wave owns TPW taps x NTA input-channel blocks x NTB output-channel blocks and a share of
// the (b, t, h) pixel rows; K runs over pixels, 4 per MFMA.
// ---------------------------------------------------------------------------------------------
template <int MODE /*FWD_REP or FWD_S2*/, int TPW, int NTA, int NTB>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  // blockIdx.z = wset ; blockIdx.y = (tap group, cin group, cout group) ; blockIdx.x = row chunk
  const int wset = blockIdx.z;
  int yy = blockIdx.y;
  const int tg = yy % a.tap_groups; yy /= a.tap_groups;
  const int gag = yy % a.ga_groups; const int gbg = yy / a.ga_groups;
  const int tap0 = tg * TPW, ga0 = gag * NTA, gb0 = gbg * NTB;

  f32x4 acc[TPW][NTA][NTB];
  float bsum[NTB];
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
    for (int i = 0; i < NTA; ++i)
#pragma unroll
      for (int n = 0; n < NTB; ++n) acc[tp][i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < NTB; ++n) bsum[n] = 0.f;

  const int nb = a.B / a.wsets;   // batch items that share this weight set
  const long total_rows = (long)nb * a.rows;
  const int steps = (a.W + 3) / 4;
  for (long row = (long)blockIdx.x * 4 + wave; row < total_rows; row += (long)gridDim.x * 4) {
    const int bi = (int)(row / a.rows), rr = (int)(row % a.rows);
    const int b = bi * a.wsets + wset;
    const int t = rr / a.H, h = rr % a.H;
    const float* dy_r = a.dy + (long)b * a.y_sB + (long)t * a.y_sT + (long)h * a.y_sH + r16;
    const float* x_b = a.x + (long)b * a.x_sB + r16;
    for (int s = 0; s < steps; ++s) {
      const int p = 4 * s + kq;
      const bool pin = p < a.W;
      float bf[NTB];
#pragma unroll
      for (int n = 0; n < NTB; ++n) {
        const int co = (gb0 + n) * 16 + r16;
        bf[n] = (pin && co < a.cout_real) ? dy_r[(long)p * a.y_sW + (a.cout_real == 1 ? -r16 : goff(gb0 + n, a.y_gpu, a.y_sU, a.y_sG))] : 0.f;
        bsum[n] += bf[n];
      }
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const int tap = tap0 + tp;
        const int ot = tap / 9, oh = (tap / 3) % 3, ow = tap % 3;
        int st, sh, sw;
        bool ok = pin;
        if (MODE == FWD_REP) {
          st = min(max(t + ot - 1, 0), a.T - 1); sh = min(max(h + oh - 1, 0), a.H - 1);
          sw = min(max(p + ow - 1, 0), a.W - 1);
        } else {
          st = a.kt * t + ot; sh = h + oh - 1; sw = p + ow - 1;
          ok = ok && sh >= 0 && sh < a.H && sw >= 0 && sw < a.W;
        }
        const float* xp = x_b + (long)st * a.x_sT + (long)sh * a.x_sH + (long)sw * a.x_sW;
#pragma unroll
        for (int i = 0; i < NTA; ++i) {
          const float af = ok ? xp[goff(ga0 + i, a.x_gpu, a.x_sU, a.x_sG)] : 0.f;
#pragma unroll
          for (int n = 0; n < NTB; ++n)
            acc[tp][i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], acc[tp][i][n], 0, 0, 0);
        }
      }
    }
  }
  // flush: D[row = cin][col = cout]
  float* dw = a.dwp + (long)wset * a.ntaps * a.Cin * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
    for (int i = 0; i < NTA; ++i)
#pragma unroll
      for (int n = 0; n < NTB; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicAdd(dw + ((long)(tap0 + tp) * a.Cin + (ga0 + i) * 16 + r0 + r) * a.CoutPad + (gb0 + n) * 16 + col,
                    acc[tp][i][n][r]);
  if (a.dbias && tg == 0 && gag == 0) {
#pragma unroll
    for (int n = 0; n < NTB; ++n) {
      float v = bsum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + (gb0 + n) * 16 + lane, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Wide (2,3,3)-stride-2 convolutions (joint classifier head, all NT = Cout/16 output blocks per wave), forward
// and data gradient, with the weights of ONE tap at a time staged through LDS and shared by the 8 waves of the
// workgroup (8 rows x 64 pixels).  The 36.9 KB tap slab (6x6 blocks) is double-buffered: while a tap is
// multiplied, the next slab travels global -> registers -> LDS.  Weight traffic from L2 drops 8x against the
// per-wave fragment loads of conv_kernel; activations still come straight from L1/L2.
// ---------------------------------------------------------------------------------------------
template <int MODE /*FWD_S2 or DGRAD_S2*/, int NT, bool RES /*all taps' weights resident in LDS: no barriers per tap*/>
__global__ __launch_bounds__(512) void conv_wide_kernel(ConvArgs a) {
  constexpr int MT = 4, WAVES = 8;
  extern __shared__ __attribute__((aligned(16))) float s_wd[];   // [2][Gk * NT * 256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const int wset = b % a.wsets;
  const int slab = a.Gk * NT * 256;                       // floats per tap
  const float* wp = a.wp + (long)wset * a.ntaps * slab;
  const int slab4 = slab / 4;
  const int NOT = (MODE == FWD_S2) ? a.kt : 1;            // T-slots

  // a (b, t) plane is cut into 16-pixel column blocks, numbered row-major; the `hgroups` workgroups of a plane take equal
  // contiguous ranges of them and walk their range in chunks of WAVES * MT blocks — a wave owns up to MT consecutive blocks
  // (they may wrap to the next row) and issues MFMAs only for the `nm` it has, so a range that is not a whole number of chunks
  // costs its last chunk pro rata (fixed tiles of 32 blocks gave 2.56 tiles per workgroup at the headline shape: 3 rounds)
  const int nblk = a.Ho * a.nsub;
  bool first = true;
  for (int wgp = blockIdx.x; wgp < a.To * a.hgroups; wgp += gridDim.x) {
   const int t = wgp / a.hgroups, rg = wgp - t * a.hgroups;
   const int r_lo = (int)((long)rg * nblk / a.hgroups), r_hi = (int)((long)(rg + 1) * nblk / a.hgroups);
   // A chunk's blocks are dealt EVENLY to the 8 waves (waves w < cextra take one more: one per SIMD first), so the ragged last chunk
   // of a range costs the most any SIMD holds of it, not a full chunk (2.54 chunks per workgroup at the headline shape ran as 3; the
   // data gradient's 5.08 as 6).  The chunk body is compiled once per block count of a wave (MT = 1..4: arrays, loads and MFMA
   // blocks sized to it; the waves of a chunk may run different copies — every copy passes the same barriers).
   auto chunk = [&](auto mtc, const int blk0, const bool row_ok) __attribute__((always_inline)) {
    constexpr int MT = decltype(mtc)::value;
    const int nm = row_ok ? MT : 0;
    int hm[MT], wm[MT];
    bool vm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int blk = blk0 + m;
      vm[m] = m < nm;
      hm[m] = blk / a.nsub;
      wm[m] = (blk - hm[m] * a.nsub) * 16 + j;
      vm[m] = vm[m] && wm[m] < a.Wo;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* in_b = a.in + (long)b * a.i_sB + 4 * kq;

    const int ntap_it = NOT * 9;
    auto tap_of = [&](int it) {                            // iteration -> weight tap index
      const int ot = it / 9, oh = (it / 3) % 3, ow = it % 3;
      const int tt = (MODE == FWD_S2) ? ot : (a.kt == 2 ? (t & 1) : 0);
      return (tt * 3 + oh) * 3 + ow;
    };
    if constexpr (RES) {
      if (first) {                                         // once per workgroup
        first = false;
        for (int u = tid; u < a.ntaps * slab4; u += 512) st4(s_wd + 4 * u, ld4(wp + 4 * u));
        __syncthreads();
      }
    } else {
      __syncthreads();                                     // previous tile's readers done
      for (int u = tid; u < slab4; u += 512) st4(s_wd + 4 * u, ld4(wp + (long)tap_of(0) * slab + 4 * u));
      __syncthreads();
    }

    // activations are prefetched one (tap, channel-block) step ahead, ACROSS tap boundaries: a source pixel outside
    // the frame loads nothing and multiplies zeros
    struct TapSrc { long off[MT]; unsigned ok; __device__ bool okm(int m) const { return (ok >> m) & 1u; } };     // (a bool array here is copied through scratch)
    auto tap_src = [&](int it) {
      TapSrc s;
      s.ok = 0;
      const int ot = it / 9, oh = (it / 3) % 3, ow = it % 3;
      const Slot st = slot_t<MODE>(ot, t, a.Ti, a.kt);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const Slot sh = slot_h<MODE>(oh, hm[m], a.Hi);
        bool ok;
        const int q = src_w<MODE>(ow, wm[m], a.Wi, ok);
        const bool okm = vm[m] && st.ok && sh.ok && ok;
        s.ok |= okm ? 1u << m : 0u;
        s.off[m] = okm ? (long)st.src * a.i_sT + (long)sh.src * a.i_sH + (long)q * a.i_sW : 0;
      }
      return s;
    };
    if constexpr (NT == 1 && RES) {
      // one output block per wave (the data gradient into a 16-channel tensor): only 16 MFMAs per (tap, channel block)
      // step.  Channel blocks OUTSIDE, taps inside: the three `ow` taps of a source row re-read the same cache lines one
      // pixel apart, and with the taps innermost that 32 KB working set per workgroup stays in L1 (taps outermost streamed
      // 196 KB per tap and re-fetched every row through L2: 5.9 TB/s of L2 fill, the bound of this launch).  Loads run
      // three steps ahead through a register ring; per step the address is a lane base + a wave-uniform tap offset and
      // the validity one bit of a per-tile mask.
      long lbase[MT];
      unsigned okbits[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        lbase[m] = (long)hm[m] * a.i_sH + (long)wm[m] * a.i_sW;
        unsigned bits = 0;
        for (int it = 0; it < ntap_it; ++it) {
          const int ot = it / 9, oh = (it / 3) % 3, ow = it % 3;
          const Slot st = slot_t<MODE>(ot, t, a.Ti, a.kt);
          const Slot sh = slot_h<MODE>(oh, hm[m], a.Hi);
          bool okw;
          (void)src_w<MODE>(ow, wm[m], a.Wi, okw);
          if (vm[m] && st.ok && sh.ok && okw) bits |= 1u << it;
        }
        okbits[m] = bits;
      }
      auto tap_delta = [&](int it) {      // wave-uniform: source offset of tap `it` relative to (h, w) of the output pixel
        const int ot = it / 9, oh = (it / 3) % 3, ow = it % 3;
        const Slot st = slot_t<MODE>(ot, t, a.Ti, a.kt);
        const int dh = (MODE == FWD_S2) ? oh - 1 : 1 - oh, dw = (MODE == FWD_S2) ? ow - 1 : 1 - ow;
        return (long)st.src * a.i_sT + (long)dh * a.i_sH + (long)dw * a.i_sW;
      };
      const int total = a.Gk * ntap_it;                       // ntap_it is 9 or 18: a multiple of the ring depth
      f32x4 xq[3][MT];
      auto issue = [&](int step, f32x4 (&dst)[MT]) __attribute__((always_inline)) {
        const int g = step / ntap_it, it = step - g * ntap_it;
        const long d = tap_delta(it) + goff(g, a.i_gpu, a.i_sU, a.i_sG);
#pragma unroll
        for (int m = 0; m < MT; ++m) dst[m] = ((okbits[m] >> it) & 1u) ? ld4(in_b + lbase[m] + d) : f32x4{0.f, 0.f, 0.f, 0.f};
      };
      issue(0, xq[0]); issue(1, xq[1]); issue(2, xq[2]);
      for (int st0 = 0; st0 < total; st0 += 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int step = st0 + r;
          const int g = step / ntap_it, it = step - g * ntap_it;
          const f32x4 wf = ld4(s_wd + tap_of(it) * slab + g * 256 + lane * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s)      // all MT blocks unconditionally: branches here break the load / MFMA interleave
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xq[r][m][s], acc[m][0], 0, 0, 0);
          if (step + 3 < total) issue(step + 3, xq[r]);
        }
      }
    } else {
      // Activation loads are UNCONDITIONAL (an absent pixel's offset is 0: a valid address) and masked when they become the current
      // operand: a load inside a branch leaves the count of outstanding loads unknown, and every wait on the previous step's
      // operands became vmcnt(0) — a wait for the loads of the NEXT step just issued (WAIT_ANY was 30 % of this kernel's wave cycles)
      TapSrc cur = tap_src(0);
      f32x4 xc[MT], xn[MT];
      unsigned okn = 0;
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int m = 0; m < MT; ++m) { xc[m] = ld4(in_b + cur.off[m]); xc[m] = cur.okm(m) ? xc[m] : zero4; }

      for (int it = 0; it < ntap_it; ++it) {
        const float* wcur = RES ? s_wd + tap_of(it) * slab : s_wd + (it & 1) * slab;
        float* wnext = s_wd + ((it + 1) & 1) * slab;
        // prefetch the next slab into registers (up to 5 x 16 bytes per thread for a 6x6 slab)
        f32x4 pre[5];
        const bool more = it + 1 < ntap_it;
        if (more && !RES) {
          const float* src = wp + (long)tap_of(it + 1) * slab;
  #pragma unroll
          for (int q = 0; q < 5; ++q) { const int u = tid + q * 512; if (q * 512 < slab4) pre[q] = ld4(src + 4 * min(u, slab4 - 1)); }
        }
        TapSrc nxt = cur;
        if (more) nxt = tap_src(it + 1);
        f32x4 wfb[2];
        for (int g = 0; g < a.Gk; ++g) {
          {   // the next step's operands: this tap's next channel block, else the next tap's first (the very last step re-reads)
            const bool same = g + 1 < a.Gk;
            const long go = same ? goff(g + 1, a.i_gpu, a.i_sU, a.i_sG) : 0;
  #pragma unroll
            for (int m = 0; m < MT; ++m) xn[m] = ld4(in_b + (same ? cur.off[m] : nxt.off[m]) + go);
            okn = same ? cur.ok : more ? nxt.ok : 0u;
          }
          if (row_ok) {
            // weight fragments one output block ahead (the read for block n + 1 — or block 0 of the next channel block — goes out
            // before block n's MFMAs: left to the compiler it sat right in front of its first use, 6 exposed LDS round trips per step)
            if (g == 0) wfb[0] = ld4(wcur + lane * 4);
  #pragma unroll
            for (int n = 0; n < NT; ++n) {
              if (n + 1 < NT) wfb[(n + 1) & 1] = ld4(wcur + (g * NT + n + 1) * 256 + lane * 4);
              else if (g + 1 < a.Gk) wfb[(n + 1) & 1] = ld4(wcur + (g + 1) * NT * 256 + lane * 4);
              __builtin_amdgcn_sched_barrier(0);
              const f32x4 wf = wfb[n & 1];
  #pragma unroll
              for (int m = 0; m < MT; ++m)
  #pragma unroll
                for (int s = 0; s < 4; ++s) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xc[m][s], acc[m][n], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (NT & 1) wfb[0] = wfb[1];
          }
  #pragma unroll
          for (int m = 0; m < MT; ++m) xc[m] = ((okn >> m) & 1u) ? xn[m] : zero4;
        }
        cur = nxt;
        if constexpr (!RES) {
          if (more) {
  #pragma unroll
            for (int q = 0; q < 5; ++q) { const int u = tid + q * 512; if (u < slab4) st4(wnext + 4 * u, pre[q]); }
          }
          __syncthreads();
        }
      }

    }

    if (!row_ok) return;
    // the NT bias vectors in one batch ahead of the stores (a load between stores waits, through vmcnt(0), for every store
    // before it: 24 serialized memory round trips per chunk)
    f32x4 biasv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = a.bias ? ld4(a.bias + (long)wset * NT * 16 + n * 16 + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (!vm[m]) continue;
      const long ro = (long)b * a.o_sB + (long)t * a.o_sT + (long)hm[m] * a.o_sH + (long)wm[m] * a.o_sW + 4 * kq;
      float* out_r = a.out + ro;
      const float* ref_r = a.relu_ref ? a.relu_ref + ro : nullptr;
      // (all loads of this pixel column ahead of its stores: see biasv)
      f32x4 rrv[NT], adv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long o = goff(n, a.o_gpu, a.o_sU, a.o_sG);
        rrv[n] = ref_r ? ld4(ref_r + o) : f32x4{1.f, 1.f, 1.f, 1.f};
        adv[n] = a.addend ? ld4(a.addend + (out_r - a.out) + o) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 v = acc[m][n];
        v += biasv[n];
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const long o = goff(n, a.o_gpu, a.o_sU, a.o_sG);
        if (ref_r) {
          const f32x4 rr = rrv[n];
          v[0] = rr[0] > 0.f ? v[0] : 0.f; v[1] = rr[1] > 0.f ? v[1] : 0.f;
          v[2] = rr[2] > 0.f ? v[2] : 0.f; v[3] = rr[3] > 0.f ? v[3] : 0.f;
        }
        v += adv[n];
        st4(out_r + o, v);
      }
    }
   };
   for (int c0 = r_lo; c0 < r_hi; c0 += WAVES * MT) {
    const int csize = min(r_hi - c0, WAVES * MT), cbase = csize / WAVES, cextra = csize - cbase * WAVES;
    const int blk0 = c0 + wave * cbase + min(wave, cextra);
    const int nm = __builtin_amdgcn_readfirstlane(cbase + (wave < cextra ? 1 : 0));
    if (nm == 4) chunk(std::integral_constant<int, 4>{}, blk0, true);
    else if (nm == 3) chunk(std::integral_constant<int, 3>{}, blk0, true);
    else if (nm == 2) chunk(std::integral_constant<int, 2>{}, blk0, true);
    else chunk(std::integral_constant<int, 1>{}, blk0, nm == 1);      // (nm == 0: staging and barriers only)
   }
  }
}


// ---------------------------------------------------------------------------------------------
// conv_wide_kernel on split-precision MFMAs (opt-in, IDEE_CONV_MATH=split): K = 32 = TWO 16-channel groups of one tap, three
// piece products per (tap, group pair, output block) — 18 f16 / bf16 MFMAs of 16 cycles where the f32 kernel issues 48 of 32.
// f16 pieces in the forward mode (operands O(1): absolute piece floor 2^-25), bf16 pieces in the data gradient.  The tap's
// weight slab is converted to piece fragments [w(group a) | w(group b)] (hi and lo) while it is staged into LDS, double
// buffered as before; the activations of the two groups are split in registers right after their loads.  Gk even.
// ---------------------------------------------------------------------------------------------
// f16 pieces (forward; data gradient under IDEE_CONV_MATH=f22, GSC): the weight set is prescaled by a power of two found in the
// kernel prologue (largest |w| of the set in [2^13, 2^14): both pieces of every weight that matters are normal numbers; the
// epilogue multiplies by its inverse).  GSC: the dy operands of a wave's chunk (4 column blocks x all taps x all channel groups)
// carry a power-of-two factor 2^k kept per wave and chunk (common.h GradScale, moving down only): every (tap, group pair)
// operand is tested before it is split (one compare and the wave's vote); an operand that would leave the window at the top
// lowers k, the chunk's accumulators are multiplied along (exact), and the epilogue multiplies by 2^-k.
// SC_ACT (forward): the ACTIVATION operands of a wave's chunk are range-guarded the same way (common.h ActGuard): split as they are
// until an operand's largest |x| reaches 2^15, then scaled by 2^k, k < 0, the chunk's accumulators multiplied along.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst);     // (LDS-DMA of 64 x 16 bytes: defined with conv16_glds_kernel)
// exponent of the scale from the set's largest |w| (bit pattern): |w| 2^k in [2^13, 2^14)
__device__ __forceinline__ int wscale_exp(unsigned bm) {
  const int e = (int)(bm >> 23) - 127;
  if (bm == 0u || e > 127) return 0;
  return 13 - e < -60 ? -60 : (13 - e > 60 ? 60 : 13 - e);
}
enum { SC_NONE = 0, SC_GRAD = 1, SC_ACT = 2 };   // scale carried by the f16-piece operand that has no a-priori range
// PRE (f16 pieces only): the piece fragments of every tap and the weight set's scale come READY from the packed buffer
// (ConvArgs::wpieces / wscale_k, written once per step by idee_conv_pack_weights): the next tap's piece slab is copied into the
// other LDS buffer by LDS-DMA while this tap is multiplied — no weight scan in the prologue (166 k floats per workgroup and
// launch), no conversion pass (48 vector instructions per thread and tap), ONE workgroup barrier per tap instead of two, two LDS
// slabs instead of three.
template <int MODE /*FWD_S2 or DGRAD_S2*/, int NT, int SC = SC_NONE, bool PRE = false>
__global__ __launch_bounds__(512) void conv_wide_split_kernel(ConvArgs a) {
  constexpr bool GSC = SC == SC_GRAD, ASC = SC == SC_ACT;
  static_assert(!GSC || MODE == DGRAD_S2, "gradient scale: data-gradient mode");
  static_assert(!ASC || MODE == FWD_S2, "activation guard: forward mode");
  constexpr int MT = 4, WAVES = 8;
  constexpr bool F16 = MODE == FWD_S2 || GSC;
  extern __shared__ __attribute__((aligned(1024))) unsigned s_ws[];   // [2][Gk/2 * NT * 2 pieces * 256 words] piece slabs, then one RAW f32 slab
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const int wset = b % a.wsets;
  const int GP = a.Gk / 2;                                // channel-group pairs
  const int slab = a.Gk * NT * 256;                       // f32 floats per tap in the packed buffer == words per split slab
  const float* wp = a.wp + (long)wset * a.ntaps * slab;
  float wscale = 1.f, winv = 1.f;
  static_assert(!PRE || F16, "pre-split pieces: f16 pieces");
  if constexpr (PRE) {
    const int k = wscale_exp((unsigned)a.wscale_k[wset]);      // (the packed buffer holds the set's largest |w| as a bit pattern)
    wscale = pow2i(k); winv = pow2i(-k);
  } else if constexpr (F16) {
    __shared__ unsigned s_wmax[WAVES];
    float mx = 0.f;
    {   // eight loads in flight per trip (one load and its wait per trip: ~80 serialized memory round trips per workgroup)
      const long nw = (long)a.ntaps * slab;
      for (long i = 4L * tid; i < nw; i += 8 * 4 * 512) {
        f32x4 f[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long iu = i + (long)u * 4 * 512; f[u] = ld4(wp + (iu < nw ? iu : i)); }
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = amax_acc(mx, f[u]);
      }
    }
    const unsigned wm_ = wave_umax(__float_as_uint(mx));
    if (lane == 0) s_wmax[wave] = wm_;
    __syncthreads();
    unsigned bm = 0u;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) bm = bm > s_wmax[w] ? bm : s_wmax[w];
    bm = __builtin_amdgcn_readfirstlane(bm);
    const int e = (int)(bm >> 23) - 127;
    if (bm != 0u && e <= 127) {
      const int k = 13 - e < -60 ? -60 : (13 - e > 60 ? 60 : 13 - e);
      wscale = pow2i(k); winv = pow2i(-k);
    }
  }
  const int npu = GP * NT * 64;                           // (pair, output block, lane) units per tap
  const int NOT = (MODE == FWD_S2) ? a.kt : 1;
  const int nblk = a.Ho * a.nsub;
  // unit u = (gp * NT + n) * 64 + l: fragments of groups 2gp and 2gp+1 for output block n, lane l
  // (a unit's two fragments sit at a per-thread 32-bit offset from the tap's slab, whose address is scalar: written with the
  //  division by NT inside, the compiler hoisted six 64-bit addresses per thread out of the tap loop, spilled them, and reloaded
  //  each from scratch — behind s_waitcnt vmcnt(0) — in front of its load: six serialized round trips per tap that also drained
  //  the activation prefetch.  18 -> 2 spilled registers; 96 -> 96 forward -7 %, data gradient -3 %)
  auto unit_off = [&](int u) { const int l = u & 63, r = u >> 6, n = r % NT, gp = r / NT; return ((2 * gp) * NT + n) * 256 + l * 4; };
  const int uo0 = unit_off(tid), uo1 = unit_off(tid + 512), uo2 = unit_off(tid + 1024);
  auto load_unit = [&](int tap, int u, f32x4& fa, f32x4& fb) __attribute__((always_inline)) {
    const float* src = wp + (long)__builtin_amdgcn_readfirstlane(tap) * slab;
    const int o = u == tid ? uo0 : u == tid + 512 ? uo1 : uo2;
    fa = ld4(src + o);
    fb = ld4(src + o + NT * 256);
  };
  auto store_unit = [&](unsigned* dst, int u, f32x4 fa, f32x4 fb) __attribute__((always_inline)) {
    const Split sa = split2<F16>(F16 ? fa * wscale : fa), sb = split2<F16>(F16 ? fb * wscale : fb);
    const int l = u & 63, r = u >> 6;
    *reinterpret_cast<u32x4*>(dst + ((r * 2 + 0) * 64 + l) * 4) = u32x4{sa.h01, sa.h23, sb.h01, sb.h23};
    *reinterpret_cast<u32x4*>(dst + ((r * 2 + 1) * 64 + l) * 4) = u32x4{sa.l01, sa.l23, sb.l01, sb.l23};
  };
  for (int wgp = blockIdx.x; wgp < a.To * a.hgroups; wgp += gridDim.x) {     // ranges and chunks: see conv_wide_kernel
   const int t = wgp / a.hgroups, rg = wgp - t * a.hgroups;
   const int r_lo = (int)((long)rg * nblk / a.hgroups), r_hi = (int)((long)(rg + 1) * nblk / a.hgroups);
   // A chunk's blocks are dealt EVENLY to the 8 waves (waves w < cextra take one more: one per SIMD first), so the ragged last chunk
   // of a range costs the most any SIMD holds of it, not a full chunk (2.54 chunks per workgroup at the headline shape ran as 3; the
   // data gradient's 5.08 as 6).  The chunk body is compiled once per block count of a wave (MT = 1..4: arrays, loads and MFMA
   // blocks sized to it; the waves of a chunk may run different copies — every copy passes the same barriers).
   auto chunk = [&](auto mtc, const int blk0, const bool row_ok) __attribute__((always_inline)) {
    constexpr int MT = decltype(mtc)::value;
    const int nm = row_ok ? MT : 0;
    int hm[MT], wm[MT];
    bool vm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int blk = blk0 + m;
      vm[m] = m < nm;
      hm[m] = blk / a.nsub;
      wm[m] = (blk - hm[m] * a.nsub) * 16 + j;
      vm[m] = vm[m] && wm[m] < a.Wo;
    }
    f32x4 acc[MT][NT];
    GradScale gs;                                        // (GSC) this wave's scale for this chunk
    ActGuard ag;                                         // (ASC) this wave's activation guard for this chunk
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* in_b = a.in + (long)b * a.i_sB + 4 * kq;
    const int ntap_it = NOT * 9;
    auto tap_of = [&](int it) {
      const int ot = it / 9, oh = (it / 3) % 3, ow = it % 3;
      const int tt = (MODE == FWD_S2) ? ot : (a.kt == 2 ? (t & 1) : 0);
      return (tt * 3 + oh) * 3 + ow;
    };
    __syncthreads();                                     // previous tile's readers done
    // (PRE) piece slab of tap `tp` -> LDS buffer `dstw` by LDS-DMA: Gk * NT blocks of 1 KB, dealt to the 8 waves
    auto dma_pieces = [&](int tp, unsigned* dstw) __attribute__((always_inline)) {
      const float* psrc = reinterpret_cast<const float*>(a.wpieces) + ((long)wset * a.ntaps + tp) * slab + lane * 4;
      const unsigned pdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dstw);
      for (int blk_ = __builtin_amdgcn_readfirstlane(wave); blk_ < a.Gk * NT; blk_ += WAVES)
        glds16(psrc + blk_ * 256, __builtin_amdgcn_readfirstlane(pdst + (unsigned)blk_ * 1024u));
    };
    if constexpr (PRE) {
      dma_pieces(tap_of(0), s_ws);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int u = tid; u < npu; u += 512) { f32x4 fa, fb; load_unit(tap_of(0), u, fa, fb); store_unit(s_ws, u, fa, fb); }
    }
    __syncthreads();

    struct TapSrc { int off[MT]; unsigned ok; __device__ bool okm(int m) const { return (ok >> m) & 1u; } };     // (element offsets inside one batch item: 32 bits, checked by the host)
    auto tap_src = [&](int it) {
      TapSrc sx;
      sx.ok = 0;
      const int ot = it / 9, oh = (it / 3) % 3, ow = it % 3;
      const Slot st = slot_t<MODE>(ot, t, a.Ti, a.kt);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const Slot sh = slot_h<MODE>(oh, hm[m], a.Hi);
        bool ok;
        const int q = src_w<MODE>(ow, wm[m], a.Wi, ok);
        const bool okm = vm[m] && st.ok && sh.ok && ok;
        sx.ok |= okm ? 1u << m : 0u;
        sx.off[m] = okm ? (int)((long)st.src * a.i_sT + (long)sh.src * a.i_sH + (long)q * a.i_sW) : 0;
      }
      return sx;
    };
    TapSrc cur = tap_src(0);
    f32x4 xa[MT], xb[MT], na[MT], nb_[MT];
    // (activation loads unconditional, masked when they become the current operand: see conv_wide_kernel)
    unsigned okn = 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xa[m] = ld4(in_b + cur.off[m]);
      xb[m] = ld4(in_b + cur.off[m] + goff(1, a.i_gpu, a.i_sU, a.i_sG));
      xa[m] = cur.okm(m) ? xa[m] : zero4;
      xb[m] = cur.okm(m) ? xb[m] : zero4;
    }
    for (int it = 0; it < ntap_it; ++it) {
      const unsigned* wcur = s_ws + (it & 1) * slab;
      unsigned* wnext = s_ws + ((it + 1) & 1) * slab;
      // prefetch the next tap's raw fragments into registers (up to 3 units = 6 x 16 bytes per thread at 6 x 6 blocks)
      // the next tap's RAW weight slab goes to LDS by DMA (one 1 KB block of 64 x 16 bytes per instruction, the blocks dealt to
      // the 8 waves) and is converted to piece fragments from there at the end of the tap: as 2 x 3 fragments per thread in
      // registers through the whole tap it was 24 registers in a kernel that spills
      const bool more = it + 1 < ntap_it;
      if (PRE && more) dma_pieces(tap_of(it + 1), wnext);      // (its readers finished at the barrier that ended tap it - 1)
      if (!PRE && more) {
        const float* rsrc = wp + (long)tap_of(it + 1) * slab + lane * 4;
        const unsigned rdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(s_ws + 2 * slab));
        for (int blk_ = __builtin_amdgcn_readfirstlane(wave); blk_ < a.Gk * NT; blk_ += WAVES)
          glds16(rsrc + blk_ * 256, __builtin_amdgcn_readfirstlane(rdst + (unsigned)blk_ * 1024u));
      }
      TapSrc nxt = cur;
      if (more) nxt = tap_src(it + 1);
      u32x4 ahb[2], alb[2];
      for (int gp = 0; gp < GP; ++gp) {
        {   // the next step's operands: this tap's next group pair, else the next tap's first (the very last step re-reads)
          const bool same = gp + 1 < GP;
          const long g0 = same ? goff(2 * gp + 2, a.i_gpu, a.i_sU, a.i_sG) : 0;
          const long g1 = same ? goff(2 * gp + 3, a.i_gpu, a.i_sU, a.i_sG) : goff(1, a.i_gpu, a.i_sU, a.i_sG);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int o = same ? cur.off[m] : nxt.off[m];
            na[m] = ld4(in_b + o + g0);
            nb_[m] = ld4(in_b + o + g1);
          }
          okn = same ? cur.ok : more ? nxt.ok : 0u;
        }
        if (row_ok) {
          u32x4 bh[MT], bl[MT];
          float gsc = 1.f;
          if constexpr (GSC) {
            float am = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) am = amax_acc(amax_acc(am, xa[m]), xb[m]);
            if (gs.set ? __any(am * gs.s() >= 256.f) : __any(am > 0.f)) {
              const int k0 = gs.k;
              const bool was = gs.set;
              gs.set_from(wave_umax(__float_as_uint(am)));
              if (was && gs.k != k0) {
                const float f = pow2i(gs.k - k0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                  for (int n = 0; n < NT; ++n) acc[m][n] *= f;
              }
            }
            gsc = gs.s();
          }
          if constexpr (ASC) {
            float am = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) am = amax_acc(amax_acc(am, xa[m]), xb[m]);
            if (ag.exceeds(am)) {         // (rare: |x| >= 2^15 at the chunk's scale)
              const int dk = ag.lower(wave_umax(__float_as_uint(ActGuard::finite(am))));
              const float f = pow2i(dk < -126 ? -126 : dk);
#pragma unroll
              for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] *= f;
            }
            if (ag.k != 0) {              // wave-uniform
              gsc = ag.s();
#pragma unroll
              for (int m = 0; m < MT; ++m) { xa[m] *= gsc; xb[m] *= gsc; }
            }
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const Split sa = split2<F16>(GSC ? xa[m] * gsc : xa[m]), sb = split2<F16>(GSC ? xb[m] * gsc : xb[m]);
            bh[m] = u32x4{sa.h01, sa.h23, sb.h01, sb.h23};
            bl[m] = u32x4{sa.l01, sa.l23, sb.l01, sb.l23};
          }
          // weight piece fragments one output block ahead (see conv_wide_kernel)
          auto wfrag = [&](int idx, int part) __attribute__((always_inline)) {
            return *reinterpret_cast<const u32x4*>(wcur + ((idx * 2 + part) * 64 + lane) * 4);
          };
          if (gp == 0) { ahb[0] = wfrag(0, 0); alb[0] = wfrag(0, 1); }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (n + 1 < NT) { ahb[(n + 1) & 1] = wfrag(gp * NT + n + 1, 0); alb[(n + 1) & 1] = wfrag(gp * NT + n + 1, 1); }
            else if (gp + 1 < GP) { ahb[(n + 1) & 1] = wfrag((gp + 1) * NT, 0); alb[(n + 1) & 1] = wfrag((gp + 1) * NT, 1); }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 ah = ahb[n & 1], al = alb[n & 1];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = mfma_s<F16>(ah, bh[m], acc[m][n]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = mfma_s<F16>(ah, bl[m], acc[m][n]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = mfma_s<F16>(al, bh[m], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (NT & 1) { ahb[0] = ahb[1]; alb[0] = alb[1]; }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) { const bool ok_ = (okn >> m) & 1u; xa[m] = ok_ ? na[m] : zero4; xb[m] = ok_ ? nb_[m] : zero4; }
      }
      cur = nxt;
      if (PRE && more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA blocks have landed (the barrier below: all)
      if (!PRE && more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA blocks have landed
        __syncthreads();                                      // ... and every other wave's
        const float* raw = reinterpret_cast<const float*>(s_ws + 2 * slab);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int u = tid + q * 512;
          if (u < npu) {
            const int l = u & 63, r = u >> 6, n = r % NT, gp = r / NT;
            const f32x4 fa = ld4(raw + ((2 * gp) * NT + n) * 256 + l * 4), fb = ld4(raw + ((2 * gp + 1) * NT + n) * 256 + l * 4);
            store_unit(wnext, u, fa, fb);
          }
        }
      }
      __syncthreads();
    }
    if (!row_ok) return;
    // the NT bias vectors in one batch ahead of the stores (a load between stores waits, through vmcnt(0), for every store
    // before it: 24 serialized memory round trips per chunk)
    f32x4 biasv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = a.bias ? ld4(a.bias + (long)wset * NT * 16 + n * 16 + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (!vm[m]) continue;
      const long ro = (long)b * a.o_sB + (long)t * a.o_sT + (long)hm[m] * a.o_sH + (long)wm[m] * a.o_sW + 4 * kq;
      float* out_r = a.out + ro;
      const float* ref_r = a.relu_ref ? a.relu_ref + ro : nullptr;
      // (all loads of this pixel column ahead of its stores: see biasv)
      f32x4 rrv[NT], adv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long o = goff(n, a.o_gpu, a.o_sU, a.o_sG);
        rrv[n] = ref_r ? ld4(ref_r + o) : f32x4{1.f, 1.f, 1.f, 1.f};
        adv[n] = a.addend ? ld4(a.addend + (out_r - a.out) + o) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 v = F16 ? acc[m][n] * (GSC ? winv * gs.inv() : ASC ? winv * ag.inv() : winv) : acc[m][n];
        v += biasv[n];
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const long o = goff(n, a.o_gpu, a.o_sU, a.o_sG);
        if (ref_r) {
          const f32x4 rr = rrv[n];
          v[0] = rr[0] > 0.f ? v[0] : 0.f; v[1] = rr[1] > 0.f ? v[1] : 0.f;
          v[2] = rr[2] > 0.f ? v[2] : 0.f; v[3] = rr[3] > 0.f ? v[3] : 0.f;
        }
        v += adv[n];
        st4(out_r + o, v);
      }
    }
   };
   for (int c0 = r_lo; c0 < r_hi; c0 += WAVES * MT) {
    const int csize = min(r_hi - c0, WAVES * MT), cbase = csize / WAVES, cextra = csize - cbase * WAVES;
    const int blk0 = c0 + wave * cbase + min(wave, cextra);
    const int nm = __builtin_amdgcn_readfirstlane(cbase + (wave < cextra ? 1 : 0));
    if (nm == 4) chunk(std::integral_constant<int, 4>{}, blk0, true);
    else if (nm == 3) chunk(std::integral_constant<int, 3>{}, blk0, true);
    else if (nm == 2) chunk(std::integral_constant<int, 2>{}, blk0, true);
    else chunk(std::integral_constant<int, 1>{}, blk0, nm == 1);      // (nm == 0: staging and barriers only)
   }
  }
}

// ---------------------------------------------------------------------------------------------
// 16 -> 16 3x3x3 replicate convolution, forward and data gradient, activations staged through LDS.
// (LDS row layout [channel quad][pixel][4 channels]: the 16 lanes of a quad read consecutive 16-byte chunks — the
// pixel-major layout [pixel][16] put every other lane on the same banks.)
// Every WAVE works alone: it owns a strip of 2 column blocks (32 pixels) and walks a run of output rows for a
// fixed (batch item, t), keeping in its private 19.6 KB of LDS, for each of the 3 T-slots, a ring of 3 source
// rows with a one-pixel halo (replicate clamp resp. zero padding resolved while staging).  Moving down one row
// stages 3 new rows.  The 27 weight fragments live in registers.  No workgroup barrier anywhere: the waves of a
// CU drift apart, so one wave's staging latency hides under another's MFMA chain.
// ---------------------------------------------------------------------------------------------
constexpr int kC16Row = 2 * 16 + 2;             // staged pixels per source row of a strip

__device__ __forceinline__ void wave_fence() {   // LDS hand-off inside one wave: DS ops of a wave execute in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Replicate-mode data gradient, ConvArgs::no_edge_atomics: frame column p = 0 (virtual column -1) is added to p = 1 and p = W + 1
// (virtual column W) to p = W by a DPP row shift — the two lanes of a pair sit in one 16-column tile — and the donor lanes drop out.
// Returns false for a donor lane.  (Called by all lanes of the wave.)
__device__ __forceinline__ bool fold_w_edges(f32x4& acc, int p, int Wo) {
  f32x4 fl, fr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ar = acc[r];     // (a scalar copy: bit-casting the vector element in place compiled to component 0 four times)
    fl[r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ar), 0x111 /*row_shr:1*/, 0xf, 0xf, true));
    fr[r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ar), 0x101 /*row_shl:1*/, 0xf, 0xf, true));
  }
  if (p == 1) acc += fl;
  if (p == Wo) acc += fr;
  return !(p == 0 || p == Wo + 1);
}

template <int MODE /*FWD_REP, DGRAD_REP: 3x3x3 replicate ; FWD_S2, DGRAD_S2: (kt,3,3) zero padding, 16 -> 16*/>
__global__ __launch_bounds__(256, 2) void conv16_lds_kernel(ConvArgs a, int nstrip, int hrun, int nhrun) {
  constexpr int MT = 2;
  constexpr bool REP = MODE == FWD_REP || MODE == DGRAD_REP;
  constexpr int TSN = REP ? 3 : (MODE == FWD_S2 ? 2 : 1);      // T-slots a row step reads
  __shared__ __attribute__((aligned(16))) float s_xw[4][9 * kC16Row * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  // (the bias once per wave: loaded in the epilogue it is a memory round trip per row, behind the row's stores)
  const f32x4 biasv = a.bias ? ld4(a.bias + (long)wset * 16 + 4 * ((threadIdx.x & 63) >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
  float* s_x = s_xw[wave];
  // wr[(ot*3+oh)*3+ow] = fragment of the weight tap that (ot, oh, ow) uses under the CURRENT edge pattern: the forward
  // conv always uses tap (ot, oh, ow); the data gradient flips a T / H tap where the replicate padding folds a source
  // row back (first / last frame, first / last row) — the 27 fragments are re-read (27 KB from L2) when that pattern
  // changes, so the tap loop below indexes registers statically and stays one straight-line, software-pipelined block
  f32x4 wr[TSN * 9];
  int wkey = -1;
  const float* wp = a.wp + (long)wset * a.ntaps * 256;
  const int Wframe = (MODE == DGRAD_REP) ? a.Wo + 2 : a.Wo;
  const int nsub = (Wframe + 15) / 16;
  const int nb = a.B / a.wsets;
  const long nunits = (long)nb * a.To * nhrun * nstrip;

  for (long un = (long)blockIdx.x * 4 + wave; un < nunits; un += (long)gridDim.x * 4) {
    // t runs fastest: the 4 waves of a workgroup take 4 consecutive output frames of the SAME strip and rows at the same
    // time, so the source rows they share (frame t feeds outputs t-1, t, t+1) are fetched from HBM once and re-read from
    // this CU's L1 / the XCD's L2 (with t slowest the three users ran far apart: 3.1x the input bytes from HBM)
    long r = un;
    const int t = (int)(r % a.To); r /= a.To;
    const int sp = (int)(r % nstrip); r /= nstrip;
    const int hr = (int)(r % nhrun);
    const int b = (int)(r / nhrun) * a.wsets + wset;
    const int my_sub = min(MT, nsub - sp * MT);                        // column blocks of this strip (1 or 2)
    const int p0 = sp * MT * 16;                                       // first frame column of the strip
    const int rowlen = my_sub * 16 + 2;
    const int h0 = hr * hrun, h1 = min(a.Ho, h0 + hrun);
    int res0 = -1, res1 = -1, res2 = -1;
    const float* in_b = a.in + (long)b * a.i_sB;
    Slot st[3];
#pragma unroll
    for (int ot = 0; ot < 3; ++ot) st[ot] = slot_t<MODE>(ot, t, a.Ti, a.kt);
    const int blk1 = my_sub > 1 ? 16 : 0;                              // a lone block re-reads block 0 (result unused)

    for (int h = h0; h < h1; ++h) {
      wave_fence();   // previous row's reads are ordered before this row's staging writes
      Slot sh[3];
      int slot_of[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        sh[oh] = slot_h<MODE>(oh, h, a.Hi);
        const int hs = sh[oh].src;                    // zero-padded modes: -1 and Hi are staged as rows of zeros
        const bool inside = sh[oh].ok;
        const int slot = (hs + 3) % 3;
        slot_of[oh] = slot;
        const int cur = slot == 0 ? res0 : slot == 1 ? res1 : res2;
        if (cur == hs + 1) continue;                  // ring key hs + 1 >= 0
        if (slot == 0) res0 = hs + 1; else if (slot == 1) res1 = hs + 1; else res2 = hs + 1;
        // all 9 loads of the 3 T-slots first (up to 3 x 16 bytes per lane and slot), then the LDS writes: a
        // load -> wait -> write loop would serialise 9 L2 round trips per row
        f32x4 stg[TSN][3];
#pragma unroll
        for (int ot = 0; ot < TSN; ++ot) {
          const float* src = in_b + (long)st[ot].src * a.i_sT + (long)hs * a.i_sH;
          const bool live = inside && st[ot].ok;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            const int px = u >> 2, q4 = u & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < rowlen * 4 && live) {
              if (MODE == FWD_REP) {
                const int w = min(max(p0 + px - 1, 0), a.Wi - 1);
                v = ld4(src + (long)w * a.i_sW + 4 * q4);
              } else {
                const int w = p0 + px - (MODE == DGRAD_REP ? 2 : 1);     // the replicate dgrad walks a frame padded by one
                if (w >= 0 && w < a.Wi) v = ld4(src + (long)w * a.i_sW + 4 * q4);
              }
            }
            stg[ot][k] = v;
          }
        }
#pragma unroll
        for (int ot = 0; ot < TSN; ++ot) {
          float* dst = s_x + (long)(ot * 3 + slot) * kC16Row * 16;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            if (u < rowlen * 4) st4(dst + (u & 3) * (kC16Row * 4) + (u >> 2) * 4, stg[ot][k]);   // [channel quad][pixel][4]
          }
        }
      }
      wave_fence();

      {
        const int key = ((((st[0].tap * 3 + st[1].tap) * 3 + st[2].tap) * 3 + sh[0].tap) * 3 + sh[1].tap) * 3 + sh[2].tap;
        if (key != wkey) {
          wkey = key;
#pragma unroll
          for (int ot = 0; ot < TSN; ++ot)
#pragma unroll
            for (int oh = 0; oh < 3; ++oh)
#pragma unroll
              for (int ow = 0; ow < 3; ++ow)
                wr[(ot * 3 + oh) * 3 + ow] = ld4(wp + ((st[ot].tap * 3 + sh[oh].tap) * 3 + ow) * 256 + lane * 4);
        }
      }
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* rowq[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) rowq[oh] = s_x + (long)slot_of[oh] * kC16Row * 16 + kq * (kC16Row * 4) + j * 4;
      // LDS fragment reads of tap tp+1 are issued before the MFMAs of tap tp (two register sets)
      f32x4 xb[2][2];
      auto rd = [&](int tp, f32x4& x0, f32x4& x1) __attribute__((always_inline)) {
        const int ot = tp / 9, oh = (tp / 3) % 3, ow = tp % 3;
        // staged index of the source pixel: fwd p + ow ; dgrad p - ow + 2   (p = column inside the strip)
        const int sh_px = (MODE == FWD_REP || MODE == FWD_S2) ? ow : 2 - ow;
        const float* rp = rowq[oh] + ot * 3 * kC16Row * 16 + sh_px * 4;
        x0 = ld4(rp);
        x1 = ld4(rp + blk1 * 4);
      };
      rd(0, xb[0][0], xb[0][1]);
#pragma unroll
      for (int tp = 0; tp < TSN * 9; ++tp) {
        if (tp + 1 < TSN * 9) rd(tp + 1, xb[(tp + 1) & 1][0], xb[(tp + 1) & 1][1]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 wf = wr[tp];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xb[tp & 1][0][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xb[tp & 1][1][s], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue
      float* out_r = a.out + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH;
      const float* ref_r = a.relu_ref ? a.relu_ref + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH : nullptr;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int p = p0 + m * 16 + j;
        bool live = true;
        if (MODE == DGRAD_REP && a.no_edge_atomics) live = fold_w_edges(acc[m], p, a.Wo);
        if (m >= my_sub || !live) continue;
        if (p >= Wframe) continue;
        int wdst = p;
        bool atomic = false;
        if (MODE == DGRAD_REP) { wdst = min(max(p - 1, 0), a.Wo - 1); atomic = ((p <= 1) || (p >= a.Wo)) && !a.no_edge_atomics; }
        f32x4 v = acc[m] + biasv;
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        const long o = (long)wdst * a.o_sW + 4 * kq;
        if (ref_r) {
          const f32x4 rr = ld4(ref_r + o);
          v[0] = rr[0] > 0.f ? v[0] : 0.f; v[1] = rr[1] > 0.f ? v[1] : 0.f;
          v[2] = rr[2] > 0.f ? v[2] : 0.f; v[3] = rr[3] > 0.f ? v[3] : 0.f;
        }
        if (atomic) {
          atomicAdd(out_r + o, v[0]); atomicAdd(out_r + o + 1, v[1]);
          atomicAdd(out_r + o + 2, v[2]); atomicAdd(out_r + o + 3, v[3]);
        } else {
          if (a.addend) v += ld4(a.addend + (out_r - a.out) + o);
          st4(out_r + o, v);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// conv16_lds_kernel for the replicate modes with the row staging taken off the critical path: the source rows of the NEXT
// output row are fetched by LDS-DMA (global_load_lds_dwordx4: no staging registers — the kernel sits at the 256-VGPR cap)
// while the current row's MFMA chain runs.  Moving down one row replaces exactly the ring slot of the oldest source row,
// so the 9 taps that read that row go FIRST and the DMA is issued right after their fragment reads have returned; the other
// 18 taps (4,600 MFMA cycles) cover its latency.  (The register-staged kernel waits out every row's loads: two waves per
// SIMD that interleave their MFMAs finish together and then wait together — 66 % MFMA-busy.)
// An LDS-DMA writes 64 consecutive 16-byte chunks, so a staged row is pixel-major [pixel][4 chunks] — the global reads
// stay coalesced — with the chunk position XOR-ed by (pixel >> 2) & 3 on the SOURCE side, which keeps the fragment reads
// (16 lanes = 16 consecutive pixels, one channel quad) on 16 distinct bank groups.  The zero columns of the data
// gradient's padded frame are written once per work unit; the DMA never touches them.
// The DMA is inline asm (hipcc would drain it at the next LDS read of the same array): completion is counted here, with an
// explicit s_waitcnt vmcnt(0) at the top of each row; asm loads older than a compiler load only make hipcc's own counted
// waits stricter.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE /*FWD_REP or DGRAD_REP*/>
__global__ __launch_bounds__(256, 2) void conv16_glds_kernel(ConvArgs a, int nstrip, int nfull) {
  constexpr int MT = 2;
  constexpr bool FWD = MODE == FWD_REP;
  constexpr int kRowB = kC16Row * 64;                 // bytes of one staged row
  __shared__ __attribute__((aligned(1024))) float s_xw[4][9 * kC16Row * 16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  float* s_x = s_xw[wave];
  const unsigned s_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)s_x);   // LDS byte address of this wave's ring
  f32x4 wr[27];
  int wkey = -1;
  const f32x4 bias = a.bias ? ld4(a.bias + (long)wset * (a.bias_set_stride ? a.bias_set_stride : 16) + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wp = a.wp + (long)wset * (a.w_set_stride ? a.w_set_stride : (long)a.ntaps * 256);
  const int wts = a.w_tap_stride ? a.w_tap_stride : 256;
  const int Wi = a.Wi;
  const long i_sT = a.i_sT, i_sH = a.i_sH, i_sW = a.i_sW;
  const int Wframe = FWD ? a.Wo : a.Wo + 2;
  const int nsub = (Wframe + 15) / 16;
  const int nb = a.B / a.wsets;
  // fragment read offsets (bytes inside a staged row) of the three W taps: pixel j + shift, chunk kq ^ ((pixel >> 2) & 3)
  int lo[3];
#pragma unroll
  for (int ow = 0; ow < 3; ++ow) {
    const int px = j + (FWD ? ow : 2 - ow);
    lo[ow] = px * 64 + ((kq ^ ((px >> 2) & 3)) * 16);
  }

  // Work partition (static, cost-balanced): the 4 waves of a workgroup take 4 CONSECUTIVE output frames (a frame group) of one
  // strip — the source rows they share are fetched from HBM once — and walk rows of the strip's (batch item, frame group, row)
  // space.  All rows of all strips are laid on one COST axis (a row of a two-block strip weighs kWF, a row of the one-block
  // last strip of an odd block count kWL: half the MFMAs, the same staging and epilogue) and every workgroup takes an equal
  // interval of it — at most two segments of consecutive rows, each inside one strip.  (Work units of whole row runs left
  // 5.3 units per wave = 6 rounds at the headline shape, and a one-block code path could not pay under them.)
  constexpr unsigned kWF = 10, kWL = 6;     // measured: a one-block row costs 0.55-0.6 of a two-block row (sweep 4..8: 6 is best)
  // 32-bit arithmetic throughout (the host checks Rtot * kWF * nstrip < 2^31): 64-bit divisions by run-time values are
  // ~150-instruction routines, and a workgroup decodes several positions per segment
  // frame groups of 4 when the frame count allows it; otherwise every WAVE takes its own interval of single frames (always
  // balanced, at the price of the shared fetches)
  const bool grouped = (a.To & 3) == 0;
  const int TG = grouped ? a.To / 4 : a.To;
  const unsigned Ho = (unsigned)a.Ho;
  const unsigned Rtot = (unsigned)nb * TG * Ho;              // rows of the (batch item, frame group, row) space
  const unsigned CF = Rtot * kWF, Cfull = (unsigned)nfull * CF, Ctot = Cfull + (nfull < nstrip ? Rtot * kWL : 0u);
  const unsigned part = grouped ? blockIdx.x : blockIdx.x * 4 + wave, nparts = grouped ? gridDim.x : gridDim.x * 4;
  const unsigned c_lo = (unsigned)((unsigned long long)Ctot * part / nparts);
  const unsigned c_hi = (unsigned)((unsigned long long)Ctot * (part + 1) / nparts);
  for (unsigned cc = c_lo; cc < c_hi;) {
   const bool in_full = cc < Cfull;
   int sp = 0;
   if (in_full) { while (cc >= (unsigned)(sp + 1) * CF) ++sp; } else sp = nfull;
   const unsigned cbase = in_full ? (unsigned)sp * CF : Cfull;
   const unsigned cend = in_full ? (cbase + CF < c_hi ? cbase + CF : c_hi) : c_hi;
   // rows whose cost position starts inside [cc, cend)
   const unsigned r_lo = in_full ? (cc - cbase + kWF - 1) / kWF : (cc - cbase + kWL - 1) / kWL;
   const unsigned r_hi = in_full ? (cend - cbase + kWF - 1) / kWF : (cend - cbase + kWL - 1) / kWL;
   cc = cend;
   for (unsigned rho = r_lo; rho < r_hi;) {
    const unsigned plane = rho / Ho;
    const int h0 = (int)(rho - plane * Ho);
    const int tg = (int)(plane % (unsigned)TG);
    const int b = (int)(plane / (unsigned)TG) * a.wsets + wset;
    const int h1 = (int)(Ho < h0 + (r_hi - rho) ? Ho : h0 + (r_hi - rho));
    rho += h1 - h0;
    const int t = grouped ? tg * 4 + wave : tg;
    const int my_sub = min(MT, nsub - sp * MT);
    const int p0 = sp * MT * 16;
    const int rowlen = my_sub * 16 + 2;
    int res0 = -1, res1 = -1, res2 = -1;
    const float* in_b = a.in + (long)b * a.i_sB;
    Slot st[3];
#pragma unroll
    for (int ot = 0; ot < 3; ++ot) st[ot] = slot_t<MODE>(ot, t, a.Ti, a.kt);

    // per-lane source of the three DMA pieces of a row: chunk c = lane + 64 k -> pixel c >> 2, stored position c & 3
    long src_off0, src_off1, src_off2;
    bool src_ok0, src_ok1, src_ok2;
    {
      auto piece = [&](int k, long& off, bool& ok) __attribute__((always_inline)) {
        const int c = lane + 64 * k, px = c >> 2, q4 = (c & 3) ^ ((px >> 2) & 3);
        int w;
        ok = c < rowlen * 4;
        if (FWD) w = min(max(p0 + px - 1, 0), Wi - 1);
        else { w = p0 + px - 2; ok = ok && w >= 0 && w < Wi; w = min(max(w, 0), Wi - 1); }
        off = (long)w * i_sW + 4 * q4;
      };
      piece(0, src_off0, src_ok0); piece(1, src_off1, src_ok1); piece(2, src_off2, src_ok2);
    }
    const int st_src[3] = {st[0].src, st[1].src, st[2].src};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_fence();                                     // the previous unit's reads and DMA are done
    if (!FWD) {                                       // zero columns of the padded frame, all 9 staged rows
      const bool zk[3] = {!src_ok0, !src_ok1, !src_ok2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = lane + 64 * k;
        if (c < rowlen * 4 && zk[k]) {
#pragma unroll
          for (int sr = 0; sr < 9; ++sr) st4(s_x + sr * (kC16Row * 16) + c * 4, f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
    }
    // STAGE_ROWS(h): DMA the source rows output row `h` needs and the ring does not hold (completion awaited by the caller)
#define STAGE_ROWS(H_)                                                                                         \
  _Pragma("unroll") for (int oh_ = 0; oh_ < 3; ++oh_) {                                                        \
    const int hs_ = slot_h<MODE>(oh_, (H_), a.Hi).src;                                                         \
    const int slot_ = hs_ % 3;                                                                                 \
    const int cur_ = slot_ == 0 ? res0 : slot_ == 1 ? res1 : res2;                                             \
    if (cur_ != hs_ + 1) {                                                                                     \
      if (slot_ == 0) res0 = hs_ + 1; else if (slot_ == 1) res1 = hs_ + 1; else res2 = hs_ + 1;                \
      _Pragma("unroll") for (int ot_ = 0; ot_ < 3; ++ot_) {                                                    \
        const float* src_ = in_b + (long)st_src[ot_] * i_sT + (long)hs_ * i_sH;                                \
        const unsigned dst_ = s_base + (unsigned)((ot_ * 3 + slot_) * kRowB);                                  \
        if (src_ok0) glds16(src_ + src_off0, dst_);                                                            \
        if (src_ok1) glds16(src_ + src_off1, dst_ + 1024);                                                     \
        if (src_ok2) glds16(src_ + src_off2, dst_ + 2048);                                                     \
      }                                                                                                        \
    }                                                                                                          \
  }
    STAGE_ROWS(h0)

    // VM operations certain to be issued after a row's DMA: the epilogue's stores / atomics (its loads are not counted: a
    // lower bound keeps the counted wait below safe).  Same for every row of the unit.
    int n_late = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m >= my_sub) continue;
      const int pb = p0 + m * 16, pe = min(pb + 15, Wframe - 1);
      if (FWD) n_late += 1;
      else { n_late += (max(pb, 2) <= min(pe, a.Wo - 1)) ? 1 : 0; n_late += (pb <= 1 || pe >= a.Wo) ? 4 : 0; }
    }

    for (int h = h0; h < h1; ++h) {
      // the DMA of this row's sources must have landed; the previous row's stores (younger, retired in order after it) need not
      if (h == h0 || n_late == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (n_late == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (n_late <= 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      wave_fence();
      Slot sh[3];
      int slot_of[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) { sh[oh] = slot_h<MODE>(oh, h, a.Hi); slot_of[oh] = sh[oh].src % 3; }
      {
        const int key = ((((st[0].tap * 3 + st[1].tap) * 3 + st[2].tap) * 3 + sh[0].tap) * 3 + sh[1].tap) * 3 + sh[2].tap;
        if (key != wkey) {
          wkey = key;
#pragma unroll
          for (int ot = 0; ot < 3; ++ot)
#pragma unroll
            for (int oh = 0; oh < 3; ++oh)
#pragma unroll
              for (int ow = 0; ow < 3; ++ow)
                wr[(ot * 3 + oh) * 3 + ow] = ld4(wp + (long)((st[ot].tap * 3 + sh[oh].tap) * 3 + ow) * wts + lane * 4);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* rowq[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) rowq[oh] = reinterpret_cast<const char*>(s_x) + slot_of[oh] * kRowB;
      // tap order: the source row the next output row replaces first (fwd: oh = 0; data gradient: oh = 2)
      f32x4 xb[2][2];
      auto rd = [&](int nb_, int tp, f32x4& x0, f32x4& x1) __attribute__((always_inline)) {
        const int oi = tp / 9, ot = (tp / 3) % 3, ow = tp % 3;
        const int oh = FWD ? oi : 2 - oi;
        const char* rp = rowq[oh] + ot * 3 * kRowB + lo[ow];
        x0 = *reinterpret_cast<const f32x4*>(rp);
        if (nb_ == 2) x1 = *reinterpret_cast<const f32x4*>(rp + 1024);
      };
#define TAPS(T0_, T1_, NB_)                                                                                    \
  _Pragma("unroll") for (int tp = (T0_); tp < (T1_); ++tp) {                                                   \
    if (tp + 1 < 27) rd(NB_, tp + 1, xb[(tp + 1) & 1][0], xb[(tp + 1) & 1][1]);                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    const int oi = tp / 9, ot = (tp / 3) % 3, ow = tp % 3;                                                     \
    const f32x4 wf = wr[(ot * 3 + (FWD ? oi : 2 - oi)) * 3 + ow];                                              \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                            \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xb[tp & 1][0][s], acc[0], 0, 0, 0);                 \
      if (NB_ == 2) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xb[tp & 1][1][s], acc[1], 0, 0, 0);   \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
      // the source row the next output row replaces is read first; its DMA goes out once those fragment reads have returned
      // (tap 9's were issued after them; DS operations complete in order).  A one-block strip runs half the MFMAs.
      if (my_sub == 2) {
        rd(2, 0, xb[0][0], xb[0][1]);
        TAPS(0, 9, 2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h + 1 < h1) { STAGE_ROWS(h + 1) }
        __builtin_amdgcn_sched_barrier(0);
        TAPS(9, 27, 2)
      } else {
        rd(1, 0, xb[0][0], xb[0][1]);
        TAPS(0, 9, 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h + 1 < h1) { STAGE_ROWS(h + 1) }
        __builtin_amdgcn_sched_barrier(0);
        TAPS(9, 27, 1)
      }
#undef TAPS
      // epilogue (as conv16_lds_kernel)
      float* out_r = a.out + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH;
      const float* ref_r = a.relu_ref ? a.relu_ref + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH : nullptr;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int p = p0 + m * 16 + j;
        bool live = true;
        if (!FWD && a.no_edge_atomics) live = fold_w_edges(acc[m], p, a.Wo);
        if (m >= my_sub || !live) continue;
        if (p >= Wframe) continue;
        int wdst = p;
        bool atomic = false;
        if (!FWD) { wdst = min(max(p - 1, 0), a.Wo - 1); atomic = ((p <= 1) || (p >= a.Wo)) && !a.no_edge_atomics; }
        f32x4 v = acc[m];
        const long o = (long)wdst * a.o_sW + 4 * kq;
        const bool add_first = FWD && a.addend_first && a.addend;       // forward partial sums: the ReLU acts on the total
        if (add_first) v += ld4(a.addend + (out_r - a.out) + o);
        v += bias;
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (ref_r) {
          const f32x4 rr = ld4(ref_r + o);
          v[0] = rr[0] > 0.f ? v[0] : 0.f; v[1] = rr[1] > 0.f ? v[1] : 0.f;
          v[2] = rr[2] > 0.f ? v[2] : 0.f; v[3] = rr[3] > 0.f ? v[3] : 0.f;
        }
        if (atomic) {
          atomicAdd(out_r + o, v[0]); atomicAdd(out_r + o + 1, v[1]);
          atomicAdd(out_r + o + 2, v[2]); atomicAdd(out_r + o + 3, v[3]);
        } else {
          if (a.addend && !add_first) v += ld4(a.addend + (out_r - a.out) + o);
          st4(out_r + o, v);
        }
      }
    }
   }
  }
#undef STAGE_ROWS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------
// Data gradient of the (1,3,3) zero-padded convs INTO a 16-channel tensor (the sign-field convs 16 -> 96 / 16 -> 192:
// dy has Gk 16-channel groups, dx one).  K = 9 taps x 16 Gk channels is deep and the output is one block wide, so the work is
// split along K: wave g of a workgroup owns channel group g — its 9 weight fragments in registers, its own LDS ring of 3 dy
// rows (one group = 64 bytes per pixel: every dy byte is fetched once, by LDS-DMA one row ahead as in conv16_glds_kernel) —
// and the Gk partial 16 x 16 tiles of a row meet in a double-buffered LDS exchange (one workgroup barrier per row); waves 0
// and 1 sum them and store.  (conv_wide_kernel's one-output-block path re-fetched dy 3x through L2: 54 TFLOP/s.)
// Rows outside the frame are staged as zeros with ds_write; columns outside it are zeroed once per work unit.
// Work partition: the cost axis of conv16_glds_kernel, one interval per workgroup (single frames).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void dgrad_k133_to16_kernel(ConvArgs a, int nstrip, int nfull) {
  constexpr int MT = 2;
  constexpr int kRowB = kC16Row * 64;                 // bytes of one staged row
  extern __shared__ __attribute__((aligned(1024))) float s_dg[];   // [G][3 rows][34 px x 16] then [2][G][2][256]
  const int G = a.Gk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  float* s_x = s_dg + (long)wave * 3 * kC16Row * 16;
  float* s_red = s_dg + (long)G * 3 * kC16Row * 16;
  const unsigned s_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)s_x);
  f32x4 wr[9];
  {
    const float* wp = a.wp + (long)wset * a.ntaps * G * 256;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wr[tp] = ld4(wp + ((long)tp * G + wave) * 256 + lane * 4);
  }
  const int Wi = a.Wi, Hi = a.Hi;
  const long i_sH = a.i_sH, i_sW = a.i_sW;
  const int nsub = (a.Wo + 15) / 16;
  const int nb = a.B / a.wsets;
  int lo[3];
#pragma unroll
  for (int ow = 0; ow < 3; ++ow) {
    const int px = j + 2 - ow;                         // staged index of source column p - (ow - 1), staged from p0 - 1
    lo[ow] = px * 64 + ((kq ^ ((px >> 2) & 3)) * 16);
  }
  constexpr unsigned kWF = 10, kWL = 6;
  const unsigned Ho = (unsigned)a.Ho;
  const unsigned Rtot = (unsigned)nb * a.To * Ho;
  const unsigned CF = Rtot * kWF, Cfull = (unsigned)nfull * CF, Ctot = Cfull + (nfull < nstrip ? Rtot * kWL : 0u);
  const unsigned c_lo = (unsigned)((unsigned long long)Ctot * blockIdx.x / gridDim.x);
  const unsigned c_hi = (unsigned)((unsigned long long)Ctot * (blockIdx.x + 1) / gridDim.x);
  int par = 0;                                         // exchange buffer of the next row (alternates over the workgroup's life)
  for (unsigned cc = c_lo; cc < c_hi;) {
   const bool in_full = cc < Cfull;
   int sp = 0;
   if (in_full) { while (cc >= (unsigned)(sp + 1) * CF) ++sp; } else sp = nfull;
   const unsigned cbase = in_full ? (unsigned)sp * CF : Cfull;
   const unsigned cend = in_full ? (cbase + CF < c_hi ? cbase + CF : c_hi) : c_hi;
   const unsigned r_lo = in_full ? (cc - cbase + kWF - 1) / kWF : (cc - cbase + kWL - 1) / kWL;
   const unsigned r_hi = in_full ? (cend - cbase + kWF - 1) / kWF : (cend - cbase + kWL - 1) / kWL;
   cc = cend;
   for (unsigned rho = r_lo; rho < r_hi;) {
    const unsigned plane = rho / Ho;
    const int h0 = (int)(rho - plane * Ho);
    const int t = (int)(plane % (unsigned)a.To);
    const int b = (int)(plane / (unsigned)a.To) * a.wsets + wset;
    const int h1 = (int)(Ho < h0 + (r_hi - rho) ? Ho : h0 + (r_hi - rho));
    rho += h1 - h0;
    const int my_sub = min(MT, nsub - sp * MT);
    const int p0 = sp * MT * 16;
    const int rowlen = my_sub * 16 + 2;
    int res0 = -9, res1 = -9, res2 = -9;               // source row held by each ring slot
    const float* in_bt = a.in + (long)b * a.i_sB + (long)t * a.i_sT + goff(wave, a.i_gpu, a.i_sU, a.i_sG);
    long src_off0, src_off1, src_off2;
    bool src_ok0, src_ok1, src_ok2, in_row0, in_row1, in_row2;
    {
      auto piece = [&](int k, long& off, bool& ok, bool& inrow) __attribute__((always_inline)) {
        const int c = lane + 64 * k, px = c >> 2, q4 = (c & 3) ^ ((px >> 2) & 3);
        const int w = p0 + px - 1;
        inrow = c < rowlen * 4;
        ok = inrow && w >= 0 && w < Wi;
        off = (long)min(max(w, 0), Wi - 1) * i_sW + 4 * q4;
      };
      piece(0, src_off0, src_ok0, in_row0); piece(1, src_off1, src_ok1, in_row1); piece(2, src_off2, src_ok2, in_row2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_fence();                                     // the previous unit's reads and DMA are done
    {                                                 // zero columns outside the frame, all 3 ring rows
      const bool zk[3] = {in_row0 && !src_ok0, in_row1 && !src_ok1, in_row2 && !src_ok2};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (zk[k]) {
#pragma unroll
          for (int sr = 0; sr < 3; ++sr) st4(s_x + sr * (kC16Row * 16) + (lane + 64 * k) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    // DG_STAGE(HS): bring dy row HS into ring slot HS mod 3 unless it is there: DMA for a row of the frame, zeros otherwise
#define DG_STAGE(HS_)                                                                                          \
  do {                                                                                                         \
    const int hs_ = (HS_);                                                                                     \
    const int slot_ = ((hs_ % 3) + 3) % 3;                                                                     \
    const int cur_ = slot_ == 0 ? res0 : slot_ == 1 ? res1 : res2;                                             \
    if (cur_ != hs_) {                                                                                         \
      if (slot_ == 0) res0 = hs_; else if (slot_ == 1) res1 = hs_; else res2 = hs_;                            \
      if (hs_ >= 0 && hs_ < Hi) {                                                                              \
        const float* src_ = in_bt + (long)hs_ * i_sH;                                                          \
        const unsigned dst_ = s_base + (unsigned)(slot_ * kRowB);                                              \
        if (src_ok0) glds16(src_ + src_off0, dst_);                                                            \
        if (src_ok1) glds16(src_ + src_off1, dst_ + 1024);                                                     \
        if (src_ok2) glds16(src_ + src_off2, dst_ + 2048);                                                     \
      } else {                                                                                                 \
        float* dz_ = s_x + slot_ * (kC16Row * 16);                                                             \
        if (in_row0) st4(dz_ + lane * 4, f32x4{0.f, 0.f, 0.f, 0.f});                                           \
        if (in_row1) st4(dz_ + (lane + 64) * 4, f32x4{0.f, 0.f, 0.f, 0.f});                                    \
        if (in_row2) st4(dz_ + (lane + 128) * 4, f32x4{0.f, 0.f, 0.f, 0.f});                                   \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
    DG_STAGE(h0 + 1); DG_STAGE(h0); DG_STAGE(h0 - 1);

    for (int h = h0; h < h1; ++h) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wave_fence();
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      // tap (oh, ow) reads source row h - (oh - 1); the next output row replaces row h - 1 (oh = 2): its taps go first
      const char* rowq[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) rowq[oh] = reinterpret_cast<const char*>(s_x) + (((h - (oh - 1)) % 3 + 3) % 3) * kRowB;
      // the summing waves request their epilogue operands ahead of the taps (read after the workgroup barrier they were a
      // memory round trip on the critical path of every row: the other waves wait at the next barrier for waves 0 / 1)
      // (tried: rotating the summing pair over the waves with the row — no gain, 235 vs 225 us)
      const int msum = wave;                            // waves 0 / 1 sum output tile 0 / 1
      const int pe = p0 + msum * 16 + j;
      const bool sum_on = msum >= 0 && msum < my_sub && pe < a.Wo;
      const long oe = (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH + (long)pe * a.o_sW + 4 * kq;
      f32x4 rrv = {1.f, 1.f, 1.f, 1.f}, adv = {0.f, 0.f, 0.f, 0.f};
      if (sum_on && a.relu_ref) rrv = ld4(a.relu_ref + oe);
      if (sum_on && a.addend) adv = ld4(a.addend + oe);
      f32x4 xb[2][2];
      auto rd = [&](int nb_, int tp, f32x4& x0, f32x4& x1) __attribute__((always_inline)) {
        const int oh = 2 - tp / 3, ow = tp % 3;
        const char* rp = rowq[oh] + lo[ow];
        x0 = *reinterpret_cast<const f32x4*>(rp);
        if (nb_ == 2) x1 = *reinterpret_cast<const f32x4*>(rp + 1024);
      };
#define DG_TAPS(T0_, T1_, NB_)                                                                                 \
  _Pragma("unroll") for (int tp = (T0_); tp < (T1_); ++tp) {                                                   \
    if (tp + 1 < 9) rd(NB_, tp + 1, xb[(tp + 1) & 1][0], xb[(tp + 1) & 1][1]);                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    const f32x4 wf = wr[(2 - tp / 3) * 3 + tp % 3];                                                            \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                            \
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xb[tp & 1][0][s], acc[0], 0, 0, 0);                 \
      if (NB_ == 2) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], xb[tp & 1][1][s], acc[1], 0, 0, 0);   \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
      if (my_sub == 2) {
        rd(2, 0, xb[0][0], xb[0][1]);
        DG_TAPS(0, 3, 2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h + 1 < h1) DG_STAGE(h + 2);
        __builtin_amdgcn_sched_barrier(0);
        DG_TAPS(3, 9, 2)
      } else {
        rd(1, 0, xb[0][0], xb[0][1]);
        DG_TAPS(0, 3, 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h + 1 < h1) DG_STAGE(h + 2);
        __builtin_amdgcn_sched_barrier(0);
        DG_TAPS(3, 9, 1)
      }
#undef DG_TAPS
      // exchange: every wave's partial tiles, then waves 0 / 1 sum block 0 / 1 over the channel groups and store
      float* red = s_red + (long)par * G * MT * 256;
#pragma unroll
      for (int m = 0; m < MT; ++m) st4(red + ((long)wave * MT + m) * 256 + lane * 4, acc[m]);
      __syncthreads();
      if (msum >= 0 && msum < my_sub) {
        const int m = msum;
        f32x4 v = ld4(red + (long)m * 256 + lane * 4);
        for (int g = 1; g < G; ++g) v += ld4(red + ((long)g * MT + m) * 256 + lane * 4);
        if (sum_on) {
          if (a.relu_ref) {
            v[0] = rrv[0] > 0.f ? v[0] : 0.f; v[1] = rrv[1] > 0.f ? v[1] : 0.f;
            v[2] = rrv[2] > 0.f ? v[2] : 0.f; v[3] = rrv[3] > 0.f ? v[3] : 0.f;
          }
          st4(a.out + oe, v + adv);
        }
      }
      par ^= 1;
    }
#undef DG_STAGE
   }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------
// conv16_lds_kernel on split-precision MFMAs (opt-in: IDEE_CONV_MATH=split; reported as its own bench line).
// Every f32 operand is cut into two 16-bit pieces x = hi + lo (common.h split2): f16 pieces in the FORWARD modes (11 + 11
// mantissa bits: f32-level, operands are O(1); the weights are pre-scaled by a power of two into f16's normal range and the
// accumulator un-scaled in the epilogue), bf16 pieces in the DATA-GRADIENT modes (8 + 8 bits, f32's exponent range).
// A product is formed from THREE piece products hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-22 resp. 2^-16 of the
// product) on v_mfma_f32_16x16x32_{f16,bf16}: K = 32 = TWO TAPS x 16 input channels, so a pair of taps costs 3 MFMAs of 16
// cycles instead of 8 f32 MFMAs of 32 cycles.  Same work decomposition, LDS ring and epilogue as conv16_lds_kernel; the
// staged rows are kept as two piece images [quad pair][pixel][quad & 1][4 channels x 16 bit] so that a lane's 4 channels
// of one piece are one 8-byte, conflict-free ds_read_b64 (32 lanes x 8 B = one 256-byte LDS row).
// ---------------------------------------------------------------------------------------------
#ifndef IDEE_DBG_SKIP
#define IDEE_DBG_SKIP 0      // timing variants of conv16_split_kernel (never set in a shipped build): 1 no row loads, 2 no MFMA loop, 4 no stores, 8 no staging
#endif
#ifndef IDEE_RD_DEPTH
#define IDEE_RD_DEPTH 1
#endif
constexpr int kSplitRowW = 2 * kC16Row * 4;     // 32-bit words of one staged row of ONE piece: [2 quad pairs][34 px][4 words]

// GSC (IDEE_CONV_MATH=f22, data-gradient modes with f16 pieces): the staged dy rows carry a power-of-two factor 2^k kept per
// WORK UNIT (32 pixels x a run of rows x the 3 source frames of one batch item) such that the unit's largest |dy| 2^k seen so
// far lies below 2^8 (common.h GradScale): every newly staged row is tested on the staging registers (one compare and the wave's
// vote); a row that would leave the window at the top lowers k (its largest |dy| 2^k to [2^4, 2^5); the rows already in the ring
// were below 2^8 at the larger scale, so they stay in range), the ring is invalidated and the output row is begun again — rare:
// the scale of a unit only moves down; the first non-zero row of a unit sets it.  The epilogue multiplies by 2^-k.
// Operand rounding max(2^-22 |dy|, 2^-26 of the unit's largest |dy| so far); weights prescaled per set as in the forward.
// SC = SC_ACT (f16 pieces, forward modes): the staged ACTIVATION rows are range-guarded instead (common.h ActGuard): split as they
// are (k = 0) until a staged row's largest |x| reaches 2^15, then the unit's rows carry 2^k, k < 0, through the same restaging
// path — an input of any finite magnitude keeps two normal f16 pieces.
// NB output blocks per wave (one input block; blockIdx.z picks the group of NB blocks: 16 -> 16 NB gridDim.z channels, the
// weights of all NB blocks in registers) and KTS temporal taps of the zero-padded forward mode: NB = 2, KTS = 1 is the (1,3,3)
// sign-field forward 16 -> 96 (K = 9 taps x 16 channels in 5 tap pairs, 3 piece products each; its f32 kernel ran at 80 TFLOP/s
// for a launch whose memory time is a quarter of that).
template <int MODE, bool F16, int SC = SC_NONE, int NB = 1, int KTS = 2>
__global__ __launch_bounds__(256, 2) void conv16_split_kernel(ConvArgs a, int nstrip, int hrun, int nhrun) {
  constexpr bool GSC = SC == SC_GRAD, ASC = SC == SC_ACT;
  static_assert(!GSC || (F16 && (MODE == DGRAD_REP || MODE == DGRAD_S2)), "gradient scale: f16 pieces, data-gradient modes");
  static_assert(!ASC || (F16 && (MODE == FWD_REP || MODE == FWD_S2)), "activation guard: f16 pieces, forward modes");
  static_assert(NB == 1 || MODE == FWD_S2, "several output blocks per wave: zero-padded forward mode");
  constexpr int MT = 2;
  constexpr bool REP = MODE == FWD_REP || MODE == DGRAD_REP;
  constexpr int TSN = REP ? 3 : (MODE == FWD_S2 ? KTS : 1);
  constexpr int NTAP = TSN * 9, NPAIR = (NTAP + 1) / 2;
  constexpr int PIECE = 3 * TSN * kSplitRowW;     // words between the hi and the lo piece image of a wave's ring
  __shared__ __attribute__((aligned(16))) unsigned s_xw[4][2 * PIECE];   // [piece][T-slot * 3 + ring slot][row]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  const int gb0 = blockIdx.z * NB;     // first output block of this workgroup
  unsigned* s_x = s_xw[wave];
  u32x4 wh[NPAIR][NB], wl[NPAIR][NB];       // A fragments of a tap pair: [piece of tap a | piece of tap b]
  int wkey = -1;
  // packed layout [tap][input block][output block][lane][4]: one input block here, so a tap is GnTot fragments
  const int wtap = NB == 1 ? 256 : a.GnTot * 256;     // (NB = 1: the 16 -> 16 convolutions, one output block in all)
  const float* wp = a.wp + (long)wset * a.ntaps * wtap;
  // power-of-two weight scale (f16 pieces only): max |w| of this weight set lands in [2^13, 2^14)
  float wscale = 1.f, winv = 1.f;
  if (F16) {
    float mx = 0.f;
    // (nine loads in flight per trip — tap counts are multiples of 9: one load and its wait per trip made this scan a chain of
    //  27 .. 54 memory round trips at the start of every wave, a quarter of the (1,3,3) forward's time)
    const int nfr = NB == 1 ? a.ntaps : a.ntaps * a.GnTot;
    for (int tp = 0; tp < nfr; tp += 9) {
      f32x4 f[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) f[u] = ld4(wp + (tp + u < nfr ? tp + u : tp) * 256 + lane * 4);
#pragma unroll
      for (int u = 0; u < 9; ++u) mx = amax_acc(mx, f[u]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    int e = 0;
    frexpf(fmaxf(mx, 1e-30f), &e);            // mx = m * 2^e, m in [0.5, 1)
    wscale = ldexpf(1.f, 14 - e);
    winv = ldexpf(1.f, e - 14);
  }
  // the bias of this lane's output channels, once (a load in the epilogue is a memory round trip per row: its s_waitcnt vmcnt(0)
  // also waits for the stores issued just before it)
  f32x4 biasv[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
    biasv[n] = a.bias ? ld4(a.bias + (NB == 1 ? (long)wset * 16 : (long)wset * a.GnTot * 16 + (gb0 + n) * 16) + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int Wframe = (MODE == DGRAD_REP) ? a.Wo + 2 : a.Wo;
  const int nsub = (Wframe + 15) / 16;
  const int nb = a.B / a.wsets;
  const long nunits = (long)nb * a.To * nhrun * nstrip;
  // word offset of this lane's 8-byte fragment piece inside a staged row, for pixel index 0: [kq >> 1][pixel][kq & 1][2 words]
  const int lane_off = (kq >> 1) * (kC16Row * 4) + (kq & 1) * 2;

  for (long un = (long)blockIdx.x * 4 + wave; un < nunits; un += (long)gridDim.x * 4) {
    long r = un;      // t fastest: see conv16_lds_kernel
    const int t = (int)(r % a.To); r /= a.To;
    const int sp = (int)(r % nstrip); r /= nstrip;
    const int hr = (int)(r % nhrun);
    const int b = (int)(r / nhrun) * a.wsets + wset;
    const int my_sub = min(MT, nsub - sp * MT);
    const int p0 = sp * MT * 16;
    const int rowlen = my_sub * 16 + 2;
    const int h0 = hr * hrun, h1 = min(a.Ho, h0 + hrun);
    int res0 = -1, res1 = -1, res2 = -1;
    const float* in_b = a.in + (long)b * a.i_sB;
    Slot st[3];
#pragma unroll
    for (int ot = 0; ot < 3; ++ot) st[ot] = slot_t<MODE>(ot, t, a.Ti, a.kt);
    const int blk1 = my_sub > 1 ? 16 : 0;
    GradScale gs;                       // (GSC) this unit's gradient scale
    ActGuard ag;                        // (ASC) this unit's activation guard

    // piece k of source row hs of source frame ot -> registers (zeros outside the frame)
    auto load_one = [&](int hs, bool inside, int ot, int k) __attribute__((always_inline)) {
      const float* src = in_b + (long)st[ot].src * a.i_sT + (long)hs * a.i_sH;
      const bool live = inside && st[ot].ok;
      const int u = lane + 64 * k;
      const int px = u >> 2, q4 = u & 3;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (u < rowlen * 4 && live && !(IDEE_DBG_SKIP & 1)) {
        if (MODE == FWD_REP) {
          const int w = min(max(p0 + px - 1, 0), a.Wi - 1);
          v = ld4(src + (long)w * a.i_sW + 4 * q4);
        } else {
          const int w = p0 + px - (MODE == DGRAD_REP ? 2 : 1);
          if (w >= 0 && w < a.Wi) v = ld4(src + (long)w * a.i_sW + 4 * q4);
        }
      }
      return v;
    };
    auto load_row = [&](int hs, bool inside, f32x4 (&stg)[TSN][3]) __attribute__((always_inline)) {
#pragma unroll
      for (int ot = 0; ot < TSN; ++ot)
#pragma unroll
        for (int k = 0; k < 3; ++k) stg[ot][k] = load_one(hs, inside, ot, k);
    };
    auto row_amax = [&](const f32x4 (&stg)[TSN][3]) __attribute__((always_inline)) {
      float am = 0.f;
#pragma unroll
      for (int ot = 0; ot < TSN; ++ot)
#pragma unroll
        for (int k = 0; k < 3; ++k) am = amax_acc(am, stg[ot][k]);
      return am;
    };

    for (int h = h0; h < h1;) {
      wave_fence();
      Slot sh[3];
      int slot_of[3];
      bool redo = false;
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) sh[oh] = slot_h<MODE>(oh, h, a.Hi);
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const int hs = sh[oh].src;
        const bool inside = sh[oh].ok;
        const int slot = (hs + 3) % 3;
        slot_of[oh] = slot;
        const int cur = slot == 0 ? res0 : slot == 1 ? res1 : res2;
        if (cur == hs + 1) continue;
        f32x4 stg[TSN][3];
        load_row(hs, inside, stg);
        if constexpr (GSC) {
          const float am = row_amax(stg);
          // (a threshold test needs no reduction: one compare and the wave's vote)
          if (gs.set ? __any(am * gs.s() >= 256.f) : __any(am > 0.f)) {
            gs.set_from(wave_umax(__float_as_uint(am)));
            res0 = res1 = res2 = -1;
            redo = true;
            break;
          }
          const float gsc = gs.s();
#pragma unroll
          for (int ot = 0; ot < TSN; ++ot)
#pragma unroll
            for (int k = 0; k < 3; ++k) stg[ot][k] *= gsc;
        }
        if constexpr (ASC) {
          const float am = row_amax(stg);
          if (ag.exceeds(am)) {           // (rare: |x| >= 2^15 at the unit's scale)
            ag.lower(wave_umax(__float_as_uint(ActGuard::finite(am))));
            res0 = res1 = res2 = -1;
            redo = true;
            break;
          }
          if (ag.k != 0) {                // wave-uniform
            const float gsc = ag.s();
#pragma unroll
            for (int ot = 0; ot < TSN; ++ot)
#pragma unroll
              for (int k = 0; k < 3; ++k) stg[ot][k] *= gsc;
          }
        }
        if (slot == 0) res0 = hs + 1; else if (slot == 1) res1 = hs + 1; else res2 = hs + 1;
#pragma unroll
        for (int ot = 0; ot < TSN; ++ot) {
          unsigned* dst = s_x + (ot * 3 + slot) * kSplitRowW;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            if (u < rowlen * 4 && (!(IDEE_DBG_SKIP & 8) || stg[ot][k][0] == 1.2345e-30f)) {
              const Split p = split2<F16>(stg[ot][k]);
              const int o = ((u & 3) >> 1) * (kC16Row * 4) + (u >> 2) * 4 + (u & 1) * 2;
              typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
              *reinterpret_cast<u32x2v*>(dst + o) = u32x2v{p.h01, p.h23};
              *reinterpret_cast<u32x2v*>(dst + PIECE + o) = u32x2v{p.l01, p.l23};
            }
          }
        }
      }
      if ((GSC || ASC) && redo) continue;   // (h not advanced: the row is staged again at the new scale)
      wave_fence();
      {
        const int key = ((((st[0].tap * 3 + st[1].tap) * 3 + st[2].tap) * 3 + sh[0].tap) * 3 + sh[1].tap) * 3 + sh[2].tap;
        if (key != wkey) {
          wkey = key;
#pragma unroll
          for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
              Split sa, sb = {0u, 0u, 0u, 0u};
              {
                const int tp = 2 * pr, ot = tp / 9, oh = (tp / 3) % 3, ow = tp % 3;
                sa = split2<F16>(ld4(wp + ((st[ot].tap * 3 + sh[oh].tap) * 3 + ow) * wtap + (gb0 + n) * 256 + lane * 4) * wscale);
              }
              if (2 * pr + 1 < NTAP) {
                const int tp = 2 * pr + 1, ot = tp / 9, oh = (tp / 3) % 3, ow = tp % 3;
                sb = split2<F16>(ld4(wp + ((st[ot].tap * 3 + sh[oh].tap) * 3 + ow) * wtap + (gb0 + n) * 256 + lane * 4) * wscale);
              }
              wh[pr][n] = u32x4{sa.h01, sa.h23, sb.h01, sb.h23};
              wl[pr][n] = u32x4{sa.l01, sa.l23, sb.l01, sb.l23};
            }
        }
      }
      f32x4 acc[MT][NB];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      // word address of (T-slot ot, tap row oh, pixel shift ow) for this lane, piece 0, tile 0
      auto addr = [&](int tp) __attribute__((always_inline)) {
        const int ot = tp / 9, oh = (tp / 3) % 3, ow = tp % 3;
        const int sh_px = (MODE == FWD_REP || MODE == FWD_S2) ? ow : 2 - ow;
        return (ot * 3 + slot_of[oh]) * kSplitRowW + lane_off + (j + sh_px) * 4;
      };
      typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
      auto rd = [&](int pr, u32x4 (&bh)[MT], u32x4 (&bl)[MT]) __attribute__((always_inline)) {
        const int aa = addr(2 * pr), ab = addr(2 * pr + 1 < NTAP ? 2 * pr + 1 : 2 * pr);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int mo = m ? blk1 * 4 : 0;
          const u32x2v ha = *reinterpret_cast<const u32x2v*>(s_x + aa + mo), hb = *reinterpret_cast<const u32x2v*>(s_x + ab + mo);
          const u32x2v la = *reinterpret_cast<const u32x2v*>(s_x + PIECE + aa + mo),
                       lb = *reinterpret_cast<const u32x2v*>(s_x + PIECE + ab + mo);
          bh[m] = u32x4{ha[0], ha[1], hb[0], hb[1]};
          bl[m] = u32x4{la[0], la[1], lb[0], lb[1]};
        }
      };
      // epilogue operands (ReLU reference, addend) requested AHEAD of the taps: loaded between the row's stores they cost a
      // memory round trip each (s_waitcnt vmcnt(0) also waits for the stores before them)
      float* out_r = a.out + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH;
      const float* ref_r = a.relu_ref ? a.relu_ref + (long)b * a.o_sB + (long)t * a.o_sT + (long)h * a.o_sH : nullptr;
      // (data-gradient modes; the forward modes take neither in practice and read them, if given, in the epilogue)
      constexpr bool EARLY = MODE == DGRAD_REP || MODE == DGRAD_S2;
      long eo[MT];
      bool eon[MT], eat[MT];
      f32x4 rrv[EARLY ? MT : 1][NB], adv[EARLY ? MT : 1][NB];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int p = p0 + m * 16 + j;
        eon[m] = m < my_sub && p < Wframe;
        int wdst = p;
        eat[m] = false;
        if (MODE == DGRAD_REP) { wdst = min(max(p - 1, 0), a.Wo - 1); eat[m] = ((p <= 1) || (p >= a.Wo)) && !a.no_edge_atomics; }
        eo[m] = (long)wdst * a.o_sW + 4 * kq;
        if constexpr (EARLY) {
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const long o = eo[m] + (NB == 1 ? 0 : goff(gb0 + n, a.o_gpu, a.o_sU, a.o_sG));
            rrv[m][n] = (ref_r && eon[m]) ? ld4(ref_r + o) : f32x4{1.f, 1.f, 1.f, 1.f};
            adv[m][n] = (a.addend && eon[m] && !eat[m]) ? ld4(a.addend + (out_r - a.out) + o) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
      constexpr int RD = IDEE_RD_DEPTH;      // tap pairs read ahead of their MFMAs
      u32x4 bh[RD + 1][MT], bl[RD + 1][MT];
#pragma unroll
      for (int pr = 0; pr < RD && pr < NPAIR && !(IDEE_DBG_SKIP & 2); ++pr) rd(pr, bh[pr], bl[pr]);
      // (scheduling fences: left alone the compiler sinks every fragment read to just before its MFMA — ds_read, s_waitcnt
      //  lgkmcnt(0..2), v_mfma — and the LDS round trip is exposed at each of the 14 tap pairs of a row: most of these kernels' time)
#pragma unroll
      for (int pr = 0; pr < ((IDEE_DBG_SKIP & 2) ? 0 : NPAIR); ++pr) {
        if (pr + RD < NPAIR) rd(pr + RD, bh[(pr + RD) % (RD + 1)], bl[(pr + RD) % (RD + 1)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = mfma_s<F16>(wh[pr][n], bh[pr % (RD + 1)][m], acc[m][n]);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = mfma_s<F16>(wh[pr][n], bl[pr % (RD + 1)][m], acc[m][n]);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][n] = mfma_s<F16>(wl[pr][n], bh[pr % (RD + 1)][m], acc[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue (as conv16_lds_kernel; the f16 weight scale is taken out first)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if constexpr (MODE == DGRAD_REP) {
          // The virtual frame columns -1 and W (replicate padding) belong to the real columns 0 and W - 1: frame column p = 0 is
          // added to p = 1 and p = W + 1 to p = W by a DPP row shift (both lanes of a pair sit in one 16-column tile: the host
          // sets no_edge_atomics only when W % 16 != 15) — no atomics and no pre-initialised edge columns (the edge lanes'
          // atomics, 12 per row in a divergent branch, were 0.08 of the 3^3 data gradient's 0.71 ms)
          if (a.no_edge_atomics && !fold_w_edges(acc[m][0], p0 + m * 16 + j, a.Wo)) eon[m] = false;      // (donors: folded into their neighbours)
        }
        if (!eon[m]) continue;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          f32x4 v = F16 ? acc[m][n] * (GSC ? winv * gs.inv() : ASC ? winv * ag.inv() : winv) : acc[m][n];
          v += biasv[n];
          if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
          const long o = eo[m] + (NB == 1 ? 0 : goff(gb0 + n, a.o_gpu, a.o_sU, a.o_sG));
          if (ref_r) {
            const f32x4 rr = EARLY ? rrv[EARLY ? m : 0][n] : ld4(ref_r + o);
            v[0] = rr[0] > 0.f ? v[0] : 0.f; v[1] = rr[1] > 0.f ? v[1] : 0.f;
            v[2] = rr[2] > 0.f ? v[2] : 0.f; v[3] = rr[3] > 0.f ? v[3] : 0.f;
          }
          if (eat[m]) {
            atomicAdd(out_r + o, v[0]); atomicAdd(out_r + o + 1, v[1]);
            atomicAdd(out_r + o + 2, v[2]); atomicAdd(out_r + o + 3, v[3]);
          } else {
            if (EARLY) v += adv[EARLY ? m : 0][n];
            else if (a.addend) v += ld4(a.addend + (out_r - a.out) + o);
            if (!(IDEE_DBG_SKIP & 4) || v[0] == 1.2345e-30f) st4(out_r + o, v);
          }
        }
      }
      ++h;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient of the 16 -> 16 3x3x3 replicate convolution, every WAVE on its own (the structure of
// conv16_lds_kernel): a wave owns a strip of 2 column blocks (32 pixels) and walks a run of rows of one (batch item, t),
// keeping a private LDS ring of 3 source rows x 3 T-slots (pixel-major [pixel][16 channels]: the A operand of a
// K = pixels MFMA is 64 consecutive floats, conflict-free).  dy fragments come straight from global memory (one
// coalesced 256-byte load per 4 pixels).  All 27 tap accumulators (16 x 16 each) stay in registers for the wave's
// whole share and are flushed once with atomics.  No workgroup barrier anywhere.
// ---------------------------------------------------------------------------------------------
// TSL T-slots per output row: 3 (3x3x3 replicate), 2 ((2,3,3) stride 2) or 1 ((1,3,3)); NB 16-channel output blocks per
// wave (blockIdx.z picks the group of NB blocks): 16 -> 16*NB*gridDim.z channels, TSL*9*NB accumulator tiles
template <int MODE /*FWD_REP: 3x3x3 replicate ; FWD_S2: (kt,3,3) zero padding*/, int TSL, int NB>
__global__ __launch_bounds__(256, 2) void wgrad16_strip_kernel(WgradArgs a, int nstrip, int hrun, int nhrun) {
  constexpr int NTAPS = TSL * 9;
  static_assert(NTAPS * NB <= 27, "accumulator tiles must fit the register file");
  __shared__ __attribute__((aligned(16))) float s_xw[4][TSL * 3 * kC16Row * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  float* s_x = s_xw[wave];
  // blockIdx.z = (input 16-channel block) * (output block groups) + (output block group): any Cin, Cout multiples of 16
  const int ngb = (a.CoutPad / 16) / NB;
  const int ga = blockIdx.z / ngb;            // input block of this workgroup
  const int gb0 = (blockIdx.z % ngb) * NB;    // first output block of this workgroup
  f32x4 acc[NTAPS][NB];
  float bsum[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    bsum[n] = 0.f;
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp) acc[tp][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nsub = (a.W + 15) / 16;
  const int nb = a.B / a.wsets;
  const long nunits = (long)nb * a.To * nhrun * nstrip;

  for (long un = (long)blockIdx.x * 4 + wave; un < nunits; un += (long)gridDim.x * 4) {
    long r = un;      // t fastest: see conv16_lds_kernel
    const int t = (int)(r % a.To); r /= a.To;
    const int sp = (int)(r % nstrip); r /= nstrip;
    const int hr = (int)(r % nhrun);
    const int b = (int)(r / nhrun) * a.wsets + wset;
    const int my_sub = min(2, nsub - sp * 2);
    const int p0 = sp * 32;
    const int rowlen = my_sub * 16 + 2;
    const int h0 = hr * hrun, h1 = min(a.H, h0 + hrun);
    int res0 = -1, res1 = -1, res2 = -1;
    const float* x_b = a.x + (long)b * a.x_sB + goff(ga, a.x_gpu, a.x_sU, a.x_sG);
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT;
    int tsrc[TSL];
#pragma unroll
    for (int ot = 0; ot < TSL; ++ot) tsrc[ot] = MODE == FWD_REP ? min(max(t + ot - 1, 0), a.T - 1) : a.kt * t + ot;

    for (int h = h0; h < h1; ++h) {
      wave_fence();   // previous row's reads are ordered before this row's staging writes
      // dy fragments of this row: lane (cout = r16, pixel 4*kb + kq); pixels past the row end contribute zero
      float bf[8][NB];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const int p = p0 + 4 * kb + kq;
        const bool on = kb < 4 * my_sub && p < a.W;
#pragma unroll
        for (int n = 0; n < NB; ++n)
          bf[kb][n] = on ? dy_b[(long)h * a.y_sH + (long)p * a.y_sW + goff(gb0 + n, a.y_gpu, a.y_sU, a.y_sG) + r16] : 0.f;
      }
      int slot_of[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        // replicate: the clamped row ; zero padding: rows -1 and H are staged as zeros (ring key hs + 1 >= 0)
        const int hs = MODE == FWD_REP ? min(max(h + oh - 1, 0), a.H - 1) : h + oh - 1;
        const bool inside = hs >= 0 && hs < a.H;
        const int slot = (hs + 3) % 3;
        slot_of[oh] = slot;
        const int cur = slot == 0 ? res0 : slot == 1 ? res1 : res2;
        if (cur == hs + 1) continue;
        if (slot == 0) res0 = hs + 1; else if (slot == 1) res1 = hs + 1; else res2 = hs + 1;
        f32x4 stg[TSL][3];      // all loads of the T-slots first, then the LDS writes
#pragma unroll
        for (int ot = 0; ot < TSL; ++ot) {
          const float* src = x_b + (long)tsrc[ot] * a.x_sT + (long)hs * a.x_sH;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < rowlen * 4 && inside) {
              const int w = p0 + (u >> 2) - 1;
              if (MODE == FWD_REP) v = ld4(src + (long)min(max(w, 0), a.W - 1) * a.x_sW + 4 * (u & 3));
              else if (w >= 0 && w < a.W) v = ld4(src + (long)w * a.x_sW + 4 * (u & 3));
            }
            stg[ot][k] = v;
          }
        }
#pragma unroll
        for (int ot = 0; ot < TSL; ++ot) {
          float* dst = s_x + (long)(ot * 3 + slot) * kC16Row * 16;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            if (u < kC16Row * 4) st4(dst + u * 4, stg[ot][k]);     // pixels past a one-block strip are written as zeros
          }
        }
      }
      wave_fence();
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int n = 0; n < NB; ++n) bsum[n] += bf[kb][n];
      // acc[tap][cin][cout] += sum over the strip's pixels of x[pixel + tap offset][cin] dy[pixel][cout]
      const float* rowq[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) rowq[oh] = s_x + ((long)slot_of[oh] * kC16Row + kq) * 16 + r16;
      // (a one-block strip multiplies its absent second block by zero dy fragments: branch-free tap loop)
#pragma unroll
      for (int ot = 0; ot < TSL; ++ot)
#pragma unroll
        for (int oh = 0; oh < 3; ++oh)
#pragma unroll
          for (int ow = 0; ow < 3; ++ow) {
            const float* ap = rowq[oh] + (ot * 3 * kC16Row + ow) * 16;
            const int tp = (ot * 3 + oh) * 3 + ow;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
              const float af = ap[kb * 64];
#pragma unroll
              for (int n = 0; n < NB; ++n) acc[tp][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[kb][n], acc[tp][n], 0, 0, 0);
            }
          }
    }
  }
  float* dw = a.dwp + ((long)wset * NTAPS * a.Cin + ga * 16) * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(dw + ((long)tp * a.Cin + r0 + r) * a.CoutPad + (gb0 + n) * 16 + col, acc[tp][n][r]);
  if (a.dbias && ga == 0) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float v = bsum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + (gb0 + n) * 16 + lane, v);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// wgrad16_strip_kernel<FWD_REP, 3, 1> (the 3x3x3 replicate 16 -> 16 weight gradient) with the x-row staging of
// conv16_glds_kernel: the next row's three new source rows arrive by LDS-DMA (no staging registers; the image is the same
// pixel-major [pixel][4 chunks] the register path writes, which is lane-linear) after the 9 taps that read the ring slot being
// replaced; the registers that frees pay for the cost-axis work partition.  One (input block, output block) pair per grid.z slice.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad16_glds_kernel(WgradArgs a, int nstrip, int nfull, int wl_arg) {
  constexpr int TSL = 3, NTAPS = 27;
  constexpr int kRowB = kC16Row * 64;
  __shared__ __attribute__((aligned(1024))) float s_xw[4][TSL * 3 * kC16Row * 16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  float* s_x = s_xw[wave];
  const unsigned s_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)s_x);
  const int ngb = a.CoutPad / 16;
  const int ga = blockIdx.z / ngb, gb0 = blockIdx.z % ngb;
  f32x4 acc[NTAPS];
  float bsum = 0.f;
#pragma unroll
  for (int tp = 0; tp < NTAPS; ++tp) acc[tp] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsub = (a.W + 15) / 16;
  const int nb = a.B / a.wsets;
  const int Wd = a.W;
  const long x_sT = a.x_sT, x_sH = a.x_sH, x_sW = a.x_sW;
  // work partition: the cost axis of conv16_glds_kernel (frame groups of 4 when To allows it)
  constexpr unsigned kWF = 10;
  const unsigned kWL = (unsigned)wl_arg;
  const bool grouped = (a.To & 3) == 0;
  const int TG = grouped ? a.To / 4 : a.To;
  const unsigned Ho = (unsigned)a.H;
  const unsigned Rtot = (unsigned)nb * TG * Ho;
  const unsigned CF = Rtot * kWF, Cfull = (unsigned)nfull * CF, Ctot = Cfull + (nfull < nstrip ? Rtot * kWL : 0u);
  const unsigned part = grouped ? blockIdx.x : blockIdx.x * 4 + wave, nparts = grouped ? gridDim.x : gridDim.x * 4;
  const unsigned c_lo = (unsigned)((unsigned long long)Ctot * part / nparts);
  const unsigned c_hi = (unsigned)((unsigned long long)Ctot * (part + 1) / nparts);
  for (unsigned cc = c_lo; cc < c_hi;) {
   const bool in_full = cc < Cfull;
   int sp = 0;
   if (in_full) { while (cc >= (unsigned)(sp + 1) * CF) ++sp; } else sp = nfull;
   const unsigned cbase = in_full ? (unsigned)sp * CF : Cfull;
   const unsigned cend = in_full ? (cbase + CF < c_hi ? cbase + CF : c_hi) : c_hi;
   const unsigned r_lo = in_full ? (cc - cbase + kWF - 1) / kWF : (cc - cbase + kWL - 1) / kWL;
   const unsigned r_hi = in_full ? (cend - cbase + kWF - 1) / kWF : (cend - cbase + kWL - 1) / kWL;
   cc = cend;
   for (unsigned rho = r_lo; rho < r_hi;) {
    const unsigned plane = rho / Ho;
    const int h0 = (int)(rho - plane * Ho);
    const int tg = (int)(plane % (unsigned)TG);
    const int b = (int)(plane / (unsigned)TG) * a.wsets + wset;
    const int h1 = (int)(Ho < h0 + (r_hi - rho) ? Ho : h0 + (r_hi - rho));
    rho += h1 - h0;
    const int t = grouped ? tg * 4 + wave : tg;
    const int my_sub = min(2, nsub - sp * 2);
    const int p0 = sp * 32;
    const int rowlen = my_sub * 16 + 2;
    int res0 = -1, res1 = -1, res2 = -1;
    const float* x_b = a.x + (long)b * a.x_sB + goff(ga, a.x_gpu, a.x_sU, a.x_sG);
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT + goff(gb0, a.y_gpu, a.y_sU, a.y_sG) + r16;
    int tsrc[TSL];
#pragma unroll
    for (int ot = 0; ot < TSL; ++ot) tsrc[ot] = min(max(t + ot - 1, 0), a.T - 1);
    // per-lane source of the three DMA pieces of a row: chunk c = lane + 64 k -> pixel c >> 2, 16-byte piece c & 3
    long src_off0, src_off1, src_off2;
    bool src_ok0, src_ok1, src_ok2;
    {
      auto piece = [&](int k, long& off, bool& ok) __attribute__((always_inline)) {
        const int c = lane + 64 * k, px = c >> 2;
        ok = c < rowlen * 4;
        off = (long)min(max(p0 + px - 1, 0), Wd - 1) * x_sW + 4 * (c & 3);
      };
      piece(0, src_off0, src_ok0); piece(1, src_off1, src_ok1); piece(2, src_off2, src_ok2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_fence();
    if (my_sub < 2) {      // pixels past a one-block strip: zeros (they meet zero dy fragments; stale LDS could hold NaN patterns)
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const int c = lane + 64 * kz;
        if (c >= rowlen * 4 && c < kC16Row * 4) {
#pragma unroll
          for (int sr = 0; sr < TSL * 3; ++sr) st4(s_x + sr * (kC16Row * 16) + c * 4, f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
    }
#define WG_STAGE(H_)                                                                                           \
  _Pragma("unroll") for (int oh_ = 0; oh_ < 3; ++oh_) {                                                        \
    const int hs_ = min(max((H_) + oh_ - 1, 0), a.H - 1);                                                      \
    const int slot_ = hs_ % 3;                                                                                 \
    const int cur_ = slot_ == 0 ? res0 : slot_ == 1 ? res1 : res2;                                             \
    if (cur_ != hs_ + 1) {                                                                                     \
      if (slot_ == 0) res0 = hs_ + 1; else if (slot_ == 1) res1 = hs_ + 1; else res2 = hs_ + 1;                \
      _Pragma("unroll") for (int ot_ = 0; ot_ < TSL; ++ot_) {                                                  \
        const float* src_ = x_b + (long)tsrc[ot_] * x_sT + (long)hs_ * x_sH;                                   \
        const unsigned dst_ = s_base + (unsigned)((ot_ * 3 + slot_) * kRowB);                                  \
        if (src_ok0) glds16(src_ + src_off0, dst_);                                                            \
        if (src_ok1) glds16(src_ + src_off1, dst_ + 1024);                                                     \
        if (src_ok2) glds16(src_ + src_off2, dst_ + 2048);                                                     \
      }                                                                                                        \
    }                                                                                                          \
  }
    WG_STAGE(h0)

    for (int h = h0; h < h1; ++h) {
      // dy fragments of this row: lane (cout = r16, pixel 4 kb + kq); pixels past the row end contribute zero.  (Ordinary
      // loads, younger than the row's DMA: the compiler's counted waits for them also retire the DMA.)
      float bf[8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const int p = p0 + 4 * kb + kq;
        bf[kb] = (kb < 4 * my_sub && p < Wd) ? dy_b[(long)h * a.y_sH + (long)p * a.y_sW] : 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wave_fence();
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) bsum += bf[kb];
      const float* rowq[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) rowq[oh] = s_x + ((long)(min(max(h + oh - 1, 0), a.H - 1) % 3) * kC16Row + kq) * 16 + r16;
      // taps of the source row the next output row replaces (oh = 0) first; then its DMA; then the rest
#define WG_TAPS(OH0_, OH1_)                                                                                    \
  _Pragma("unroll") for (int oh = (OH0_); oh < (OH1_); ++oh)                                                   \
    _Pragma("unroll") for (int ot = 0; ot < TSL; ++ot)                                                         \
      _Pragma("unroll") for (int ow = 0; ow < 3; ++ow) {                                                       \
        const float* ap = rowq[oh] + (ot * 3 * kC16Row + ow) * 16;                                             \
        const int tp = (ot * 3 + oh) * 3 + ow;                                                                 \
        _Pragma("unroll") for (int kb = 0; kb < 8; ++kb)                                                       \
          acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kb * 64], bf[kb], acc[tp], 0, 0, 0);               \
      }
      // (a one-block last strip multiplies its absent pixels by zero dy fragments: skipping them — a second code path, or a
      // branch around the second half of the K-chunks — cost more in registers / scheduling than the 12 % of MFMAs it saves)
      WG_TAPS(0, 1)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (h + 1 < h1) { WG_STAGE(h + 1) }
      WG_TAPS(1, 3)
#undef WG_TAPS
    }
#undef WG_STAGE
   }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float* dw = a.dwp + ((long)wset * NTAPS * a.Cin + ga * 16) * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(dw + ((long)tp * a.Cin + r0 + r) * a.CoutPad + gb0 * 16 + col, acc[tp][r]);
  if (a.dbias && ga == 0) {
    float v = bsum;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + gb0 * 16 + lane, v);
  }
}


// ---------------------------------------------------------------------------------------------
// wgrad16_strip_kernel on split-precision MFMAs (opt-in, IDEE_CONV_MATH=split): bf16 pieces (gradients have no a-priori
// scale), three piece products per tap, K = the strip's 32 pixels in ONE v_mfma_f32_16x16x32_bf16: 81 MFMAs of 16 cycles
// per output row instead of 216 f32 MFMAs of 32 cycles.  The x rows are staged as token-major piece images
// [pixel][16 channels hi | 16 channels lo] (64 B per pixel) and read back TRANSPOSED (ds_read_b64_tr_b16: lane (channel i,
// group g) gets pixels 8g..8g+7 of channel i, shifted by the tap); dy comes from global memory as before and is split in
// registers.  16 -> 16 channels only (NB = 1).
// ---------------------------------------------------------------------------------------------
// F16 (IDEE_CONV_MATH=f22): f16 pieces; the dy row of every output row (16 channels x 32 pixels, in registers) is multiplied by
// a power of two kept per wave with common.h GradScale — the row's largest |dy| 2^k in [2^1, 2^8), the 9 TSL tap accumulators
// multiplied along when k moves, by 2^-k at the flush; the x rows (activations) are split as they are under a range guard
// (common.h ActGuard: per work unit, k = 0 until a staged row's largest |x| reaches 2^15; then the unit's rows are staged again
// with 2^k, k < 0, and the accumulators, which carry 2^(k_dy + k_x), are multiplied along).
// NB output blocks per wave (blockIdx.z picks the group; 16 input channels): NB = 3 for the (1,3,3) sign-field convs 16 -> 48 k.
// (NB = 1, the 16 -> 16 convolutions, now runs on wgrad16_shift_kernel below: a third of the LDS reads; this kernel keeps NB = 3,
//  whose three shifted dy operands per output block would not fit the register file.)
template <int MODE /*FWD_REP | FWD_S2*/, int TSL, bool F16 = false, int NB = 1>
__global__ __launch_bounds__(256, 2) void wgrad16_split_kernel(WgradArgs a, int nstrip, int hrun, int nhrun) {
  constexpr int NTAPS = TSL * 9;
  static_assert(NTAPS * NB <= 27, "accumulator tiles must fit the register file");
  const int gb0 = blockIdx.z * NB;                 // first output block of this workgroup (one input block: Cin = 16)
  constexpr int ROWW = kC16Row * 16;              // 32-bit words per staged row: 34 pixels x (8 words hi + 8 words lo)
  __shared__ __attribute__((aligned(16))) unsigned s_xw[4][TSL * 3 * ROWW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  unsigned* s_x = s_xw[wave];
  f32x4 acc[NTAPS][NB];
  float bsum[NB];
  GradScale gs;
  ActGuard ag;                                     // (F16) range guard of the x rows, moving down only over the wave's whole share
  auto scale_acc = [&](float f) __attribute__((always_inline)) {
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[tp][n] *= f;
  };
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    bsum[n] = 0.f;
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp) acc[tp][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nsub = (a.W + 15) / 16;
  const int nb = a.B / a.wsets;
  const long nunits = (long)nb * a.To * nhrun * nstrip;
  // transposed-read address of this lane inside a staged row, for pixel shift 0: lane 4q + p of group g supplies row
  // (pixel) 8g + q [+ 4 for the second read], columns (channels) 4p .. 4p + 3
  const int tr_off = (8 * kq + (r16 >> 2)) * 16 + 2 * (r16 & 3);
  typedef short s16x4t __attribute__((ext_vector_type(4)));
  auto tr = [&](const unsigned* p) __attribute__((always_inline)) {
    const s16x4t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4t __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(unsigned long long, t);
  };

  for (long un = (long)blockIdx.x * 4 + wave; un < nunits; un += (long)gridDim.x * 4) {
    long r = un;      // t fastest: see conv16_lds_kernel
    const int t = (int)(r % a.To); r /= a.To;
    const int sp = (int)(r % nstrip); r /= nstrip;
    const int hr = (int)(r % nhrun);
    const int b = (int)(r / nhrun) * a.wsets + wset;
    const int my_sub = min(2, nsub - sp * 2);
    const int p0 = sp * 32;
    const int rowlen = my_sub * 16 + 2;
    const int h0 = hr * hrun, h1 = min(a.H, h0 + hrun);
    int res0 = -1, res1 = -1, res2 = -1;
    const float* x_b = a.x + (long)b * a.x_sB;
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT;
    int tsrc[TSL];
#pragma unroll
    for (int ot = 0; ot < TSL; ++ot) tsrc[ot] = MODE == FWD_REP ? min(max(t + ot - 1, 0), a.T - 1) : a.kt * t + ot;
    // (the activation guard is NOT reset between the units of a wave: its scale only moves down over the wave's whole share and
    //  is taken out once, at the flush, separately from the gradient scale.  Multiplying the persistent accumulators back up at a
    //  unit boundary — by up to 2^116 while they still carry the gradient scale — overflowed for large finite activations whose true
    //  dW is finite; a quiet unit behind a loud one adds its terms at the loud unit's scale, i.e. rounded relative to the total.)

    for (int h = h0; h < h1;) {
      wave_fence();
      // dy of this row: lane (cout = r16, pixels p0 + 8 kq + {0..7}); pixels past the row end contribute zero.  All 8 NB loads are
      // issued from clamped addresses and masked after the staging below (`on ? load : 0` was a branch around every load)
      f32x4 d0[NB], d1[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int p = min(p0 + 8 * kq + jj, a.W - 1);
          const float v = dy_b[(long)h * a.y_sH + (long)p * a.y_sW + goff(gb0 + n, a.y_gpu, a.y_sU, a.y_sG) + r16];
          if (jj < 4) d0[n][jj] = v; else d1[n][jj - 4] = v;
        }
      int slot_of[3];
      bool redo = false;
      int redo_dk = 0;
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const int hs = MODE == FWD_REP ? min(max(h + oh - 1, 0), a.H - 1) : h + oh - 1;
        const bool inside = hs >= 0 && hs < a.H;
        const int slot = (hs + 3) % 3;
        slot_of[oh] = slot;
        const int cur = slot == 0 ? res0 : slot == 1 ? res1 : res2;
        if (cur == hs + 1) continue;
        if (slot == 0) res0 = hs + 1; else if (slot == 1) res1 = hs + 1; else res2 = hs + 1;
        f32x4 stg[TSL][3];
#pragma unroll
        for (int ot = 0; ot < TSL; ++ot) {
          const float* src = x_b + (long)tsrc[ot] * a.x_sT + (long)hs * a.x_sH;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < rowlen * 4 && inside) {
              const int w = p0 + (u >> 2) - 1;
              if (MODE == FWD_REP) v = ld4(src + (long)min(max(w, 0), a.W - 1) * a.x_sW + 4 * (u & 3));
              else if (w >= 0 && w < a.W) v = ld4(src + (long)w * a.x_sW + 4 * (u & 3));
            }
            stg[ot][k] = v;
          }
        }
        if constexpr (F16) {
          float am = 0.f;
#pragma unroll
          for (int ot = 0; ot < TSL; ++ot)
#pragma unroll
            for (int k = 0; k < 3; ++k) am = amax_acc(am, stg[ot][k]);
          if (ag.exceeds(am)) {             // (rare: |x| >= 2^15 at the unit's scale) lower the scale, stage the ring again
            redo_dk = ag.lower(wave_umax(__float_as_uint(ActGuard::finite(am))));
            res0 = res1 = res2 = -1;
            redo = true;
            break;
          }
          const float xsc = ag.s();         // (1 unless the guard has fired in this unit)
#pragma unroll
          for (int ot = 0; ot < TSL; ++ot)
#pragma unroll
            for (int k = 0; k < 3; ++k) stg[ot][k] *= xsc;
        }
#pragma unroll
        for (int ot = 0; ot < TSL; ++ot) {
          unsigned* dst = s_x + (ot * 3 + slot) * ROWW;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int u = lane + 64 * k;
            if (u < kC16Row * 4) {          // pixels past a one-block strip are written as zeros
              const Split p = split2<F16>(stg[ot][k]);
              typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
              *reinterpret_cast<u32x2v*>(dst + (u >> 2) * 16 + (u & 3) * 2) = u32x2v{p.h01, p.h23};
              *reinterpret_cast<u32x2v*>(dst + (u >> 2) * 16 + 8 + (u & 3) * 2) = u32x2v{p.l01, p.l23};
            }
          }
        }
      }
      if (F16 && redo) {                  // (h not advanced: the row's ring is staged again at the new scale, which the
        scale_acc(pow2i(redo_dk < -126 ? -126 : redo_dk));   //  accumulators follow)
        continue;
      }
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const bool on = 8 * kq + jj < 16 * my_sub && p0 + 8 * kq + jj < a.W;
          if (jj < 4) d0[n][jj] = on ? d0[n][jj] : 0.f; else d1[n][jj - 4] = on ? d1[n][jj - 4] : 0.f;
        }
#pragma unroll
      for (int n = 0; n < NB; ++n) bsum[n] += hsum4(d0[n]) + hsum4(d1[n]);
      if constexpr (F16) {
        float am = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) am = amax_acc(amax_acc(am, d0[n]), d1[n]);
        const float f = gs.update(wave_umax(__float_as_uint(am)));
        if (f != 1.f) scale_acc(f);
        const float gsc = gs.s();
#pragma unroll
        for (int n = 0; n < NB; ++n) { d0[n] *= gsc; d1[n] *= gsc; }
      }
      u32x4 Bh[NB], Bl[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const Split s0 = split2<F16>(d0[n]), s1 = split2<F16>(d1[n]);
        Bh[n] = u32x4{s0.h01, s0.h23, s1.h01, s1.h23};
        Bl[n] = u32x4{s0.l01, s0.l23, s1.l01, s1.l23};
      }
      wave_fence();
      // acc[tap][cin][cout] += sum over the strip's 32 pixels of x[pixel + tap offset][cin] dy[pixel][cout]
      // (the x fragments of tap tp + 1 are read while tap tp is multiplied, with scheduling fences: the compiler's own order is
      //  ds_read_tr x 4, s_waitcnt lgkmcnt(0), 3 MFMAs per tap — an exposed LDS round trip at each of the 27 taps)
      auto rdA = [&](int tp, u32x4& Ah, u32x4& Al) __attribute__((always_inline)) {
        const int ot = tp / 9, oh = (tp / 3) % 3, ow = tp % 3;
        const unsigned* ap = s_x + (ot * 3 + slot_of[oh]) * ROWW + ow * 16 + tr_off;
        const unsigned long long h0_ = tr(ap), h1_ = tr(ap + 4 * 16), l0_ = tr(ap + 8), l1_ = tr(ap + 4 * 16 + 8);
        Ah = u32x4{(unsigned)h0_, (unsigned)(h0_ >> 32), (unsigned)h1_, (unsigned)(h1_ >> 32)};
        Al = u32x4{(unsigned)l0_, (unsigned)(l0_ >> 32), (unsigned)l1_, (unsigned)(l1_ >> 32)};
      };
      u32x4 Ahb[2], Alb[2];
      rdA(0, Ahb[0], Alb[0]);
#pragma unroll
      for (int tp = 0; tp < NTAPS; ++tp) {
        if (tp + 1 < NTAPS) rdA(tp + 1, Ahb[(tp + 1) & 1], Alb[(tp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          acc[tp][n] = mfma_s<F16>(Ahb[tp & 1], Bh[n], acc[tp][n]);
          acc[tp][n] = mfma_s<F16>(Ahb[tp & 1], Bl[n], acc[tp][n]);
          acc[tp][n] = mfma_s<F16>(Alb[tp & 1], Bh[n], acc[tp][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ++h;
    }
  }
  float* dw = a.dwp + (long)(blockIdx.x % kWgradCopies) * a.copy_stride + (long)wset * NTAPS * 16 * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
  const float un = F16 ? gs.inv() : 1.f, unx = F16 ? ag.inv() : 1.f;   // (two factors: their product may leave f32's range)
#pragma unroll
  for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dw + ((long)tp * 16 + r0 + r) * a.CoutPad + (gb0 + n) * 16 + col, F16 ? acc[tp][n][r] * un * unx : acc[tp][n][r]);
  if (a.dbias) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float v = bsum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + (gb0 + n) * 16 + lane, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad16_shift_kernel: the 16 -> 16 weight gradient of wgrad16_split_kernel with the W taps taken on the dy operand.
//   acc[ot][oh][ow][cin][cout] += sum_q x_pad[row(oh), frame(ot)][q][cin] dy[q - ow][cout]     (q: padded W coordinate, x_pad[q] = x[q - 1])
// A strip is 32 PADDED pixels q0 .. q0 + 31 (strips tile [0, W + 2): the same 7 strips at W = 200): the x fragments of a staged
// row are the same for the three W taps, so a row is read from LDS ONCE per (ot, oh) — 36 transposing reads per output row where
// wgrad16_split_kernel issues 108, each now feeding 9 MFMAs instead of 3 (profiles r04: that kernel's LDS pipe was busy half of
// its time, a quarter of it on bank conflicts) — and the shift lives in the registers of the dy operand: a lane loads the 10
// pixels q0 + 8 kq - 2 .. + 7 of its output channel and forms the three shifted 8-pixel operands by register naming.  No halo
// pixels in the staged rows (32 x 64 B); the hi / lo halves of a pixel's 64 bytes swap places with (pixel bit 1) xor (pixel bit 3),
// which makes the 8-byte writes (banks mod 32, 4 pixels per service group) and the transposing reads (banks mod 64, pixels
// 8 kq .. 8 kq + 3 of two kq per group) both conflict-free; the reads' swizzle is static per lane because nothing shifts.
// Scales (F16): as wgrad16_split_kernel — GradScale on the dy row (all 10 pixels of every lane), ActGuard on the x rows.
// ---------------------------------------------------------------------------------------------
template <int MODE /*FWD_REP | FWD_S2*/, int TSL, bool F16 = false>
__global__ __launch_bounds__(256, 2) void wgrad16_shift_kernel(WgradArgs a, int nstrip, int hrun, int nhrun) {
  constexpr int NTAPS = TSL * 9, NR = TSL * 3;
  constexpr int ROWW = 32 * 16;                    // 32-bit words per staged row: 32 pixels x (8 words hi + 8 words lo)
  __shared__ __attribute__((aligned(16))) unsigned s_xw[4][NR * ROWW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y, gb0 = blockIdx.z;
  unsigned* s_x = s_xw[wave];
  f32x4 acc[NTAPS];
  float bsum = 0.f;
  GradScale gs;
  ActGuard ag;
  auto scale_acc = [&](float f) __attribute__((always_inline)) {
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp) acc[tp] *= f;
  };
#pragma unroll
  for (int tp = 0; tp < NTAPS; ++tp) acc[tp] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nb = a.B / a.wsets;
  const long nunits = (long)nb * a.To * nhrun * nstrip;
  // transposed read of this lane: pixel 8 kq + (r16 >> 2) [+ 4 for the second read], channels 4 (r16 & 3) ..; lo piece: ^ 8 words
  const int px0 = 8 * kq + (r16 >> 2);
  const int tr_hi = px0 * 16 + (((px0 >> 1) ^ (px0 >> 3)) & 1) * 8 + 2 * (r16 & 3);
  typedef short s16x4t __attribute__((ext_vector_type(4)));
  auto tr = [&](const unsigned* p) __attribute__((always_inline)) {
    const s16x4t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4t __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(unsigned long long, t);
  };

  for (long un = (long)blockIdx.x * 4 + wave; un < nunits; un += (long)gridDim.x * 4) {
    long r = un;      // t fastest: see conv16_lds_kernel
    const int t = (int)(r % a.To); r /= a.To;
    const int sp = (int)(r % nstrip); r /= nstrip;
    const int hr = (int)(r % nhrun);
    const int b = (int)(r / nhrun) * a.wsets + wset;
    const int q0 = sp * 32;
    const int qn = min(32, a.W + 2 - q0);           // padded pixels of this strip
    const int h0 = hr * hrun, h1 = min(a.H, h0 + hrun);
    int res0 = -1, res1 = -1, res2 = -1;
    const float* x_b = a.x + (long)b * a.x_sB;
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT + goff(gb0, a.y_gpu, a.y_sU, a.y_sG) + r16;
    int tsrc[TSL];
#pragma unroll
    for (int ot = 0; ot < TSL; ++ot) tsrc[ot] = MODE == FWD_REP ? min(max(t + ot - 1, 0), a.T - 1) : a.kt * t + ot;
    // (the activation guard is NOT reset between the units of a wave: its scale only moves down over the wave's whole share and
    //  is taken out once, at the flush, separately from the gradient scale.  Multiplying the persistent accumulators back up at a
    //  unit boundary — by up to 2^116 while they still carry the gradient scale — overflowed for large finite activations whose true
    //  dW is finite; a quiet unit behind a loud one adds its terms at the loud unit's scale, i.e. rounded relative to the total.)

    // dy of a row: e[i] = dy[q0 + 8 kq - 2 + i] of output channel r16; ten loads from clamped addresses, masked at their use
    auto load_dy = [&](int hh, float (&ee)[10]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int p = q0 + 8 * kq - 2 + i;
        ee[i] = dy_b[(long)hh * a.y_sH + (long)min(max(p, 0), a.W - 1) * a.y_sW];
      }
    };
    // one source row of all T slots, raw (two 16-byte units per lane and slot; zero outside the frame / the strip)
    auto load_row = [&](int hs, bool inside, f32x4 (&st)[TSL][2]) __attribute__((always_inline)) {
#pragma unroll
      for (int ot = 0; ot < TSL; ++ot) {
        const float* src = x_b + (long)tsrc[ot] * a.x_sT + (long)hs * a.x_sH;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int u = lane + 64 * k;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (u < qn * 4 && inside) {
            const int w = q0 + (u >> 2) - 1;
            if (MODE == FWD_REP) v = ld4(src + (long)min(max(w, 0), a.W - 1) * a.x_sW + 4 * (u & 3));
            else if (w >= 0 && w < a.W) v = ld4(src + (long)w * a.x_sW + 4 * (u & 3));
          }
          st[ot][k] = v;
        }
      }
    };
    // The NEXT output row's operands are requested before this row's MFMA loop and used after it: its dy row and the one source
    // row it adds to the ring (tag pf_row = that row + 1; a row the ring already holds is never taken from here).  Their
    // registers are live over the MFMA loop only, where the staging registers are dead; without this every output row exposed
    // one memory round trip per wave (loads issued and waited for at the top of the row).
    float e[10];
    f32x4 pf[TSL][2];
    int pf_row = -1;
    load_dy(h0, e);

    for (int h = h0; h < h1;) {
      wave_fence();
      int slot_of[3];
      bool redo = false;
      int redo_dk = 0;
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const int hs = MODE == FWD_REP ? min(max(h + oh - 1, 0), a.H - 1) : h + oh - 1;
        const bool inside = hs >= 0 && hs < a.H;
        const int slot = (hs + 3) % 3;
        slot_of[oh] = slot;
        const int cur = slot == 0 ? res0 : slot == 1 ? res1 : res2;
        if (cur == hs + 1) continue;
        if (slot == 0) res0 = hs + 1; else if (slot == 1) res1 = hs + 1; else res2 = hs + 1;
        f32x4 stg[TSL][2];
        if (pf_row == hs + 1) {           // (wave-uniform) requested a row ago
#pragma unroll
          for (int ot = 0; ot < TSL; ++ot)
#pragma unroll
            for (int k = 0; k < 2; ++k) stg[ot][k] = pf[ot][k];
        } else {
          load_row(hs, inside, stg);
        }
        if constexpr (F16) {
          float am = 0.f;
#pragma unroll
          for (int ot = 0; ot < TSL; ++ot)
#pragma unroll
            for (int k = 0; k < 2; ++k) am = amax_acc(am, stg[ot][k]);
          if (ag.exceeds(am)) {             // (rare: |x| >= 2^15 at the unit's scale) lower the scale, stage the ring again
            redo_dk = ag.lower(wave_umax(__float_as_uint(ActGuard::finite(am))));
            res0 = res1 = res2 = -1;
            redo = true;
            break;
          }
          const float xsc = ag.s();         // (1 unless the guard has fired in this unit)
#pragma unroll
          for (int ot = 0; ot < TSL; ++ot)
#pragma unroll
            for (int k = 0; k < 2; ++k) stg[ot][k] *= xsc;
        }
#pragma unroll
        for (int ot = 0; ot < TSL; ++ot) {
          unsigned* dst = s_x + (ot * 3 + slot) * ROWW;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int u = lane + 64 * k, px = u >> 2;
            const int o = px * 16 + (((px >> 1) ^ (px >> 3)) & 1) * 8 + (u & 3) * 2;
            const Split p = split2<F16>(stg[ot][k]);
            typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2v*>(dst + o) = u32x2v{p.h01, p.h23};
            *reinterpret_cast<u32x2v*>(dst + (o ^ 8)) = u32x2v{p.l01, p.l23};
          }
        }
      }
      if (F16 && redo) {                  // (h not advanced: the row's ring is staged again at the new scale, which the
        scale_acc(pow2i(redo_dk < -126 ? -126 : redo_dk));   //  accumulators follow)
        continue;
      }
      float ev[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int p = q0 + 8 * kq - 2 + i;
        ev[i] = (p >= 0 && p < a.W) ? e[i] : 0.f;
      }
      bsum += ((ev[1] + ev[2]) + (ev[3] + ev[4])) + ((ev[5] + ev[6]) + (ev[7] + ev[8]));      // the ow = 1 operand: every dy pixel once
      if constexpr (F16) {
        float am = 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i) am = fmaxf(am, __builtin_fabsf(ev[i]));
        const float f = gs.update(wave_umax(__float_as_uint(am)));
        if (f != 1.f) scale_acc(f);
        const float gsc = gs.s();
#pragma unroll
        for (int i = 0; i < 10; ++i) ev[i] *= gsc;
      }
      u32x4 Bh[3], Bl[3];                 // dy shifted by ow pixels: K index jj <-> dy pixel q0 + 8 kq + jj - ow = ev[jj + 2 - ow]
#pragma unroll
      for (int ow = 0; ow < 3; ++ow) {
        const Split s0 = split2<F16>(f32x4{ev[2 - ow], ev[3 - ow], ev[4 - ow], ev[5 - ow]});
        const Split s1 = split2<F16>(f32x4{ev[6 - ow], ev[7 - ow], ev[8 - ow], ev[9 - ow]});
        Bh[ow] = u32x4{s0.h01, s0.h23, s1.h01, s1.h23};
        Bl[ow] = u32x4{s0.l01, s0.l23, s1.l01, s1.l23};
      }
      const bool more = h + 1 < h1;       // (wave-uniform)
      if (more) {
        load_dy(h + 1, e);
        const int hs = MODE == FWD_REP ? min(h + 2, a.H - 1) : h + 2;
        pf_row = hs + 1;
        load_row(hs, hs < a.H, pf);
      }
      wave_fence();
      // the x fragments of staged row r + 1 are read while row r is multiplied (9 MFMAs per read; scheduling fences as in
      // wgrad16_split_kernel); the three W taps' accumulators alternate so that no MFMA waits for its predecessor
      auto rdA = [&](int rr, u32x4& Ah, u32x4& Al) __attribute__((always_inline)) {
        const int o = (rr / 3 * 3 + slot_of[rr % 3]) * ROWW + tr_hi;
        const unsigned long long h0_ = tr(s_x + o), h1_ = tr(s_x + o + 64), l0_ = tr(s_x + (o ^ 8)), l1_ = tr(s_x + (o ^ 8) + 64);
        Ah = u32x4{(unsigned)h0_, (unsigned)(h0_ >> 32), (unsigned)h1_, (unsigned)(h1_ >> 32)};
        Al = u32x4{(unsigned)l0_, (unsigned)(l0_ >> 32), (unsigned)l1_, (unsigned)(l1_ >> 32)};
      };
      u32x4 Ahb[2], Alb[2];
      rdA(0, Ahb[0], Alb[0]);
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        if (rr + 1 < NR) rdA(rr + 1, Ahb[(rr + 1) & 1], Alb[(rr + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int tp0 = (rr / 3) * 9 + (rr % 3) * 3;
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) acc[tp0 + ow] = mfma_s<F16>(Ahb[rr & 1], Bh[ow], acc[tp0 + ow]);
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) acc[tp0 + ow] = mfma_s<F16>(Ahb[rr & 1], Bl[ow], acc[tp0 + ow]);
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) acc[tp0 + ow] = mfma_s<F16>(Alb[rr & 1], Bh[ow], acc[tp0 + ow]);
        __builtin_amdgcn_sched_barrier(0);
      }
      ++h;
    }
  }
  float* dw = a.dwp + (long)(blockIdx.x % kWgradCopies) * a.copy_stride + (long)wset * NTAPS * 16 * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
  const float un = F16 ? gs.inv() : 1.f, unx = F16 ? ag.inv() : 1.f;   // (two factors: their product may leave f32's range)
#pragma unroll
  for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      atomicAdd(dw + ((long)tp * 16 + r0 + r) * a.CoutPad + gb0 * 16 + col, F16 ? acc[tp][r] * un * unx : acc[tp][r]);
  if (a.dbias) {
    float v = bsum;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + gb0 * 16 + lane, v);
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient of the wide (2,3,3)-stride-2 convolutions (joint classifier head: 16*GA -> 16*GB
// channels), LDS-staged.  A workgroup owns one (batch item, t_out, kt) and a run of rows of one row
// segment; LDS holds a ring of 3 source rows of x (one-pixel zero halo) and the dy row.  The 9*GA
// (tap, input-channel-block) A fragments are dealt round-robin to the 8 waves; every wave forms all GB
// output-channel blocks for its fragments: 7 x 6 MFMAs per 13 four-byte LDS reads at GA = GB = 6.
// ---------------------------------------------------------------------------------------------
template <int GA, int GB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wgrad_wide_lds_kernel(WgradArgs a, int seg, int nseg, int hrun, int nhrun) {
  constexpr int CI = 16 * GA, CO = 16 * GB;
  constexpr int PSX = CI + ((48 - CI % 32) % 32);   // pixel stride == 16 (mod 32) floats: conflict-free 4-byte reads
  constexpr int PSY = CO + ((48 - CO % 32) % 32);
  constexpr int NFRAG = 9 * GA;
  constexpr int FPW = (NFRAG + WAVES - 1) / WAVES;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int rowlen = seg + 2, segp = (seg + 3) & ~3;
  float* s_x = s_dyn;                               // [3][rowlen][PSX]
  float* s_dy = s_dyn + 3 * rowlen * PSX;           // [segp][PSY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;

  f32x4 acc[FPW][GB];
  float bsum[GB];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int n = 0; n < GB; ++n) acc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < GB; ++n) bsum[n] = 0.f;

  // every workgroup keeps ONE kt for its whole life (so its accumulators belong to 9 fixed taps) and one (input, output)
  // channel sub-block of GA x GB 16-channel groups: blockIdx.z = kt + a.kt * (xsub * (CoutTot / CO) + ysub)
  const int kt = blockIdx.z % a.kt;
  const int nys = a.CoutPad / CO;
  const int xg0 = ((blockIdx.z / a.kt) / nys) * GA, yg0 = ((blockIdx.z / a.kt) % nys) * GB;
  const int nb = a.B / a.wsets;
  const long nchunks = (long)nb * a.To * nhrun * nseg;
  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    long r = ch;
    const int sg = (int)(r % nseg); r /= nseg;
    const int hr = (int)(r % nhrun); r /= nhrun;
    const int t = (int)(r % a.To);
    const int b = (int)(r / a.To) * a.wsets + wset;
    const int w0 = sg * seg;
    const int h0 = hr * hrun, h1 = min(a.H, h0 + hrun);
    int res0 = -1, res1 = -1, res2 = -1;
    const float* x_bt = a.x + (long)b * a.x_sB + (long)(a.kt * t + kt) * a.x_sT;
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT;

    for (int h = h0; h < h1; ++h) {
      __syncthreads();
      int slot_of[3];
      bool ok_h[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const int hs = h + oh - 1;
        ok_h[oh] = hs >= 0 && hs < a.H;
        const int slot = ((hs % 3) + 3) % 3;
        slot_of[oh] = slot;
        if (!ok_h[oh]) continue;
        const int cur = slot == 0 ? res0 : slot == 1 ? res1 : res2;
        if (cur == hs) continue;
        if (slot == 0) res0 = hs; else if (slot == 1) res1 = hs; else res2 = hs;
        const float* src = x_bt + (long)hs * a.x_sH;
        float* dst = s_x + (long)slot * rowlen * PSX;
        for (int u = tid; u < rowlen * (CI / 4); u += 64 * WAVES) {
          const int px = u / (CI / 4), q = u % (CI / 4);       // q: 16-byte unit inside the pixel's channels
          const int w = w0 + px - 1;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (w >= 0 && w < a.W) v = ld4(src + (long)w * a.x_sW + goff(xg0 + (q >> 2), a.x_gpu, a.x_sU, a.x_sG) + 4 * (q & 3));
          st4(dst + px * PSX + 4 * q, v);
        }
      }
      {
        const float* src = dy_b + (long)h * a.y_sH;
        for (int u = tid; u < segp * (CO / 4); u += 64 * WAVES) {
          const int px = u / (CO / 4), q = u % (CO / 4);
          const int w = w0 + px;
          const bool in = px < seg && w < a.W;
          st4(s_dy + px * PSY + 4 * q, in ? ld4(src + (long)w * a.y_sW + goff(yg0 + (q >> 2), a.y_gpu, a.y_sU, a.y_sG) + 4 * (q & 3)) : f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
      __syncthreads();

      int base[FPW];
      bool okf[FPW];
#pragma unroll
      for (int f = 0; f < FPW; ++f) {
        const int fr = wave + f * WAVES;            // fragment id = tap9 * GA + ga
        const int tap9 = fr / GA, ga = fr % GA;
        const int oh = (tap9 / 3) % 3, ow = tap9 % 3;
        okf[f] = fr < NFRAG && ok_h[oh];
        base[f] = (slot_of[oh] * rowlen + ow + kq) * PSX + 16 * ga + r16;
      }
      const int steps = segp >> 2;
#pragma unroll 2
      for (int s = 0; s < steps; ++s) {
        float bf[GB];
#pragma unroll
        for (int n = 0; n < GB; ++n) { bf[n] = s_dy[(4 * s + kq) * PSY + 16 * n + r16]; bsum[n] += bf[n]; }
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
          if (!okf[f]) continue;
          const float af = s_x[base[f] + s * 4 * PSX];
#pragma unroll
          for (int n = 0; n < GB; ++n) acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], acc[f][n], 0, 0, 0);
        }
      }
    }
  }
  float* dw = a.dwp + (long)wset * a.ntaps * a.Cin * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    const int fr = wave + f * WAVES;
    if (fr >= NFRAG) continue;
    const int tap = kt * 9 + fr / GA, ga = fr % GA;
#pragma unroll
    for (int n = 0; n < GB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dw + ((long)tap * a.Cin + 16 * (xg0 + ga) + r0 + r) * a.CoutPad + 16 * (yg0 + n) + col, acc[f][n][r]);
  }
  if (a.dbias && wave == 0 && kt == 0 && xg0 == 0) {
#pragma unroll
    for (int n = 0; n < GB; ++n) {
      float v = bsum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + 16 * (yg0 + n) + lane, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad_wide_dma_kernel: wgrad_wide_lds_kernel (exact f32 MFMA) with the NEXT row's operands staged by LDS-DMA while this row is
// multiplied.  The register-staged kernel loads a row, writes it to LDS and waits at two workgroup barriers per output row with
// its one workgroup per CU idle on the MFMA side meanwhile (MFMA-busy 0.60, SQ_WAIT_ANY 27 %, round 4); a register-staged
// prefetch spilled (round 2).  Here: a ring of FOUR x rows and TWO dy rows, pixel-major WITHOUT padding (the DMA writes 64
// consecutive 16-byte chunks), the row of iteration h + 1 in flight during iteration h, ONE barrier per row.  Bank conflicts of the
// 4-byte operand reads (16 lanes = 16 consecutive channels, the 4 lane groups = 4 consecutive pixels; pixel stride 16 G floats)
// are avoided on the SOURCE side: with an even group count G the 16-channel group g of pixel p sits at position g ^ ((p >> 1) & 1)
// — the four pixels of a read then cover four different 16-bank groups (32 (p & 1) + 16 ((p >> 1) & 1) mod 64); with G = 3 the
// stride 48 does that by itself.  Out-of-frame pixels read a 16-byte zero constant.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const float kZeroQuad[4] = {0.f, 0.f, 0.f, 0.f};

template <int GA, int GB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wgrad_wide_dma_kernel(WgradArgs a, int seg, int nseg, int hrun, int nhrun) {
  constexpr int CI = 16 * GA, CO = 16 * GB;
  constexpr bool ROTX = GA % 2 == 0, ROTY = GB % 2 == 0;
  constexpr int NFRAG = 9 * GA;
  constexpr int FPW = (NFRAG + WAVES - 1) / WAVES;
  extern __shared__ __attribute__((aligned(1024))) float s_dyn[];
  const int rowlen = seg + 2, segp = (seg + 3) & ~3;
  const int xrow = rowlen * CI, yrow = segp * CO;         // floats per staged row
  float* s_x = s_dyn;                                     // [4][rowlen][CI]
  float* s_dy = s_dyn + 4 * xrow;                         // [2][segp][CO]
  // (wave index as a SCALAR: with `tid >> 6` in a vector register the per-fragment validity test below compiled to an EXEC-mask
  //  branch per fragment — s_and_saveexec / s_cbranch_execz around every group of GB MFMAs, each group behind its own ds_read +
  //  s_waitcnt lgkmcnt(0): the LDS round trip exposed 7 times per K step, MFMA-busy 0.60 in the register-staged kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;

  f32x4 acc[FPW][GB];
  float bsum[GB];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int n = 0; n < GB; ++n) acc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < GB; ++n) bsum[n] = 0.f;

  const int kt = blockIdx.z % a.kt;
  const int nys = a.CoutPad / CO;
  const int xg0 = ((blockIdx.z / a.kt) / nys) * GA, yg0 = ((blockIdx.z / a.kt) % nys) * GB;
  const int nb = a.B / a.wsets;
  const long nchunks = (long)nb * a.To * nhrun * nseg;
  const unsigned lds_x = (unsigned)(size_t)s_x, lds_y = (unsigned)(size_t)s_dy;
  const int nqx = rowlen * (CI / 4), nqy = segp * (CO / 4);     // 16-byte chunks per staged row

  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    long r = ch;
    const int sg = (int)(r % nseg); r /= nseg;
    const int hr = (int)(r % nhrun); r /= nhrun;
    const int t = (int)(r % a.To);
    const int b = (int)(r / a.To) * a.wsets + wset;
    const int w0 = sg * seg;
    const int h0 = hr * hrun, h1 = min(a.H, h0 + hrun);
    const float* x_bt = a.x + (long)b * a.x_sB + (long)(a.kt * t + kt) * a.x_sT;
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT;

    // source row hs of x -> ring slot hs & 3 (rows outside the frame are never staged: their taps are skipped)
    // (the per-lane source offsets do not depend on the row, but kept in registers across the row loop — 8 values — they spilled
    //  next to the 168 accumulators: recomputed per row)
    auto dma_x = [&](int hs) __attribute__((always_inline)) {
      if (hs < 0 || hs >= a.H) return;
      const float* src = x_bt + (long)hs * a.x_sH;
      const unsigned dst = lds_x + (unsigned)((hs & 3) * xrow) * 4u;
      for (int q0 = wave * 64; q0 < nqx; q0 += WAVES * 64) {
        const int q = q0 + lane;
        const int px = q / (CI / 4), rem = q - px * (CI / 4);
        const int g = (rem >> 2) ^ (ROTX ? ((px >> 1) & 1) : 0);
        const int w = w0 + px - 1;
        const float* sp = (w >= 0 && w < a.W) ? src + (long)w * a.x_sW + goff(xg0 + g, a.x_gpu, a.x_sU, a.x_sG) + 4 * (rem & 3) : kZeroQuad;
        if (q < nqx) glds16(sp, __builtin_amdgcn_readfirstlane(dst + (unsigned)q0 * 16u));
      }
    };
    auto dma_dy = [&](int h) __attribute__((always_inline)) {
      const float* src = dy_b + (long)h * a.y_sH;
      const unsigned dst = lds_y + (unsigned)((h & 1) * yrow) * 4u;
      for (int q0 = wave * 64; q0 < nqy; q0 += WAVES * 64) {
        const int q = q0 + lane;
        const int px = q / (CO / 4), rem = q - px * (CO / 4);
        const int g = (rem >> 2) ^ (ROTY ? ((px >> 1) & 1) : 0);
        const int w = w0 + px;
        const float* sp = (px < seg && w < a.W) ? src + (long)w * a.y_sW + goff(yg0 + g, a.y_gpu, a.y_sU, a.y_sG) + 4 * (rem & 3) : kZeroQuad;
        if (q < nqy) glds16(sp, __builtin_amdgcn_readfirstlane(dst + (unsigned)q0 * 16u));
      }
    };
    __syncthreads();                                      // the previous chunk's readers are done
    dma_x(h0 - 1); dma_x(h0); dma_x(h0 + 1); dma_dy(h0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int h = h0; h < h1; ++h) {
      if (h + 1 < h1) { dma_x(h + 2); dma_dy(h + 1); }    // (slots (h + 2) & 3 and (h + 1) & 1: last read in iteration h - 1)
      int base[FPW];
      bool okf[FPW];
#pragma unroll
      for (int f = 0; f < FPW; ++f) {
        const int fr = wave + f * WAVES;            // fragment id = tap9 * GA + ga
        const int tap9 = fr / GA, ga = fr % GA;
        const int oh = (tap9 / 3) % 3, ow = tap9 % 3;
        const int hs = h + oh - 1;
        okf[f] = fr < NFRAG && hs >= 0 && hs < a.H;
        const int px = ow + kq;                     // (+ 4 s: does not change bit 1 of the pixel index... see rot below)
        base[f] = ((hs & 3) * rowlen + px) * CI + 16 * (ga ^ (ROTX ? ((px >> 1) & 1) : 0)) + r16;
      }
      const float* yb = s_dy + (h & 1) * yrow + kq * CO + r16;
      const int yrot = ROTY ? ((kq >> 1) & 1) : 0;
      const int steps = segp >> 2;
      // the operands of K step s + 1 are read from LDS while step s is multiplied (scheduling fences: left alone the compiler
      // sinks every read to just before its MFMAs); a fragment whose source row
      // lies outside the frame multiplies zeros — only the frame's first and last row have such fragments, and only the waves
      // whose last fragment id exceeds the fragment count: the masking selects and the bias sums live in wave-uniform branches
      auto rd = [&](int s_, float (&af_)[FPW], float (&bf_)[GB]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < FPW; ++f) af_[f] = s_x[base[f] + s_ * 4 * CI];
#pragma unroll
        for (int n = 0; n < GB; ++n) bf_[n] = yb[s_ * 4 * CO + 16 * (n ^ yrot)];
      };
      bool all_ok = true;
#pragma unroll
      for (int f = 0; f < FPW; ++f) { all_ok = all_ok && okf[f]; if (!okf[f]) base[f] = base[0]; }   // (any valid address; masked below)
      const bool want_bias = a.dbias && wave == 0 && kt == 0 && xg0 == 0;                              // (scalar)
      auto mma = [&](float (&af_)[FPW], float (&bf_)[GB]) __attribute__((always_inline)) {
        if (!all_ok) {
#pragma unroll
          for (int f = 0; f < FPW; ++f) af_[f] = okf[f] ? af_[f] : 0.f;
        }
        if (want_bias) {
#pragma unroll
          for (int n = 0; n < GB; ++n) bsum[n] += bf_[n];
        }
#pragma unroll
        for (int f = 0; f < FPW; ++f)
#pragma unroll
          for (int n = 0; n < GB; ++n) acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af_[f], bf_[n], acc[f][n], 0, 0, 0);
      };
      // (two alternating register sets instead of the copy spilled 17 - 19 registers next to the 168 accumulators)
      float afn[FPW], bfn[GB];
      rd(0, afn, bfn);
      for (int s = 0; s < steps; ++s) {
        float af[FPW], bf[GB];
#pragma unroll
        for (int f = 0; f < FPW; ++f) af[f] = afn[f];
#pragma unroll
        for (int n = 0; n < GB; ++n) bf[n] = bfn[n];
        if (s + 1 < steps) rd(s + 1, afn, bfn);
        __builtin_amdgcn_sched_barrier(0);
        mma(af, bf);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA blocks of the next row have landed
      __syncthreads();                                    // ... every wave's; and this row's readers are done
    }
  }
  float* dw = a.dwp + (long)wset * a.ntaps * a.Cin * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    const int fr = wave + f * WAVES;
    if (fr >= NFRAG) continue;
    const int tap = kt * 9 + fr / GA, ga = fr % GA;
#pragma unroll
    for (int n = 0; n < GB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dw + ((long)tap * a.Cin + 16 * (xg0 + ga) + r0 + r) * a.CoutPad + 16 * (yg0 + n) + col, acc[f][n][r]);
  }
  if (a.dbias && wave == 0 && kt == 0 && xg0 == 0) {
#pragma unroll
    for (int n = 0; n < GB; ++n) {
      float v = bsum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) atomicAdd(a.dbias + (long)wset * a.CoutPad + 16 * (yg0 + n) + lane, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad_wide_lds_kernel on f16 pieces at f32 level (IDEE_CONV_MATH=f22): the same work decomposition (a workgroup owns one
// (batch item, t_out, kt) and a run of rows of one 64-pixel segment; the 9 GA (tap, input block) fragments dealt round-robin to
// the 8 waves, every wave forming all GB output blocks for its fragments), K = 32 PIXELS per v_mfma_f32_16x16x32_f16 and three
// piece products per tile: 126 MFMAs of 16 cycles per wave and 32 pixels where the f32 kernel issues 336 of 32.  The x rows and
// the dy row live in LDS as pixel-major f16 piece images [pixel][block][16 ch hi | 16 ch lo] (64 B per pixel and block); BOTH
// K = pixels operands come out of them with the transposing read (ds_read_b64_tr_b16: lane (channel i, group g) gets pixels
// 8g..8g+7 of channel i).  x (activations, O(1)) is split unscaled; the dy row carries a power of two chosen per ROW for the whole
// workgroup (largest |dy| of the row over all threads: wave max, LDS, barrier; common.h GradScale — every wave follows the same
// sequence, so each rescales its own accumulators when the scale moves) and the flush multiplies by 2^-k.  The bias gradient is
// summed from the unscaled f32 values in the staging threads' registers (a thread stages the same channel quads in every row).
// The x rows are range-guarded per work unit (common.h ActGuard, workgroup-wide through the same LDS exchange as the dy maximum):
// k = 0 until a row's largest |x| reaches 2^15; then the two rows already in the ring are staged again at the lower scale and
// every wave multiplies its accumulators along.
// ---------------------------------------------------------------------------------------------
template <int GA, int GB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wgrad_wide_split_kernel(WgradArgs a, int seg /*multiple of 32, <= 64*/, int nseg, int hrun,
                                                                      int nhrun) {
  constexpr int CI = 16 * GA, CO = 16 * GB;
  // 32-bit words per pixel of the x / dy piece images: 16 per channel block + 8 of padding — with a stride of 96 words the 4
  // pixels a 16-lane group of the transposing read touches fall on 2 bank groups (96 = 32 mod 64); 104 = 40 mod 64 spreads them
  constexpr int PX = 16 * GA + 8, PY = 16 * GB + 8;
  constexpr int NFRAG = 9 * GA;
  constexpr int FPW = (NFRAG + WAVES - 1) / WAVES;
  constexpr int NTH = 64 * WAVES;
  constexpr int DYU = (64 * (CO / 4) + NTH - 1) / NTH;   // dy staging units (16 bytes) per thread and row
  extern __shared__ __attribute__((aligned(16))) unsigned s_dynu[];
  __shared__ unsigned s_red[WAVES], s_redx[WAVES], s_redx0[WAVES];
  __shared__ float s_bs[CO];
  const int rowlen = seg + 2;
  unsigned* s_x = s_dynu;                            // [3][rowlen][PX]
  unsigned* s_dy = s_dynu + 3 * rowlen * PX;         // [seg][PY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wset = blockIdx.y;
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  typedef short s16x4t __attribute__((ext_vector_type(4)));
  auto tr = [&](const unsigned* p) __attribute__((always_inline)) {
    const s16x4t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4t __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(unsigned long long, t);
  };
  auto pack2 = [](unsigned long long lo_, unsigned long long hi_) {
    return u32x4{(unsigned)lo_, (unsigned)(lo_ >> 32), (unsigned)hi_, (unsigned)(hi_ >> 32)};
  };

  f32x4 acc[FPW][GB];
  GradScale gs;
  ActGuard ag;                                       // range guard of the x rows (workgroup-uniform), moving down only over the workgroup's whole share
  auto scale_acc = [&](float f) __attribute__((always_inline)) {
#pragma unroll
    for (int ff = 0; ff < FPW; ++ff)
#pragma unroll
      for (int n = 0; n < GB; ++n) acc[ff][n] *= f;
  };
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int n = 0; n < GB; ++n) acc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient: unit i of a thread is the same (pixel, channel quad) of every row, so its partial sums stay in registers
  f32x4 bsum[DYU];
#pragma unroll
  for (int i = 0; i < DYU; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kt = blockIdx.z % a.kt;
  const int nys = a.CoutPad / CO;
  const int xg0 = ((blockIdx.z / a.kt) / nys) * GA, yg0 = ((blockIdx.z / a.kt) % nys) * GB;
  const bool do_bias = a.dbias && kt == 0 && xg0 == 0;
  const int nb = a.B / a.wsets;
  const long nchunks = (long)nb * a.To * nhrun * nseg;
  // transposed-read word offset of this lane inside an image row block, pixel shift 0, hi piece
  const int tr_px = 8 * kq + (r16 >> 2), tr_w = 2 * (r16 & 3);
  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    long r = ch;
    const int sg = (int)(r % nseg); r /= nseg;
    const int hr = (int)(r % nhrun); r /= nhrun;
    const int t = (int)(r % a.To);
    const int b = (int)(r / a.To) * a.wsets + wset;
    const int w0 = sg * seg;
    const int nvalid = min(seg, a.W - w0);
    const int ksteps = (nvalid + 31) >> 5;
    const int h0 = hr * hrun, h1 = min(a.H, h0 + hrun);
    const float* x_bt = a.x + (long)b * a.x_sB + (long)(a.kt * t + kt) * a.x_sT;
    const float* dy_b = a.dy + (long)b * a.y_sB + (long)t * a.y_sT;
    // x row hs -> registers (zeros outside the frame), registers -> piece image slot hs mod 3; dy row likewise.  The rows of the
    // NEXT iteration are requested right before this iteration's MFMAs (a workgroup is alone on its CU: nothing else hides the
    // memory latency between the barriers).
    constexpr int XU = (66 * (CI / 4) + NTH - 1) / NTH;
    // per-thread staging units: their in-row source offsets (elements; -1 = outside the frame or past the row) and LDS word
    // offsets are the same for every row of the chunk — computed once (the per-row index arithmetic, with its divisions by the
    // run-time channel-group layout, was ~1,800 instructions per row and wave: most of this kernel's time)
    int xo[XU], xd[XU], yo[DYU], yd[DYU];
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      const int u = tid + i * NTH;
      const int px = u / (CI / 4), q = u % (CI / 4);
      const int w = w0 + px - 1;
      const bool st_ = u < rowlen * (CI / 4);
      xd[i] = st_ ? px * PX + (q >> 2) * 16 + (q & 3) * 2 : -1;
      xo[i] = (st_ && w >= 0 && w < a.W) ? (int)((long)w * a.x_sW + goff(xg0 + (q >> 2), a.x_gpu, a.x_sU, a.x_sG) + 4 * (q & 3)) : -1;
    }
#pragma unroll
    for (int i = 0; i < DYU; ++i) {
      const int u = tid + i * NTH;
      const int px = u / (CO / 4), q = u % (CO / 4);
      const int w = w0 + px;
      const bool st_ = u < seg * (CO / 4);
      yd[i] = st_ ? px * PY + (q >> 2) * 16 + (q & 3) * 2 : -1;
      yo[i] = (st_ && w < a.W) ? (int)((long)w * a.y_sW + goff(yg0 + (q >> 2), a.y_gpu, a.y_sU, a.y_sG) + 4 * (q & 3)) : -1;
    }
    auto load_x = [&](int hs, f32x4 (&xr)[XU]) __attribute__((always_inline)) {
      const float* src = x_bt + (long)hs * a.x_sH;
      const bool row_in = hs >= 0 && hs < a.H;
#pragma unroll
      for (int i = 0; i < XU; ++i) xr[i] = (row_in && xo[i] >= 0) ? ld4(src + xo[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_x = [&](int hs, const f32x4 (&xr)[XU]) __attribute__((always_inline)) {
      unsigned* dst = s_x + (long)(((hs % 3) + 3) % 3) * rowlen * PX;
      const float xsc = ag.s();                      // (1 unless the guard has fired in this unit)
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        if (xd[i] < 0) continue;
        const Split p = split2<PM_F16X2>(xr[i] * xsc);
        unsigned* d = dst + xd[i];
        *reinterpret_cast<u32x2v*>(d) = u32x2v{p.h01, p.h23};
        *reinterpret_cast<u32x2v*>(d + 8) = u32x2v{p.l01, p.l23};
      }
    };
    auto load_dy = [&](int h, f32x4 (&dv)[DYU]) __attribute__((always_inline)) {
      const float* src = dy_b + (long)h * a.y_sH;
#pragma unroll
      for (int i = 0; i < DYU; ++i) dv[i] = (h < a.H && yo[i] >= 0) ? ld4(src + yo[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // staging registers, TWO rows deep: (xr, dv) = x row h + 1 and dy row h for the top of iteration h, (xr2, dv2) the pair for
    // iteration h + 1, requested a full iteration before they are needed (one row deep, a row's loads had only the ~1 us of its
    // MFMA loop to land before the row maximum needed them: 0.67 -> see DESIGN ms)
    f32x4 xr[XU], dv[DYU], xr2[XU], dv2[DYU];
    auto x_amax = [&](const f32x4 (&xv)[XU]) __attribute__((always_inline)) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < XU; ++i) am = amax_acc(am, xv[i]);
      return ActGuard::finite(am);
    };
    auto wg_max = [&](const unsigned (&red)[WAVES]) __attribute__((always_inline)) {
      unsigned m = 0u;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) m = m > red[w] ? m : red[w];
      return m;
    };
    // (the activation guard is not reset between the units of a workgroup: see wgrad16_split_kernel)
    // A unit is walked in one segment — or, when the guard fires for a row (workgroup-uniform, rare), begun again from that row
    // at the lower scale through the same segment-start code.
    int pend_dk = 0;
    for (int hb = h0;;) {
    bool again = false;
    {   // segment start: rows hb - 1 and hb into the ring, at the unit's guard scale (k = 0 unless they hold |x| >= 2^15)
      if (pend_dk != 0) { scale_acc(pow2i(imax_dev(pend_dk, -126))); pend_dk = 0; }
      load_x(hb - 1, xr); load_x(hb, xr2);
      const unsigned wm0 = wave_umax(__float_as_uint(fmaxf(x_amax(xr), x_amax(xr2))));
      if (lane == 0) s_redx0[wave] = wm0;
      __syncthreads();                               // the previous readers are done; s_redx0 holds the two rows' maxima
      const unsigned m0 = wg_max(s_redx0);
      if (__uint_as_float(m0) >= ag.thr) scale_acc(pow2i(imax_dev(ag.lower(m0), -126)));
      store_x(hb - 1, xr);
      store_x(hb, xr2);
      load_x(hb + 1, xr);  load_dy(hb, dv);
      load_x(hb + 2, xr2); load_dy(hb + 1, dv2);
    }
    // the row's largest |dy| (and the largest |x| of the x row staged with it) goes to s_red / s_redx right after the MFMAs of
    // the row before (its loads have landed by then), so ONE barrier both retires the previous row's readers and publishes
    // them: two barriers per row
    auto post_amax = [&]() __attribute__((always_inline)) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < DYU; ++i) am = amax_acc(am, dv[i]);
      const unsigned wm = wave_umax(__float_as_uint(am));
      const unsigned wx = wave_umax(__float_as_uint(x_amax(xr)));
      if (lane == 0) { s_red[wave] = wm; s_redx[wave] = wx; }
    };
    post_amax();
    for (int h = hb; h < h1; ++h) {
      __syncthreads();                               // the previous row's readers are done; s_red / s_redx hold this row's maxima
      int slot_of[3];
      bool ok_h[3];
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const int hs = h + oh - 1;
        ok_h[oh] = hs >= 0 && hs < a.H;
        slot_of[oh] = ((hs % 3) + 3) % 3;
      }
      {
        const unsigned mx = wg_max(s_redx);
        if (__uint_as_float(mx) >= ag.thr) {         // (workgroup-uniform, rare) row h + 1 holds |x| >= 2^15 at the unit's scale:
          pend_dk = ag.lower(mx);                    // lower it and begin the segment again at row h (the accumulators
          hb = h;                                    // follow at the segment start)
          again = true;
          break;
        }
      }
      store_x(h + 1, xr);                            // (row h + 1 replaces row h - 2 in the ring)
      {
        const float f = gs.update(wg_max(s_red));
        if (f != 1.f) scale_acc(f);
        const float gsc = gs.s();
#pragma unroll
        for (int i = 0; i < DYU; ++i) {
          if (yd[i] < 0) continue;
          bsum[i] += dv[i];                       // (unit i of this thread is channel quad (tid + i NTH) % (CO / 4) in every row)
          const Split p = split2<PM_F16X2>(dv[i] * gsc);
          unsigned* d = s_dy + yd[i];
          *reinterpret_cast<u32x2v*>(d) = u32x2v{p.h01, p.h23};
          *reinterpret_cast<u32x2v*>(d + 8) = u32x2v{p.l01, p.l23};
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < XU; ++i) xr[i] = xr2[i];
#pragma unroll
      for (int i = 0; i < DYU; ++i) dv[i] = dv2[i];
      if (h + 2 < h1) { load_x(h + 3, xr2); load_dy(h + 2, dv2); }   // the rows of iteration h + 2

      const unsigned* abase[FPW];
      bool okf[FPW];
#pragma unroll
      for (int f = 0; f < FPW; ++f) {
        const int fr = wave + f * WAVES;            // fragment id = tap9 * GA + ga
        const int tap9 = fr / GA, ga = fr % GA;
        const int oh = (tap9 / 3) % 3, ow = tap9 % 3;
        okf[f] = fr < NFRAG;                     // (a row outside the frame sits in the ring as zeros: no per-row validity)
        abase[f] = s_x + ((long)slot_of[oh] * rowlen + ow + tr_px) * PX + ga * 16 + tr_w;
      }
      const unsigned* bbase = s_dy + tr_px * PY + tr_w;
      // (the output blocks in passes of at most 3: 168 accumulator registers leave room for 3 B fragment pairs, not 6; the A
      //  fragments are read once per pass)
      constexpr int NBP = GB > 3 ? 3 : GB;
      for (int s = 0; s < ksteps; ++s) {
#pragma unroll
        for (int n0 = 0; n0 < GB; n0 += NBP) {
          u32x4 Bh[NBP], Bl[NBP];
#pragma unroll
          for (int n = 0; n < NBP; ++n) {
            const unsigned* bp = bbase + (long)(32 * s) * PY + (n0 + n) * 16;
            Bh[n] = pack2(tr(bp), tr(bp + 4 * PY));
            Bl[n] = pack2(tr(bp + 8), tr(bp + 4 * PY + 8));
          }
          // the A fragment of f + 1 is read while the MFMAs of f run (left alone the compiler emits ds_read, s_waitcnt
          // lgkmcnt(0), v_mfma per fragment: one exposed LDS round trip each, at two waves per SIMD)
          u32x4 Ah[2], Al[2];
          auto rdA = [&](int f, u32x4& h_, u32x4& l_) __attribute__((always_inline)) {
            const unsigned* ap = abase[f] + (long)(32 * s) * PX;
            h_ = pack2(tr(ap), tr(ap + 4 * PX));
            l_ = pack2(tr(ap + 8), tr(ap + 4 * PX + 8));
          };
          rdA(0, Ah[0], Al[0]);
#pragma unroll
          for (int f = 0; f < FPW; ++f) {
            if (f + 1 < FPW) rdA(f + 1, Ah[(f + 1) & 1], Al[(f + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (!okf[f]) continue;
#pragma unroll
            for (int n = 0; n < NBP; ++n) {
              acc[f][n0 + n] = mfma_s<PM_F16X2>(Ah[f & 1], Bh[n], acc[f][n0 + n]);
              acc[f][n0 + n] = mfma_s<PM_F16X2>(Ah[f & 1], Bl[n], acc[f][n0 + n]);
              acc[f][n0 + n] = mfma_s<PM_F16X2>(Al[f & 1], Bh[n], acc[f][n0 + n]);
            }
          }
        }
      }
      if (h + 1 < h1) post_amax();
    }
    if (!again) break;
    }
  }
  float* dw = a.dwp + (long)(blockIdx.x % kWgradCopies) * a.copy_stride + (long)wset * a.ntaps * a.Cin * a.CoutPad;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
  const float un = gs.inv(), unx = ag.inv();         // (two factors: their product may leave f32's range)
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    const int fr = wave + f * WAVES;
    if (fr >= NFRAG) continue;
    const int tap = kt * 9 + fr / GA, ga = fr % GA;
#pragma unroll
    for (int n = 0; n < GB; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dw + ((long)tap * a.Cin + 16 * (xg0 + ga) + r0 + r) * a.CoutPad + 16 * (yg0 + n) + col, acc[f][n][r] * un * unx);
  }
  if (do_bias) {      // workgroup-uniform.  The 512 threads' partial sums meet in LDS first: ONE global atomic per channel and
    __syncthreads();  // workgroup (per-thread global atomics — 4,096 per workgroup onto 48 addresses — were 0.29 of the kernel's 0.69 ms)
    for (int i = tid; i < CO; i += NTH) s_bs[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DYU; ++i) {
      const int u = tid + i * NTH;
      if (u >= seg * (CO / 4)) continue;
      const int q = u % (CO / 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&s_bs[4 * q + r], bsum[i][r]);
    }
    __syncthreads();
    for (int i = tid; i < CO; i += NTH) atomicAdd(a.dbias + (long)wset * a.CoutPad + 16 * yg0 + i, s_bs[i]);
  }
}

// the replicate-mode data gradient accumulates its two edge columns (w = 0, W-1) with atomics: they start from zero,
// or from the addend's edge columns
__global__ void init_w_edges_kernel(float* __restrict__ dx, const float* __restrict__ addend, int B, int T, int H, int W, int Cin,
                                    long sB, long sT, long sH, long sW) {
  const int q4 = Cin / 4;
  const long total = (long)B * T * H * 2 * q4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e % q4); long r = e / q4;
    const int side = (int)(r % 2); r /= 2;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T); const long b = r / T;
    const long o = b * sB + t * sT + h * sH + (side ? W - 1 : 0) * sW + 4 * q;
    st4(dx + o, addend ? ld4(addend + o) : f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing (reference layout (wsets, Cout, Cin, kt, kh, kw) -> MFMA fragment order)
// ---------------------------------------------------------------------------------------------
// f16 piece fragments for conv_wide_split_kernel (PRE): one power-of-two scale per weight set (largest |w| 2^k in [2^13, 2^14), as the
// kernel's own prologue scan chose it) ...
__global__ __launch_bounds__(256) void pack_wscale_kernel(const float* __restrict__ pf, unsigned* __restrict__ mout, long per_set,
                                                          int parts) {
  // grid = (parts, wsets): every workgroup scans 1 / parts of a set, eight loads in flight per trip, and adds its maximum (the
  // bit pattern of a non-negative float orders like an unsigned) to the set's slot, zeroed by the caller
  __shared__ unsigned s_m[4];
  const float* w = pf + (long)blockIdx.y * per_set;
  const long n4 = per_set / 4, lo = n4 * blockIdx.x / parts, hi = n4 * (blockIdx.x + 1) / parts;
  float mx = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 8 * 256) {
    f32x4 f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long iu = i + (long)u * 256; f[u] = ld4(w + 4 * (iu < hi ? iu : i)); }
#pragma unroll
    for (int u = 0; u < 8; ++u) mx = amax_acc(mx, f[u]);
  }
  const unsigned m = wave_umax(__float_as_uint(mx));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned bm = 0u;
    for (int i = 0; i < 4; ++i) bm = bm > s_m[i] ? bm : s_m[i];
    atomicMax(mout + blockIdx.y, bm);
  }
}
// ... and the piece image of every (weight set, tap): [group pair gp][output block n][hi | lo][lane][4 words], the two pieces of
// the fragments of K groups 2 gp and 2 gp + 1 side by side — exactly what the kernel's conversion pass wrote to LDS per tap
__global__ __launch_bounds__(256) void pack_pieces_kernel(const float* __restrict__ pf, const unsigned* __restrict__ mx_bits,
                                                          unsigned* __restrict__ pieces, int wsets, int ntaps, int Gk, int NT) {
  const int GP = Gk / 2;
  const long units = (long)wsets * ntaps * GP * NT * 64;
  const long slab = (long)Gk * NT * 256;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
    const int l = (int)(u & 63);
    long r = u >> 6;
    const int n = (int)(r % NT); r /= NT;
    const int gp = (int)(r % GP); r /= GP;                 // r = wset * ntaps + tap
    const int ws = (int)(r / ntaps);
    const float sc = pow2i(wscale_exp(mx_bits[ws]));
    const float* src = pf + r * slab + ((2 * gp) * NT + n) * 256 + l * 4;
    const Split sa = split2<PM_F16X2>(ld4(src) * sc), sb = split2<PM_F16X2>(ld4(src + NT * 256) * sc);
    unsigned* dst = pieces + r * slab + (((long)(gp * NT + n) * 2) * 64 + l) * 4;
    *reinterpret_cast<u32x4*>(dst) = u32x4{sa.h01, sa.h23, sb.h01, sb.h23};
    *reinterpret_cast<u32x4*>(dst + 256) = u32x4{sa.l01, sa.l23, sb.l01, sb.l23};
  }
}
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ pf, float* __restrict__ pd, int wsets,
                            int Cout, int Cin, int ntaps) {
  const int Gi = Cin / 16, Go = (Cout + 15) / 16;
  const long per_set = (long)ntaps * Gi * Go * 256;
  const long total = (long)wsets * per_set;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int s = (int)(e & 3), lane = (int)((e >> 2) & 63);
    long r = e >> 8;
    const int ws = (int)(r / ((long)ntaps * Gi * Go));
    r %= (long)ntaps * Gi * Go;
    const int tap = (int)(r / (Gi * Go));
    const int rem = (int)(r % (Gi * Go));
    const int l15 = lane & 15, kq = lane >> 4;
    if (pf) {   // [tap][gin][gout]: W[cout = gout*16 + l15][cin = gin*16 + 4kq + s]
      const int gin = rem / Go, gout = rem % Go;
      const int co = gout * 16 + l15, ci = gin * 16 + 4 * kq + s;
      pf[e] = co < Cout ? w[(((long)ws * Cout + co) * Cin + ci) * ntaps + tap] : 0.f;
    }
    if (pd) {   // [tap][gK = gout][gN = gin]: W[cout = gK*16 + 4kq + s][cin = gN*16 + l15]
      const int gk = rem / Gi, gn = rem % Gi;
      const int co = gk * 16 + 4 * kq + s, ci = gn * 16 + l15;
      pd[e] = co < Cout ? w[(((long)ws * Cout + co) * Cin + ci) * ntaps + tap] : 0.f;
    }
  }
}

// dWp (wsets, ntaps, Cin, CoutPad) -> dW (wsets, Cout, Cin, ntaps)
__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int wsets, int Cout, int Cin,
                                    int ntaps, int CoutPad) {
  const long total = (long)wsets * Cout * Cin * ntaps;
  const long copy_stride = (long)wsets * ntaps * Cin * CoutPad;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % ntaps);
    long r = e / ntaps;
    const int ci = (int)(r % Cin); r /= Cin;
    const int co = (int)(r % Cout);
    const int ws = (int)(r / Cout);
    const float* src = dwp + (((long)ws * ntaps + tap) * Cin + ci) * CoutPad + co;
    float v = src[0];
#pragma unroll
    for (int c = 1; c < kWgradCopies; ++c) v += src[c * copy_stride];      // the partial copies (conv_common.h kWgradCopies)
    dw[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// The single-class output convs (Cout == 1, (2,3,3) stride 2 over T = 2: classifier conv3,
// classifier/CNN_3D.py:40,85).  They are memory-bound dot products, not GEMMs: streaming kernels, lanes
// of a pixel own 4 channels each (16-byte accesses), weights read from the packed fragment buffers.
// ---------------------------------------------------------------------------------------------
struct C1Args {
  const void* x;       // (B, 2, H, W, C) by strides; XT = float | bf16_t (act_dtype)
  const float* w;      // packed fragments (fwd or dgrad order)
  const float* bias;   // (wsets) or null
  const float* dy;     // (B, H, W)
  const void* relu_ref;   // XT, like x
  float* y;            // fwd: (B, H, W)
  void* dx;            // dgrad: like x (XT)
  float* dwp;          // wgrad: (wsets, 18, C, 16)
  float* dbias;        // (wsets, 16) or null
  int B, wsets, H, W, C;
  long x_sB, x_sT, x_sH, x_sW, x_sG, x_sU;
  int x_gpu;
  int accumulate;
  int To;              // output frames: frame `to` of item b reads input frames 2 to, 2 to + 1 (x has 2 To frames, y / dy has To)
};
// weight of (tap, cin) for cout 0 in the packed buffers
__device__ __forceinline__ f32x4 c1_w4_fwd(const float* wp, int Gi, int tap, int c4 /* cin = 4*c4 .. +3 */) {
  const int cin = 4 * c4;
  return ld4(wp + ((long)(tap * Gi + cin / 16) * 64 + ((cin % 16) / 4) * 16) * 4);
}
__device__ __forceinline__ float c1_w_dgrad(const float* wp, int Gi, int tap, int cin) {
  return wp[((long)(tap * Gi + cin / 16) * 64 + cin % 16) * 4];
}

// LPP lanes per pixel (power of two >= C/4); lanes q >= C/4 idle
template <int LPP, typename XT>
__global__ __launch_bounds__(256) void c1_fwd_kernel(C1Args a) {
  // LPP lanes (4 channels each) own FOUR consecutive output pixels of a row: per (kt, kh) they load the 6 source pixels
  // once (9 loads per output instead of 18) and keep f32x4 partial products, summed over channels only at the end
  constexpr int PXT = 4;
  const int QC = a.C / 4, Gi = a.C / 16;
  const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP, GPB = 256 / LPP;
  const int WG = (a.W + PXT - 1) / PXT;
  // 32-bit pixel-group arithmetic (launch_c1_t checks the count): the 64-bit divisions by run-time values this loop used to do
  // per lane and trip are ~150-instruction routines each — more vector work than the dot products themselves
  const unsigned ngrp = (unsigned)a.B * a.To * a.H * WG;       // (item, output frame) pairs walk like a batch of a.B * a.To items
  // this lane's 4 channels of all 18 taps stay in registers (re-read when the weight set changes: per-variable heads)
  f32x4 wr[18];
  int wcur = -1;
  for (unsigned g0 = blockIdx.x * GPB; g0 < ngrp; g0 += gridDim.x * GPB) {
    const unsigned g = g0 + pl;
    f32x4 accv[PXT];
#pragma unroll
    for (int o = 0; o < PXT; ++o) accv[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    int b = 0, h = 0, w0 = 0;
    float biasv = 0.f;                            // (read with the source pixels: between the stores it is a round trip per pixel)
    if (g < ngrp && q < QC) {
      const unsigned gr = g / (unsigned)WG;
      w0 = (int)(g - gr * WG) * PXT;
      b = (int)(gr / (unsigned)a.H);             // (item, frame) pair index; item = b / To
      h = (int)(gr - (unsigned)b * a.H);
      const int wset = (b / a.To) % a.wsets;
      if (a.bias) biasv = a.bias[wset];
      if (wset != wcur) {
        wcur = wset;
        const float* wp = a.w + (long)wset * 18 * Gi * 256;
#pragma unroll
        for (int tap = 0; tap < 18; ++tap) wr[tap] = c1_w4_fwd(wp, Gi, tap, q);
      }
      const XT* xb = reinterpret_cast<const XT*>(a.x) + (long)(b / a.To) * a.x_sB + (long)(b % a.To) * 2 * a.x_sT + goff(q / 4, a.x_gpu, a.x_sU, a.x_sG) + 4 * (q % 4);
      // per frame: the 18 source vectors (3 rows x 6 pixels, clamped addresses) are ALL requested before the first is used —
      // with the out-of-frame select written right behind each load the compiler waited (vmcnt(0)) after every single one:
      // 36 serialized memory round trips per 4 output pixels
      typedef typename Raw4<XT>::type raw_t;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        raw_t xr[3][PXT + 2];
#pragma unroll
        for (int oh = 0; oh < 3; ++oh) {
          const XT* row = xb + (long)kt * a.x_sT + (long)min(max(h + oh - 1, 0), a.H - 1) * a.x_sH;
#pragma unroll
          for (int i = 0; i < PXT + 2; ++i) xr[oh][i] = ldraw4<XT>(row + (long)min(max(w0 - 1 + i, 0), a.W - 1) * a.x_sW);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int oh = 0; oh < 3; ++oh) {
          const int hs = h + oh - 1;
          const bool rok = hs >= 0 && hs < a.H;
          f32x4 xv[PXT + 2];
#pragma unroll
          for (int i = 0; i < PXT + 2; ++i) {            // source pixels w0-1 .. w0+PXT, zeroed outside the frame
            const int ws = w0 - 1 + i;
            xv[i] = (rok && ws >= 0 && ws < a.W) ? cvtraw4(xr[oh][i]) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int ow = 0; ow < 3; ++ow) {
            const f32x4 wv = wr[(kt * 3 + oh) * 3 + ow];
#pragma unroll
            for (int o = 0; o < PXT; ++o) accv[o] = __builtin_elementwise_fma(xv[o + ow], wv, accv[o]);
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < PXT; ++o) {
      float acc = (accv[o][0] + accv[o][1]) + (accv[o][2] + accv[o][3]);
#pragma unroll
      for (int sft = 1; sft < LPP; sft <<= 1) acc += __shfl_xor(acc, sft, 64);
      if (g < ngrp && q == 0 && w0 + o < a.W) a.y[((long)b * a.H + h) * a.W + w0 + o] = acc + biasv;
    }
  }
}

template <int LPP, typename XT>
__global__ __launch_bounds__(256) void c1_dgrad_kernel(C1Args a) {
  const int QC = a.C / 4, Gi = a.C / 16;
  const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP, PPB = 256 / LPP;
  const long npix = (long)a.B * a.To * 2 * a.H * a.W;       // input pixels (b, to, kt, h, w)
  // this lane's 4 channels of all 18 taps, once (requires one weight set per launch column: blockIdx.y = wset)
  const int wset = blockIdx.y;
  f32x4 wr[18];
  if (q < QC) {
    const float* wp = a.w + (long)wset * 18 * Gi * 256;
#pragma unroll
    for (int tap = 0; tap < 18; ++tap)
#pragma unroll
      for (int u = 0; u < 4; ++u) wr[tap][u] = c1_w_dgrad(wp, Gi, tap, 4 * q + u);
  }
  const int nb = a.B / a.wsets;
  const unsigned npw = (unsigned)nb * a.To * a.H * a.W;           // output pixels (b, to, h, w) of this weight set (32-bit index
  (void)npix;                                                     // arithmetic: see c1_fwd_kernel)
  // A lane group owns BOTH input frames (kt = 0, 1) of UN pixels per trip: the 9 dy values of a pixel serve both frames, and
  // every load of the trip — dy (clamped addresses, masked afterwards), the ReLU references, the addends — is requested before
  // the first use (one pixel and frame per trip, loaded at the points of use: three dependent memory round trips per 16-byte
  // store, 1.1 - 1.7 TB/s).
  constexpr int UN = 2;
  typedef typename Raw4<XT>::type raw_t;
  const unsigned pstride = gridDim.x * PPB;
  for (unsigned p0 = blockIdx.x * PPB; p0 < npw; p0 += UN * pstride) {
    float gv[UN][9];
    raw_t rr_raw[UN][2], ad_raw[UN][2];
    long off[UN];
    int hh[UN], ww[UN];
    bool on[UN];
#pragma unroll
    for (int r = 0; r < UN; ++r) {
      const unsigned p = p0 + r * pstride + pl;
      on[r] = p < npw && q < QC;
      const unsigned pc = on[r] ? p : 0u;
      const unsigned pr = pc / (unsigned)a.W, bt = pr / (unsigned)a.H;
      const int w = (int)(pc - pr * a.W), h = (int)(pr - bt * a.H);
      const unsigned bi = bt / (unsigned)a.To;
      const int to = (int)(bt - bi * a.To);
      const int b = (int)bi * a.wsets + wset;
      hh[r] = h; ww[r] = w;
      const float* dyb = a.dy + ((long)b * a.To + to) * a.H * a.W;
      const int qe = on[r] ? q : 0;
      off[r] = (long)b * a.x_sB + (long)(2 * to) * a.x_sT + (long)h * a.x_sH + (long)w * a.x_sW + goff(qe / 4, a.x_gpu, a.x_sU, a.x_sG) + 4 * (qe % 4);
#pragma unroll
      for (int oh = 0; oh < 3; ++oh)
#pragma unroll
        for (int ow = 0; ow < 3; ++ow)
          gv[r][oh * 3 + ow] = dyb[(long)min(max(h - (oh - 1), 0), a.H - 1) * a.W + min(max(w - (ow - 1), 0), a.W - 1)];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        rr_raw[r][kt] = raw_t(); ad_raw[r][kt] = raw_t();
        if (a.relu_ref) rr_raw[r][kt] = ldraw4<XT>(reinterpret_cast<const XT*>(a.relu_ref) + off[r] + (long)kt * a.x_sT);
        if (a.accumulate) ad_raw[r][kt] = ldraw4<XT>(reinterpret_cast<const XT*>(a.dx) + off[r] + (long)kt * a.x_sT);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < UN; ++r) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int oh = 0; oh < 3; ++oh) {
        const int hq = hh[r] - (oh - 1);
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) {
          const int wq = ww[r] - (ow - 1);
          const bool ok = hq >= 0 && hq < a.H && wq >= 0 && wq < a.W;
          const float g = ok ? gv[r][oh * 3 + ow] : 0.f;
          acc[0] += wr[oh * 3 + ow] * g;
          acc[1] += wr[9 + oh * 3 + ow] * g;
        }
      }
      if (!on[r]) continue;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (a.relu_ref) {
          const f32x4 rr = cvtraw4(rr_raw[r][kt]);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[kt][u] = rr[u] > 0.f ? acc[kt][u] : 0.f;
        }
        if (a.accumulate) acc[kt] += cvtraw4(ad_raw[r][kt]);
        stx4<XT>(reinterpret_cast<XT*>(a.dx) + off[r] + (long)kt * a.x_sT, acc[kt]);
      }
    }
  }
}

// dW[tap][c] = sum_{b,h,w} x[b][kt][h+oh-1][w+ow-1][c] dy[b][h][w]: every x value is read once and spread over its 9 taps
template <int LPP, typename XT>
__global__ __launch_bounds__(256) void c1_wgrad_kernel(C1Args a) {
  __shared__ __attribute__((aligned(16))) float s_red18[18][4][LPP][4];
  const int QC = a.C / 4;
  const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP, PPB = 256 / LPP;
  const int wset = blockIdx.y;
  const int nb = a.B / a.wsets;
  const unsigned npix = (unsigned)nb * a.To * a.H * a.W;     // source pixels per kt for this weight set (32-bit: see c1_fwd_kernel)
  f32x4 acc[2][9];
  float bsum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) acc[kt][t9] = f32x4{0.f, 0.f, 0.f, 0.f};
  // four source pixels per lane and trip: their x and dy loads are all in flight before the first fma
  constexpr int UN = 4;
  const unsigned pstride = gridDim.x * PPB;
  for (unsigned p0 = blockIdx.x * PPB; p0 < npix; p0 += UN * pstride) {
    f32x4 xq[UN][2];
    float gq[UN][9];
    bool on[UN];
#pragma unroll
    for (int r = 0; r < UN; ++r) {
      const unsigned p = p0 + r * pstride + pl;
      on[r] = p < npix && q < QC;
      const unsigned pc = on[r] ? p : 0u;
      const unsigned pr = pc / (unsigned)a.W, bt = pr / (unsigned)a.H;
      const int ws = (int)(pc - pr * a.W), hs = (int)(pr - bt * a.H);
      const unsigned bi = bt / (unsigned)a.To;
      const int to = (int)(bt - bi * a.To);
      const int b = (int)bi * a.wsets + wset;
      const float* dyb = a.dy + ((long)b * a.To + to) * a.H * a.W;
#pragma unroll
      for (int oh = 0; oh < 3; ++oh)
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) {
          const int h = hs - (oh - 1), w = ws - (ow - 1);      // output pixel this source feeds through tap (oh, ow)
          gq[r][oh * 3 + ow] = dyb[(long)min(max(h, 0), a.H - 1) * a.W + min(max(w, 0), a.W - 1)];   // (masked after the loads)
        }
      const XT* xb = reinterpret_cast<const XT*>(a.x) + (long)b * a.x_sB + (long)(2 * to) * a.x_sT + (long)hs * a.x_sH + (long)ws * a.x_sW + goff((on[r] ? q : 0) / 4, a.x_gpu, a.x_sU, a.x_sG) + 4 * ((on[r] ? q : 0) % 4);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) xq[r][kt] = ldx4<XT>(xb + (long)kt * a.x_sT);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < UN; ++r) {
      const unsigned p = p0 + r * pstride + pl;
      const unsigned pc = on[r] ? p : 0u;
      const unsigned pr = pc / (unsigned)a.W;
      const int ws = (int)(pc - pr * a.W), hs = (int)(pr % (unsigned)a.H);
#pragma unroll
      for (int oh = 0; oh < 3; ++oh)
#pragma unroll
        for (int ow = 0; ow < 3; ++ow) {
          const int h = hs - (oh - 1), w = ws - (ow - 1);
          if (!(on[r] && h >= 0 && h < a.H && w >= 0 && w < a.W)) gq[r][oh * 3 + ow] = 0.f;    // pixels that do not exist
        }
      if (q == 0) bsum += gq[r][4];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) acc[kt][t9] += xq[r][kt] * gq[r][t9];
    }
  }
  // reduce over the lanes of a wave that own the same channels, then over the 4 waves, then one atomic per value
  float* dw = a.dwp + (long)wset * 18 * a.C * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* red = &s_red18[0][0][0][0];       // [18 taps][4 waves][LPP][4]
  // all 18 tap sums of a wave go to LDS in one pass (two workgroup barriers in all: a barrier pair per tap was most of this
  // kernel's time — a workgroup streams only ~2,000 pixels)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      f32x4 v = acc[kt][t9];
#pragma unroll
      for (int o = LPP; o < 64; o <<= 1) {
        v[0] += __shfl_xor(v[0], o, 64); v[1] += __shfl_xor(v[1], o, 64);
        v[2] += __shfl_xor(v[2], o, 64); v[3] += __shfl_xor(v[3], o, 64);
      }
      if (lane < LPP) st4(red + (((kt * 9 + t9) * 4 + wave) * LPP + lane) * 4, v);
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 18 * LPP * 4; e += 256) {
    const int tap = e / (LPP * 4), qq = (e / 4) % LPP, u = e & 3;
    if (qq >= QC) continue;
    const float* rp = red + (tap * 4 * LPP + qq) * 4 + u;
    atomicAdd(dw + ((long)tap * a.C + 4 * qq + u) * 16, (rp[0] + rp[LPP * 4]) + (rp[2 * LPP * 4] + rp[3 * LPP * 4]));
  }
  if (a.dbias) {      // (uniform) one atomic per WORKGROUP: per-wave atomics onto this single address form a chain of ~17 ns each
    bsum = wave_sum(bsum);
    __syncthreads();
    if (lane == 0) red[wave] = bsum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.dbias + (long)wset * 16, (red[0] + red[1]) + (red[2] + red[3]));
  }
}

template <typename XT>
int launch_c1_t(int which, C1Args& a, hipStream_t st) {
  const int QC = a.C / 4;
  const int LPP = QC <= 4 ? 4 : QC <= 8 ? 8 : QC <= 16 ? 16 : QC <= 32 ? 32 : 64;
  if (QC > 64) return idee::fail(IDEE_ERR_UNSUPPORTED, "conv (Cout=1): Cin %d > 256 is not built", a.C);
  const int PPB = 256 / LPP;
  const long nbw = a.B / a.wsets;
  const long npix = (which == 0 ? (long)a.B * a.H * ((a.W + 3) / 4) : nbw * a.H * a.W) * a.To;
  if ((long)a.B * a.To * 2 * a.H * a.W + 4L * 256 * kCUs * 8 >= (1L << 31))      // the kernels index pixels (and their grid strides) in 32 bits
    return idee::fail(IDEE_ERR_UNSUPPORTED, "conv (Cout=1): more than 2^31 pixels per launch");
  const int per = which == 0 ? 1 : a.wsets;                       // dgrad / wgrad: one weight set per grid column
  int gx = imin(cdiv(npix, PPB), cdiv((long)kCUs * (which == 2 ? 2 : which == 1 ? 4 : 8), per));
  // weight gradient: a workgroup ends with an 18-tap reduction and 72 C atomics — at least 8 trips of 4 PPB pixels each, or a
  // small input (config #1: 8,192 pixels) is all epilogue
  if (which == 2) gx = imin(gx, imax(npix / (PPB * 4 * 8), 1));
  const dim3 grid(gx < 1 ? 1 : gx, per);
#define C1(K) do { if (LPP == 4) hipLaunchKernelGGL((K<4, XT>), grid, dim3(256), 0, st, a); else if (LPP == 8) hipLaunchKernelGGL((K<8, XT>), grid, dim3(256), 0, st, a); \
  else if (LPP == 16) hipLaunchKernelGGL((K<16, XT>), grid, dim3(256), 0, st, a); else if (LPP == 32) hipLaunchKernelGGL((K<32, XT>), grid, dim3(256), 0, st, a); \
  else hipLaunchKernelGGL((K<64, XT>), grid, dim3(256), 0, st, a); } while (0)
  if (which == 0) C1(c1_fwd_kernel); else if (which == 1) C1(c1_dgrad_kernel); else C1(c1_wgrad_kernel);
#undef C1
  return IDEE_OK;
}
int launch_c1(int which, C1Args& a, bool bf16, hipStream_t st) {
  return bf16 ? launch_c1_t<bf16_t>(which, a, st) : launch_c1_t<float>(which, a, st);
}
// Cout = 1 (2,3,3) stride-2 convs: streaming kernels, any even frame count (T = 2: the reference's last classifier conv at T = 8;
// T = 4: the same conv at T = 16), y / dy a contiguous (B, T / 2, H, W)
bool is_c1(const idee_conv_desc* d) {
  return d->Cout == 1 && d->mode == IDEE_CONV_K233_S2_ZERO && d->T >= 2 && d->T % 2 == 0 && d->y_sW == 1 && d->y_sH == d->W &&
         d->y_sB == (long)(d->T / 2) * d->H * d->W && (d->T == 2 || d->y_sT == (long)d->H * d->W);
}
void fill_c1(C1Args& a, const idee_conv_desc* d) {
  memset(&a, 0, sizeof(a));
  a.B = d->B; a.wsets = d->wsets; a.H = d->H; a.W = d->W; a.C = d->Cin; a.To = d->T / 2;
  a.x_sB = d->x_sB; a.x_sT = d->x_sT; a.x_sH = d->x_sH; a.x_sW = d->x_sW; a.x_sG = d->x_sG;
  a.x_sU = d->x_sU; a.x_gpu = d->x_gpu > 0 ? d->x_gpu : (1 << 30);
}

int check_desc(const idee_conv_desc* d) {
  IDEE_CHECK_ARG(d, "conv: null descriptor");
  IDEE_CHECK_ARG(d->mode == IDEE_CONV_K333_REPLICATE || d->mode == IDEE_CONV_K233_S2_ZERO || d->mode == IDEE_CONV_K133_ZERO,
                 "conv: bad mode %d", d->mode);
  IDEE_CHECK_ARG(d->B > 0 && d->wsets > 0 && d->B % d->wsets == 0, "conv: B %d must be a multiple of wsets %d", d->B, d->wsets);
  IDEE_CHECK_ARG(d->T > 0 && d->H > 0 && d->W > 0, "conv: empty extent");
  IDEE_CHECK_ARG(d->Cin > 0 && d->Cin % 16 == 0, "conv: Cin %d must be a multiple of 16", d->Cin);
  IDEE_CHECK_ARG(d->Cout == 1 || (d->Cout > 0 && d->Cout % 16 == 0), "conv: Cout %d must be 1 or a multiple of 16", d->Cout);
  if (d->mode == IDEE_CONV_K233_S2_ZERO) IDEE_CHECK_ARG(d->T >= 2, "conv (2,3,3): T %d < kernel 2", d->T);
  return IDEE_OK;
}
// act_dtype == IDEE_DTYPE_BF16 outside the streaming Cout == 1 case: what conv_bf16.hip is built for
int check_bf16(const idee_conv_desc* d) {
  IDEE_CHECK_ARG(d->Cout % 16 == 0, "conv (bf16): Cout %d must be a multiple of 16 (or the (2,3,3) Cout = 1 logit conv)", d->Cout);
  IDEE_CHECK_ARG(d->x_sW % 8 == 0 && d->y_sW % 8 == 0 && d->x_sG % 8 == 0 && d->y_sG % 8 == 0 && d->x_sU % 8 == 0 && d->y_sU % 8 == 0,
                 "conv (bf16): pixel / group strides must keep 16-byte alignment");
  return IDEE_OK;
}
inline int out_T(const idee_conv_desc* d) { return d->mode == IDEE_CONV_K233_S2_ZERO ? (d->T - 2) / 2 + 1 : d->T; }
inline int n_taps(const idee_conv_desc* d) { return d->mode == IDEE_CONV_K333_REPLICATE ? 27 : d->mode == IDEE_CONV_K233_S2_ZERO ? 18 : 9; }
inline int mode_kt(const idee_conv_desc* d) { return d->mode == IDEE_CONV_K133_ZERO ? 1 : 2; }
inline int pick_nt(int G) { return G % 6 == 0 ? 6 : G % 3 == 0 ? 3 : G % 2 == 0 ? 2 : 1; }
// (Gk K groups, Gn output groups) that conv_wide_split_kernel takes (launch_conv): all output blocks in one workgroup, K in
// pairs, both piece slabs + the raw slab within the LDS
inline bool wide_split_shape(int Gk, int Gn) {
  const int NT = pick_nt(Gn);
  const long slab_bytes = (long)Gk * NT * 256 * 4;
  return Gn / NT == 1 && (NT == 6 || NT == 3 || NT == 2) && Gk % 2 == 0 && Gk * NT <= 36 && 3 * slab_bytes <= 156 * 1024 &&
         Gk * NT * 64 <= 5 * 512;
}
// the packed weight buffers of a convolution carry ready f16 piece fragments (+ 64 ints: the sets' scale exponents) behind the
// f32 fragments when either direction can run on conv_wide_split_kernel
inline bool packs_pieces(const idee_conv_desc* d) {
  if (d->act_dtype != IDEE_DTYPE_F32 || d->mode == IDEE_CONV_K333_REPLICATE || d->Cin % 16 || d->Cout % 16 || d->wsets > 64) return false;
  return wide_split_shape(d->Cin / 16, d->Cout / 16) || wide_split_shape(d->Cout / 16, d->Cin / 16);
}

// resident workgroups per CU of a kernel (occupancy API; advisory, only steers the grid size)
template <class K>
int wg_per_cu(K kernel, int block) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) n = 1;
  return n;
}
// grid.x such that gx * others fills the chip in whole rounds of resident workgroups
inline int one_round(long work_items, long others, int per_cu) {
  long gx = ((long)kCUs * per_cu) / (others > 0 ? others : 1);
  if (gx < 1) gx = 1;
  if (gx > work_items) gx = work_items;
  return (int)gx;
}

// Rows per run for the wave-private strip kernels: a work unit is one run of `hrun` rows of one (strip, frame); every wave of a
// grid slice takes ceil(units / waves) of them, so the makespan is that count x (hrun + the unit's start-up, ~1.5 row times)
// — minimised over the number of runs (runs of at least 8 rows).  (A fixed "about 5 units per wave" rule cost 12 % at the
// headline shape: 5.27 units per wave means some waves run 6.)
inline int pick_hrun(int H, long base_units, long slices, int per_cu) {
  if (const int v = opt_int("IDEE_CONV_HRUN")) {        // test knob: force short runs so small shapes cross run boundaries
    if (v > 0) return v < H ? v : H;
  }
  long gx_full = ((long)kCUs * per_cu) / (slices > 0 ? slices : 1);
  if (gx_full < 1) gx_full = 1;
  int best = H;
  double best_cost = 1e30;
  for (int nhr = 1; nhr <= (H + 7) / 8; ++nhr) {
    const int hrun = cdiv(H, nhr);
    const long units = base_units * cdiv(H, hrun);
    const long gx = gx_full < cdiv(units, 4) ? gx_full : cdiv(units, 4);
    const long per = cdiv(units, gx * 4);
    const double cost = (double)per * (hrun + 1.5);
    if (cost < best_cost * 0.995) { best_cost = cost; best = hrun; }
  }
  return best;
}

void split_tiles(ConvArgs& a, int Wframe) {
  a.nsub = cdiv(Wframe, 16);
  a.wtiles = cdiv(a.nsub, 4);
  a.sub_base = a.nsub / a.wtiles;
  a.sub_rem = a.nsub % a.wtiles;
  a.hgroups = cdiv(a.Ho, 4);
  a.tiles = a.To * a.hgroups * a.wtiles;
}

// IDEE_CONV_MATH=split selects the split-precision 16 -> 16 kernels (default: exact f32 MFMA)
// (IDEE_CONV_MATH=fwd22: only the forward passes, whose f16 pieces round operands at 2^-22; see swin_mfma.hip split_math)
// (IDEE_CONV_MATH=f22: the forward passes as fwd22; the 16 -> 16 data and weight gradients on f16 pieces with a power-of-two
//  gradient scale per work unit / per row — conv16_split_kernel<.., GSC>, wgrad16_split_kernel<.., F16>; the other gradient
//  kernels stay exact f32)
bool conv_f22_math() { return env_is("IDEE_CONV_MATH", "f22"); }
bool conv_split_math(bool forward) {
  return env_is("IDEE_CONV_MATH", "split") || (forward && (env_is("IDEE_CONV_MATH", "fwd22") || conv_f22_math()));
}

template <int MODE>
int launch_conv(ConvArgs& a, bool cin1, hipStream_t stream) {
  if constexpr (MODE == FWD_REP || MODE == DGRAD_REP) {
    // 32 -> 32 replicate convolutions (config #4's width) as 2 x 2 sub-problems on the 16 -> 16 strip kernel: output block n =
    // sum over input blocks g, the second pass adding to the first in place (the generic kernel ran them at 73-78 TFLOP/s)
    const bool no_split = env_is("IDEE_CONV32", "generic"), reg16 = env_is("IDEE_CONV16", "regs");   // read at every launch
    if (!cin1 && !no_split && !reg16 && !conv_split_math(MODE == FWD_REP || MODE == FWD_S2) && a.Gk == 2 && a.GnTot == 2 && a.cout_real == 32 && !a.w_tap_stride &&
        !(a.addend && a.relu)) {
      const int GA = a.Gk, GB = a.GnTot;        // fwd: input, output blocks ; dgrad: dy, dx blocks (packed [tap][GA][GB])
      for (int n = 0; n < GB; ++n)
        for (int g = 0; g < GA; ++g) {
          ConvArgs sub = a;
          sub.Gk = 1; sub.GnTot = 1; sub.cout_real = 16;
          sub.in = a.in + (long)(g / a.i_gpu) * a.i_sU + (long)(g % a.i_gpu) * a.i_sG;
          const long og = (long)(n / a.o_gpu) * a.o_sU + (long)(n % a.o_gpu) * a.o_sG;
          sub.out = a.out + og;
          sub.wp = a.wp + ((long)g * GB + n) * 256;
          sub.w_tap_stride = GA * GB * 256;
          sub.w_set_stride = (long)a.ntaps * GA * GB * 256;
          sub.bias = (g == 0 && a.bias) ? a.bias + n * 16 : nullptr;
          sub.bias_set_stride = GB * 16;
          sub.relu = (g == GA - 1) ? a.relu : 0;
          sub.relu_ref = a.relu_ref ? a.relu_ref + og : nullptr;
          sub.addend = g == 0 ? (a.addend ? a.addend + og : nullptr) : sub.out;
          sub.addend_first = g > 0;
          split_tiles(sub, MODE == DGRAD_REP ? sub.Wo + 2 : sub.Wo);
          if (int rc = launch_conv<MODE>(sub, false, stream)) return rc;
        }
      return IDEE_OK;
    }
  }
  const int NT = pick_nt(a.GnTot);
  const bool wlds = (a.GnTot == 1 && a.Gk == 1 && !cin1);
  const int gz = a.GnTot / NT;
  const dim3 block(256);
  {
    if (wlds && a.cout_real == 16 && (MODE == FWD_REP || MODE == DGRAD_REP || a.kt == 2)) {
      const int Wframe = (MODE == DGRAD_REP) ? a.Wo + 2 : a.Wo;
      const int nstrip = cdiv(cdiv(Wframe, 16), 2);
      const long bt = (long)(a.B / a.wsets) * a.To;
      const int hrun = pick_hrun(a.Ho, bt * nstrip, a.wsets, 2);
      const int nhrun = cdiv(a.Ho, hrun);
      const long nunits = bt * nhrun * nstrip;
      if constexpr (MODE == DGRAD_REP || MODE == DGRAD_S2) {
        // f16 pieces with the per-unit gradient scale (not the 2 x 2 sub-problems of a 32 -> 32 conv: the split kernels read
        // the packed weights without a tap stride)
        if (conv_f22_math() && !a.w_tap_stride && !a.addend_first) {
          auto kern = conv16_split_kernel<MODE, true, SC_GRAD>;
          const dim3 grid(one_round(cdiv(nunits, 4), a.wsets, wg_per_cu(kern, 256)), a.wsets);
          hipLaunchKernelGGL(kern, grid, block, 0, stream, a, nstrip, hrun, nhrun);
          return IDEE_OK;
        }
      }
      if (conv_split_math(MODE == FWD_REP || MODE == FWD_S2)) {      // opt-in split-precision path: f16 pieces forward, bf16 pieces in the data gradient
        // (forward: f16 pieces with the activation range guard; data gradient: bf16 pieces)
        auto kern = conv16_split_kernel<MODE, (MODE == FWD_REP || MODE == FWD_S2), (MODE == FWD_REP || MODE == FWD_S2) ? SC_ACT : SC_NONE>;
        const dim3 grid(one_round(cdiv(nunits, 4), a.wsets, wg_per_cu(kern, 256)), a.wsets);
        hipLaunchKernelGGL(kern, grid, block, 0, stream, a, nstrip, hrun, nhrun);
        return IDEE_OK;
      }
      if constexpr (MODE == FWD_REP || MODE == DGRAD_REP) {
        const bool reg_staged = env_is("IDEE_CONV16", "regs");
        if (!reg_staged) {
          auto kern = conv16_glds_kernel<MODE>;
          // workgroups per weight set: one round of resident workgroups, at least ~8 rows each; IDEE_CONV_HRUN (test knob)
          // caps the rows per workgroup instead
          const int nsubw = cdiv(Wframe, 16), lone = nsubw & 1, nfull = nstrip - lone;
          const long rows = (long)(a.B / a.wsets) * ((a.To & 3) ? a.To * 4L : a.To) / 4 * a.Ho * nstrip;   // per workgroup-slot
          long g = imax(((long)kCUs * wg_per_cu(kern, 256)) / a.wsets, 1);
          long cap = imax(rows / 8, 1);
          if (const int v = opt_int("IDEE_CONV_HRUN")) { if (v > 0) { cap = cdiv(rows, v); g = cap; } }
          if (rows * 10 >= (1L << 31)) return idee::fail(IDEE_ERR_UNSUPPORTED, "conv: %ld strip rows exceed the 32-bit work partition", rows);
          const dim3 grid((unsigned)imax(imin(g, cap), 1), a.wsets);
          hipLaunchKernelGGL(kern, grid, block, 0, stream, a, nstrip, nfull);
          return IDEE_OK;
        }
      }
      auto kern = conv16_lds_kernel<MODE>;
      const dim3 grid(one_round(cdiv(nunits, 4), a.wsets, wg_per_cu(kern, 256)), a.wsets);
      hipLaunchKernelGGL(kern, grid, block, 0, stream, a, nstrip, hrun, nhrun);
      return IDEE_OK;
    }
  }
  if constexpr (MODE == FWD_S2) {
    // (1,3,3) forward from ONE input block into 2 n output blocks (the sign-field convs 16 -> 96 / 192) on f16 pieces: the strip
    // kernel with the weights of 2 output blocks per wave in registers, grid.z = n
    if (!cin1 && conv_split_math(true) && a.Gk == 1 && a.kt == 1 && a.GnTot % 2 == 0 && a.cout_real == 16 * a.GnTot && !a.w_tap_stride &&
        !a.addend_first) {
      const int nstrip = cdiv(cdiv(a.Wo, 16), 2);
      const long bt = (long)(a.B / a.wsets) * a.To;
      const int gzs = a.GnTot / 2;       // (two blocks = one 128-byte line of a 96-channel pixel; three blocks per wave spilled)
      const int hrun = pick_hrun(a.Ho, bt * nstrip, (long)a.wsets * gzs, 2);
      const int nhrun = cdiv(a.Ho, hrun);
      const long nunits = bt * nhrun * nstrip;
      auto kern = conv16_split_kernel<FWD_S2, true, SC_ACT, 2, 1>;
      const dim3 grid(one_round(cdiv(nunits, 4), (long)a.wsets * gzs, wg_per_cu(kern, 256)), a.wsets, gzs);
      hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, a, nstrip, hrun, nhrun);
      return IDEE_OK;
    }
  }
  if constexpr (MODE == DGRAD_S2) {
    const bool old_path = env_is("IDEE_DGRAD_TO16", "wide");
    if (!cin1 && !old_path && a.kt == 1 && a.GnTot == 1 && a.cout_real == 16 && a.Gk >= 2 && a.Gk <= 12 && a.o_sW % 4 == 0) {
      // (1,3,3) data gradient into one 16-channel block: K split over the waves of a workgroup
      const int nstrip = cdiv(cdiv(a.Wo, 16), 2);
      const int nsubw = cdiv(a.Wo, 16), lone = nsubw & 1, nfull = nstrip - lone;
      const size_t lds = (size_t)a.Gk * (3 * kC16Row * 16 + 2 * 2 * 256) * 4;
      auto kern = dgrad_k133_to16_kernel;
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return idee::fail(IDEE_ERR_LAUNCH, "conv_dgrad: cannot reserve %zu bytes of LDS", lds);
      int per_cu = 1;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * a.Gk, lds) != hipSuccess || per_cu < 1) per_cu = 1;
      const long rows = (long)(a.B / a.wsets) * a.To * a.Ho * nstrip;
      if (rows * 10 >= (1L << 31)) return idee::fail(IDEE_ERR_UNSUPPORTED, "conv_dgrad: %ld strip rows exceed the 32-bit work partition", rows);
      long g = imax(((long)kCUs * per_cu) / a.wsets, 1);
      long cap = imax(rows / 8, 1);
      if (const int v = opt_int("IDEE_CONV_HRUN")) { if (v > 0) { cap = cdiv(rows, v); g = cap; } }
      hipLaunchKernelGGL(kern, dim3((unsigned)imax(imin(g, cap), 1), a.wsets), dim3(64 * a.Gk), lds, stream, a, nstrip, nfull);
      return IDEE_OK;
    }
  }
  if constexpr (MODE == FWD_S2 || MODE == DGRAD_S2) {
    const long slab_bytes = (long)a.Gk * NT * 256 * 4;
    if (!cin1 && gz == 1 && (NT >= 2 || (a.Gk >= 2 && a.Gk <= 6)) && a.cout_real != 1 && 2 * slab_bytes <= 150 * 1024 && a.Gk * NT * 64 <= 5 * 512) {
      {   // workgroups per (b, t) plane: the chip's workgroups of this batch item spread over its To planes, and no more than
          // the plane has chunks of 32 column blocks (8 waves x 4)
        const int g_b = kCUs / a.B > 0 ? kCUs / a.B : 1;
        a.hgroups = imin(imax(g_b / a.To, 1), cdiv((long)a.Ho * a.nsub, 32));
        a.tiles = a.To * a.hgroups;
      }
      const bool wide_f22 = (MODE == DGRAD_S2) && conv_f22_math();    // f16 pieces with the per-chunk gradient scale
      // (the split kernel keeps a tap's source offsets inside one batch item as 32-bit element counts: a larger batch item runs
      // on the exact kernel below, whose offsets are 64-bit)
      const bool off32 = (long)(a.Ti - 1) * a.i_sT + (long)(a.Hi - 1) * a.i_sH + (long)(a.Wi - 1) * a.i_sW < (1L << 31) - (1L << 24);
      if ((wide_f22 || conv_split_math(MODE == FWD_REP || MODE == FWD_S2)) && off32 && wide_split_shape(a.Gk, a.GnTot)) {
        // f16 pieces (forward; data gradient under f22) with the piece fragments of the packed buffer: two piece slabs; bf16 pieces
        // (data gradient under `split`) and buffers packed without pieces: two piece slabs + one raw slab
        const bool f16p = MODE == FWD_S2 || wide_f22;
        const bool pre = f16p && a.wpieces != nullptr;
        const long lds2 = pre ? 2 * slab_bytes : 3 * slab_bytes;
#define LCS(NT_)                                                                                               \
  do {                                                                                                         \
    auto kern = pre ? conv_wide_split_kernel<MODE, NT_, MODE == DGRAD_S2 ? SC_GRAD : SC_ACT, true> \
                : wide_f22 ? conv_wide_split_kernel<MODE, NT_, MODE == DGRAD_S2 ? SC_GRAD : SC_ACT> \
                         : conv_wide_split_kernel<MODE, NT_, MODE == FWD_S2 ? SC_ACT : SC_NONE>; \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) \
      return idee::fail(IDEE_ERR_LAUNCH, "conv: cannot reserve %ld bytes of LDS", lds2);                         \
    const dim3 grid(one_round(a.tiles, a.B, 1), a.B);                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds2, stream, a);                                                \
  } while (0)
        if (NT == 6) LCS(6); else if (NT == 3) LCS(3); else LCS(2);
#undef LCS
        return IDEE_OK;
      }
      const bool res = a.ntaps * slab_bytes <= 150 * 1024;      // every tap's weights stay in LDS
      const long lds_bytes = res ? a.ntaps * slab_bytes : 2 * slab_bytes;
#define LCW2(NT_, RES_)                                                                                        \
  do {                                                                                                         \
    auto kern = conv_wide_kernel<MODE, NT_, RES_>;                                                             \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) \
      return idee::fail(IDEE_ERR_LAUNCH, "conv: cannot reserve %ld bytes of LDS", lds_bytes);                    \
    const dim3 grid(one_round(a.tiles, a.B, 1), a.B);                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, stream, a);                                           \
  } while (0)
#define LCW(NT_) do { if (res) LCW2(NT_, true); else LCW2(NT_, false); } while (0)
      if (NT == 6) LCW(6); else if (NT == 3) LCW(3); else if (NT == 2) LCW(2); else LCW(1);
#undef LCW
#undef LCW2
      return IDEE_OK;
    }
  }
  if (a.no_edge_atomics) return idee::fail(IDEE_ERR_INVALID_ARG, "conv_dgrad: edge folding was requested for a kernel that does not implement it");
#define LC(NT_, WL_, C1_)                                                                                     \
  do {                                                                                                        \
    auto kern = conv_kernel<MODE, NT_, WL_, C1_>;                                                             \
    const dim3 grid(one_round(a.tiles, (long)a.B * gz, wg_per_cu(kern, 256)), a.B, gz);                       \
    hipLaunchKernelGGL(kern, grid, block, 0, stream, a);                                                      \
  } while (0)
  if (cin1) { if (NT == 1) LC(1, false, true); else if (NT == 2) LC(2, false, true); else if (NT == 3) LC(3, false, true); else LC(6, false, true); }
  else if (wlds) LC(1, true, false);
  else if (NT == 1) LC(1, false, false);
  else if (NT == 2) LC(2, false, false);
  else if (NT == 3) LC(3, false, false);
  else LC(6, false, false);
#undef LC
  return IDEE_OK;
}

}  // namespace

static long frag_floats(const idee_conv_desc* d) { return (long)d->wsets * n_taps(d) * (d->Cin / 16) * ((d->Cout + 15) / 16) * 256; }
extern "C" long idee_conv_packed_floats(const idee_conv_desc* d) {
  if (!d || d->Cin % 16) return -1;
  const long n = frag_floats(d);
  return packs_pieces(d) ? 2 * n + 64 : n;      // [f32 fragments][f16 piece fragments][scale exponents]
}
extern "C" long idee_conv_wgrad_floats(const idee_conv_desc* d) {
  if (!d) return -1;
  return (long)kWgradCopies * d->wsets * n_taps(d) * d->Cin * (((d->Cout + 15) / 16) * 16);      // see conv_common.h kWgradCopies
}

extern "C" int idee_conv_pack_weights(const idee_conv_desc* d, const float* w, void* packed_fwd_, void* packed_dgrad_,
                                      idee_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  IDEE_CHECK_ARG(w && (packed_fwd_ || packed_dgrad_), "conv_pack_weights: null argument");
  if (d->act_dtype == IDEE_DTYPE_BF16 && d->Cout != 1) {      // (Cout == 1: the streaming kernels read the f32 fragments)
    if (int rc = bf16_pack(d, w, packed_fwd_, packed_dgrad_, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_pack_weights");
    return IDEE_OK;
  }
  float* packed_fwd = (float*)packed_fwd_;
  float* packed_dgrad = (float*)packed_dgrad_;
  const long total = frag_floats(d);
  hipLaunchKernelGGL(pack_kernel, dim3(imin(cdiv(total, 256), 2048)), dim3(256), 0, (hipStream_t)stream, w, packed_fwd,
                     packed_dgrad, d->wsets, d->Cout, d->Cin, n_taps(d));
  if (packs_pieces(d)) {      // ready f16 piece fragments for the wide split kernels (whatever the arithmetic option says NOW: the
                              // library looks its options up at launch time)
    const int Gi = d->Cin / 16, Go = d->Cout / 16;
    float* bufs[2] = {packed_fwd, packed_dgrad};
    const int gk[2] = {Gi, Go}, gn[2] = {Go, Gi};
    for (int i = 0; i < 2; ++i) {
      if (!bufs[i] || !wide_split_shape(gk[i], gn[i])) continue;
      unsigned* mbuf = reinterpret_cast<unsigned*>(bufs[i] + 2 * total);
      unsigned* pieces = reinterpret_cast<unsigned*>(bufs[i] + total);
      if (hipMemsetAsync(mbuf, 0, 64 * sizeof(unsigned), (hipStream_t)stream) != hipSuccess)
        return idee::fail(IDEE_ERR_LAUNCH, "conv_pack_weights: hipMemsetAsync failed");
      const long per_set = total / d->wsets;
      const int parts = (int)imin(imax(per_set / 8192, 1), 32);
      hipLaunchKernelGGL(pack_wscale_kernel, dim3(parts, d->wsets), dim3(256), 0, (hipStream_t)stream, bufs[i], mbuf, per_set, parts);
      const long units = total / 8;
      hipLaunchKernelGGL(pack_pieces_kernel, dim3(imin(cdiv(units, 256), 1024)), dim3(256), 0, (hipStream_t)stream, bufs[i], mbuf,
                         pieces, d->wsets, n_taps(d), gk[i], pick_nt(gn[i]));
    }
  }
  IDEE_LAUNCH_CHECK("conv_pack_weights");
  return IDEE_OK;
}

extern "C" int idee_conv_unpack_wgrad(const idee_conv_desc* d, const float* dw_packed, float* dw, idee_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  IDEE_CHECK_ARG(dw_packed && dw, "conv_unpack_wgrad: null argument");
  const long total = (long)d->wsets * d->Cout * d->Cin * n_taps(d);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(imin(cdiv(total, 256), 2048)), dim3(256), 0, (hipStream_t)stream,
                     dw_packed, dw, d->wsets, d->Cout, d->Cin, n_taps(d), ((d->Cout + 15) / 16) * 16);
  IDEE_LAUNCH_CHECK("conv_unpack_wgrad");
  return IDEE_OK;
}

extern "C" int idee_conv_fwd(const idee_conv_desc* d, const void* x_, const void* w_packed_, const float* bias, void* y_,
                             idee_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  IDEE_CHECK_ARG(x_ && w_packed_ && y_, "conv_fwd: null argument");
  const bool bf = d->act_dtype == IDEE_DTYPE_BF16;
  const float* x = (const float*)x_; const float* w_packed = (const float*)w_packed_; float* y = (float*)y_;
  if (is_c1(d)) {
    C1Args c; fill_c1(c, d);
    c.x = x_; c.w = w_packed; c.bias = bias; c.y = y;
    if (int rc = launch_c1(0, c, bf, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_fwd");
    return IDEE_OK;
  }
  if (bf) {
    if (int rc = check_bf16(d)) return rc;
    if (int rc = bf16_fwd(d, x_, w_packed_, bias, y_, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_fwd");
    return IDEE_OK;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.wp = w_packed; a.bias = bias; a.relu_ref = nullptr; a.out = y;
  if (packs_pieces(d) && wide_split_shape(d->Cin / 16, d->Cout / 16)) {
    a.wpieces = reinterpret_cast<const unsigned*>(w_packed + frag_floats(d));
    a.wscale_k = reinterpret_cast<const int*>(w_packed + 2 * frag_floats(d));
  }
  a.B = d->B; a.wsets = d->wsets;
  a.Ti = d->T; a.Hi = d->H; a.Wi = d->W;
  a.To = out_T(d); a.Ho = d->H; a.Wo = d->W;
  a.Gk = d->Cin / 16; a.GnTot = (d->Cout + 15) / 16; a.cout_real = d->Cout; a.relu = d->relu; a.ntaps = n_taps(d); a.kt = mode_kt(d);
  a.i_sB = d->x_sB; a.i_sT = d->x_sT; a.i_sH = d->x_sH; a.i_sW = d->x_sW; a.i_sG = d->x_sG;
  a.o_sB = d->y_sB; a.o_sT = d->y_sT; a.o_sH = d->y_sH; a.o_sW = d->y_sW; a.o_sG = d->y_sG;
  a.i_sU = d->x_sU; a.i_gpu = d->x_gpu > 0 ? d->x_gpu : (1 << 30); a.o_sU = d->y_sU; a.o_gpu = d->y_gpu > 0 ? d->y_gpu : (1 << 30);
  split_tiles(a, a.Wo);
  if (d->mode == IDEE_CONV_K333_REPLICATE) { if (int rc = launch_conv<FWD_REP>(a, false, (hipStream_t)stream)) return rc; }
  else if (int rc = launch_conv<FWD_S2>(a, false, (hipStream_t)stream)) return rc;
  IDEE_LAUNCH_CHECK("conv_fwd");
  return IDEE_OK;
}

extern "C" int idee_conv_dgrad(const idee_conv_desc* d, const void* dy, const void* w_packed_dgrad, const void* relu_ref,
                               void* dx, int accumulate, idee_stream_t stream) {
  return idee_conv_dgrad_add(d, dy, w_packed_dgrad, relu_ref, accumulate ? dx : nullptr, dx, stream);
}

extern "C" int idee_conv_dgrad_add(const idee_conv_desc* d, const void* dy_, const void* w_packed_dgrad_, const void* relu_ref_,
                                   const void* addend_, void* dx_, idee_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  IDEE_CHECK_ARG(dy_ && w_packed_dgrad_ && dx_, "conv_dgrad: null argument");
  const bool bf = d->act_dtype == IDEE_DTYPE_BF16;
  const float* dy = (const float*)dy_; const float* w_packed_dgrad = (const float*)w_packed_dgrad_;
  const float* relu_ref = (const float*)relu_ref_; const float* addend = (const float*)addend_; float* dx = (float*)dx_;
  if (is_c1(d) && (d->T == 2 || bf)) {      // (f32, T = 4: the padded MFMA path is faster for the data gradient: 0.35 vs 0.53 ms at config #4)
    IDEE_CHECK_ARG(!addend || addend == dx, "conv_dgrad (Cout = 1): a separate addend is not built");
    const int accumulate = addend != nullptr;
    C1Args c; fill_c1(c, d);
    c.dy = dy; c.w = w_packed_dgrad; c.relu_ref = relu_ref_; c.dx = dx_; c.accumulate = accumulate;
    if (int rc = launch_c1(1, c, bf, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_dgrad");
    return IDEE_OK;
  }
  if (bf) {
    if (int rc = check_bf16(d)) return rc;
    if (int rc = bf16_dgrad(d, dy_, w_packed_dgrad_, relu_ref_, addend_, dx_, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_dgrad");
    return IDEE_OK;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = dy; a.wp = w_packed_dgrad; a.bias = nullptr; a.relu_ref = relu_ref; a.out = dx; a.addend = addend;
  if (packs_pieces(d) && wide_split_shape(d->Cout / 16, d->Cin / 16)) {
    a.wpieces = reinterpret_cast<const unsigned*>(w_packed_dgrad + frag_floats(d));
    a.wscale_k = reinterpret_cast<const int*>(w_packed_dgrad + 2 * frag_floats(d));
  }
  a.B = d->B; a.wsets = d->wsets;
  a.Ti = out_T(d); a.Hi = d->H; a.Wi = d->W;
  a.To = d->T; a.Ho = d->H; a.Wo = d->W;
  a.Gk = (d->Cout + 15) / 16; a.GnTot = d->Cin / 16; a.cout_real = d->Cin; a.relu = 0; a.ntaps = n_taps(d); a.kt = mode_kt(d);
  a.i_sB = d->y_sB; a.i_sT = d->y_sT; a.i_sH = d->y_sH; a.i_sW = d->y_sW; a.i_sG = d->y_sG;
  a.o_sB = d->x_sB; a.o_sT = d->x_sT; a.o_sH = d->x_sH; a.o_sW = d->x_sW; a.o_sG = d->x_sG;
  a.i_sU = d->y_sU; a.i_gpu = d->y_gpu > 0 ? d->y_gpu : (1 << 30); a.o_sU = d->x_sU; a.o_gpu = d->x_gpu > 0 ? d->x_gpu : (1 << 30);
  const bool cin1 = d->Cout == 1;
  if (d->mode == IDEE_CONV_K333_REPLICATE) {
    IDEE_CHECK_ARG(!cin1, "conv_dgrad: Cout == 1 is only built for the (2,3,3) mode");
    split_tiles(a, a.Wo + 2);
    // the 16 -> 16 strip kernels (exact and split) fold the two virtual frame columns inside the wave (ConvArgs::no_edge_atomics);
    // this mirrors launch_conv<DGRAD_REP>'s choice of those kernels, which refuses the flag for any other
    a.no_edge_atomics = a.GnTot == 1 && a.Gk == 1 && a.cout_real == 16 && a.Wo % 16 != 15 && a.Wo >= 2;
    if (addend != dx && !a.no_edge_atomics) {   // interior columns are plain stores; the two edge columns are atomics on top of 0 / the addend
      IDEE_CHECK_ARG(d->x_sG == 16 && d->x_gpu <= 0, "conv_dgrad (replicate): grouped dx layouts are not built");
      const long total = (long)d->B * d->T * d->H * 2 * (d->Cin / 4);
      hipLaunchKernelGGL(init_w_edges_kernel, dim3(imin(cdiv(total, 256), 2048)), dim3(256), 0, (hipStream_t)stream, dx, addend,
                         d->B, d->T, d->H, d->W, d->Cin, d->x_sB, d->x_sT, d->x_sH, d->x_sW);
    }
    if (int rc = launch_conv<DGRAD_REP>(a, false, (hipStream_t)stream)) return rc;
  } else {
    split_tiles(a, a.Wo);
    if (int rc = launch_conv<DGRAD_S2>(a, cin1, (hipStream_t)stream)) return rc;
  }
  IDEE_LAUNCH_CHECK("conv_dgrad");
  return IDEE_OK;
}

extern "C" int idee_conv_wgrad(const idee_conv_desc* d, const void* x_, const void* dy_, float* dw_packed, float* dbias,
                               idee_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  IDEE_CHECK_ARG(x_ && dy_ && dw_packed, "conv_wgrad: null argument");
  const bool bf = d->act_dtype == IDEE_DTYPE_BF16;
  const float* x = (const float*)x_; const float* dy = (const float*)dy_;
  if (is_c1(d)) {
    C1Args c; fill_c1(c, d);
    c.x = x_; c.dy = dy; c.dwp = dw_packed; c.dbias = dbias;
    if (int rc = launch_c1(2, c, bf, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_wgrad");
    return IDEE_OK;
  }
  if (bf) {
    if (int rc = check_bf16(d)) return rc;
    if (int rc = bf16_wgrad(d, x_, dy_, dw_packed, dbias, (hipStream_t)stream)) return rc;
    IDEE_LAUNCH_CHECK("conv_wgrad");
    return IDEE_OK;
  }
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dy = dy; a.dwp = dw_packed; a.dbias = dbias;
  a.copy_stride = (long)d->wsets * n_taps(d) * d->Cin * (((d->Cout + 15) / 16) * 16);
  a.B = d->B; a.wsets = d->wsets; a.T = d->T; a.H = d->H; a.W = d->W; a.To = out_T(d);
  a.Cin = d->Cin; a.CoutPad = ((d->Cout + 15) / 16) * 16; a.cout_real = d->Cout; a.ntaps = n_taps(d); a.kt = mode_kt(d);
  a.x_sB = d->x_sB; a.x_sT = d->x_sT; a.x_sH = d->x_sH; a.x_sW = d->x_sW; a.x_sG = d->x_sG;
  a.y_sB = d->y_sB; a.y_sT = d->y_sT; a.y_sH = d->y_sH; a.y_sW = d->y_sW; a.y_sG = d->y_sG;
  a.x_sU = d->x_sU; a.x_gpu = d->x_gpu > 0 ? d->x_gpu : (1 << 30); a.y_sU = d->y_sU; a.y_gpu = d->y_gpu > 0 ? d->y_gpu : (1 << 30);
  a.rows = a.To * d->H;
  const int Ga = d->Cin / 16, Gb = a.CoutPad / 16;
  const long total_rows = (long)(d->B / d->wsets) * a.rows;
  const bool rep = d->mode == IDEE_CONV_K333_REPLICATE;
  hipStream_t st = (hipStream_t)stream;
#define LW(MODE_, TPW_, NTA_, NTB_)                                                                          \
  do {                                                                                                       \
    a.tap_groups = a.ntaps / TPW_; a.ga_groups = Ga / NTA_;                                                  \
    const int gy = a.tap_groups * a.ga_groups * (Gb / NTB_);                                                 \
    int gx = imin(cdiv(total_rows, 4), cdiv((long)kCUs * 4, (long)gy * d->wsets));                           \
    if (gx < 1) gx = 1;                                                                                      \
    a.row_chunks = gx;                                                                                       \
    hipLaunchKernelGGL((wgrad_kernel<MODE_, TPW_, NTA_, NTB_>), dim3(gx, gy, d->wsets), dim3(256), 0, st, a); \
  } while (0)
  if (d->Cout % 16 == 0 && ((Ga <= 2 && Gb <= 2 && d->mode != IDEE_CONV_K133_ZERO) || (Ga == 1 && (Gb == 1 || (d->mode == IDEE_CONV_K133_ZERO && Gb % 3 == 0))))) {
    // wave-private strip kernel: strips of 32 pixels, runs of rows per (b, t); 16 -> 16 in every mode, and the
    // (1,3,3) sign-field convs 16 -> 48k with 3 output blocks per wave
    const int nstrip = cdiv(cdiv(d->W, 16), 2);
    const int gz = (d->mode == IDEE_CONV_K133_ZERO && Gb > 1) ? Gb / 3 : Ga * Gb;      // (input block, output block group) pairs
    const long btr = (long)(d->B / d->wsets) * a.To;
    const int hr2 = pick_hrun(d->H, btr * nstrip, (long)d->wsets * gz, 2);
    const int nhr2 = cdiv(d->H, hr2);
    const long nunits = btr * nhr2 * nstrip;
#define LSTRIP(KERN)                                                                                          \
  do {                                                                                                        \
    auto kern = KERN;                                                                                         \
    hipLaunchKernelGGL(kern, dim3(one_round(cdiv(nunits, 4), (long)d->wsets * gz, wg_per_cu(kern, 256)), d->wsets, gz), \
                       dim3(256), 0, st, a, nstrip, hr2, nhr2);                                               \
  } while (0)
    const bool split = conv_split_math(false) && Gb == 1 && Ga == 1;
    const bool f22 = conv_f22_math() && Gb == 1 && Ga == 1;
    const bool f22_k133 = conv_f22_math() && Ga == 1 && d->mode == IDEE_CONV_K133_ZERO && Gb > 1 && Gb % 3 == 0 && !bf;
    // the 16 -> 16 split kernels take the W taps on the dy operand: strips of 32 PADDED pixels (wgrad16_shift_kernel)
    const int nstrip_s = cdiv(d->W + 2, 32);
    const int hr2_s = pick_hrun(d->H, btr * nstrip_s, (long)d->wsets * gz, 2);
    const int nhr2_s = cdiv(d->H, hr2_s);
    const long nunits_s = btr * nhr2_s * nstrip_s;
#define LSHIFT(KERN)                                                                                          \
  do {                                                                                                        \
    auto kern = KERN;                                                                                         \
    hipLaunchKernelGGL(kern, dim3(one_round(cdiv(nunits_s, 4), (long)d->wsets * gz, wg_per_cu(kern, 256)), d->wsets, gz), \
                       dim3(256), 0, st, a, nstrip_s, hr2_s, nhr2_s);                                         \
  } while (0)
    if (f22_k133) LSTRIP((wgrad16_split_kernel<FWD_S2, 1, true, 3>));     // the sign-field convs 16 -> 48 k (gz = Gb / 3 groups)
    else if (rep && f22) LSHIFT((wgrad16_shift_kernel<FWD_REP, 3, true>));
    else if (d->mode == IDEE_CONV_K233_S2_ZERO && f22) LSHIFT((wgrad16_shift_kernel<FWD_S2, 2, true>));
    else if (rep && split) LSHIFT((wgrad16_shift_kernel<FWD_REP, 3>));
    else if (d->mode == IDEE_CONV_K233_S2_ZERO && split) LSHIFT((wgrad16_shift_kernel<FWD_S2, 2>));
    else if (rep && !env_is("IDEE_WGRAD16_REGS", "1")) {
      // LDS-DMA staged kernel, cost-axis partition (see conv16_glds_kernel); a one-block last strip's rows weigh by their K-chunks
      auto kern = wgrad16_glds_kernel;
      const int nsubw = cdiv(d->W, 16), lone = nsubw & 1, nfull = nstrip - lone;
      const int kbn_l = imin(8, (imin(d->W - (nstrip - 1) * 32, 16) + 3) / 4);
      const int wl = 10; (void)kbn_l; (void)lone;     // (a one-block strip still runs all 8 K-chunks: see the kernel)
      const long rows_s = (long)(d->B / d->wsets) * ((a.To & 3) ? a.To * 4L : a.To) / 4 * d->H * nstrip;
      if (rows_s * 10 >= (1L << 31)) return idee::fail(IDEE_ERR_UNSUPPORTED, "conv_wgrad: %ld strip rows exceed the 32-bit work partition", rows_s);
      long g = imax(((long)kCUs * wg_per_cu(kern, 256)) / ((long)d->wsets * gz), 1);
      long cap = imax(rows_s / 8, 1);
      if (const int v = opt_int("IDEE_CONV_HRUN")) { if (v > 0) { cap = cdiv(rows_s, v); g = cap; } }
      hipLaunchKernelGGL(kern, dim3((unsigned)imax(imin(g, cap), 1), d->wsets, gz), dim3(256), 0, st, a, nstrip, nfull, wl);
    }
    else if (rep) LSTRIP((wgrad16_strip_kernel<FWD_REP, 3, 1>));
    else if (d->mode == IDEE_CONV_K233_S2_ZERO) LSTRIP((wgrad16_strip_kernel<FWD_S2, 2, 1>));
    else if (Gb == 1) LSTRIP((wgrad16_strip_kernel<FWD_S2, 1, 1>));
    else LSTRIP((wgrad16_strip_kernel<FWD_S2, 1, 3>));
#undef LSTRIP
#undef LSHIFT
  }
  else if (!rep && (Ga == Gb || Ga == 1) && (Gb == 6 || Gb == 3 || Gb == 2 || (Gb % 6 == 0 && Ga % 6 == 0)) && d->Cout % 16 == 0) {
    // wide LDS-staged kernel: segment width from the LDS budget (ring of 3 x rows + dy row, padded pixel strides)
    // channel sub-blocks of at most 6 x 6 groups per workgroup (wider convolutions: a grid of sub-blocks)
    const int ga_k = Ga > 6 ? 6 : Ga, gb_k = Gb > 6 ? 6 : Gb, nsubz = (Ga / ga_k) * (Gb / gb_k);
    const int PSX = 16 * ga_k + ((48 - (16 * ga_k) % 32) % 32), PSY = 16 * gb_k + ((48 - (16 * gb_k) % 32) % 32);
    const long budget = 150 * 1024 / 4;                                   // floats
    int segmax = (int)((budget - 6L * PSX) / (3L * PSX + PSY)) & ~3;
    if (segmax > 256) segmax = 256;
    const int nseg = cdiv(d->W, segmax);
    const int seg = ((cdiv(d->W, nseg) + 3) / 4) * 4;
    const long btk = (long)(d->B / d->wsets) * a.To * a.kt;
    int hrun = d->H;
    while (hrun > 8 && btk * cdiv(d->H, hrun) * nseg * d->wsets * nsubz < 3L * kCUs) hrun = (hrun + 1) / 2;
    const int nhrun = cdiv(d->H, hrun);
    const long nchunks = (long)(d->B / d->wsets) * a.To * nhrun * nseg;
    const size_t lds = (size_t)(3L * (seg + 2) * PSX + (long)((seg + 3) & ~3) * PSY) * 4;
    const dim3 grid(one_round(nchunks, (long)a.kt * d->wsets * nsubz, 1), d->wsets, a.kt * nsubz);
#define LWW(GA_, GB_, WAVES_)                                                                                         \
  do {                                                                                                                \
    auto kern = wgrad_wide_lds_kernel<GA_, GB_, WAVES_>;                                                              \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)   \
      return idee::fail(IDEE_ERR_LAUNCH, "conv_wgrad: cannot reserve %zu bytes of LDS", lds);                          \
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_), lds, st, a, seg, nseg, hrun, nhrun);                            \
  } while (0)
    // (the f16 kernel keeps its in-row source offsets as 32-bit element counts)
    const bool off32 = (long)d->W * a.x_sW + (long)(Ga + 1) * (a.x_sU > a.x_sG ? a.x_sU : a.x_sG) < (1L << 31) &&
                       (long)d->W * a.y_sW + (long)(Gb + 1) * (a.y_sU > a.y_sG ? a.y_sU : a.y_sG) < (1L << 31);
    if (conv_f22_math() && Ga == Gb && (ga_k == 6 || Ga == 3 || Ga == 2) && !bf && off32) {
      // f16 pieces at f32 level: 32- or 64-pixel segments (K = 32 pixels per MFMA), piece images of 64 B per pixel and channel
      // block.  6 x 6 blocks run as two 6 x 3 halves (84 accumulator registers instead of 168).  (32-pixel segments — 49 KB of LDS,
      // more workgroups per CU — measured slower, 1.26 vs 0.93 ms: the per-row barriers and latencies dominate, not the MFMAs)
      const int gb2 = ga_k == 6 ? 3 : gb_k, nsubz2 = (Ga / ga_k) * (Gb / gb2);
      int seg2 = 64;
      if (const int v = opt_int("IDEE_WGRAD_WIDE_SEG")) { if (v == 32 || v == 64) seg2 = v; }   // (test / tuning knob)
      const int nseg2 = cdiv(d->W, seg2);
      int hrun2 = d->H;
      while (hrun2 > 8 && btk * cdiv(d->H, hrun2) * nseg2 * d->wsets * nsubz2 < 6L * kCUs) hrun2 = (hrun2 + 1) / 2;
      const int nhrun2 = cdiv(d->H, hrun2);
      const long nchunks2 = (long)(d->B / d->wsets) * a.To * nhrun2 * nseg2;
      const size_t lds2 = (size_t)(3L * (seg2 + 2) * (16 * ga_k + 8) + (long)seg2 * (16 * gb2 + 8)) * 4;
      const int wgcu = imax(1, imin(3, (int)(150 * 1024 / (lds2 + 64))));
      const dim3 grid2(one_round(nchunks2, (long)a.kt * d->wsets * nsubz2, wgcu), d->wsets, a.kt * nsubz2);
#define LWS(GA_, GB_)                                                                                                 \
  do {                                                                                                                \
    auto kern = wgrad_wide_split_kernel<GA_, GB_, 8>;                                                                 \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)  \
      return idee::fail(IDEE_ERR_LAUNCH, "conv_wgrad: cannot reserve %zu bytes of LDS", lds2);                         \
    hipLaunchKernelGGL(kern, grid2, dim3(512), lds2, st, a, seg2, nseg2, hrun2, nhrun2);                              \
  } while (0)
      if (ga_k == 6) LWS(6, 3); else if (Ga == 3) LWS(3, 3); else LWS(2, 2);
#undef LWS
    } else
    // 16 input channels (the sign field): one tap per wave, 9 waves
    if (Ga == 1) { if (Gb == 6) LWW(1, 6, 9); else if (Gb == 3) LWW(1, 3, 9); else LWW(1, 2, 9); }
    else if (!env_is("IDEE_WGRAD_WIDE", "regs")) {
      // exact f32, LDS-DMA staged one row ahead (wgrad_wide_dma_kernel): 4 x rows + 2 dy rows, unpadded
      const int ci = 16 * ga_k, co = 16 * gb_k;
      int segm = (int)((150L * 1024 / 4 - 8L * ci) / (4L * ci + 2L * co)) & ~3;
      if (segm > 256) segm = 256;
      const int nsegd = cdiv(d->W, segm);
      const int segd = ((cdiv(d->W, nsegd) + 3) / 4) * 4;
      int hrund = d->H;
      while (hrund > 8 && btk * cdiv(d->H, hrund) * nsegd * d->wsets * nsubz < 3L * kCUs) hrund = (hrund + 1) / 2;
      const int nhrund = cdiv(d->H, hrund);
      const long nchunksd = (long)(d->B / d->wsets) * a.To * nhrund * nsegd;
      const size_t ldsd = (size_t)(4L * (segd + 2) * ci + 2L * segd * co) * 4;
      const dim3 gridd(one_round(nchunksd, (long)a.kt * d->wsets * nsubz, 1), d->wsets, a.kt * nsubz);
#define LWD(GA_, GB_)                                                                                                 \
  do {                                                                                                                \
    auto kern = wgrad_wide_dma_kernel<GA_, GB_, 8>;                                                                   \
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd) != hipSuccess)  \
      return idee::fail(IDEE_ERR_LAUNCH, "conv_wgrad: cannot reserve %zu bytes of LDS", ldsd);                         \
    hipLaunchKernelGGL(kern, gridd, dim3(512), ldsd, st, a, segd, nsegd, hrund, nhrund);                              \
  } while (0)
      if (ga_k == 6) LWD(6, 6); else if (Ga == 3) LWD(3, 3); else LWD(2, 2);
#undef LWD
    }
    else if (ga_k == 6) LWW(6, 6, 8); else if (Ga == 3) LWW(3, 3, 8); else LWW(2, 2, 8);
#undef LWW
  }
  else if (Ga == 1 && Gb == 1) { if (rep) LW(FWD_REP, 27, 1, 1); else if (a.kt == 2) LW(FWD_S2, 18, 1, 1); else LW(FWD_S2, 9, 1, 1); }
  else {
    const int na = pick_nt(Ga), nb = pick_nt(Gb);
    // one tap per wave; channel blocks up to 6x6
#define LWT(NA_, NB_) do { if (rep) LW(FWD_REP, 1, NA_, NB_); else LW(FWD_S2, 1, NA_, NB_); } while (0)
    if (na == 6 && nb == 6) LWT(6, 6); else if (na == 6 && nb == 1) LWT(6, 1);
    else if (na == 3 && nb == 3) LWT(3, 3); else if (na == 3 && nb == 1) LWT(3, 1);
    else if (na == 2 && nb == 2) LWT(2, 2); else if (na == 2 && nb == 1) LWT(2, 1);
    else LWT(1, 1);   // any other block counts: one 16x16 block per wave
#undef LWT
  }
#undef LW
  IDEE_LAUNCH_CHECK("conv_wgrad");
  return IDEE_OK;
}
