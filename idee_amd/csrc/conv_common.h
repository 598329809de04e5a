// Definitions shared by the convolution translation units (conv_mfma.hip: exact-f32 / split MFMA kernels on f32 tensors;
// conv_bf16.hip: the bf16-storage kernels on v_mfma_f32_16x16x32_bf16): argument blocks, the tap / source-row maps of the
// four convolution modes, channel-group addressing.
#pragma once
#include "common.h"

namespace idee_conv {

enum { FWD_REP = 0, DGRAD_REP = 1, FWD_S2 = 2, DGRAD_S2 = 3 };

struct ConvArgs {
  const float* in;
  const float* wp;        // packed fragments
  const float* bias;      // fwd: (wsets, CoutPad) or null
  const float* relu_ref;  // dgrad: multiply by (relu_ref > 0), same layout as out, or null
  float* out;
  int B, wsets;
  int Ti, Hi, Wi;         // extent of `in`
  int To, Ho, Wo;         // extent of `out`
  int Gk;                 // K groups (16 channels each) of `in`
  int GnTot;              // output channel groups in total
  int cout_real;          // real output channels (1 => scalar store)
  int relu;
  const float* addend;    // out = result + addend (same layout as out; may BE out) or null
  int ntaps;
  int kt;                 // temporal taps of the (kt,3,3) zero-padded modes: 2 (stride 2 in T) or 1 (stride 1)
  long i_sB, i_sT, i_sH, i_sW, i_sG, i_sU;
  long o_sB, o_sT, o_sH, o_sW, o_sG, o_sU;
  int i_gpu, o_gpu;                      // 16-channel groups per unit: group g lives at (g / gpu) * sU + (g % gpu) * sG
  int nsub, wtiles, sub_base, sub_rem;   // 16-pixel column blocks per row and their split into wave tiles
  int hgroups, tiles;                    // ceil(Ho/4); To*hgroups*wtiles
  // 16 -> 16 sub-problems of a wider convolution run on the 16 -> 16 strip kernel (conv16_glds_kernel): where the (tap) fragments
  // of the sub-block live in the packed buffer, and the epilogue order (0 everywhere = the plain 16 -> 16 layout)
  int w_tap_stride;                      // floats between consecutive taps (default 256)
  long w_set_stride;                     // floats between weight sets (default ntaps * 256)
  int bias_set_stride;                   // floats between the bias vectors of weight sets (default 16)
  int addend_first;                      // 1: out = relu(acc + addend + bias) (a partial sum), 0: relu(acc + bias) + addend
  // replicate-mode data gradient: the two virtual frame columns (-1, W) are folded onto columns 0 / W - 1 INSIDE the wave (a DPP
  // row shift between the two lanes that hold them — they share a 16-column tile unless W % 16 == 15) instead of with atomics on
  // top of a pre-initialised edge: set by idee_conv_dgrad_add for the kernels that implement it (conv16_split_kernel)
  int no_edge_atomics;
  // wide split kernels (conv_wide_split_kernel, f16 pieces): the weights as READY piece fragments, written once per step by
  // idee_conv_pack_weights behind the f32 fragments of the same buffer — per weight set and tap the LDS image of the kernel's piece
  // slab [group pair][output block][hi | lo][lane][4 words] — and the sets' power-of-two scale exponents; null: not packed
  const unsigned* wpieces;
  const int* wscale_k;
};

struct Slot { int src, tap; bool ok; };

// offset of 16-channel group g: units of `gpu` groups (e.g. the V slices of z_q), sU apart; groups sG apart inside a unit
__device__ __forceinline__ long goff(int g, int gpu, long sU, long sG) { return (long)(g / gpu) * sU + (long)(g % gpu) * sG; }

template <int MODE>
__device__ __forceinline__ Slot slot_t(int o, int t, int Ti, int kt) {
  Slot s;
  if (MODE == FWD_REP) { s.src = min(max(t + o - 1, 0), Ti - 1); s.tap = o; s.ok = true; }
  else if (MODE == DGRAD_REP) {
    const int q = t - (o - 1);
    const bool in = q >= 0 && q < Ti;
    s.src = min(max(q, 0), Ti - 1); s.tap = in ? o : 2 - o; s.ok = true;
  } else if (MODE == FWD_S2) { s.src = kt * t + o; s.tap = o; s.ok = (o < kt) && (s.src < Ti); }
  else { s.src = kt == 2 ? t >> 1 : t; s.tap = kt == 2 ? t & 1 : 0; s.ok = (o == 0) && (s.src < Ti); }
  return s;
}
template <int MODE>
__device__ __forceinline__ Slot slot_h(int o, int h, int Hi) {
  Slot s;
  if (MODE == FWD_REP) { s.src = min(max(h + o - 1, 0), Hi - 1); s.tap = o; s.ok = true; }
  else if (MODE == DGRAD_REP) {
    const int q = h - (o - 1);
    const bool in = q >= 0 && q < Hi;
    s.src = min(max(q, 0), Hi - 1); s.tap = in ? o : 2 - o; s.ok = true;
  } else if (MODE == FWD_S2) { s.src = h + o - 1; s.tap = o; s.ok = s.src >= 0 && s.src < Hi; }
  else { s.src = h - (o - 1); s.tap = o; s.ok = s.src >= 0 && s.src < Hi; }
  return s;
}
// per-lane source column for output column index p (p counts the padded frame in DGRAD_REP)
template <int MODE>
__device__ __forceinline__ int src_w(int o, int p, int Wi, bool& ok) {
  int q;
  if (MODE == FWD_REP) { q = min(max(p + o - 1, 0), Wi - 1); ok = true; }
  else if (MODE == DGRAD_REP) { q = (p - 1) - (o - 1); ok = q >= 0 && q < Wi; }
  else if (MODE == FWD_S2) { q = p + o - 1; ok = q >= 0 && q < Wi; }
  else { q = p - (o - 1); ok = q >= 0 && q < Wi; }
  return q;
}

// The packed weight-gradient buffer holds kWgradCopies partial copies of (wsets, ntaps, Cin, CoutPad): a workgroup may flush
// its register accumulators into copy blockIdx.x % kWgradCopies, idee_conv_unpack_wgrad sums the copies.  Measured on the
// 96 -> 96 weight gradient with 8 copies: NO gain (0.691 ms either way) — the tile flush (11 M float atomics onto 166 k
// addresses, 64 workgroups per address) is not what was slow; what was, 0.29 of 0.69 ms, were 4,096 per-thread atomics per
// workgroup onto the 48 bias-gradient addresses (now reduced in LDS first).  Hence ONE copy; the mechanism stays for shapes
// where few addresses take many flushes.
constexpr int kWgradCopies = 1;
struct WgradArgs {
  const float* x;
  const float* dy;
  float* dwp;     // kWgradCopies x (wsets, ntaps, Cin, CoutPad); kernels that do not spread their flush use copy 0
  long copy_stride;   // floats per copy
  float* dbias;   // (wsets, CoutPad) or null
  int B, wsets;
  int T, H, W;    // x extent
  int To;         // dy extent in T (H, W equal)
  int Cin, CoutPad, cout_real;
  int ntaps;
  int kt;         // temporal taps (= temporal stride) of the zero-padded modes: 2 or 1
  long x_sB, x_sT, x_sH, x_sW, x_sG, x_sU;
  long y_sB, y_sT, y_sH, y_sW, y_sG, y_sU;
  int x_gpu, y_gpu;
  int rows;       // To * H  (pixel rows per batch item)
  int row_chunks; // gridDim.x
  int tap_groups; // ntaps / TPW
  int ga_groups;  // (Cin/16) / NTA
};



// conv_bf16.hip: the act_dtype == IDEE_DTYPE_BF16 implementations behind idee_conv_* (Cout a multiple of 16)
int bf16_pack(const idee_conv_desc* d, const float* w, void* packed_fwd, void* packed_dgrad, hipStream_t stream);
int bf16_fwd(const idee_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y, hipStream_t stream);
int bf16_dgrad(const idee_conv_desc* d, const void* dy, const void* w_packed_dgrad, const void* relu_ref, const void* addend,
               void* dx, hipStream_t stream);
int bf16_wgrad(const idee_conv_desc* d, const void* x, const void* dy, float* dw_packed, float* dbias, hipStream_t stream);

}  // namespace idee_conv
