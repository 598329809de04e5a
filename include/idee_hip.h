/*
 * idee_hip.h — C-ABI of libidee_hip.so: the MI355X (gfx950) kernels behind the IDEE
 * encoder -> codebook -> classifier hot path.
 *
 * The reference (HakamShams/IDEE) is pure PyTorch: it has no native extension and no FFI
 * surface to mirror.  The entry points below are therefore the ones a binding for this path
 * WOULD bind — one per fused unit of the path — and each cites the reference code it replaces
 * (paths relative to the reference repo).  The reference-side binding is the ctypes stub in
 * INTEGRATION.md; idee_amd/_lib.py is the one this repo ships.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is a DEVICE pointer to contiguous f32 unless noted;
 *   - activations are channels-last tokens: (N, V, T, H, W, C)  [reference: (N, V, C, T, H, W)];
 *   - the V per-variable weight sets are stacked on a leading V axis; batch item b = n*V + v
 *     uses weight set b % V;
 *   - the caller allocates every output and workspace (size queries are provided);
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); no call
 *     allocates, synchronises or touches host memory asynchronously;
 *   - return value 0 = ok; non-zero = IDEE_ERR_*; message via idee_last_error() (thread-local);
 *   - gradient outputs named d* that are documented "accumulate" must be zero-initialised by
 *     the caller (they are added to with float atomics); all others are overwritten;
 *   - re-entrant: no global mutable state on the data path; the only process-wide state is the
 *     kernel-selection option table (idee_set_option), read under a reader lock at each launch.
 */
#ifndef IDEE_HIP_H
#define IDEE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* idee_stream_t; /* hipStream_t */

enum { IDEE_OK = 0, IDEE_ERR_INVALID_ARG = 1, IDEE_ERR_UNSUPPORTED = 2, IDEE_ERR_LAUNCH = 3 };

/* Activation storage of the tensors that travel between the fused units (tokens, conv inputs / outputs and their
 * gradients): f32 (the reference's arithmetic, default) or bf16 (BASELINE config #4: "bf16 activations/GEMMs with fp32
 * accumulate and fp32 LFQ").  Under IDEE_DTYPE_BF16 every activation pointer of a call that carries an `act_dtype` field or
 * argument addresses bf16 elements (strides stay in ELEMENTS), GEMM-shaped work runs on v_mfma_f32_16x16x32_bf16 with f32
 * accumulation, and parameters, biases, parameter gradients, LFQ's projections, logits and losses stay f32. */
enum { IDEE_DTYPE_F32 = 0, IDEE_DTYPE_BF16 = 1 };

const char* idee_version(void);
const char* idee_last_error(void);
/* IDEE_OK iff HIP device `dev` is a gfx950. */
int idee_check_device(int dev);

/* Kernel-selection options.  Every GEMM-shaped unit exists in more than one arithmetic form / staging variant; which one a
 * launch uses is looked up in a process-wide table at that launch.  The environment variable of the same name gives an
 * option its INITIAL value only (read once); afterwards the table changes through idee_set_option alone, which is safe
 * against concurrent launches from other host threads.  value == NULL or "" resets an option to its default.
 *   IDEE_SWIN_MATH, IDEE_CONV_MATH : "f32" (default: exact-f32 MFMA, the reference's arithmetic) | "f22" (two f16 pieces per
 *       operand at f32 level, range-guarded) | "fwd22" (f22 in forward passes only) | "split" (bf16 pieces in the backward)
 *   IDEE_CONV16 "regs", IDEE_WGRAD16_REGS "1", IDEE_CONV32 "generic", IDEE_DGRAD_TO16 "wide", IDEE_CONV_BF16 "direct" | "lds" :
 *       alternative staging variants of single kernels (kept for A/B timing; each is covered by the parity suite)
 *   IDEE_CONV_HRUN, IDEE_WGRAD_WIDE_SEG : integer test / tuning knobs
 * Unknown names return IDEE_ERR_INVALID_ARG.  idee_get_option copies the current value ("" = default) into buf. */
int idee_set_option(const char* name, const char* value);
int idee_get_option(const char* name, char* buf, int buflen);

/* ------------------------------------------------------------------------------------------
 * PatchEmbed3D with patch_size (1,1,1) + non-affine LayerNorm over channels.
 * Replaces PatchEmbed3D.forward, models/encoder/Swin_3D.py:473-491 (bias) and
 * models/encoder/CNN_3D.py:44-71 (b == NULL).
 *   x (N, V, Cin, L) [L = T*H*W, the reference input layout], w (V, E, Cin), b (V, E) | NULL
 *   y (N, V, L, E)
 * A patch (pt, ph, pw) != (1,1,1) (en_patch_size; Swin_3D.py:473-491 incl. its zero padding) is the same call on the input
 * gathered per patch by the caller: Cin := Cin pt ph pw values per output token in the weight's own (Cin, pt, ph, pw) order
 * (<= 128), L := the output tokens.
 * bwd: dw (V, E, Cin), db (V, E) | NULL accumulate.  No dx: x is the model input.
 * ---------------------------------------------------------------------------------------- */
/* act_dtype: storage of y / dy (IDEE_DTYPE_*); x is the f32 model input */
int idee_patch_embed_ln_fwd(const float* x, const float* w, const float* b, void* y, int N, int V, int Cin, long L,
                            int E, int act_dtype, idee_stream_t stream);
int idee_patch_embed_ln_bwd(const float* x, const float* w, const float* b, const void* dy, float* dw, float* db, int N,
                            int V, int Cin, long L, int E, int act_dtype, idee_stream_t stream);

/* The same PatchEmbed3D (patch (1,1,1), bias, non-affine LayerNorm) BETWEEN two Swin stages of different width — BasicLayer
 * builds it when in_dim != dim (models/encoder/Swin_3D.py:414-419, forward :473-491).  Input is already channels-last tokens.
 *   x (B, L, Cin), w (V, E, Cin), b (V, E) | NULL, y (B, L, E); item b uses weight set b % V; Cin, E in {16, 32}
 * bwd: dx (B, L, Cin) overwritten; dw (V, E, Cin), db (V, E) | NULL accumulate. */
int idee_stage_embed_ln_fwd(const float* x, const float* w, const float* b, float* y, int B, int V, int Cin, long L, int E,
                            idee_stream_t stream);
int idee_stage_embed_ln_bwd(const float* x, const float* w, const float* b, const float* dy, float* dx, float* dw, float* db,
                            int B, int V, int Cin, long L, int E, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One fused SwinTransformerBlock3D (norm1, pad, cyclic shift, window partition, WindowAttention3D
 * with relative-position bias and shift mask, window reverse, un-shift, crop, residual, norm2,
 * MLP with erf-GELU, residual).  Replaces SwinTransformerBlock3D.forward, Swin_3D.py:224-287,
 * WindowAttention3D.forward :145-178, window_partition/reverse :45-74, compute_mask :340-352,
 * Mlp.forward :36-42.  Backward recomputes the block from x and the saved pre-MLP state h.
 *   x, y, dy, dx: (N, V, D, H, W, C)
 * Window and shift are the values AFTER get_window_size clamping (Swin_3D.py:77-90).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int N, V;       /* batch, variables (weight sets) */
  int D, H, W;    /* token grid */
  int C, heads, hidden;
  int wd, wh, ww; /* window (clamped) */
  int sd, sh, sw; /* shift (clamped) */
  float scale;    /* qk scale: head_dim^-0.5 unless overridden (Swin_3D.py:113) */
  int act_dtype;  /* IDEE_DTYPE_F32 | IDEE_DTYPE_BF16: storage of x, y, h_save, dy, dx, dh_workspace.  BF16: the linear layers,
                     P.V and the token-reduction weight-gradient GEMMs run on v_mfma_f32_16x16x32_bf16 with one-piece bf16
                     operands (f32 accumulate); LayerNorm, softmax, GELU and the attention scores stay f32 */
  /* Element-wise dropout inside the block, training only (rates 0 otherwise): `drop` = Mlp.drop after GELU and after fc2
   * (Swin_3D.py:34-41) and WindowAttention3D.proj_drop (:140, :176); `attn_drop` on the attention probabilities (:138, :172).
   * The keep masks come from a counter-based generator keyed by `drop_seed` (the backward must be given the same seed);
   * built in the exact-f32 kernels (IDEE_UNSUPPORTED under IDEE_SWIN_MATH=split or bf16 storage). */
  float drop, attn_drop;
  uint32_t drop_seed;
} idee_swin_geom;

typedef struct {
  const float *qkv_w, *qkv_b;   /* (V, 3C, C), (V, 3C)       attn.qkv    */
  const float *proj_w, *proj_b; /* (V, C, C), (V, C)         attn.proj   */
  const float *fc1_w, *fc1_b;   /* (V, hidden, C), (V, hidden) mlp.fc1   */
  const float *fc2_w, *fc2_b;   /* (V, C, hidden), (V, C)    mlp.fc2     */
  const float* rel_bias;        /* (V, heads, Nw, Nw) dense bias from idee_relpos_bias_gather */
  const float* path_scale;      /* (N*V, 2) | NULL: DropPath factors mask / keep_prob of the attention and the MLP branch of
                                   every batch item (timm DropPath, Swin_3D.py:16, :280-285); NULL = no stochastic depth */
} idee_swin_params;

typedef struct { /* all accumulate */
  float *qkv_w, *qkv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *rel_bias;
} idee_swin_grads;

/* h_save (same shape as x) | NULL: the block state before the MLP (x + attention), which the backward needs. */
int idee_swin_block_fwd(const void* x, void* y, void* h_save, const idee_swin_params* p, const idee_swin_geom* g,
                        idee_stream_t stream);
/* two launches: MLP + norm2 (h_saved, dy -> dh_workspace), then attention + norm1 (x, dh_workspace -> dx).
 * dh_workspace: same shape as x. */
int idee_swin_block_bwd(const void* x, const void* h_saved, const void* dy, void* dx, void* dh_workspace,
                        const idee_swin_params* p, const idee_swin_grads* d, const idee_swin_geom* g, idee_stream_t stream);

/* relative_position_bias_table[relative_position_index] (Swin_3D.py:158-162):
 * table (V, table_len, heads), index (Nw, Nw) int64 -> dense (V, heads, Nw, Nw);
 * scatter: dtable accumulates. */
int idee_relpos_bias_gather(const float* table, const int64_t* index, float* dense, int V, int table_len, int heads,
                            int Nw, idee_stream_t stream);
int idee_relpos_bias_scatter(const float* ddense, const int64_t* index, float* dtable, int V, int table_len, int heads,
                             int Nw, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PatchMerging (Swin_3D.py:290-335): 2x2 spatial (and, as the reference writes it, d-parity-coupled) gather of 4
 * neighbours -> LayerNorm(4C) -> Linear(4C -> 2C, no bias).  x (B, D, H, W, C) -> y (B, D2, ceil(H/2), ceil(W/2), 2C),
 * D2 = 1 if D == 1 else ceil(D/2); odd extents zero-padded (:312-317).  gamma, beta (4C); w (2C, 4C).
 * bwd: dx must be zero-filled by the caller (positions the gather never reads keep 0); dgamma, dbeta, dw accumulate.
 * ---------------------------------------------------------------------------------------- */
int idee_patch_merging_fwd(const float* x, const float* gamma, const float* beta, const float* w, float* y, int B, int D, int H,
                           int W, int C, idee_stream_t stream);
int idee_patch_merging_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* w, float* dx,
                           float* dgamma, float* dbeta, float* dw, int B, int D, int H, int W, int C, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3-D convolutions as implicit GEMM on the f32 MFMA.
 *   IDEE_CONV_K333_REPLICATE: Conv3d(k=3, s=1, p=1, padding_mode='replicate')
 *       proj_var (Swin_3D.py:586-592, encoder/CNN_3D.py:185-191), conv_block convs (:99-106)
 *   IDEE_CONV_K233_S2_ZERO:   Conv3d((2,3,3), (2,1,1), (0,1,1)) zero padding
 *       classifier heads (classifier/CNN_3D.py:38-40, :83-85)
 *   IDEE_CONV_K133_ZERO:      Conv3d((1,3,3), 1, (0,1,1)) zero padding — the same classifier conv with its two
 *       temporal taps folded into the channels (frame pairs side by side); used on LFQ's paired sign field below
 * The descriptor speaks about the FORWARD conv: x (input, extent T,H,W) -> y (output extent
 * To,H,W with To = T, or (T-2)/2+1 for the stride-2 mode).  Tensors are addressed by float strides so one kernel
 * serves per-variable convs (B = N*V, wsets = V) and the joint head whose 16-channel groups
 * are the V slices of z_q (see x_gpu / x_sU).  Channel group g, channel c: + g*sG + c.
 * Cin must be a multiple of 16; Cout a multiple of 16, or 1 (final class conv).
 * ---------------------------------------------------------------------------------------- */
enum { IDEE_CONV_K333_REPLICATE = 0, IDEE_CONV_K233_S2_ZERO = 1, IDEE_CONV_K133_ZERO = 2 };

typedef struct {
  int mode;
  int B, wsets; /* batch items; weight sets (item b uses set b % wsets) */
  int T, H, W;  /* input extent */
  int Cin, Cout;
  long x_sB, x_sT, x_sH, x_sW, x_sG;
  long y_sB, y_sT, y_sH, y_sW, y_sG;
  int relu; /* fwd: ReLU after the bias */
  /* optional two-level channel-group addressing: group g at (g / gpu) * sU + (g % gpu) * sG.
   * gpu <= 0 means one level (g * sG).  The joint head reads z_q (N,V,T,H,W,C): gpu = C/16, sU = T*H*W*C, sG = 16. */
  int x_gpu, y_gpu;
  long x_sU, y_sU;
  /* IDEE_DTYPE_F32 | IDEE_DTYPE_BF16: storage of x, y, dy, dx, relu_ref, addend (a Cout == 1 output and its gradient stay
   * f32: logits).  BF16: the packed weight buffers hold bf16 (same element count; Cout == 1: f32 as before), bias, dw_packed
   * and dbias stay f32; Cout must be a multiple of 16 or the streaming Cout == 1 case. */
  int act_dtype;
} idee_conv_desc;

/* floats in a packed (fragment-ordered) weight buffer / in a packed weight-gradient buffer (the latter holds several partial
 * copies over which the kernels spread their atomic flushes; zero it before idee_conv_wgrad, idee_conv_unpack_wgrad sums them).
 * A packed WEIGHT buffer of a wide zero-padded convolution (e.g. 96 -> 96) carries, behind its f32 fragments, the same weights as
 * ready f16 piece fragments (hi | lo per K group pair and output block, the LDS image of conv_wide_split_kernel's per-tap slab) and
 * 64 words with each weight set's largest |w|: idee_conv_pack_weights writes all three, the size below covers them. */
long idee_conv_packed_floats(const idee_conv_desc* d);
long idee_conv_wgrad_floats(const idee_conv_desc* d);
/* w: reference layout (wsets, Cout, Cin, kt, kh, kw) -> fragment order for fwd and/or dgrad */
int idee_conv_pack_weights(const idee_conv_desc* d, const float* w, void* packed_fwd, void* packed_dgrad,
                           idee_stream_t stream);
/* bias: (wsets, Cout) | NULL */
int idee_conv_fwd(const idee_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                  idee_stream_t stream);
/* dx = conv^T(dy) [* (relu_ref > 0) if relu_ref != NULL; same layout as x]; accumulate != 0: dx += ...
 * accumulate == 0: dx need not be initialised (REPLICATE mode: the two edge columns, which are accumulated with atomics,
 * are cleared by the call itself); accumulate != 0: dx holds the addend (e.g. a residual gradient). */
int idee_conv_dgrad(const idee_conv_desc* d, const void* dy, const void* w_packed_dgrad, const void* relu_ref,
                    void* dx, int accumulate, idee_stream_t stream);
/* dx = addend + conv^T(dy) [masked like idee_conv_dgrad]: the shortcut gradient of a residual unit added without a copy.
 * addend has dx's layout and may be dx itself (== accumulate) or NULL (== plain). */
int idee_conv_dgrad_add(const idee_conv_desc* d, const void* dy, const void* w_packed_dgrad, const void* relu_ref,
                        const void* addend, void* dx, idee_stream_t stream);
/* dw_packed (wsets, taps, Cin, CoutPad) and dbias (wsets, CoutPad) | NULL accumulate */
int idee_conv_wgrad(const idee_conv_desc* d, const void* x, const void* dy, float* dw_packed, float* dbias,
                    idee_stream_t stream);
/* packed weight gradient -> reference layout (wsets, Cout, Cin, kt, kh, kw) (overwrites) */
int idee_conv_unpack_wgrad(const idee_conv_desc* d, const float* dw_packed, float* dw, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Tail of the CNN_3D encoder's conv_block unit: y = shortcut + ReLU(LayerNorm_affine(c)), c = conv output.
 * Replaces conv_block.forward, models/encoder/CNN_3D.py:131-142 (the permute/view pairs disappear).
 *   c, shortcut, y, dy, dc: (B, L, E) channels-last; gamma, beta: (V, E); item b uses set b % V
 * bwd: dc overwritten; dgamma, dbeta accumulate; d(shortcut) = dy is the caller's.
 * ---------------------------------------------------------------------------------------- */
int idee_ln_relu_res_fwd(const float* c, const float* gamma, const float* beta, const float* shortcut, float* y, int B,
                         int V, long L, int E, const float* item_scale, idee_stream_t stream);
int idee_ln_relu_res_bwd(const float* c, const float* gamma, const float* beta, const float* dy, float* dc, float* dgamma,
                         float* dbeta, int B, int V, long L, int E, const float* item_scale, idee_stream_t stream);
/* item_scale (B) | NULL: DropPath factor mask / keep_prob of each batch item on the ReLU(LN(c)) branch (CNN_3D.py:97,140,142) */

/* nn.Dropout after the first ReLU of the classifier heads (classifier/CNN_3D.py:41,53): x <- x * keep-mask IN PLACE (the caller
 * folds 1 / (1 - p) into the next convolution's weights); element i keeps iff hash(seed, i) >= p * 2^32.  n % 4 == 0. */
int idee_dropout_mask_(void* x, long n, float p, uint64_t seed, int act_dtype, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Lookup-free quantisation with projections (dim -> num_codebooks * log2 K -> dim), straight-through gradient,
 * entropy + commitment auxiliary loss.  Replaces LFQ.forward, models/codebook/LFQ.py:183-307 (num_codebooks: :92-101, :215).
 *   z, zq, dz, dzq: (M, E) tokens; indices int64 (M): the num_codebooks per-codebook indices of a token packed most significant
 *   codebook first, log2 K bits each (num_codebooks = 1: the reference's index)
 *   w_in (D, E), b_in (D), w_out (E, D), b_out (E),  D = num_codebooks * log2(codebook_size) code bits per token
 *   stats (4 + num_codebooks * K floats): [aux loss, per-sample entropy (mean over tokens and codebooks), codebook entropy (mean over
 *   codebooks), commit, avg prob[codebook][K]]
 *   workspace: idee_lfq_workspace_floats(M, K, num_codebooks) floats
 * bwd: gloss = device pointer to d(loss)/d(aux loss); dw_in, db_in, dw_out, db_out accumulate.
 * ---------------------------------------------------------------------------------------- */
long idee_lfq_workspace_floats(long M, int codebook_size, int num_codebooks);
/* signs (M, D) | NULL: the +-1 code bits per token; dsigns (M, D) | NULL: an additional gradient on them (straight-through),
 * coming from consumers that read the sign field instead of z_q (the rank-d classifier input and anomaly loss below).
 * dzq | NULL: NULL when z_q has no dense consumer (every gradient arrives through dsigns): the 64-byte-per-token read is skipped. */
/* act_dtype: storage of the token tensors z, zq, dz, dzq (IDEE_DTYPE_*); the projections, the code bits, the statistics and
 * every parameter gradient are f32 in either case ("fp32 LFQ") */
int idee_lfq_fwd(const void* z, const float* w_in, const float* b_in, const float* w_out, const float* b_out, void* zq,
                 int64_t* indices, float* stats, float* workspace, float* signs, long M, int E, int codebook_size,
                 int num_codebooks, int training, float w_entropy, float gamma, float w_commit, int act_dtype, idee_stream_t stream);
int idee_lfq_bwd(const void* z, const void* dzq, const float* w_in, const float* b_in, const float* w_out,
                 const float* stats, const float* gloss, const float* dsigns, void* dz, float* dw_in, float* db_in,
                 float* dw_out, float* db_out, long M, int E, int codebook_size, int num_codebooks, float w_entropy, float gamma,
                 float w_commit, int act_dtype, idee_stream_t stream);

/* LFQ WITHOUT projections (LFQ.py:98-101: dim == num_codebooks * log2(codebook_size), project_in / project_out = nn.Identity):
 * the token's E channels are its code bits, log2(codebook_size) in {1, 2, 4} consecutive channels per codebook.
 *   z, zq, dz, dzq, dsigns: (M, E); indices int64 (M): the E bits, channel 0 most significant (per-codebook indices = its bit fields)
 *   stats (4 + (E / log2 K) * K floats) as idee_lfq_fwd's; workspace idee_lfq_noproj_workspace_floats(M, E, K) floats */
long idee_lfq_noproj_workspace_floats(long M, int E, int codebook_size);
int idee_lfq_noproj_fwd(const void* z, void* zq, int64_t* indices, float* stats, float* workspace, long M, int E, int codebook_size,
                        int training, float w_entropy, float gamma, float w_commit, int act_dtype, idee_stream_t stream);
int idee_lfq_noproj_bwd(const void* z, const void* dzq, const float* stats, const float* gloss, const float* dsigns, void* dz, long M,
                        int E, int codebook_size, float w_entropy, float gamma, float w_commit, int act_dtype, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Rank-d classifier input (algebraic refactoring; exact in real arithmetic, disclosed in DESIGN.md):
 * z_q = b_out + sum_i q_i w_out[:, i], so the first (2,3,3)-stride-2 convolutions of the classifier
 * (classifier/CNN_3D.py:83,38 applied to z_q at build.py:157) equal IDEE_CONV_K133_ZERO convolutions on the paired sign
 * field S (N,T/2,H,W,CP): with R = V*D+1, channel kt*R + v*D+i = q_i of variable v in frame 2*t1+kt, channel kt*R + V*D = 1,
 * others 0; CP a multiple of 16 >= 2R — with weights folded through [w_out | b_out].
 *   sign_field: signs (N,V,T,HW,D) <-> S (N,T/2,HW,CP)
 *   fold: w (Co, V*C, 2,3,3) joint | (V, Co, C, 2,3,3) per_var -> weff (rows, CP, 1,3,3), rows = Co | V*Co;
 *         bwd: dw overwritten, dw_out (C,D) and db_out (C) accumulate.
 * ---------------------------------------------------------------------------------------- */
/* act_dtype: storage of the sign field sp / its gradient dsp (+-1, 0 and 1 are exact in bf16) */
int idee_lfq_sign_field_fwd(const float* signs, void* sp, int N, int V, int T, long HW, int D, int CP, int act_dtype, idee_stream_t stream);
int idee_lfq_sign_field_bwd(const void* dsp, float* dsigns, int N, int V, int T, long HW, int D, int CP, int act_dtype, idee_stream_t stream);
int idee_lfq_fold_weights_fwd(const float* w, const float* w_out, const float* b_out, float* weff, int Co, int V, int C, int D,
                              int CP, int per_var, idee_stream_t stream);
int idee_lfq_fold_weights_bwd(const float* dweff, const float* w, const float* w_out, const float* b_out, float* dw,
                              float* dw_out, float* db_out, int Co, int V, int C, int D, int CP, int per_var,
                              idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Euclidean nearest-neighbour vector quantisation with straight-through gradient and commitment loss — the
 * reference VQ's default path: cdist + argmax (models/codebook/VQ.py:44-48, :477-555), STE (:986), commit mse
 * (:1056).  (EMA / k-means / code expiry / gumbel / multi-head are outside the path: build.py:87 wires LFQ.)
 *   x, q, dq, dx: (M, E); embed, dembed: (K, E) with K*E <= 8192; indices int64 (M)
 *   loss (2): [sum (q-x)^2 over all elements (the caller scales by w_commit / (M*E)), block count];
 *   workspace idee_vq_workspace_floats(M).   bwd: dx overwritten; dembed accumulates; gloss: d/d(loss value).
 * ---------------------------------------------------------------------------------------- */
long idee_vq_workspace_floats(long M);
int idee_vq_nn_fwd(const float* x, const float* embed, float* q, int64_t* indices, float* loss, float* workspace, long M,
                   int E, int K, int training, float w_commit, idee_stream_t stream);
int idee_vq_nn_bwd(const float* x, const float* embed, const int64_t* indices, const float* dq, const float* gloss, float* dx,
                   float* dembed, long M, int E, int K, float w_commit, idee_stream_t stream);

/* EuclideanCodebook's codebook maintenance (models/codebook/VQ.py): EMA update :519-545, k-means init :213-253 + :354-377,
 * dead-code expiry :449-475.  All buffers f32; x (M, E) tokens, indices int64 (M) from idee_vq_nn_fwd.
 *   cluster_stats : counts (K), sums (K, E) ACCUMULATE #{idx = k} and sum_{idx = k} x  (the caller zeroes them, and
 *                   all-reduces them across data-parallel ranks where the reference calls all_reduce_fn)
 *   ema_update    : cluster_size <- lerp(cluster_size, new_size, 1 - decay); embed_avg likewise; embed <- embed_avg /
 *                   (laplace_smoothing(cluster_size, K, eps) * sum(cluster_size))
 *   kmeans_update : means[k] <- sums[k] / bins[k] where bins[k] != 0
 *   expire_codes  : the r-th code with cluster_size < threshold <- x[candidates[r]]; cluster_size <- reset, embed_avg <- sample *
 *                   reset; n_expired (device int32 | NULL) receives the count.  candidates: int64 (>= K entries) sample rows. */
int idee_vq_cluster_stats(const float* x, const int64_t* indices, float* counts, float* sums, long M, int E, int K,
                          idee_stream_t stream);
int idee_vq_ema_update(float* cluster_size, float* embed_avg, float* embed, const float* new_size, const float* new_sum, int K,
                       int E, float decay, float eps, idee_stream_t stream);
int idee_vq_kmeans_update(float* means, const float* bins, const float* sums, int K, int E, idee_stream_t stream);
int idee_vq_expire_codes(float* embed, float* cluster_size, float* embed_avg, const float* x, const int64_t* candidates, int K,
                         int E, float threshold, float reset, int32_t* n_expired, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Losses (models/losses.py).
 * BCE with logits weighted by log(1.1 + class_freq^-0.5), class counts from torch.histc(bins=2)
 * semantics: BCE_loss_synthetic :98-124 (valid == NULL) and BCE_loss :64-95 (valid mask).
 *   counts (2) <- idee_bce_class_counts(target (M), valid (M) | NULL)
 *   pred (P, M): P predictions sharing one target; loss (P + 1): P means then the denominator
 * ---------------------------------------------------------------------------------------- */
int idee_bce_class_counts(const float* target, const float* valid, long M, float* counts, idee_stream_t stream);
long idee_bce_workspace_floats(int P, long M);
int idee_bce_weighted_fwd(const float* pred, const float* target, const float* valid, const float* counts, float* loss,
                          float* workspace, int P, long M, idee_stream_t stream);
/* gout (P): upstream gradients; denom (1): loss[P] from the forward */
int idee_bce_weighted_bwd(const float* pred, const float* target, const float* valid, const float* counts,
                          const float* gout, const float* denom, float* dpred, int P, long M, idee_stream_t stream);

/* Masked L1 pulling z_q to the "normal" code outside extremes: Anomaly_L1_loss_synthetic
 * :127-168 (mask_valid == NULL) and Anomaly_L1_loss :15-61.
 *   zq (N, V, T, HW, E); mask_extreme (N, HW); mask_valid (N, T, HW) | NULL; vq0 (E)
 *   out (2): [loss, denominator]; workspace idee_anomaly_workspace_floats(N*V*T*HW) */
long idee_anomaly_workspace_floats(long tokens);
int idee_anomaly_l1_fwd(const float* zq, const float* mask_extreme, const float* mask_valid, const float* vq0,
                        float* out, float* workspace, int N, int V, int T, long HW, int E, idee_stream_t stream);
int idee_anomaly_l1_bwd(const float* zq, const float* mask_extreme, const float* mask_valid, const float* vq0,
                        const float* gout, const float* denom, float* dzq, int N, int V, int T, long HW, int E,
                        idee_stream_t stream);
/* The same loss read from LFQ's code bits instead of z_q (z_q = b_out + sum_i signs_i w_out[:, i] has rank log2 K per token:
 * the rank-d form of DESIGN.md 3b; values and gradients equal those through z_q up to f32 summation order).
 *   signs (N, V, T, HW, D) as idee_lfq_fwd writes them, D = log2(codebook_size); w_out (E, D), b_out (E): LFQ.project_out;
 *   vq0 (E): the code-0 vector (idee_lfq_code_vectors), a constant of the loss (losses.py detaches it)
 *   bwd: dsigns (N, V, T, HW, D) is the straight-through gradient idee_lfq_bwd takes as `dsigns`; dw_out (E, D) and db_out (E)
 *   are ONE buffer of E (D + 1) floats (db_out = dw_out + E D), overwritten; workspace
 *   idee_anomaly_signs_workspace_floats(N*V*T*HW, E, codebook_size) for either call */
long idee_anomaly_signs_workspace_floats(long tokens, int E, int codebook_size);
int idee_anomaly_l1_signs_fwd(const float* signs, const float* mask_extreme, const float* mask_valid, const float* w_out,
                              const float* b_out, const float* vq0, float* out, float* workspace, int N, int V, int T, long HW,
                              int E, int codebook_size, idee_stream_t stream);
int idee_anomaly_l1_signs_bwd(const float* signs, const float* mask_extreme, const float* mask_valid, const float* w_out,
                              const float* b_out, const float* vq0, const float* gout, const float* denom, float* dsigns,
                              float* dw_out, float* db_out, float* workspace, int N, int V, int T, long HW, int E,
                              int codebook_size, idee_stream_t stream);

/* Step glue of the training drivers.
 *   lfq_code_vectors: LFQ.indices_to_codes(indices) with project_out (models/codebook/LFQ.py:152-181; e.g. the code-0 target of
 *       the anomaly loss, train_synthetic.py:193-194): indices int64 (n) -> out (n, E); w_out (E, log2 K), b_out (E)
 *   loss_combine: total = mean_t bce[0,t] + lambda_anomaly * l_anom + sum_{p>=1} mean_t bce[p,t] + loss_z_q
 *       (train_synthetic.py:182-201, train_ERA5_Land.py:181-198); bce (P, T3) from idee_bce_weighted_fwd with the joint
 *       prediction first; out5 = [total, extreme, anomaly, var, z_q]; bwd: gtotal (1) -> dbce (P, T3), d_anom (1), d_zq (1) */
int idee_lfq_code_vectors(const int64_t* indices, const float* w_out, const float* b_out, float* out, int n, int E,
                          int codebook_size, idee_stream_t stream);
int idee_loss_combine_fwd(const float* bce, int P, int T3, const float* l_anom, float lambda_anomaly, const float* loss_zq,
                          float* out5, idee_stream_t stream);
int idee_loss_combine_bwd(const float* gtotal, int P, int T3, float lambda_anomaly, float* dbce, float* d_anom, float* d_zq,
                          idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * torch.optim.Adam step (train_synthetic.py:127-129,205) over one flat parameter buffer.
 * g is multiplied by grad_scale first (1/world for data-parallel averaging).
 * ---------------------------------------------------------------------------------------- */
int idee_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, idee_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Device-side bookkeeping of the training drivers: replaces the per-iteration
 * `.cpu().numpy()` + numpy confusion counts of train_synthetic.py:207-215 / utils/utils_train.py:175-348 and the
 * majority-vote anomaly collector utils_train.py:529-554.
 *   counts (9 x uint64, accumulate): [label*4 + {correct, seen, union, predicted}] for labels 0,1; [8] = valid pixels
 *   prediction = sigmoid(logit) > threshold (0.5 synthetic, 0.35 real data); valid (M) | NULL
 *   vote: anomaly int64 (n, V, dt, HW); idx_end int32 (n) = position of each clip's last step in the timeline;
 *   acc, cnt (V, TT, HW) accumulate the time-flipped votes; finalize: out = acc/cnt >= 0.5 (NaN where never seen)
 * ---------------------------------------------------------------------------------------- */
int idee_confusion_counts(const float* logits, const float* gt, const float* valid, long M, float threshold,
                          uint64_t* counts, idee_stream_t stream);
int idee_anomaly_vote_accumulate(const int64_t* anomaly, const int32_t* idx_end, float* acc, float* cnt, int n_batch, int V,
                                 int dt, long HW, int TT, idee_stream_t stream);
int idee_anomaly_vote_finalize(const float* acc, const float* cnt, float* out, long n, idee_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IDEE_HIP_H */
