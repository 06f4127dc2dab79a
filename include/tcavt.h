/*
 * tcavt.h -- C ABI of the MI355X (gfx950) hot path for the multimodal-LLM
 * vehicle-trajectory-prediction forward/backward pass.
 *
 * The reference (imjaegyun/Traffic-Context-Augmented-Vehicle-Trajectory-
 * Prediction-Framework-Using-Multimodal-LLM) has no FFI of its own: its only
 * shared interface is the Python class surface of scripts/train.py:352-964.
 * Each entry point below names the reference code (file:line) whose arithmetic
 * it replaces.  The Python host side (package `tcavt_amd`, module model.py)
 * mirrors the reference classes and calls these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()) unless marked host
 *   - no allocation, no synchronisation, no host<->device copy inside a call:
 *     the caller owns all buffers incl. workspaces; calls are stream-ordered
 *     on `stream` (a hipStream_t passed as void*) and graph-capturable
 *   - return value 0 = ok; otherwise a TCAVT_ERR_* code, and
 *     tcavt_last_error() returns a thread-local message
 *   - 16-bit buffers are raw uint16 storage of either IEEE half (TCAVT_F16: the forward path's default storage
 *     type -- 11 significant bits keep the whole model within 1e-3 of the fp32 reference, profiles/
 *     r02_error_budget_full_size.json) or brain-float (TCAVT_BF16: upper half of an IEEE f32; gradient-side tensors
 *     and the LoRA-trainable variant).  Parameters still called "..._bf16" take either; the entry point's `dtype16`
 *     / `in_dtype` / `out_dtype` argument says which.  fp32 accumulation everywhere.
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS
 */
#ifndef TCAVT_H
#define TCAVT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tcavt_stream_t; /* hipStream_t */

#define TCAVT_ABI_VERSION 4

#define TCAVT_OK 0
#define TCAVT_ERR_ARG 1  /* shape / alignment / null-pointer contract violated */
#define TCAVT_ERR_HIP 2  /* a HIP runtime call failed */
#define TCAVT_ERR_ARCH 3 /* device is not gfx950 */

#define TCAVT_F32 0
#define TCAVT_BF16 1
#define TCAVT_F16 2

int tcavt_abi_version(void);
const char* tcavt_last_error(void);
/* Selects `device`, verifies it is gfx950, reports its CU count. */
int tcavt_init(int device, int* num_cus);

/* ------------------------------------------------------------------------
 * Dense contraction  C[M,N] = A[M,K] . W[N,K]^T (+ A2[M,K2] . W2[N,K2]^T)
 * bf16 operands, fp32 MFMA accumulation, fused epilogue.
 * Replaces every nn.Linear on the bf16 part of the path:
 *   - Llama q/k/v/o/gate/up/down projections (transformers modeling_llama.py
 *     :174-176,254-256,279-280 as called from scripts/train.py:446-452)
 *   - the LoRA update  y += (alpha/r) * B(A(x))  on q_proj/v_proj
 *     (scripts/train.py:433-440) via the second K-source (A2 = 4*x.A^T, W2 = B)
 *   - BlipQFormer linears (scripts/train.py:401-406), LlamaMultiModal.q_proj
 *     (:493,521), cross_attn in/out projections and dec_proj/dec_unproj
 *     (:754-757,794-799)
 * Epilogue flags (combinable where it makes sense):
 *   (first)   acc *= acc_scale  (LoRA alpha/r on the down-projection)
 *   BIAS      acc += bias[n]
 *   RELU      acc = max(acc, 0)
 *   RESIDUAL  acc += residual[m][n]            (fp32, may alias C when C is fp32)
 *   SILU_MUL  W holds gate/up rows interleaved in blocks of 16
 *             (rows 32j..32j+15 = gate features 16j.., rows 32j+16..32j+31 = up
 *             features 16j..); C[m][f] = silu(gate_f) * up_f, C has N/2 columns
 *   ROPE      rotary embedding (half-split convention, head_dim 64) applied to
 *             columns [0, rope_cols) with position = m % rope_L, cos/sin tables
 *             [rope_L][32] fp32 (modeling_llama.py:113-160)
 * out_dtype may be TCAVT_F32, TCAVT_BF16 or TCAVT_F16.
 * Constraints: K % 64 == 0, K2 % 64 == 0, N % 16 == 0, ld* % 8 == 0,
 *              16-byte aligned base pointers; SILU_MUL/ROPE need N % 128 == 0.
 * ---------------------------------------------------------------------- */
#define TCAVT_EPI_BIAS 1
#define TCAVT_EPI_RELU 2
#define TCAVT_EPI_RESIDUAL 4
#define TCAVT_EPI_SILU_MUL 8
#define TCAVT_EPI_ROPE 16
#define TCAVT_EPI_BIAS_ROW 32 /* acc += bias[m] (bias per output ROW; used when the roles of
                                 activations and weights are swapped to emit a transposed result) */
#define TCAVT_EPI_ACCUM 64    /* tcavt_gemm_f32[_strided] only: C already holds a valid partial result (zeros or an
                                 earlier gradient contribution); the product is ADDED to it and no memset is issued */
/* Llama RMSNorm (modeling_llama.py:62-67) fused into the GEMMs on either side of it -- "fused RMSNorm+RoPE+QKV-proj":
 *   y = (x * rsqrt(mean(x^2) + eps) * gamma) . W^T  ==  rsqrt(mean(x^2) + eps) * ( x . (W * gamma)^T )
 * NORM_OUT  (producer: o_proj / down_proj; fp32 output, optionally + RESIDUAL): besides C the epilogue writes the
 *           16-bit copy of C's rows to `norm_h16` (leading dimension ldc; it is the next projection's A operand) and,
 *           for every row and 64-column group, the sum of squares of the fp32 values to norm_part[M][N / 64].
 *           C == NULL (and residual == NULL) selects the 16-bit residual stream: norm_h16 itself is the stream, with
 *           RESIDUAL it is read, added to the accumulator and rewritten in place (leading dimension ldc), and the
 *           partial sums are those of the ROUNDED values -- 4 bytes per element through HBM instead of 10.
 * ROWSCALE  (consumer: ROPE or SILU_MUL epilogue; gamma is folded into W by the caller): the accumulator row m is
 *           multiplied by rsqrt(sum_i rowscale_part[m][i] / rowscale_h + rowscale_eps) first; the partials are added in
 *           index order (no atomics: bit-reproducible).  rowscale_npart % 4 == 0. */
#define TCAVT_EPI_NORM_OUT 128
#define TCAVT_EPI_ROWSCALE 256
/* SILU_BWD (dgrad of down_proj in the LoRA-trainable variant's backward; the 4-wave kernel only: M, N % 256 == 0, 16-bit
 *           output of the operand type, ldc and ld_preact % 8 == 0, no other flag): the accumulator is d = dL/d(silu(gate)*up)
 *           [M][N]; with the forward's gate|up pre-activations `silu_preact` [M][2N] (interleaved layout of SILU_MUL) the
 *           epilogue writes dL/dgate | dL/dup to C [M][2N] in the same layout -- tcavt_silu_mul_bwd without the round trip
 *           of d through memory.  C may be `silu_preact` itself (in place: a lane reads its 16 bytes before it writes them). */
#define TCAVT_EPI_SILU_BWD 512

typedef struct tcavt_gemm_args {
  const void* A;   int64_t lda;  /* bf16 [M][K]  */
  const void* W;   int64_t ldw;  /* bf16 [N][K]  */
  const void* A2;  int64_t lda2; /* bf16 [M][K2] or NULL */
  const void* W2;  int64_t ldw2; /* bf16 [N][K2] or NULL */
  void* C;         int64_t ldc;  /* out_dtype [M][N] ([M][N/2] for SILU_MUL) */
  const float* bias;             /* fp32 [N] or NULL */
  const float* residual; int64_t ldr; /* fp32 [M][N] or NULL */
  const float* rope_cos;         /* fp32 [rope_L][32] */
  const float* rope_sin;
  int32_t M, N, K, K2;
  int32_t out_dtype;             /* TCAVT_F32 | TCAVT_BF16 */
  int32_t epilogue;              /* TCAVT_EPI_* flags */
  int32_t rope_L, rope_cols;
  int32_t tile;                  /* 0 = auto (recommended).  Forcing a kernel form (all forms give bit-identical results):
                                    64 / 128 = small-launch kernels (4-stage pipeline), 256 = 8-wave 256x256,
                                    257 = 4-wave 256x256 (whole tiles only), 271 = 4-wave 256x192 (N % 192 == 0),
                                    272 = 4-wave two-barrier deep-prefetch form (long K).  The 4-wave forms write their SILU_MUL / ROPE /
                                    in-place NORM_OUT results with 16-byte accesses: 16-bit output of the operand type and
                                    ldc % 8 == 0 (refused otherwise; auto picks the 8-wave form).  Any other code is refused by
                                    the product library (measured-and-rejected variants and timing experiments live in
                                    the -DTCAVT_EXPERIMENTS build that tools/ makes for itself) */
  float acc_scale;               /* accumulator is multiplied by this first; 0 means 1 */
  int32_t in_dtype;              /* operand type of A/W/A2/W2: 0 or TCAVT_BF16, or TCAVT_F16 (every epilogue and kernel form) */
  /* Batched form (generic epilogue only): batch > 1 runs `batch` independent products; product i uses
   * A + (i / batch_inner) * sAo + (i % batch_inner) * sAi, likewise W and C (strides in ELEMENTS,
   * every offset must keep 16-byte alignment).  bias / residual are shared by all products.
   * This is how the per-(sample, head) products of the head_dim-1024 cross-attention
   * (scripts/train.py:795-798) are issued: scores = q_bh . k_bh^T and out = p_bh . (v^T_bh)^T. */
  int32_t batch, batch_inner;
  int64_t sAo, sAi, sWo, sWi, sCo, sCi;
  /* Train-mode dropout fused into the generic epilogue, applied after bias / ReLU and before the residual
   * (nn.Dropout placement of nn.Transformer*Layer and the LTSF blocks, scripts/train.py:358,402,405,664-671,749):
   * element (m, n) is kept iff the 16-bit Philox4x32-10 draw of flat index e = m*N+n at (seed, site) -- half-word e & 7 of
   * the call with counter e >> 3 -- is >= ceil(p * 65536), and
   * scaled by 1/(1-p).  dropout_p == 0 disables it.  See csrc/philox.hpp. */
  float dropout_p;
  uint32_t dropout_site;
  uint64_t dropout_seed;
  /* Batched form, grouped operands: when > 1, W uses (i % batch_inner) / batch_w_group in place of i % batch_inner --
   * `batch_w_group` consecutive products share one W (the query heads of a grouped-query attention group share a key /
   * value head).  0 or 1: off. */
  int32_t batch_w_group;
  /* Layout of W.  0: row-major [N][ldw].  TCAVT_W_FRAG16 (1): the fragment-major copy tcavt_pack_weight16 makes -- skinny form
   * only (M <= 32, tile 0: the decode step; refused elsewhere), ldw == K.  Same arithmetic in the same order: bit-identical
   * results; what changes is that one wave instruction reads 1 KiB of consecutive bytes instead of 16 rows x 64 bytes that are
   * K * 2 bytes apart (decode step 1.09 -> 0.91 ms at B = 8: the weight stream was bound by its access pattern, not by HBM) */
  int32_t w_layout;
  /* TCAVT_EPI_SILU_MUL only, optional: a bf16 copy of the gate|up pre-activations [M, N] (interleaved layout, leading
   * dimension ld_preact) next to the activated output -- what the backward of silu(gate)*up needs (tcavt_silu_mul_bwd) */
  void* silu_preact;
  int64_t ld_preact;
  /* TCAVT_EPI_NORM_OUT / TCAVT_EPI_ROWSCALE (see the flag definitions) */
  void* norm_h16;
  float* norm_part;
  const float* rowscale_part;
  int32_t rowscale_npart;
  int32_t rowscale_h;
  float rowscale_eps;
  /* TCAVT_EPI_NORM_OUT, optional (0 means 1): SCALED 16-bit image of the residual stream.  Every 16-bit image of the stream
   * (the stream itself when C == NULL, its copy norm_h16 otherwise) holds norm_scale * x, and norm_part the sums of squares
   * of those scaled values:  C == NULL: h16 <- round(norm_scale * acc + h16);  C != NULL: C = acc + residual (unscaled fp32),
   * norm_h16 = round(norm_scale * C).  A power of two <= 1 costs no precision (fp16 is a floating-point format) and moves the
   * overflow limit of the image from 65504 to 65504 / norm_scale -- what real checkpoints' outlier channels need.  RMSNorm is
   * scale-invariant up to its eps: the consumers (TCAVT_EPI_ROWSCALE, tcavt_rmsnorm16) are given rowscale_eps = eps *
   * norm_scale^2 and need nothing else.  (tcavt_llama_stack_args.stream_scale does all of this for a decoder pass.) */
  float norm_scale;
  /* TCAVT_EPI_ROPE, optional: int32 [M] device array, the position of row m (a decode step has one row per sample, each
   * at its own position; cos / sin tables then hold rope_L >= max position + 1 rows).  NULL: position = m % rope_L */
  const int32_t* rope_pos;
  /* TCAVT_EPI_NORM_OUT, optional: device int32 word that receives `nonfinite_tag` (compare-and-swap from 0: the first
   * launch that sees one wins) when a partial sum of squares of the rows it writes is not finite, or -- fp16 16-bit copy of
   * an fp32 stream -- when a rounded element is +-inf.  This is how an fp16 overflow anywhere upstream (the stream itself,
   * `act`, q|k|v, attention output) becomes observable: it reaches the next residual epilogue as inf / NaN.  Only ever
   * SET by the kernels; the host reads and clears it (LlamaMultiModal.check_flags).  NULL: no check. */
  int32_t* nonfinite_flag;
  int32_t nonfinite_tag;
  /* Skinny form only (M <= 32, tile 0: the decode step; refused elsewhere): 16-bit operands / results in FRAGMENT-MAJOR order, so
   * that the activation fragments of a k-step are 1 KiB of consecutive bytes as well (w_layout does it for the weights).  Element
   * (m, f) of a [<= 32][K] operand lives at  (m >> 4) * 16 K + (f >> 5) * 512 + ((f >> 3) & 3) * 128 + (m & 15) * 8 + (f & 7)
   * elements; the buffer holds 16 (M <= 16) or 32 whole rows.  Flags:
   *   TCAVT_ACT_A_FRAG16   (1)  A is in this order (lda ignored; K % 32 == 0)
   *   TCAVT_ACT_OUT_FRAG16 (2)  the 16-bit result is: C of TCAVT_EPI_SILU_MUL (row length N / 2), or the in-place 16-bit stream
   *                             norm_h16 / norm_res16 of TCAVT_EPI_NORM_OUT with C == NULL (row length N)
   *   TCAVT_ACT_BLOCK8     (4)  with either: M <= 8, ONE block of 8 tokens -- element (m, f) at (f >> 5) * 256 + ((f >> 3) & 3) * 64
   *                             + m * 8 + (f & 7), the buffer holds 8 whole rows (a k-step is 512 consecutive bytes: at M <= 8 a
   *                             16-token fragment would be half padding)
   * Same values into the same MFMAs: results are bit-identical to the row-major call. */
  int32_t act_layout;
  /* TCAVT_EPI_NORM_OUT with C == NULL (16-bit residual stream), optional: the 16-bit residual is READ from here
   * (leading dimension ldc) and the updated stream written to norm_h16 -- out of place, so that a caller can keep the
   * stream of every layer (the LoRA-trainable variant's tape).  NULL: read from norm_h16 (in place). */
  const void* norm_res16;
  /* Optional device workspace for splitting K.  Layout: 4096 int32 tickets, ZEROED ONCE by the caller (every launch leaves them
   * zero), then fp32 slabs; >= 64 KiB.  Launches that share it must be stream-ordered.
   *  - skinny form (M <= 32: the decode step): several workgroups per block of output columns, partial sums in the slabs,
   *    combined by the last arriver in slice order (bit-reproducible); 8.2 MiB covers every shape of the Llama-3.2-1B decode step;
   *  - the in-place 16-bit residual form (TCAVT_EPI_NORM_OUT with C == NULL, tile 0) on a grid of fewer than ~256 128 x 128
   *    tiles (M > 32, e.g. B * L = 1024): TWO launches -- S <= 8 partial products over K / S as a batched launch into S slabs
   *    [M][N] (needs 16 KiB + S * M * N * 4 bytes, else it is not used), then one kernel that adds the slabs in slice order and
   *    runs the residual epilogue (bit-reproducible; sums in another order than the one-launch form).
   * NULL: no split. */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  /* Skinny form only (decode step), optional: the NEXT layer's LoRA down-projection t = scale * h16 . a_cat^T without a launch
   * of its own (r <= 8: adapter rows 0..7 = A_q and 16..23 = A_v of a_cat [>= 24, lda]).
   *   producer = a residual GEMM (TCAVT_EPI_NORM_OUT) with lora_part + lora_part_a: workgroup b (16 output columns) also writes
   *     lora_part[b][m][16] (fp32) = the dot products of ITS columns of the rounded 16-bit stream with the 16 adapter rows;
   *   consumer = the q|k|v GEMM (TCAVT_EPI_ROPE) with lora_part + lora_part_np (= producer's N / 16) + W2 (= b_ext [N, ldw2 >= 32])
   *     and A2 == NULL, K2 == 0: t[m][j] = round16(lora_part_scale * sum_b lora_part[b][m][j]), partials added in index
   *     order (bit-reproducible), then used as the second K source (32 deep).
   * lora_part: M * 16 * (N / 16) floats, 16-byte aligned. */
  void* lora_part;
  const void* lora_part_a;
  int64_t lora_part_lda;
  int32_t lora_part_np;
  float lora_part_scale;
} tcavt_gemm_args;

int tcavt_gemm_bf16(const tcavt_gemm_args* args, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Llama RMSNorm: out = bf16( x * rsqrt(mean(x^2) + eps) * gamma ), x fp32 [M][H]
 * (modeling_llama.py:62-67; F1/F7 of SURVEY.md section 8a).
 * out_f32 (optional) receives the un-rounded fp32 result (final norm ->
 * hidden_states[-1], scripts/train.py:553).  H % 8 == 0, H <= 8192.
 * out_drop_bf16 (optional, train mode) additionally receives dropout(out_bf16) with the Philox mask of
 * (dropout_seed, dropout_site, element m * H + n) -- the LoRA branch input (PEFT lora_dropout on x,
 * scripts/train.py:433-440); identical to tcavt_dropout applied to out_bf16.
 * ---------------------------------------------------------------------- */
int tcavt_rmsnorm(const float* x, const float* gamma, float eps, void* out_bf16,
                  float* out_f32, int M, int H, void* out_drop_bf16, float dropout_p,
                  uint64_t dropout_seed, uint32_t dropout_site, int dtype16 /* of out_bf16 / out_drop_bf16 */,
                  tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * nn.LayerNorm over the last dim with optional fused residual add:
 *   y = LN(x + residual) * gamma + beta      (post-LN blocks of
 * nn.TransformerEncoderLayer / DecoderLayer used at scripts/train.py:358,402,405
 * and LayerNorms at :662,672,760).  Writes fp32 and/or bf16 copies.
 * D % 4 == 0, D <= 4096.
 * ---------------------------------------------------------------------- */
int tcavt_layernorm(const float* x, const float* residual, const float* gamma,
                    const float* beta, float eps, float* out_f32, void* out_bf16,
                    int M, int D, int dtype16 /* of out_bf16 */, tcavt_stream_t stream);

/* Elementwise train-mode dropout: out[i] = x[i] * keep(i) / (1-p), dtype TCAVT_F32, TCAVT_BF16 or TCAVT_F16 (in place
 * allowed).  Used for the LoRA branch input (lora_dropout, scripts/train.py:433-439). */
int tcavt_dropout(const void* x, void* out, int64_t n, int dtype, float p, uint64_t seed, uint32_t site,
                  const void* add /* optional, same dtype: out[i] = dropout(x)[i] + add[i] (two masked gradient addends) */,
                  tcavt_stream_t stream);

/* Dropout masks under hipGraph replay.  Every dropout site takes (p, seed, site) as launch arguments, which a captured
 * graph bakes in.  tcavt_set_dropout_epoch(ptr) registers a device-resident uint64 (process-wide; NULL = off, the default)
 * whose value every kernel launched afterwards ADDS to its seed when it runs; tcavt_dropout_epoch_advance (one tiny launch,
 * put it at the head of the captured step) increments it.  Each replay of a captured train-mode step then draws fresh
 * masks, while forward and backward of one step still regenerate the same ones. */
int tcavt_set_dropout_epoch(const uint64_t* epoch_dev);
int tcavt_dropout_epoch_advance(uint64_t* epoch_dev, tcavt_stream_t stream);

/* Batched copy as ONE kernel launch: items i < n (n <= 16) copy bytes[i] bytes from src[i] to dst[i] (16-byte aligned; dst, src,
 * bytes are HOST arrays read at call time).  src may be pinned, device-mapped HOST memory (hipHostMalloc / a torch tensor made
 * with pin_memory=True): the upload of a training batch (scripts/train.py:1153-1166) as one launch in the stream's queue rather
 * than nine copy-engine transfers (tcavt_amd.data.DeviceFeeder). */
int tcavt_copy_batch(void* const* dst, const void* const* src, const int64_t* bytes, int n, tcavt_stream_t stream);

/* fp32 -> fp16 / bf16 (round-to-nearest-even) copy of n elements, n % 8 == 0 not required */
int tcavt_cast_f32_16(const float* x, void* out16, int64_t n, int dtype16, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused input embedding build (scripts/train.py:521-528):
 *   h[b][i]      = img[b][i] + vis_mod               i <  Nq   (img = q_proj output)
 *   h[b][Nq + j] = table[ids[b][j]] + txt_mod        j <  Lt
 * table fp16 / bf16 (table_dtype) [V][H]; ids int64 [B][Lt]; img fp32 [B][Nq][H]; h fp32 [B][Nq+Lt][H].
 * Ids outside [0,V) are reported through *bad_id_flag (device int, set to 1).
 * h16 / part (optional, both or neither): what the first decoder layer's fused RMSNorm needs (TCAVT_EPI_ROWSCALE) --
 * the 16-bit copy of h (same type as the table) and part[row][npart] with the row's sum of squares in slot 0 and zeros
 * in the others (npart = H / 64, the layout TCAVT_EPI_NORM_OUT writes).
 * h == NULL (h16 given): the 16-bit residual stream of tcavt_llama_stack_forward -- no fp32 copy is written and the sum
 * of squares is that of the ROUNDED values (the stream's own content).
 * ---------------------------------------------------------------------- */
int tcavt_embed_fuse(const void* table_bf16, const int64_t* ids, const float* img,
                     const float* vis_mod, const float* txt_mod, float* h, int B,
                     int Nq, int Lt, int H, int V, int* bad_id_flag, int table_dtype,
                     void* h16, float* part, int npart, float stream_scale /* 0 means 1: h16 = round(stream_scale * h),
                     part = sums of squares of the scaled values (tcavt_llama_stack_args.stream_scale); h stays unscaled */,
                     tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * attention_mask -> per-sample valid key count of the fused sequence
 * (scripts/train.py:531-532: mask = [ones(B,Nq), attention_mask]):
 *   kv_len[b] = Nq + sum_j mask[b][j]
 * The collate function right-pads (train.py:330-331), so a valid mask is a prefix
 * of ones; *not_prefix_flag (device int) is set to 1 if some mask is not.
 * ---------------------------------------------------------------------- */
int tcavt_mask_to_kvlen(const int64_t* mask, int B, int Lt, int Nq, int32_t* kv_len,
                        int* not_prefix_flag, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Causal grouped-query attention over the fused sequence
 * (modeling_llama.py:191-213 semantics = HF sdpa path with a causal AND
 * key-valid mask).
 *   qkv  fp16 / bf16 (dtype16) [B*L][(nq + 2*nkv) * 64]   (q heads | k heads | v heads, RoPE applied)
 *   out  same type [B*L][nq * 64]
 *   kv_len int32 [B]: keys j < kv_len[b] are valid (right padding); a query i
 *   attends keys j <= i with j < kv_len[b]; padded queries are still computed.
 * head_dim is 64; nq % nkv == 0; L <= 544.  softmax in fp32, scale given.
 * ---------------------------------------------------------------------- */
int tcavt_attn_causal_gqa(const void* qkv, void* out, const int32_t* kv_len, int B,
                          int L, int nq, int nkv, float scale, int dtype16, tcavt_stream_t stream);
/* The same, also leaving lse fp32 [B][nq][L] (NULL: none) = log sum_j exp(scale * q_i . k_j) over the keys query i attends
 * (0 for a query without one): with it and the output, tcavt_attn_bwd_scores needs one sweep over the keys instead of two
 * (LoRA-trainable variant, modify_scripts/modify_train.py:512-528). */
int tcavt_attn_causal_gqa_lse(const void* qkv, void* out, float* lse, const int32_t* kv_len, int B,
                              int L, int nq, int nkv, float scale, int dtype16, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Row softmax for the batched cross-attention: P[r][c] = softmax_c(S[r][c]) over c < n_valid,
 * P[r][c] = 0 for n_valid <= c < n_out.  S fp32 (ld lds), P fp16 or bf16 (ld ldp).  dropout_p > 0 applies
 * attention-weight dropout to P (flat index r*n_out + c; nn.MultiheadAttention(dropout=...), train.py:754).
 * (nn.MultiheadAttention's softmax, no mask: scripts/train.py:798.)
 * ---------------------------------------------------------------------- */
int tcavt_softmax_rows(const float* S, int64_t lds, void* P, int64_t ldp, int out_dtype, int rows,
                       int n_valid, int n_out, float dropout_p, uint64_t dropout_seed, uint32_t dropout_site,
                       tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Generic small multi-head attention, fp32 softmax (nn.MultiheadAttention
 * core after the in-projection: scripts/train.py:359,403,406,663,754).
 *   q  [B][Lq] rows of stride ldq, head h at column offset h*dh
 *   k,v [B][Lk] rows of stride ldk / ldv
 *   key_len int32 [B] or NULL: keys j >= key_len[b] are masked (src_key_padding_mask)
 *   out [B][Lq][nh*dh], leading dim ldo
 * in_dtype/out_dtype: TCAVT_F32, TCAVT_BF16 or TCAVT_F16 (q,k,v share in_dtype).
 * No causal mask.  Lk <= 544, Lq*Lk*4 bytes must fit 64 KiB.  dropout_p > 0: dropout on the attention
 * probabilities, flat index ((b*nh + h)*Lq + i)*Lk + j.
 * ---------------------------------------------------------------------- */
int tcavt_mha(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
              int64_t ldv, void* out, int64_t ldo, const int32_t* key_len, int B,
              int Lq, int Lk, int nh, int dh, float scale, int in_dtype,
              int out_dtype, float dropout_p, uint64_t dropout_seed, uint32_t dropout_site,
              tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * fp32 dense layer  C[M,N] = A[M,K] . W[N,K]^T + bias (+ReLU) (+residual)
 * for the parts of the path that must stay fp32 (raw-pixel lane polygons,
 * LTSF head: scripts/train.py:357-365,741-765).  Any M,N,K; flags = EPI_BIAS |
 * EPI_RELU | EPI_RESIDUAL.  dropout_p > 0: dropout after bias / ReLU, before the residual (flat index m*N+n).
 * ---------------------------------------------------------------------- */
int tcavt_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw,
                   const float* bias, const float* residual, int64_t ldr, float* C,
                   int64_t ldc, int M, int N, int K, int flags, float dropout_p,
                   uint64_t dropout_seed, uint32_t dropout_site, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * LanePolygonEncoder front and back (scripts/train.py:362-383):
 *   poly_embed:  x[b][p] = W_in . polygon[b][p] + b_in + pos[p]      (Linear(2,d))
 *   masked_mean: emb[b] = mean_{p < len[b]} enc[b][p]  (zeros when len[b] == 0)
 * ---------------------------------------------------------------------- */
int tcavt_poly_embed(const float* polygon, const float* w_in, const float* b_in,
                     const float* pos, float* x, int B, int P, int D,
                     tcavt_stream_t stream);
int tcavt_masked_mean(const float* enc, const int32_t* len, float* emb, int B, int P,
                      int D, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * TransformerLTSF front (scripts/train.py:837-839, 701-716):
 *   xp[b][c][t] = conv_w[c][0]*x[b][0][t] + conv_w[c][1]*x[b][1][t] + conv_b[c]
 *   enc[b][c][s] = sum_t enc_w[c][s][t]*(xp[b][c][t]-xp[b][c][T-1]) + enc_b[c][s]
 *                  + xp[b][c][T-1] + pos[c][s]
 * x [B][2][T]; out enc_tok token-major [B][T][C]
 * ---------------------------------------------------------------------- */
int tcavt_ltsf_front(const float* x, const float* conv_w, const float* conv_b,
                     const float* enc_w, const float* enc_b, const float* pos,
                     float* enc_tok, float* xp_tok /* optional [B][T][C]: xp, kept for the backward */,
                     int B, int C, int T, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * LTSF_NLinearDecoder front (scripts/train.py:768-785):
 *   dec[b][c][s] = sum_t dec_w[c][s][t]*(e[b][c][t]-e[b][c][T-1]) + dec_b[c][s]
 *                  + e[b][c][T-1] + lane_adj[b][c*To+s]
 * e is given token-major: e_tok [B][T][C] (output layout of the attention block).
 * out dec [B][C*To]
 * ---------------------------------------------------------------------- */
int tcavt_ltsf_decode(const float* e_tok, const float* dec_w, const float* dec_b,
                      const float* lane_adj, float* dec, int B, int C, int T, int To,
                      tcavt_stream_t stream);

/* [B][C][To] -> [B][To][C] transpose (decoded.permute(0,2,1), train.py:793),
 * optional 16-bit copy (dtype16) for the following 16-bit contraction */
int tcavt_transpose_ct(const float* in, float* out_f32, void* out_bf16, int B, int C,
                       int To, int dtype16, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Final head (scripts/train.py:804-805, 941-943):
 *   out[b][f][s] = w[f] . fused[b][s] + bias[f] (+ x[b][f][T-1] when add_last != 0)
 * fused [B][To][C]; x [B][F][T]; out [B][F][To]
 * ---------------------------------------------------------------------- */
int tcavt_out_head(const float* fused, const float* w, const float* bias,
                   const float* x, float* out, int B, int To, int C, int F, int T,
                   int add_last, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * De-normalise + loss + metrics (scripts/train.py:945-962, 1302-1325;
 * scripts/test.py:1342-1382; ablation_study_without_lora.py:1233-1238).
 *   pred [B][K][2][To] (K candidates; K = 1 for the plain path), gt [B][2][To],
 *   norm_stat [B][4] = (min_x, max_x, min_y, max_y)
 *   sums[0] = sum_b sum_t (dx^2)   sums[1] = sum_b sum_t (dy^2)   (K must be 1; MSE numerators)
 *   sums[2] = sum_b min_k ADE      sums[3] = sum_b min_k FDE      sums[4] = sum_b min_k RMSE
 *   argmin [B][3] int32: index of the minimising candidate for ADE/FDE/RMSE
 *   (first minimum, as torch.min / np.argmin)
 * sums must be zeroed by the caller; accumulated with one atomicAdd per block.
 * ---------------------------------------------------------------------- */
int tcavt_traj_metrics(const float* pred, const float* gt, const float* norm_stat,
                       float* sums, int32_t* argmin, float* per_sample, int B, int K,
                       int To, tcavt_stream_t stream);

/* fp32 dense layer with general element strides: A[m][k] = A[m*rsA + k*csA], W[n][k] = W[n*rsW + k*csW];
 * C[M,N] = A . W^T (+bias)(+relu)(+residual).  Lets the backward of an fp32 nn.Linear run without
 * transposes: gx = gy . W  (rsW = 1, csW = ldw)  and  gW = gy^T . x  (rsA = 1, csA = ld_gy; rsW = 1, csW = ld_x). */
int tcavt_gemm_f32_strided(const float* A, int64_t rsA, int64_t csA, const float* W, int64_t rsW,
                           int64_t csW, const float* bias, const float* residual, int64_t ldr, float* C,
                           int64_t ldc, int M, int N, int K, int flags, tcavt_stream_t stream);

/* ========================================================================
 * Training step of scripts/train.py (:1168-1183): backward through the trainable part (lane-polygon
 * encoder + TransformerLTSF; the MLLM is frozen, :1140-1145) and AdamW (:1145).  What autograd does
 * for the reference is spelled out as kernels here; heavy contractions reuse tcavt_gemm_bf16 on
 * transposed operands.
 * ====================================================================== */

/* out[c][r] = in[r][c] for 16-bit elements (bf16 / fp16), `batch` matrices s_in / s_out elements apart;
 * rows r in [rows, rows_pad) of the source are written as zeros (K-padding of the following GEMM).
 * f16_to_bf16 != 0: the source is fp16 (a forward activation) and the copy is converted to bf16 -- the other operand of
 * a gradient-side contraction is a bf16 gradient, and the MFMA wants one type. */
int tcavt_transpose16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols,
                      int rows_pad, int batch, int64_t s_in, int64_t s_out, int f16_to_bf16, tcavt_stream_t stream);
/* fp32 [rows][cols] -> bf16 [cols][rows_pad], zero padded */
int tcavt_transpose_f32_bf16(const float* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                             int cols, int rows_pad, tcavt_stream_t stream);
/* out[n] (+)= sum_m g[m][n]  (bias gradient); dtype of g: TCAVT_F32 or TCAVT_BF16 */
int tcavt_colsum(const void* g, int64_t ld, int dtype, float* out, int M, int N, int accumulate,
                 tcavt_stream_t stream);
/* g[i] = 0 where the saved post-ReLU activation y[i] <= 0 */
int tcavt_relu_bwd(float* g, const void* y, int y_dtype, int64_t n, tcavt_stream_t stream);
/* a[i] += b[i] */
int tcavt_add_inplace(float* a, const float* b, int64_t n, tcavt_stream_t stream);
/* ---- decoder-layer backward for the LoRA-trainable variant (modify_scripts/modify_train.py:512-528) ---- */
/* d(silu(gate) * up): gu [M, 2I] bf16 in the interleaved TCAVT_EPI_SILU_MUL layout, g_act [M, I] bf16 -> g_gu [M, 2I] bf16 */
int tcavt_silu_mul_bwd(const void* gu_bf16, const void* g_act_bf16, void* g_gu_bf16, int64_t M, int I, int dtype16,
                       tcavt_stream_t stream);
/* The 16-bit tensors of this group (names say _bf16 for history) are of the type `dtype16` names: TCAVT_BF16, or TCAVT_F16 --
   the forward's storage contract; the gradients among them then carry the power-of-two scale of tcavt_grad_scale_pick. */
/* LlamaRMSNorm backward w.r.t. its input x [M, H] (H % 8 == 0; x_dtype: TCAVT_F32, or a 16-bit type when the forward kept
   the 16-bit residual stream itself); gy (+ gy2, optional) 16-bit [M, H] of gy_dtype,
   multiplied by *gy_scale when gy_scale != NULL (device scalar: where the backward enters its scale);
   gx = or += (accumulate); gx_bf16 (optional): a 16-bit copy (out_dtype) of the updated gx, the next dgrad GEMM's operand */
int tcavt_rmsnorm_bwd(const void* x, const float* gamma, const void* gy_bf16, const void* gy2_bf16, float eps,
                      float* gx, void* gx_bf16, int accumulate, int M, int H, int gy_dtype, int out_dtype,
                      const float* gy_scale, int x_dtype, tcavt_stream_t stream);
/* scale[0] = S = 2^k with max|g_a, g_b| * S in [target / 2, target], scale[1] = 1 / S, decided on the device (S = 1 for
   all-zero or non-finite input); g_a, g_b (optional) 16-bit [n] of dtype16; scratch: one uint32, zero-initialised once.
   backoff (optional device int32): S is divided by 2^*backoff -- dynamic loss scaling: tcavt_adamw_gated raises its ctl[6] by
   four when it skips an update for a non-finite gradient norm and gives one back per 256 applied updates */
int tcavt_grad_scale_pick(const void* g_a, const void* g_b, int64_t n, int dtype16, float target, float* scale,
                          uint32_t* scratch, const int32_t* backoff, tcavt_stream_t stream);
/* fp32 gradient of the rotated q|k|v [M, ncols] -> bf16 gradient of the projection outputs: transposed RoPE rotation on
   the first rope_cols columns (heads of 64), plain conversion on the rest; tables as for TCAVT_EPI_ROPE ([L, 32]) */
int tcavt_rope_bwd_pack(const float* g32, void* out_bf16, const float* rope_cos, const float* rope_sin, int64_t M,
                        int ncols, int rope_cols, int L, int dtype16, tcavt_stream_t stream);
/* backward of tcavt_attn_causal_gqa (head_dim 64, T <= 280): qkv = the forward's rotated q|k|v [B*T, (nq+2nkv)*64] bf16,
   dO [B*T, nq*64] bf16; g32 [B*T, (nq+2nkv)*64] fp32 in the same layout, ZEROED by the caller (k/v parts are accumulated
   with float atomics) */
int tcavt_attn_causal_gqa_bwd(const void* qkv_bf16, const void* dO_bf16, float* g32, const int32_t* kv_len, int B, int T,
                              int nq, int nkv, int head_dim, float scale, tcavt_stream_t stream);
/* row-wise middle of the composed attention backward: S (scaled scores) and dP = dO V^T, fp32 [B*nq*T, Tp];
   P = causal softmax (keys < min(i+1, kv_len[b])), dS = scale * P * (dP - sum P dP); both bf16 [.., Tp], zero-filled */
int tcavt_causal_softmax_bwd_rows(const float* S, const float* dP, void* P_bf16, void* dS_bf16, const int32_t* kv_len,
                                  int B, int T, int Tp, int nq, float scale, tcavt_stream_t stream);
/* the same arithmetic, tiled (the production form): writes dS row-major [B*nq*T, Tp] and the transposed P^T, dS^T
   [B*nq*Tp, Tp] (rows = keys, columns = queries) directly; Tp = T rounded up to 64.  All three outputs must be
   ZERO-INITIALISED once by the caller: key blocks above the causal diagonal are never written */
int tcavt_causal_softmax_bwd_tiles(const float* S, const float* dP, void* dS_bf16, void* PT_bf16, void* dST_bf16,
                                   const int32_t* kv_len, int B, int T, int Tp, int nq, float scale, tcavt_stream_t stream);
/* production form of the middle of the attention backward: S = scale q K^T and dP = dO V^T on the matrix cores inside
   the kernel (never stored), softmax backward, outputs as tcavt_causal_softmax_bwd_tiles (dS, P^T, dS^T; ZERO-INITIALISED
   once by the caller).  qkv = the forward's rotated q|k|v [B*T, (nq+2nkv)*64] bf16, dO [B*T, nq*64] bf16.
   dQ (optional): fp32 [B*T, ld_dq], head h at columns 64 h, receives dQ = dS K computed in the same kernel; dS_bf16
   (optional) is the row-major dS for an external product; at least one of the two.  PT / dST are optional (both or
   neither; the GEMM form of dK, dV); stats (optional; required without PT/dST): fp32 [B*nq*T, 4] = row maximum of the
   scaled scores, 1 / row sum, sum(P dP), 0 -- the input of tcavt_attn_bwd_dkv.
   lse + att (optional, both or neither): the forward's log-sum-exp fp32 [B*nq*T] and output 16-bit [B*T, nq*64]
   (tcavt_attn_causal_gqa_lse).  With them the row statistics come from the forward -- P = exp(s - lse), sum(P dP) = dO . O --
   and the kernel makes one sweep over the key blocks instead of two (stats then holds lse, 1, dO . O, 0) */
int tcavt_attn_bwd_scores(const void* qkv_bf16, const void* dO_bf16, void* dS_bf16, void* PT_bf16, void* dST_bf16,
                          float* dQ, int64_t ld_dq, float* stats, const int32_t* kv_len, int B, int T, int Tp, int nq,
                          int nkv, int head_dim, float scale, int dtype16, const float* lse, const void* att,
                          tcavt_stream_t stream);
/* The attention backward of the LoRA-trainable variant in two launches when a head's keys / queries fit in LDS
   (tcavt_attn_bwd_resident_ok: T <= 256, 16 % (nq / nkv) == 0): K, V, K^T resident for dQ; q, dO and their transposes resident
   for dK, dV; row statistics from the forward (lse, att: tcavt_attn_causal_gqa_lse).  Writes g_qkv16 [B*T][(nq + 2 nkv) * 64]
   = the 16-bit gradient of the q|k|v projection's output with the RoPE rotation of q and k undone (rope_cos / rope_sin
   fp32 [T][32], position = row inside the sample) -- what tcavt_attn_bwd_scores + tcavt_attn_bwd_dkv + tcavt_rope_bwd_pack
   produce through an fp32 buffer.  stats: fp32 [B*nq*T][4] scratch (lse, 1, dO . O, 0 per query row). */
int tcavt_attn_bwd_resident_ok(int T, int nq, int nkv);
int tcavt_attn_bwd_resident(const void* qkv16, const void* dO16, const void* att16, const float* lse, void* g_qkv16,
                            float* stats, const float* rope_cos, const float* rope_sin, const int32_t* kv_len, int B, int T,
                            int nq, int nkv, int head_dim, float scale, int dtype16, tcavt_stream_t stream);
/* dK, dV of the attention backward, key-major on the matrix cores (one workgroup per sample, key/value head and 64 keys;
   P^T, dS^T rebuilt from `stats`, the query heads of the group summed in registers): writes the k and v columns of
   g32 [B*T, (nq+2nkv)*64] fp32 (every row; no zero-initialisation needed) */
int tcavt_attn_bwd_dkv(const void* qkv_bf16, const void* dO_bf16, const float* stats, float* g32, const int32_t* kv_len,
                       int B, int T, int Tp, int nq, int nkv, int head_dim, float scale, int dtype16, tcavt_stream_t stream);
/* G3 fp32 [M, 3*nq*64] = dQ | dK per query head | dV per query head -> bf16 [M, (nq+2nkv)*64]: group sums + RoPE^T */
int tcavt_gqa_rope_bwd_pack(const float* G3, void* out_bf16, const float* rope_cos, const float* rope_sin, int64_t M,
                            int nq, int nkv, int head_dim, int L, tcavt_stream_t stream);
/* Weight gradient with a skinny output, no physical transposes (LoRA adapters: dA = g_t^T x, dB^T = t^T g_qkv):
 *   C[i][h] += sum_m G[m][g_col0 + i] * X[m][h]     i < n (16, 32, 48 or 64), h < H, contraction over the M tokens
 * G bf16 [M][ldg] (a gradient), X 16-bit [M][ldx] of x_dtype (an fp16 forward activation is converted to bf16 on the way),
 * C fp32, ACCUMULATED into with float atomics (zero it or let it hold an earlier contribution): [n][ldc], or, with
 * trans_out != 0, the transpose [H][ldc].
 * g_dtype: 0 / TCAVT_BF16 (as described), or TCAVT_F16 together with an fp16 X: both operands as they are on the f16 MFMA.
 * rs_part (optional, NULL: none): fp32 [M][rs_npart] partial sums of squares of the fused RMSNorm over rs_h features
 *   (tcavt_llama_layer.tape_part): row m of G is multiplied by rsqrt(sum / rs_h + rs_eps) on the way in -- dB of the adapters
 *   from the taped, un-normalised t. */
int tcavt_wgrad_tn(const void* G, int64_t ldg, int g_col0, int n, const void* X, int64_t ldx, int x_dtype, float* C,
                   int64_t ldc, int M, int H, int trans_out, int g_dtype, const float* rs_part, int rs_npart, int rs_h,
                   float rs_eps, tcavt_stream_t stream);
/* dA of both adapters in one pass over the taped residual stream (LoRA-trainable variant; modify_scripts/modify_train.py:512-528:
 * PEFT lora_A of q_proj and v_proj behind their own lora_dropout modules, input = input_layernorm(h)):
 *   dA[r][n]      += gamma[n] * sum_m g_t[m][r]      * drop_q(round16(rs[m] * x16[m][n]))        r < 16
 *   dA[16 + r][n] += gamma[n] * sum_m g_t[m][16 + r] * drop_v(round16(rs[m] * x16[m][n]))
 * x16 16-bit [M][H] = the layer's input stream, part fp32 [M][npart] = its partial sums of squares (rs as in tcavt_wgrad_tn),
 * gamma fp32 [H], g_t 16-bit [M][64], dA fp32 [>= 32][ldc] ACCUMULATED into with float atomics; masks of sites site_q / site_v
 * as tcavt_lora_down draws them (dropout_p == 0: none).  Replaces rmsnorm + two mask kernels + two tcavt_wgrad_tn launches. */
int tcavt_lora_wgrad_a(const void* x16, const float* part, int npart, float eps, const float* gamma, const void* g_t, float* dA,
                       int64_t ldc, int M, int H, float dropout_p, uint64_t dropout_seed, uint32_t site_q, uint32_t site_v,
                       int dtype16, tcavt_stream_t stream);

/* clip_grad_norm_ on a flat fp32 gradient vector (modify_scripts/modify_train.py:1192):
     g *= grad_scale;  g *= min(1, max_norm / (||g|| + 1e-6))
   grad_scale = 1 / world turns a SUM-all-reduced data-parallel gradient into the DDP-averaged one the reference clips
   (pass grad_scale = 1 to tcavt_adamw afterwards).  Fixed summation order, no host synchronisation; scratch: >= 1026
   floats (scratch[1024] = the total factor applied, [1025] = the norm of the scaled gradient before clipping) */
int tcavt_clip_grad_norm(float* g, int64_t n, float max_norm, float grad_scale, float* scratch, tcavt_stream_t stream);
/* nn.LayerNorm backward; x is the LayerNorm input; ggamma / gbeta are ACCUMULATED (zero them first) */
int tcavt_layernorm_bwd(const float* x, const float* gamma, const float* gy, float eps, float* gx,
                        float* ggamma, float* gbeta, int M, int D, tcavt_stream_t stream);
/* backward of tcavt_mha (fp32 operands): gq, gk, gv share leading dimension ldg; dropout_p > 0 regenerates the
   attention-weight mask of the forward call with the same (seed, site) */
int tcavt_mha_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                  const float* go, int64_t ldo, float* gq, float* gk, float* gv, int64_t ldg,
                  const int32_t* key_len, int B, int Lq, int Lk, int nh, int dh, float scale,
                  float dropout_p, uint64_t dropout_seed, uint32_t dropout_site,
                  tcavt_stream_t stream);
/* backward of tcavt_softmax_rows (+ the 1/sqrt(dh) score scale): dS bf16, zero padded to n_out */
int tcavt_softmax_bwd_rows(const void* P_f16, int64_t ldp, const float* dP, int64_t ldd, void* dS_bf16,
                           int64_t lds, float scale, int rows, int n_valid, int n_out,
                           tcavt_stream_t stream);
/* d loss / d decoded for loss = MSE_x + MSE_y in pixel space (train.py:945-961) */
int tcavt_mse_grad(const float* pred, const float* gt, const float* norm_stat, float* g, int B, int To,
                   tcavt_stream_t stream);
/* backward of tcavt_out_head: gf [B][To][C], gw [F][C], gb [F] */
int tcavt_out_head_bwd(const float* g, const float* fused, const float* w, float* gf, float* gw, float* gb,
                       int B, int To, int C, int F, tcavt_stream_t stream);
/* backward of the per-channel N-Linear blocks (tcavt_ltsf_front's encoder, tcavt_ltsf_decode):
 * in_tok / gin_tok token-major [B][T][C]; g addressed as g[b*g_sb + c*g_sc + s*g_ss]; gin_tok may be NULL */
int tcavt_nlinear_bwd(const float* in_tok, const float* W, const float* g, int64_t g_sb, int64_t g_sc,
                      int64_t g_ss, float* gW, float* gbias, float* gin_tok, int B, int C, int T, int S,
                      tcavt_stream_t stream);
/* Conv1d(k=1) token projection backward: gxp_tok [B][T][C], x [B][F][T] -> gw [C][F], gb [C] */
int tcavt_conv1x1_bwd(const float* gxp_tok, const float* x, float* gw, float* gb, int B, int C, int T,
                      int F, tcavt_stream_t stream);
int tcavt_poly_embed_bwd(const float* g, const float* polygon, float* gw, float* gb, float* gpos, int B,
                         int P, int D, tcavt_stream_t stream);
int tcavt_masked_mean_bwd(const float* gemb, const int32_t* len, float* genc, int B, int P, int D,
                          tcavt_stream_t stream);
/* torch.optim.AdamW step over a flat fp32 vector; grad_scale multiplies g first (1/world for DP mean) */
int tcavt_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int step, float grad_scale, tcavt_stream_t stream);

/* ========================================================================
 * Stage-level entry points: the decoder stack issued from C++ -- one call enqueues the whole
 * HF LlamaModel.forward (modeling_llama.py:376-424 as called from scripts/train.py:446-452) for a batch: per layer
 *   [LoRA down-projection(s)]  q|k|v projection with the input RMSNorm fused in (gamma folded into the weights, 1 / rms
 *   as a row scale in the RoPE epilogue) + LoRA update  ->  causal GQA attention  ->  o_proj + residual (emitting the next
 *   norm's 16-bit input and partial sums of squares)  ->  gate|up with the post-attention RMSNorm fused in the same way,
 *   SiLU * up epilogue  ->  down_proj + residual (again emitting the next norm's inputs);  then the final RMSNorm.
 * No allocation, no synchronisation, graph-capturable; every buffer is the caller's.  Host cost: a few microseconds
 * per launch instead of a Python call each.
 * ====================================================================== */
typedef struct tcavt_llama_layer {
  const void* w_qkv;  /* 16-bit [(nq + 2 nkv) * 64][H], rows q | k | v, input_layernorm.weight folded in (W * gamma) */
  const void* a_cat;  /* 16-bit [64][H] or NULL (no LoRA): A_q in rows [0, r), A_v in rows [16, 16 + r), gamma folded in */
  const void* b_ext;  /* 16-bit [(nq + 2 nkv) * 64][64]: B_q in columns [0, r) of the q rows, B_v in [16, 16 + r) of the v rows */
  const void* w_o;    /* 16-bit [H][nq * 64] */
  const void* w_gu;   /* 16-bit [2 I][H], gate / up rows interleaved in blocks of 16 (TCAVT_EPI_SILU_MUL), post_attention_layernorm.weight folded in */
  const void* w_d;    /* 16-bit [H][I] */
  /* LoRA-trainable variant (optional, all NULL otherwise): per-layer buffers the backward reads (csrc/llm_backward.hip) */
  float* tape_h_mid;  /* [M][H]: residual stream after the attention half (then h_in stays untouched): fp32, or -- 16-bit
                         residual stream, tcavt_llama_stack_args.h == NULL -- of the 16-bit storage type (cast the pointer) */
  float* tape_h_out;  /* [M][H]: residual stream after the MLP half (same type) */
  void* tape_qkv;     /* 16-bit [M (+ pad)][(nq + 2 nkv) * 64]: rotated q|k|v of this layer */
  void* tape_gu;      /* 16-bit [M][2 I]: gate|up pre-activations (interleaved layout) */
  void* tape_t;       /* 16-bit [M][64]: LoRA down-projection */
  void* tape_att;     /* optional, 16-bit [M][nq * 64]: the attention's output (instead of the shared workspace `att`) ... */
  float* tape_lse;    /* ... and fp32 [B][nq][L]: its log-sum-exp per query row (tcavt_attn_causal_gqa_lse); both or neither:
                         with them tcavt_attn_bwd_scores runs one sweep over the keys instead of two */
  float* tape_part;   /* optional, fp32 [M][npart_in]: the partial sums of squares of this layer's INPUT stream (what its fused
                         input RMSNorm reads; otherwise they live in the shared `part` and are overwritten by the o_proj
                         epilogue) -- 1 / rms of every token for the adapters' weight gradients */
} tcavt_llama_layer;

typedef struct tcavt_llama_stack_args {
  const tcavt_llama_layer* layers; /* HOST array of n_layers entries */
  const float* gamma_final;        /* fp32 [H]: model.norm.weight */
  const float* rope_cos;           /* fp32 [L][32] */
  const float* rope_sin;
  float* h;                        /* fp32 [M][H]: the fused input embeddings; updated in place unless a tape is kept.
                                      NULL: 16-bit residual stream -- h16 IS the stream (its partial sums those of the rounded
                                      values: tcavt_embed_fuse with h == NULL, tcavt_rownorm_prep with rounded_sums = 1); every
                                      residual epilogue adds to it in place (TCAVT_EPI_NORM_OUT with C == NULL) and the final
                                      norm is tcavt_rmsnorm16.  4 instead of 10 bytes per element and epilogue through HBM;
                                      accumulation, norms, softmax stay fp32.  Not with a tape (the backward reads fp32 streams) */
  void* h16;                       /* 16-bit [M][H]: copy of h (tcavt_embed_fuse writes it); rewritten by every residual epilogue */
  float* part;                     /* fp32 [M][>= npart_in], at least [M][H / 16]: partial sums of squares of h's rows (same producers) */
  const int32_t* kv_len;           /* int32 [B] */
  /* workspaces, 16-bit */
  void* qkv;                       /* [M][(nq + 2 nkv) * 64] (unused when the layers carry tape_qkv) */
  void* att;                       /* [M][nq * 64] */
  void* act;                       /* [M][I] */
  void* t;                         /* [M][64], zero-initialised once by the caller (columns >= 32 are never written) */
  void* xq;                        /* unused (kept for layout stability): the adapters' dropout masks are applied inside tcavt_lora_down */
  void* xv;
  /* outputs */
  float* out_f32;                  /* fp32 [M][H] or NULL: hidden_states[-1] (scripts/train.py:553) */
  void* out16;                     /* 16-bit [M][H] or NULL */
  /* optional: rotated keys / values of every layer, for the decode steps that follow a prefill (text generation,
     scripts/train.py:577-654): 16-bit [n_layers][B][kv_lmax][nkv * 64] each */
  void* k_cache;
  void* v_cache;
  /* optional: HOST array of hipEvent_t, 10 per layer (start / stop around q|k|v, attention, o, gate|up, down) --
     in-situ kernel timing on the launching stream (bench.py's roofline leg) */
  void* const* events;
  int32_t n_layers, B, L, H, I, nq, nkv, dtype16;
  int32_t kv_lmax;
  int32_t gemm_tile;               /* tcavt_gemm_args.tile for the four big projections (0 = auto) */
  int32_t npart_in;                /* partials per row in `part` on entry: must equal tcavt_norm_npart(M, H, I) */
  float stream_scale;              /* 0 means 1.  A power of two <= 1: h16 and `part` arrive holding stream_scale * x (tcavt_embed_fuse /
                                      tcavt_rownorm_prep with the same value) and every 16-bit image of the residual stream is kept at
                                      that scale (tcavt_gemm_args.norm_scale): the overflow limit of an fp16 stream moves from 65504 to
                                      65504 / stream_scale at no cost in precision.  The fused norms run with eps * stream_scale^2; the adapters'
                                      un-normalised t = lora_scale * (stream_scale x) . A^T is at the stream's scale like the main term of the
                                      q|k|v accumulator, and the row scale takes it out of both.  out_f32 / out16 are the true-scale final
                                      hidden states.  Not with a tape (the backward reads the streams at scale 1) */
  float rms_eps, lora_scale;       /* lora_scale = alpha / r */
  float lora_dropout_p;            /* > 0: train mode; sites first_site + 2 l (q_proj), first_site + 2 l + 1 (v_proj) */
  uint32_t lora_first_site;
  uint64_t dropout_seed;
  /* optional: device int32 word; a residual epilogue that produces a non-finite value stores 1 + 2 * layer (o_proj) or
     2 + 2 * layer (down_proj) into it if it is still 0 (tcavt_gemm_args.nonfinite_flag): the first layer whose output left
     the 16-bit range, or whose inputs already had */
  int32_t* nonfinite_flag;
  /* optional: workspace for the two-launch split K of the residual projections on grids that leave CUs idle (small B * L:
     tcavt_gemm_args.splitk_ws; 16 KiB + S * M * H * 4 bytes, S <= 8: 64 MiB covers M <= 2048).  NULL: one launch each */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
} tcavt_llama_stack_args;

int tcavt_llama_stack_forward(const tcavt_llama_stack_args* args, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * The LoRA-trainable variant's decoder backward as ONE call (modify_scripts/modify_train.py:512-528: lora_A / lora_B of
 * q_proj, v_proj trainable, every base weight frozen; what loss.backward() does between the final hidden states and the
 * adapters' .grad, and -- input_grad -- the decoder's input embeddings).  Per layer, last to first:
 *   dgrad of down_proj with d(silu(gate) * up) in its epilogue (TCAVT_EPI_SILU_BWD, in place on the taped pre-activations)
 *   -> dgrad of gate|up -> RMSNorm backward (post-attention norm; accumulates into the fp32 residual gradient g_h)
 *   -> dgrad of o_proj -> attention backward (tcavt_attn_bwd_resident) -> g_t = s * g_qkv . B_ext
 *   -> [leaf stream] adapter weight gradients (tcavt_lora_wgrad_a, row-scaled tcavt_wgrad_tn), written into the caller's
 *      gradient tensors times scale[1]
 *   -> adapters' input gradient (tcavt_lora_dgrad) -> dgrad of q|k|v -> RMSNorm backward (input norm).
 * Requires the forward's tape with 16-bit streams, tape_att / tape_lse / tape_part (tcavt_llama_layer), L <= 256,
 * 16 % (nq / nkv) == 0, M % 256 == 0, I % 256 == 0, H % 128 == 0, adapters of rank <= 16, head_dim 64.  dtype16 must be
 * TCAVT_F16 (16-bit stream tapes exist for fp16 storage only; anything else is refused): the incoming gradients g_final_a / b
 * are bf16 and the walk runs under the power-of-two scale picked here (tcavt_grad_scale_pick into scale[0..1]).  Every layer's
 * pointers are checked before the first launch; should a launch fail in the middle of the walk, the caller's stream still
 * joins the leaf stream before the error is returned.  No allocation, no synchronisation; the leaf work is ordered against the
 * main stream with the caller's events.  Weights: transposes of the forward's (dgrad operands, 16-bit): W^T stored [in][out].
 * ---------------------------------------------------------------------- */
typedef struct tcavt_llama_bwd_layer {
  const void* w_dT;    /* [I][H]:   down_proj.weight^T */
  const void* w_guT;   /* [H][2 I]: gate|up (interleaved rows of the forward's w_gu, post-attention gain folded in) transposed */
  const void* w_oT;    /* [nq * 64][H] */
  const void* w_qkvT;  /* [H][(nq + 2 nkv) * 64], input gain folded in */
  const void* b_extT;  /* [64][(nq + 2 nkv) * 64]: B_ext^T */
  const void* a_qT;    /* [H][64]: plain A_q^T in columns [0, r), zeros elsewhere */
  const void* a_vT;    /* [H][64]: plain A_v^T in columns [16, 16 + r) */
  const float* g1;     /* fp32 [H]: input_layernorm.weight */
  const float* g2;     /* fp32 [H]: post_attention_layernorm.weight */
  /* the forward's tape of this layer (tcavt_llama_layer.tape_*; h_in = the previous layer's tape_h_out, or the fused embeddings) */
  const void* h_in;    /* 16-bit [M][H] */
  const void* h_mid;   /* 16-bit [M][H] */
  const void* qkv;     /* 16-bit [M + 64][(nq + 2 nkv) * 64] */
  void* gu;            /* 16-bit [M][2 I]: overwritten with d(gate|up) */
  const void* att;     /* 16-bit [M][nq * 64] */
  const float* lse;    /* fp32 [B][nq][L] */
  const float* part;   /* fp32 [M][npart] */
  const void* t;       /* 16-bit [M][64] */
  /* outputs: the adapters' gradients, fp32, OVERWRITTEN: lora_A [r][H] (leading dimension H), lora_B [out][r] (leading dimension r) */
  float* g_Aq;
  float* g_Av;
  float* g_Bq;
  float* g_Bv;
} tcavt_llama_bwd_layer;

typedef struct tcavt_llama_backward_args {
  const tcavt_llama_bwd_layer* layers; /* HOST array of n_layers entries */
  const void* h_last;                  /* 16-bit [M][H]: the last layer's tape_h_out (input of the final norm) */
  const float* gamma_final;            /* fp32 [H] */
  const void* g_final_a;               /* bf16 / 16-bit [M][H]: gradient of the post-final-norm hidden states */
  const void* g_final_b;               /* optional second summand of it */
  const float* rope_cos;               /* fp32 [L][32] */
  const float* rope_sin;
  const int32_t* kv_len;               /* int32 [B] */
  float* scale;                        /* fp32 [2] */
  uint32_t* scale_scratch;             /* one uint32, zero-initialised once */
  /* scratch, caller-owned */
  float* g_h;                          /* fp32 [M][H]: the residual gradient; on return dL/d(input embeddings) * scale[0] if input_grad */
  void* g_hb;                          /* 16-bit [M][H] */
  void* g_xn;                          /* 16-bit [M][H] */
  void* g_xl;                          /* 16-bit [M][H] */
  void* g_att;                         /* 16-bit [M][nq * 64] */
  void* g_qkv0;                        /* 16-bit [M][(nq + 2 nkv) * 64] each (layer parity) */
  void* g_qkv1;
  void* g_t0;                          /* 16-bit [M][64] each */
  void* g_t1;
  float* dA;                           /* fp32 [64][H] */
  float* dB;                           /* fp32 [(nq + 2 nkv) * 64][64] */
  float* stats;                        /* fp32 [B * nq * L][4] */
  /* leaf work: a second stream and four events (hipEvent_t: ready[0..1], done[0..1]); leaf_stream == NULL: on `stream` itself */
  tcavt_stream_t leaf_stream;
  void* const* events;
  int32_t n_layers, B, L, H, I, nq, nkv, dtype16;
  int32_t npart;                       /* partials per row in `part` */
  int32_t lora_rank;
  int32_t input_grad;                  /* != 0: also walk through layer 0's projections (g_h then holds the input gradient) */
  int32_t reserved0;
  float rms_eps, lora_scale;
  float lora_dropout_p;                /* > 0: the forward's masks, sites lora_first_site + 2 l (q_proj), + 2 l + 1 (v_proj) */
  uint32_t lora_first_site;
  uint64_t dropout_seed;
  const int32_t* scale_backoff;        /* optional device int32: extra binary orders of headroom under the picked scale
                                          (tcavt_grad_scale_pick's `backoff`; tcavt_adamw_gated keeps it in ctl[6]) */
} tcavt_llama_backward_args;

int tcavt_llama_stack_backward(const tcavt_llama_backward_args* args, tcavt_stream_t stream);

/* Backward of tcavt_lora_down w.r.t. its input, both adapters and their masks in one pass (LoRA-trainable variant,
 * modify_scripts/modify_train.py:512-528):
 *   out[m][n] = mask_q[m][n] * sum_r g_t[m][r] A_q[r][n]  +  mask_v[m][n] * sum_r g_t[m][16 + r] A_v[r][n]
 * g_t 16-bit [M][64] (columns 0-15: dL/dt of the q adapter, 16-31: of the v adapter), aqT / avT 16-bit [H][64]: the plain
 * (no folded gain) A matrices transposed, rank r of the q adapter in column r / of the v adapter in column 16 + r, the other
 * adapter's columns ZERO; out 16-bit [M][H].  Masks as in tcavt_lora_down (dropout_p == 0: none).  H % 128 == 0. */
int tcavt_lora_dgrad(const void* g_t, const void* aqT, const void* avT, void* out, int M, int H, float dropout_p,
                     uint64_t dropout_seed, uint32_t site_q, uint32_t site_v, int dtype16, tcavt_stream_t stream);

/* ------------------------------------------------------------------------
 * Post-LN nn.TransformerEncoderLayer / nn.TransformerDecoderLayer stacks as one call: the Q-Former's encoder and decoder
 * (scripts/train.py:388-414; dtype16 = TCAVT_F16 / TCAVT_BF16: 16-bit weights and activations on MFMA) and the lane-polygon
 * encoder's layers (scripts/train.py:352-383; dtype16 = 0: fp32 end to end).  Per layer: self-attention (in_proj, tcavt_mha,
 * out_proj + residual, LayerNorm), for decoder layers (w_q != NULL) cross-attention over `mem`, feed-forward (ReLU), each
 * residual sum followed by its LayerNorm.  Train-mode dropout (dropout_p > 0) at the reference's sites -- attention weights,
 * after out_proj, after ReLU, after linear2 (+ the cross-attention's two) -- numbered in call order from first_site.
 * All buffers are caller-owned: shared between layers in inference, one set per layer when a backward will read them.
 * ---------------------------------------------------------------------- */
typedef struct tcavt_tlayer {
  const void* w_in;  const float* b_in;   /* self-attention in_proj  [3E][E], [3E]  (weights: 16-bit or fp32 per dtype16) */
  const void* w_out; const float* b_out;  /* self-attention out_proj [E][E], [E] */
  const void* w_q;   const float* b_q;    /* cross-attention q projection [E][E] -- NULL: encoder layer */
  const void* w_kv;  const float* b_kv;   /* cross-attention k|v projection [2E][E] */
  const void* w_co;  const float* b_co;   /* cross-attention out_proj */
  const void* w1;    const float* b1;     /* linear1 [FF][E] */
  const void* w2;    const float* b2;     /* linear2 [E][FF] */
  const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b; /* LayerNorms (n3: decoder layers) */
  /* activations of this layer, [M = B*L rows] unless noted; "16" = dtype16 in 16-bit mode, fp32 otherwise */
  float* qkv;            /* [M][3E] fp32 */
  void* att;             /* [M][E] 16 */
  float* y;              /* [M][E] x + self-attention */
  float* x1; void* x1b;  /* LayerNorm1 output, fp32 + 16-bit copy (16-bit mode) */
  float* cq;             /* [M][E] fp32 (decoder) */
  float* ckv;            /* [B*Lk][2E] fp32 (decoder) */
  void* catt;            /* [M][E] 16 (decoder) */
  float* cy;             /* [M][E] x1 + cross-attention (decoder) */
  float* x2; void* x2b;  /* LayerNorm2 output (decoder) */
  void* ffh;             /* [M][FF] 16 */
  float* y2;             /* [M][E] feed-forward residual sum */
  float* out; void* outb;/* layer output (LayerNorm2 / LayerNorm3), fp32 + 16-bit copy */
} tcavt_tlayer;

typedef struct tcavt_tstack_args {
  const tcavt_tlayer* layers;  /* HOST array */
  const float* x;              /* input tokens fp32 [B*L][E] */
  const void* xb;              /* their 16-bit copy (16-bit mode) */
  const float* mem;            /* decoder memory fp32 [B*Lk][E] (decoder layers) */
  const void* memb;
  const int32_t* key_len;      /* int32 [B] valid self-attention keys per sample, or NULL */
  int32_t n_layers, B, L, Lk, E, FF, nhead;
  int32_t dtype16;             /* 0: fp32 layers; TCAVT_F16 / TCAVT_BF16 */
  float dropout_p;
  uint32_t first_site;
  uint64_t dropout_seed;
} tcavt_tstack_args;

int tcavt_tlayer_stack_forward(const tcavt_tstack_args* args, tcavt_stream_t stream);

/* Backward of tcavt_tlayer_stack_forward for fp32 ENCODER layers (dtype16 == 0, no cross-attention): the lane-polygon
 * encoder's layers in the training step (scripts/train.py:352-383 under loss.backward(), :1168-1183).  `fwd` is the forward
 * call's argument block with every layer's activations still in place (a forward that feeds a backward gives each layer its
 * own buffers); parameter gradients are fp32 and ADDED to (zero them first).  The 16-bit stacks (Q-Former, only trainable
 * in modify_scripts/modify_train.py) are composed by the caller from the kernel-level entry points. */
typedef struct tcavt_tlayer_grads {
  float *g_w_in, *g_b_in, *g_w_out, *g_b_out, *g_w1, *g_b1, *g_w2, *g_b2, *g_n1_w, *g_n1_b, *g_n2_w, *g_n2_b;
} tcavt_tlayer_grads;

typedef struct tcavt_tstack_bwd_args {
  const tcavt_tstack_args* fwd;
  const tcavt_tlayer_grads* grads; /* HOST array, fwd->n_layers entries */
  const float* g_out;              /* [M][E]: dL/d (stack output), M = B*L */
  float* g_x;                      /* [M][E]: dL/d (stack input) (out) */
  /* workspaces, shared by the layers */
  float *g_tmp, *g_y2, *g_y2d, *g_x1, *g_y, *g_yd, *g_att; /* [M][E] each (g_y2d / g_yd: train mode only) */
  float* g_f;                      /* [M][FF] */
  float* g_qkv;                    /* [M][3E] */
} tcavt_tstack_bwd_args;

int tcavt_tlayer_stack_backward(const tcavt_tstack_bwd_args* args, tcavt_stream_t stream);

/* The trajectory head's cross-attention over the LLM's final hidden states (scripts/train.py:795-798) in absorbed form, one
 * call: q'_h = q_h W_k[h], scores = q' F^T / sqrt(dh), P = dropout(softmax), ctx = P F, att_h = ctx_h W_v[h]^T + b_v[h]
 * (identical to nn.MultiheadAttention in exact arithmetic: b_k is softmax-invariant, rows of P sum to 1).  fp16 storage. */
typedef struct tcavt_cross_attn_args {
  const void* q;       /* fp16 [B*To][H]: projected queries (in_proj_weight[:H], bias included) */
  const void* wk_t;    /* fp16 [nhead][H][dh]: W_k per head, transposed */
  const void* w_v;     /* fp16 [H][H]: in_proj_weight[2H:] */
  const float* b_v;    /* fp32 [H] */
  const void* fh;      /* fp16 [B*L (+ >= Lp - L zeroed rows)][H]: final hidden states */
  void* fh_t;          /* fp16 [H][B*Lp] workspace: receives F^T per sample */
  void* qp;            /* fp16 [nhead][B*To][H] workspace (kept for the backward) */
  float* scores;       /* fp32 [B*nhead*To][Lp] */
  void* probs;         /* fp16 [B*nhead*To][Lp]: dropped probabilities, zero beyond L */
  void* ctx;           /* fp16 [nhead][B*To][H] (kept for the backward) */
  void* att;           /* fp16 [B*To][H]: output, heads side by side */
  int32_t B, To, L, Lp, H, nhead, dtype16;
  float dropout_p;     /* attention-weight dropout (train mode) */
  uint32_t dropout_site;
  uint32_t reserved0;
  uint64_t dropout_seed;
} tcavt_cross_attn_args;

int tcavt_cross_attn_forward(const tcavt_cross_attn_args* args, tcavt_stream_t stream);

/* Backward of tcavt_cross_attn_forward for constant hidden states (frozen MLLM): weight / bias gradients of the packed
 * in_proj ([3H][H] / [3H]: rows H.. of W_k, 2H.. of W_v; the key bias gets no gradient: softmax-invariant) and the gradient of
 * the projected queries.  Gradient-side tensors are bf16. */
typedef struct tcavt_cross_attn_bwd_args {
  const tcavt_cross_attn_args* fwd; /* the forward call's arguments (q, ctx, probs, scores, fh, shapes, dropout) */
  const void* g_att;     /* bf16 [B*To][H]: dL/d att */
  const float* w_in;     /* fp32 [3H][H]: cross_attn.in_proj_weight (the parameter itself) */
  float* gw_in;          /* fp32 [3H][H]: its gradient; rows H..3H are WRITTEN */
  float* gb_in;          /* fp32 [3H]: bias gradient; entries 2H..3H are ADDED to */
  void* g_q;             /* bf16 [B*To][H]: dL/d q (out) */
  /* workspaces (Mp = B*To rounded up to 64) */
  void* fh_tb;           /* bf16 [H][B*Lp] */
  void* fh_b;            /* bf16 [B*Lp][H] */
  void* ga_t;            /* bf16 [H][Mp] */
  void* g_ctx;           /* bf16 [nhead][B*To][H] */
  void* w_t;             /* bf16 [H*dh] */
  void* x_t;             /* bf16 [H][Mp] */
  float* d_p;            /* fp32 [B*nhead*To][Lp] */
  void* d_s;             /* bf16 [B*nhead*To][Lp] */
  void* g_qp;            /* bf16 [nhead][B*To][H] */
  void* p_undropped;     /* fp16 [B*nhead*To][Lp]: needed when the forward ran with dropout */
} tcavt_cross_attn_bwd_args;

int tcavt_cross_attn_backward(const tcavt_cross_attn_bwd_args* args, tcavt_stream_t stream);

/* TransformerLTSF.forward (scripts/train.py:808-842), one call per phase: 1 = the LLM-independent front (token projection,
 * N-Linear encoder, SelfAttentionBlock), 2 = the head (N-Linear decoder .. cross-attention .. output head), 3 = both.
 * fp16 storage; every buffer caller-owned. */
typedef struct tcavt_ltsf_args {
  /* ---- phase 1: LLM-independent front (fp32) */
  const float* x;        /* [B][F][T] normalised input trajectories (also the head's "last position" residual) */
  const float* conv_w;   /* token_proj.weight [C][F] */
  const float* conv_b;   /* [C] */
  const float* enc_w;    /* N-Linear encoder weights, stacked [C][T][T] */
  const float* enc_b;    /* [C][T] */
  const float* pos;      /* pos_encoding [C][T] */
  float* tok;            /* [B*T][C] tokens */
  float* xp_tok;         /* optional [B*T][C]: projected input, kept for the backward */
  const float *sa_n1_w, *sa_n1_b, *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b, *sa_n2_w, *sa_n2_b, *sa_f0_w, *sa_f0_b, *sa_f3_w, *sa_f3_b;
  float *sa_xn, *sa_qkv, *sa_att, *sa_res1, *sa_rn, *sa_f; /* SelfAttentionBlock activations: [B*T][C], [.][3C], [.][C], [.][C], [.][C], [.][4C] */
  float* e;              /* [B*T][C]: output of the front (input of phase 2) */
  /* ---- phase 2: the head */
  const float* poly_emb; /* [B][poly_dim] lane-polygon embedding */
  const float *lane_w, *lane_b;  /* lane_fc [C*To][poly_dim], [C*To] */
  const float *dec_w, *dec_b;    /* N-Linear decoder, stacked [C][To][T], [C][To] */
  float *lane, *d0;              /* [B][C*To] each */
  const float *pm0_w, *pm0_b, *pm3_w, *pm3_b; /* post-MLP (post_hidden > 0): [post_hidden][C*To], [C*To][post_hidden] */
  float *hid, *d1;               /* [B][post_hidden], [B][C*To] */
  float* dec_t;                  /* [B*To][C] fp32 */
  void* dec_tb;                  /* its fp16 copy */
  const void* w_dp; const float* b_dp;  /* dec_proj fp16 [H][C] */
  void* proj;                    /* fp16 [B*To][H] */
  const void* w_q; const float* b_q;    /* cross-attention q projection fp16 [H][H] */
  tcavt_cross_attn_args xattn;   /* its q / att buffers are the stage's */
  const void* w_co; const float* b_co;  /* cross-attention out_proj fp16 [H][H] */
  void* cross;                   /* fp16 [B*To][H] */
  const void* w_un; const float* b_un;  /* dec_unproj fp16 [C][H] */
  float* fused;                  /* [B*To][C] = dec_t + unproj */
  const float *fl_n_w, *fl_n_b; float* fn;          /* fusion LayerNorm */
  const float *fl1_w, *fl1_b; float* f1;            /* fusion Linear + ReLU */
  const float *fl3_w, *fl3_b; float* f2;            /* fusion Linear */
  const float *out_w, *out_b;    /* out_proj [F][C] */
  float* out;                    /* [B][F][To] */
  int32_t B, C, T, To, F, H, nhead_sa, poly_dim, post_hidden, add_last;
  float dropout_p;               /* LTSF dropout (train mode); sites first_site + 0..3 (self-attention block), + 4 (post-MLP) */
  uint32_t first_site;
  uint64_t dropout_seed;
} tcavt_ltsf_args;

int tcavt_ltsf_forward(const tcavt_ltsf_args* args, int phase, tcavt_stream_t stream);

/* Backward of tcavt_ltsf_forward (SURVEY.md 8b "ltsf_backward"; what autograd does for TransformerLTSF in the training step of
 * scripts/train.py:1168-1183), one call per phase:
 *   1 = the head, from dL/d out down to the lane-polygon embedding's gradient g_poly (output head, fusion layer, dec_unproj,
 *       cross-attention out_proj, tcavt_cross_attn_backward, q projection, dec_proj, transpose, post-MLP, lane_fc) -- after
 *       it the caller may start the lane-polygon encoder's backward on another stream;
 *   2 = the rest: N-Linear decoder, SelfAttentionBlock, N-Linear encoder, positional term, token projection;
 *   3 = both.
 * `fwd` carries the pointers of BOTH forward phases (activations as the forward left them, the forward's weights, shapes and
 * dropout sites).  Parameter gradients are fp32 and must be zeroed by the caller (most are ADDED to); the per-channel
 * nn.Linear gradients of the two N-Linear blocks are written with a stride between channels, so that they can live inside a
 * flat gradient buffer in parameter order (weight, bias, weight, bias, ...).  Gradient-side 16-bit tensors are bf16. */
typedef struct tcavt_ltsf_bwd_args {
  const tcavt_ltsf_args* fwd;
  tcavt_cross_attn_bwd_args xattn; /* .fwd = &fwd->xattn; .g_att / .g_q are this stage's buffers; .w_in / .gw_in / .gb_in: the packed in_proj */
  const float* g_out;              /* [B][F][To]: dL/d out */
  const float *w_un, *w_co, *w_dp; /* fp32 master weights of the head's 16-bit projections: [C][H], [H][H], [H][C] */
  /* ---- parameter gradients */
  float *g_out_w, *g_out_b, *g_fl3_w, *g_fl3_b, *g_fl1_w, *g_fl1_b, *g_fl_n_w, *g_fl_n_b;
  float *g_un_w, *g_un_b, *g_co_w, *g_co_b, *g_dp_w, *g_dp_b;
  float *g_pm3_w, *g_pm3_b, *g_pm0_w, *g_pm0_b, *g_lane_w, *g_lane_b;
  float *g_dec_w, *g_dec_b;        /* decoder_linears.c.{weight [To][T], bias [To]} of channel c at + c * dec_stride */
  float *g_sa_n1_w, *g_sa_n1_b, *g_sa_in_w, *g_sa_in_b, *g_sa_out_w, *g_sa_out_b, *g_sa_n2_w, *g_sa_n2_b, *g_sa_f0_w, *g_sa_f0_b,
      *g_sa_f3_w, *g_sa_f3_b;
  float *g_enc_w, *g_enc_b;        /* encoder_linears.c.{weight [T][T], bias [T]} of channel c at + c * enc_stride */
  float* g_pos;                    /* pos_encoding gradient [C][pos_ld]; columns 0..T-1 are written */
  float *g_conv_w, *g_conv_b;      /* token_proj [C][F], [C] */
  /* ---- output */
  float* g_poly;                   /* [B][poly_dim]: dL/d lane-polygon embedding */
  /* ---- gradient workspaces (Mo = B*To, Mt = B*T, Mp = Mo rounded up to 64) */
  float *g_f2, *g_f1, *g_fn, *g_dec_t, *g_dt2; /* [Mo][C] each */
  void *g_cross, *g_proj;          /* bf16 [Mo][H] */
  float *g_d1, *g_hid, *g_d0;      /* [B][C*To], [B][post_hidden], [B][C*To] (g_hid / g_d0: with the post-MLP only) */
  float *g_dw, *g_db, *g_e;        /* [C][To][T], [C][To], [Mt][C] */
  float *g_e_d, *g_res1_d;         /* [Mt][C] each: masked copies (train mode only) */
  float *g_ff, *g_rn, *g_res1, *g_att_sa, *g_qkv, *g_xn, *g_tok; /* [Mt][4C], [Mt][C] x3, [Mt][3C], [Mt][C] x2 */
  float *g_ew, *g_eb, *g_xp;       /* [C][T][T], [C][T], [Mt][C] */
  void *s_gyb, *s_gyt, *s_xt, *s_wt; /* bf16 scratch of the 16-bit linear backwards: [Mo][H], [H][Mp], [H][Mp], [H][H] */
  int64_t dec_stride, enc_stride;
  int32_t pos_ld, reserved0;
} tcavt_ltsf_bwd_args;

int tcavt_ltsf_backward(const tcavt_ltsf_bwd_args* args, int phase, tcavt_stream_t stream);

/* SUM all-reduce, in place, of a flat fp32 buffer on the caller's RCCL communicator (`nccl_comm` is an ncclComm_t) and
 * stream: one gradient bucket of the data-parallel step (the DistributedDataParallel wrap of scripts/train.py:1127 does
 * this during backward; tcavt_amd.training.Trainer issues the same exchange through torch.distributed).  The mean is taken
 * afterwards (tcavt_clip_grad_norm / tcavt_adamw grad_scale = 1 / world).  librccl is opened on first use. */
int tcavt_allreduce_flat(float* buf, int64_t n, void* nccl_comm, tcavt_stream_t stream);

/* LoRA down-projection of both adapters in one pass over x16 [M][H] (the 16-bit residual-stream copy):
 *   t[m][0:16] = scale * (mask_q o x16[m]) . a_cat[0:16]^T,   t[m][16:32] = scale * (mask_v o x16[m]) . a_cat[16:32]^T
 * t is 16-bit [M][64] (columns >= 32 untouched).  dropout_p > 0: PEFT's per-adapter lora_dropout (scripts/train.py:433-439)
 * with the Philox masks of (dropout_seed, site_q / site_v, element m * H + k), the masked operand rounded as tcavt_dropout
 * rounds it; dropout_p == 0: no masks. */
int tcavt_lora_down(const void* x16, const void* a_cat, void* t, int M, int H, float scale, float dropout_p,
                    uint64_t dropout_seed, uint32_t site_q, uint32_t site_v, int dtype16, tcavt_stream_t stream);

/* Number of partial sums of squares per row that a TCAVT_EPI_NORM_OUT product [M][N] over K writes (N / 64, or N / 16 in the
 * skinny form tcavt_gemm_bf16 selects for M <= 32, K % 256 == 0) -- the `npart` to give tcavt_embed_fuse / tcavt_rownorm_prep
 * so that the first fused norm of tcavt_llama_stack_forward / tcavt_llama_decode_step reads what it expects (N = H, K = I) */
int tcavt_norm_npart(int M, int N, int K);

/* x16 = 16-bit copy of x fp32 [M][H], part[M][npart] = (sum of squares of the row, 0, 0, ...): the h16 / part inputs of
 * tcavt_llama_stack_forward for embeddings that do not come from tcavt_embed_fuse (HF-style inputs_embeds call) */
/* rounded_sums != 0: the sums are those of the rounded 16-bit values (the 16-bit residual stream, h == NULL) */
/* stream_scale (0 means 1): x16 = round(stream_scale * x), sums of the scaled values (tcavt_llama_stack_args.stream_scale) */
int tcavt_rownorm_prep(const float* x, void* x16, float* part, int64_t M, int H, int npart, int dtype16, int rounded_sums,
                       float stream_scale, tcavt_stream_t stream);

/* RMSNorm of 16-bit rows x16 [M][H] (fp32 arithmetic; H % 256 == 0): out16 and / or out_f32 -- the final norm of the
 * 16-bit residual stream (HF modeling_llama.py:62-67 on the stream's stored values) */
int tcavt_rmsnorm16(const void* x16, const float* gamma, float eps, void* out16, float* out_f32, int M, int H, int dtype16,
                    tcavt_stream_t stream);

/* ========================================================================
 * Autoregressive text generation (LlamaMultiModal.generate_batch, scripts/train.py:577-654;
 * scripts/check_generation.py:152-222).  Prefill = tcavt_llama_stack_forward with k_cache / v_cache set;
 * tcavt_gather_last picks every sample's last valid hidden state; one tcavt_llama_decode_step per further token;
 * tcavt_sample_logits turns logits into the next token and advances the device-resident state, so that a decode
 * step is a fixed launch sequence (hipGraph-capturable, BASELINE.json configs[4]).
 * ====================================================================== */
typedef struct tcavt_sample_params {
  float temperature;          /* TemperatureLogitsWarper (reference: 0.9); sampling only */
  float top_p;                /* TopPLogitsWarper (0.9); sampling only; 1 disables */
  float repetition_penalty;   /* RepetitionPenaltyLogitsProcessor (1.2); 1 disables */
  int32_t top_k;              /* TopKLogitsWarper (40); sampling only; 1 .. 256 */
  int32_t no_repeat_ngram_size; /* NoRepeatNGramLogitsProcessor (3); 0 disables */
  int32_t do_sample;          /* 0: greedy arg-max of the processed scores (first maximum), bit-reproducible */
  int64_t eos_token_id;       /* < 0: none */
  int64_t pad_token_id;       /* emitted for samples that have finished */
  uint64_t seed;              /* Philox key of the sampling draws (counter = step, sample) */
} tcavt_sample_params;

/* logits fp32 [B][V] (modified in place); history int64 [B][hist_cap] + hist_len int32 [B]: the tokens the repetition
 * penalty and the n-gram ban look at (prompt ids + generated); *step (device int32): index of the token being chosen --
 * out_tokens[b][*step] receives it, cur_tok[b] too, the history grows by it, pos[b] += 1 when advance_pos != 0,
 * finished[b] is set on EOS, and *step is incremented at the end. */
/* workspace (optional; tcavt_sample_workspace_bytes(B) bytes, 16-byte aligned, its first 64 + 4 B bytes ZEROED ONCE by the caller:
 * every call leaves them zero): the two-stage form -- B x 16 workgroups process and pre-select slices of the vocabulary that
 * they hold in registers, one workgroup per sample merges, ranks and draws; *step is advanced by the same launch.  The token
 * selected is the one the one-stage form (workspace NULL: one workgroup per sample, three passes over the row) selects.
 * Calls that share a workspace must be stream-ordered. */
int64_t tcavt_sample_workspace_bytes(int B);
int tcavt_sample_logits(float* logits, int B, int V, int64_t* history, int hist_cap, int32_t* hist_len,
                        const tcavt_sample_params* params, int32_t* step, int64_t* cur_tok, int32_t* pos,
                        int32_t* finished, int64_t* out_tokens, int out_cap, int advance_pos, void* workspace,
                        int64_t workspace_bytes, tcavt_stream_t stream);

/* out16[b] = src16[b * L + kv_len[b] - 1]: 16-bit rows of width H */
int tcavt_gather_last(const void* src16, const int32_t* kv_len, void* out16, int B, int L, int H, tcavt_stream_t stream);

typedef struct tcavt_decode_args {
  const tcavt_llama_layer* layers; /* HOST array, as for tcavt_llama_stack_forward (tape fields unused) */
  const float* gamma_final;
  const float* rope_cos;           /* fp32 [rope_L][32], rope_L >= kv_lmax */
  const float* rope_sin;
  const void* table;               /* 16-bit [V][H]: embedding table = lm_head (tied) */
  const float* txt_mod;            /* fp32 [H]: text_modality_embedding (generated tokens are text tokens, train.py:526-527) */
  const int64_t* cur_tok;          /* int64 [B]: the token fed in this step */
  const int32_t* pos;              /* int32 [B]: its position = number of keys already in the cache */
  float* h;                        /* fp32 [B][H] workspace, or NULL: 16-bit residual stream (as tcavt_llama_stack_args.h) */
  void* h16;                       /* 16-bit [B][H] */
  float* part;                     /* fp32 [B][H / 16] */
  void* qkv;                       /* 16-bit [B][(nq + 2 nkv) * 64] */
  void* att;                       /* 16-bit [B][nq * 64] */
  void* act;                       /* 16-bit [B][I] */
  void* t;                         /* 16-bit [B][64] (LoRA; zero-initialised once) */
  void* k_cache;                   /* 16-bit [n_layers][B][kv_lmax][nkv * 64], filled up to pos[b] by the prefill / earlier steps */
  void* v_cache;
  void* x16;                       /* 16-bit [B][H]: final-norm output */
  float* logits;                   /* fp32 [B][V] */
  int32_t* bad_id_flag;
  int32_t n_layers, B, H, I, nq, nkv, V, dtype16, kv_lmax, rope_L;
  float rms_eps, lora_scale;
  int32_t* nonfinite_flag;         /* optional, as tcavt_llama_stack_args.nonfinite_flag */
  void* splitk_ws;                 /* optional: tcavt_gemm_args.splitk_ws for the step's projections (tickets zeroed once) */
  int64_t splitk_ws_bytes;
  void* lora_part;                 /* optional, B * H floats: layers 1.. take their LoRA down-projection from partial sums the
                                      previous layer's down-projection GEMM leaves here (tcavt_gemm_args.lora_part: one launch
                                      per layer less); adapters of rank <= 8 only (lora_rank), B <= 32, NULL = a launch per layer */
  int32_t lora_rank;
  float stream_scale;              /* as tcavt_llama_stack_args.stream_scale (0 means 1) */
  int32_t w_layout;                /* 0, or TCAVT_W_FRAG16: layers[].w_qkv / w_o / w_gu / w_d point to tcavt_pack_weight16 copies */
  int32_t act_layout;              /* 0; 1: h16, att, act and x16 are kept in fragment-major order (tcavt_gemm_args.act_layout; each
                                      buffer then holds 16 (B <= 16) or 32 whole rows, B <= 32); 2: the same in ONE block of 8
                                      tokens (TCAVT_ACT_BLOCK8; buffers of 8 whole rows, B <= 8).  Needs h == NULL */
  const void* table_packed;        /* optional: tcavt_pack_weight16 copy of `table` for the lm_head product (the token lookup
                                      keeps reading `table`) */
} tcavt_decode_args;

int tcavt_llama_decode_step(const tcavt_decode_args* args, tcavt_stream_t stream);

/* Fragment-major copy of a 16-bit weight matrix W [N][ldw] (N % 16 == 0, K % 32 == 0) for the skinny form of tcavt_gemm_bf16
 * (tcavt_gemm_args.w_layout = TCAVT_W_FRAG16): out holds N * K elements; block b (rows 16 b .. 16 b + 15), k-step j (columns
 * 32 j .. 32 j + 31) is the 1 KiB chunk (b * K / 32 + j), and inside it lane l = 16 q + r of the consuming wave finds its eight
 * elements W[16 b + r][32 j + 8 q .. + 7] at byte 16 l -- the operand layout of v_mfma_f32_16x16x32, so that every load
 * instruction of the weight stream reads consecutive bytes.  The Hugging Face generate() of the reference (train.py:628-643)
 * has no counterpart: this is a layout of the frozen weights made once per checkpoint. */
#define TCAVT_W_FRAG16 1
#define TCAVT_ACT_A_FRAG16 1
#define TCAVT_ACT_OUT_FRAG16 2
#define TCAVT_ACT_BLOCK8 4
int tcavt_pack_weight16(const void* W, int64_t ldw, void* out, int N, int K, tcavt_stream_t stream);

/* hipEvent helpers for tcavt_llama_stack_args.events (timing enabled); elapsed time in milliseconds between two
 * recorded events after the stream has been synchronised by the caller */
int tcavt_events_create(void** events, int n);
int tcavt_events_destroy(void** events, int n);
int tcavt_event_elapsed_ms(void* start, void* stop, float* ms);

/* tcavt_adamw gated on a finite loss, as the LoRA-trainable loop does (modify_scripts/modify_train.py:1190-1196:
 * `if torch.isfinite(loss): clip; optimizer.step()  else: skip`), decided on the device: the update is skipped when
 * *loss (device fp32 scalar) is not finite -- or, if grad_norm != NULL (e.g. scratch + 1025 of tcavt_clip_grad_norm),
 * when the exchanged gradient's norm is not finite, which keeps data-parallel ranks in step where the reference's
 * rank-local test would not.  The optimizer's step count lives on the device: ctl int32[8], zero-initialised by the
 * caller once; ctl[0] = applied updates so far, ctl[1] = skipped updates, ctl[2] = 1 if this call was skipped, ctl[6] = the
 * fp16 backward's scale back-off (+4 per skipped update, capped at 24; -1 per 256 applied updates: tcavt_grad_scale_pick). */
int tcavt_adamw_gated(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float grad_scale, const float* loss, const float* grad_norm,
                      int32_t* ctl, tcavt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TCAVT_H */
