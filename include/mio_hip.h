/*
 * mio_hip.h -- C ABI of libmio_hip.so, the MI355X (gfx950 / CDNA4) implementation of the
 * transformer-inference hot path of aslitaser/ml-inference-optimizer.
 *
 * Every entry point replaces one Python->Triton launch boundary of the reference (cited per
 * function as reference file:line).  Conventions:
 *   - plain C: pointers, sizes, strides; no C++ or torch types.
 *   - every function returns 0 on success, <0 on error; mio_last_error() gives the message
 *     (thread-local).  Nothing is allocated, no ownership changes hands: outputs and workspaces
 *     are caller-owned device buffers.
 *   - asynchronous on the given HIP stream (`stream` is a hipStream_t passed as void*); safe to
 *     call concurrently from several threads/streams; graph-capturable (no sync, no malloc).
 *   - strides are in ELEMENTS of the tensor's dtype.  The innermost (head_dim / feature)
 *     dimension must be contiguous (stride 1) and every row start must be 16-byte aligned.
 */
#ifndef MIO_HIP_H
#define MIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIO_VERSION 105 /* 0.1.0 */

typedef enum { MIO_BF16 = 0, MIO_FP16 = 1 } mio_dtype_t;

typedef enum {
  MIO_ACT_NONE = 0,
  MIO_ACT_GELU_TANH = 1, /* kernels/triton/mlp_kernels.py:144-161, kernels/mlp/fused_mlp.py:227-231 */
  MIO_ACT_GELU_ERF = 2,  /* kernels/triton/mlp_kernels.py:782-783, kernels/mlp/fused_mlp.py:162-163 */
  MIO_ACT_RELU = 3,      /* kernels/triton/mlp_kernels.py:233-414 */
  MIO_ACT_SILU = 4,      /* kernels/mlp/fused_mlp.py:166-167 */
  MIO_ACT_SWIGLU = 5     /* kernels/triton/mlp_kernels.py:417-641 */
} mio_act_t;

typedef enum {
  MIO_MASK_NONE = 0,
  MIO_MASK_KEEP_U8 = 1, /* 1 = attend, 0 -> score := -1e9  (flash_attention_kernels.py:257-273) */
  MIO_MASK_ADD_F32 = 2  /* score += mask                   (attention_kernels.py:1565-1566)      */
} mio_mask_kind_t;

int mio_version(void);
const char* mio_last_error(void);

/* ------------------------------------------------------------------------------------------
 * FlashAttention-3 style tiled attention forward (prefill), online softmax, fp32 accumulators.
 * Replaces: triton_flash_attention -> _flash_attention_forward_kernel launch
 *           (kernels/triton/flash_attention_kernels.py:1291-1310; kernel :38-325) and
 *           triton_ring_attention_forward -> _ring_attention_forward_kernel launch
 *           (kernels/triton/attention_kernels.py:979-995; kernel :35-202).
 *
 * q [B,Sq,H,D], k/v [B,Sk,Hkv,D], o [B,Sq,H,D] given by (b, s, h) strides (d stride is 1), so
 * both the reference's seq-major [B,S,H,D] (flash) and head-major [B,H,S,D] (ring) layouts are
 * accepted.  H % Hkv == 0 (GQA: kv head = h / (H/Hkv), flash_attention.py:894-912).
 * D in [8,128], D % 8 == 0.
 *
 * causal: key position (k_offset + j) > query position (q_offset + i) is excluded.  Without a
 *   user mask excluded blocks are skipped (exact: the reference's -1e9 fill underflows to 0).
 * mask: kind KEEP_U8 (uint8) or ADD_F32 (float), addressed with 4 element strides
 *   (b, h, q, k); 0 strides broadcast.
 * lse (nullable): [B,H,Sq] fp32, natural-log softmax denominator (m + log l of
 *   flash_attention_kernels.py:308-325).  Rows with no visible key get lse=-inf, o=0.
 * Ring carry (nullable o_acc): fp32 [B,Sq,H,D] contiguous running output state.
 *   carry_in != 0: start from (o_acc, lse) instead of empty;  o_acc != NULL: the normalised
 *   fp32 output is also written to o_acc (and lse must be non-NULL).  `o` may be NULL when
 *   o_acc is given (intermediate ring steps).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  float* o_acc;
  const void* mask;
  int64_t q_stride[3]; /* b, s, h */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int64_t o_stride[3];
  int64_t mask_stride[4]; /* b, h, q, k */
  int32_t B, Sq, Sk, H, Hkv, D;
  int32_t dtype;     /* mio_dtype_t */
  int32_t causal;    /* 0/1 */
  int32_t mask_kind; /* mio_mask_kind_t */
  int32_t carry_in;  /* 0/1 */
  int32_t q_offset, k_offset;
  float softmax_scale; /* > 0 */
  int32_t k_prescaled; /* 0/1: k already holds K * softmax_scale * log2(e), scaled in fp32 BEFORE its rounding to 16 bits
                          (mio_gemm_bias_act_bw col_scale: the epilogue of the projection that produced it).  Only for
                          launches mio_fa3_k_prescaled_ok() accepts; the kernel then skips the per-score multiply. */
  int32_t o_blocked;   /* 0/1: `o` is the [B*Sq, H*D] output in the blocked ACTIVATION layout of the GEMM entry points below
                          (ceil(B*Sq/256)*256 x H*D elements, o_stride ignored): the attention epilogue hands the output
                          projection contiguous K-tiles.  Only for launches mio_fa3_o_blocked_ok() accepts. */
} mio_fa3_fwd_params_t;

int mio_fa3_fwd(const mio_fa3_fwd_params_t* p, void* stream);
/* 1 iff a launch with these parameters (k_prescaled ignored) may set k_prescaled = 1: no user mask, Sq > 128, K / V rows
 * within 4 GiB of their (batch, head) base, head dim <= 96; with the (o_acc, lse) ring carry (o_acc and lse given,
 * carry_in 0 / 1, o optional) only at head dim <= 64. */
int32_t mio_fa3_k_prescaled_ok(const mio_fa3_fwd_params_t* p);
/* 1 iff a launch with these parameters (o_blocked ignored) may set o_blocked = 1: a k_prescaled launch without the ring
 * carry at head dim <= 64 with (H * D) % 32 == 0. */
int32_t mio_fa3_o_blocked_ok(const mio_fa3_fwd_params_t* p);

/* Merge two normalised partial attention states over disjoint key sets (ring / split-KV):
 * o = w_a*o_a + w_b*o_b, lse = logaddexp(lse_a, lse_b), w_x = exp(lse_x - lse).
 * Restates the (alpha, beta) update of kernels/triton/attention_kernels.py:1573-1585.
 * o_* fp32 [rows, D] contiguous with rows = B*Sq*H laid out [B,Sq,H]; lse_* fp32 [B,H,Sq].
 * Result in (o_a, lse_a); if o_out != NULL the merged output is also written there in `dtype`. */
int mio_attn_merge(float* o_a, float* lse_a, const float* o_b, const float* lse_b, void* o_out,
                   int32_t B, int32_t Sq, int32_t H, int32_t D, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * y[M,N] = epilogue( x[M,K] @ w[N,K]^T ), bf16/fp16 in, fp32 accumulate on MFMA.
 *   act != SWIGLU:  y = act(x w^T + bias) (+ residual)
 *   act == SWIGLU:  y = silu(x w_gate^T + bias_gate) * (x w^T + bias)
 * This is the GEMM under both FusedMLP stages and the q/k/v/o projections
 * (F.linear call sites: kernels/mlp/fused_mlp.py:159,176,225,236,265-266,274;
 *  kernels/attention/flash_attention.py:629-631,654-657).
 * ldx/ldw/ldy/ldr: row strides in elements (multiples of 8); K % 8 == 0; bias/residual nullable.
 * ------------------------------------------------------------------------------------------ */
int mio_gemm_bias_act(const void* x, const void* w, const void* bias, const void* w_gate,
                      const void* bias_gate, const void* residual, void* y, int64_t M, int32_t N,
                      int32_t K, int64_t ldx, int64_t ldw, int64_t ldy, int64_t ldr, int32_t act,
                      int32_t dtype, void* stream);

/* FusedMLP forward: y = fc2(act(fc1(x))) (+ residual).
 * Replaces: triton_fused_mlp -> _fused_mlp_{gelu,relu,swiglu}_kernel launch
 *           (kernels/triton/mlp_kernels.py:648-756, launch :711) and FusedMLP._forward_triton's
 *           fused_mlp_forward call (kernels/mlp/fused_mlp.py:131-141).
 * x [M,d], w1/wg [I,d], w2 [d,I], biases nullable; workspace >= mio_fused_mlp_workspace_bytes()
 * holds the bf16/fp16 [M,I] activation (written once by stage 1's epilogue, read once by stage 2; rows rounded up
 * to whole 256-row blocks: at sizes where both GEMMs run the 256x256-tile kernels the library keeps it in a blocked
 * layout of contiguous 16 KiB K-tiles). */
size_t mio_fused_mlp_workspace_bytes(int64_t M, int32_t d, int32_t I, int32_t act);
int mio_fused_mlp_fwd(const void* x, const void* w1, const void* b1, const void* wg, const void* bg,
                      const void* w2, const void* b2, const void* residual, void* y, void* workspace,
                      int64_t M, int32_t d, int32_t I, int32_t act, int32_t dtype, void* stream);

/* Blocked weights: a one-time repack of an nn.Linear weight [N, K] (K % 32 == 0) so that every (256-row, 32-column)
 * K-tile the 256x256-tile GEMM kernels fetch is one contiguous 16 KiB block instead of 256 pieces of 64 B at a stride of
 * 2 K bytes:   wb[((n / 256) * (K / 32) + k / 32) * 256 + n % 256][k % 32],  rows padded with zeros to a multiple of 256
 * (mio_weight_blocked_bytes).  The reference has no counterpart (its weights stay nn.Linear tensors read by tl.load,
 * kernels/triton/mlp_kernels.py:91-126); the plain-weight entry points above remain the drop-in boundary.
 * *_ok() == 0: the shape does not take those kernels -- keep the plain weight and call the plain entry point
 * (the *_bw entry points return an error then). */
size_t mio_weight_blocked_bytes(int32_t N, int32_t K);
int mio_weight_block(const void* w, int64_t ldw, void* wb, int32_t N, int32_t K, int32_t dtype, void* stream);
int32_t mio_gemm_blocked_weight_ok(int64_t M, int32_t N, int32_t K, int32_t act);
int mio_gemm_bias_act_bw(const void* x, const void* wb, const void* bias, const void* residual, void* y, int64_t M,
                         int32_t N, int32_t K, int64_t ldx, int64_t ldy, int64_t ldr, int32_t act, int32_t dtype,
                         int32_t x_blocked, void* stream);
/* The same (no residual) with output columns [cs_lo, cs_hi) multiplied by cs_val in fp32 after bias / activation and before
 * the rounding to 16 bits (cs_lo, cs_hi multiples of 128).  The fused q/k/v projection uses it to hand mio_fa3_fwd a K that
 * already carries softmax_scale * log2(e) with a single rounding (k_prescaled).  Only where mio_gemm_col_scale_ok() != 0
 * (the persistent 256x256-tile kernel takes the launch). */
int32_t mio_gemm_col_scale_ok(int64_t M, int32_t N, int32_t K, int32_t act);
int mio_gemm_bias_act_bw_cs(const void* x, const void* wb, const void* bias, void* y, int64_t M, int32_t N, int32_t K,
                            int64_t ldx, int64_t ldy, int32_t act, int32_t dtype, int32_t x_blocked, int32_t cs_lo,
                            int32_t cs_hi, float cs_val, void* stream);
int32_t mio_fused_mlp_blocked_weight_ok(int64_t M, int32_t d, int32_t I, int32_t act);
int mio_fused_mlp_fwd_bw(const void* x, const void* w1b, const void* b1, const void* w2b, const void* b2,
                         const void* residual, void* y, void* workspace, int64_t M, int32_t d, int32_t I, int32_t act,
                         int32_t dtype, int32_t x_blocked, void* stream);
/* SwiGLU on the 256x256-tile kernels (reference kernels/triton/mlp_kernels.py:417-641 _fused_mlp_swiglu_kernel;
 * kernels/mlp/fused_mlp.py:262-275): the gate and up weights [I, K] are repacked ONCE into one blocked weight whose 256-row
 * tiles interleave, per 64-row wave slice, 32 gate rows and the 32 up rows of the same output columns
 *   row (tn * 256 + wn * 64 + h * 32 + j)  <-  (h ? w_up : w_gate)[tn * 128 + wn * 32 + j],   rows padded with zeros,
 * so that silu(gate) * up is local to a lane of the accumulator and stage 1 writes act once.  Use where
 * mio_fused_mlp_blocked_weight_ok(M, d, I, MIO_ACT_SWIGLU) != 0; w2b is mio_weight_block(w2). */
size_t mio_weight_blocked_glu_bytes(int32_t I, int32_t K);
int mio_weight_block_glu(const void* w_gate, const void* w_up, int64_t ldw, void* wb, int32_t I, int32_t K, int32_t dtype,
                         void* stream);
int mio_fused_mlp_glu_fwd_bw(const void* x, const void* wgu_b, const void* b_up, const void* b_gate, const void* w2b,
                             const void* b2, const void* residual, void* y, void* workspace, int64_t M, int32_t d, int32_t I,
                             int32_t dtype, int32_t x_blocked, void* stream);
/* x_blocked != 0: the activation operand x is in the same blocked layout (m in the place of n; ceil(M/256)*256 x K
 * elements, ldx ignored) -- what mio_layernorm_fwd_bx writes, so that LayerNorm -> GEMM hands over contiguous K-tiles. */
int mio_layernorm_fwd_bx(const void* x, const void* residual, const void* weight, const void* bias, void* yb,
                         void* sum_out, int64_t rows, int32_t cols, float eps, float alpha, int32_t dtype, void* stream);

/* LayerNorm folded into the GEMMs on either side of it.  Replaces the reference's LayerNorm -> QKV prologue
 * fusion (kernels/triton/fused_layernorm_qkv.py:37-420, triton_fused_layernorm_qkv) and its residual + LayerNorm pass
 * (kernels/triton/layernorm_kernels.py:35-188) between two GEMMs of a transformer block:
 *   producer = the GEMM that writes the residual stream (out-proj / fc2 with the residual epilogue): stats_out != NULL makes it
 *     also write, per output row and 256-column tile, (sum, sum of squares) of the ROUNDED row: [N / 256][ceil(M/256)*256][2]
 *     fp32 (mio_ln_stats_bytes(M, N)); deterministic (fixed summation order, no atomics);
 *   consumer = the projection behind the LayerNorm: ln_stats (a producer's stats_out of width K) != NULL: x is the raw stream,
 *     wb = mio_weight_block of the gamma-scaled, row-centred weight and bias the beta-folded bias (both from mio_ln_fold_weight:
 *     (x - mean 1) . w = x . (w - mean(w) 1), so centring the weight rows makes the plain product the centred one); the
 *     read-out computes rstd * acc + bias, then act / column scale.  Rounding: mio_ln_fold_weight chooses the rounding direction
 *     of a few elements per row so that the 16-bit row sums to zero within an ulp or two (plain rounding would leave ~sqrt(K/12)
 *     ulp, and the product would carry mean(x) times that); measured error equals the LayerNorm kernel + GEMM's (2.3e-3 bf16)
 *     for streams whose row mean is up to 4x their deviation.
 * flags: the operands in the blocked activation layout ((256-row, 32-column) blocks of 16 KiB, rows padded to 256; ld* ignored
 * for a blocked operand): x (as mio_gemm_bias_act_bw's x_blocked), y (what the next GEMM takes as blocked x), residual.
 * Shapes: mio_gemm_ln_ok(M, N, K, act, fold_in, stats_out) != 0 (blocked-weight shapes; fold_in: K % 256 == 0, act none /
 * gelu_tanh / swiglu; stats_out: N % 256 == 0, act none).  Without ln_stats and stats_out it is
 * mio_gemm_bias_act_bw (+ column scale) with blocked y / residual.  act == MIO_ACT_SWIGLU: wb is mio_weight_block_glu of the two
 * folded weights (gate, up), bias the up bias, bias_gate the gate bias, N the number of OUTPUT columns (I). */
#define MIO_GEMM_X_BLOCKED 1
#define MIO_GEMM_Y_BLOCKED 2
#define MIO_GEMM_RES_BLOCKED 4
size_t mio_ln_stats_bytes(int64_t M, int32_t width);
int32_t mio_gemm_ln_ok(int64_t M, int32_t N, int32_t K, int32_t act, int32_t fold_in, int32_t stats_out);
/* w [N, K] (row stride ldw), gamma / beta [K] (beta nullable), bias [N] (nullable), all in dtype; w_scaled [N, K] contiguous
 * = w * gamma - mean_k(w * gamma) (rounded once), bias_out [N] = bias + w beta.  One-time weight preparation. */
int mio_ln_fold_weight(const void* w, int64_t ldw, const void* gamma, const void* beta, const void* bias, void* w_scaled,
                       void* bias_out, int32_t N, int32_t K, int32_t dtype, void* stream);
int mio_gemm_ln_bw(const void* x, const void* wb, const void* bias, const void* bias_gate, const void* residual, void* y, int64_t M, int32_t N, int32_t K,
                   int64_t ldx, int64_t ldy, int64_t ldr, int32_t act, int32_t dtype, int32_t flags, const float* ln_stats,
                   int32_t ln_slots, float ln_eps, float* stats_out, int32_t cs_lo, int32_t cs_hi, float cs_val, void* stream);
/* ln_slots: statistic slots in ln_stats (0: K / 256, what a producer of width K writes); at most 8.  A wider stream (K > 2048)
 * goes through mio_ln_stats_reduce first: stats_out[s'] = sum of slots_in / slots_out consecutive slots, same row padding. */
int mio_ln_stats_reduce(const float* stats_in, int32_t slots_in, float* stats_out, int32_t slots_out, int64_t M, void* stream);

/* LayerNorm / residual+LayerNorm rows (the step either side of attention):
 * sum = x + alpha*residual (if residual), y = (sum-mean)/sqrt(var+eps)*weight + bias.
 * Replaces triton_layernorm -> _layernorm_fwd_kernel / _layernorm_residual_fwd_kernel
 * (kernels/triton/layernorm_kernels.py:191-276; kernels :35-188).  sum_out nullable. */
int mio_layernorm_fwd(const void* x, const void* residual, const void* weight, const void* bias,
                      void* y, void* sum_out, int64_t rows, int32_t cols, float eps, float alpha,
                      int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Paged-KV decode attention.  Replaces triton_paged_attention_forward ->
 * _paged_attention_fwd_kernel (kernels/triton/attention_kernels.py:1206-1311; kernel :628-808).
 * q/o [B,H,q_len,D] (strides b,h,s; d contiguous); caches [num_blocks, L, block_size, Hkv, D]
 * contiguous; block_tables [B,max_blocks] int32; context_lengths [B] int32.  No causal mask
 * (the reference's is commented out, :774-776).  workspace: mio_fa3_decode_workspace_bytes().
 * ------------------------------------------------------------------------------------------ */
size_t mio_fa3_decode_workspace_bytes(int32_t B, int32_t H, int32_t q_len, int32_t D, int32_t max_ctx);
int mio_fa3_decode_paged(const void* q, void* o, const void* k_cache, const void* v_cache,
                         const int32_t* block_tables, const int32_t* context_lengths,
                         const int64_t q_stride[3], const int64_t o_stride[3], int32_t B, int32_t H,
                         int32_t Hkv, int32_t q_len, int32_t D, int32_t num_layers, int32_t layer_idx,
                         int32_t block_size, int32_t max_blocks_per_seq, int32_t max_ctx, float scale,
                         int32_t dtype, void* workspace, void* stream);

/* Scatter the current token's K/V into the paged cache at position context_len-1.
 * Replaces triton_reshape_and_cache -> _reshape_and_cache_kernel
 * (kernels/triton/attention_kernels.py:1314-1407; kernel :811-905).  key/value [B,1,Hkv,D]. */
int mio_reshape_and_cache(const void* key, const void* value, void* k_cache, void* v_cache,
                          const int32_t* block_tables, const int32_t* context_lengths,
                          const int64_t k_stride[2], const int64_t v_stride[2], /* b, h */
                          int32_t B, int32_t Hkv, int32_t D, int32_t num_layers, int32_t layer_idx,
                          int32_t block_size, int32_t max_blocks_per_seq, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIO_HIP_H */
