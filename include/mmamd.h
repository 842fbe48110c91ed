/*
 * mmamd.h — C-ABI of libmmamd.so: the MI355X (gfx950) kernels behind the dual-encoder contrastive
 * forward + loss path of facebookresearch/multimodal (TorchMultimodal).
 *
 * The reference has no FFI of its own (it is pure Python over ATen); the boundary it
 * exposes is the nn.Module API.  Each entry point below therefore replaces one group of ATen calls
 * that the reference's Python issues on this path; the citation after each prototype names the
 * reference call site (paths relative to the reference checkout).  The host-side mirror of the
 * nn.Module API that binds these symbols with ctypes lives in multimodal_amd/ (see INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; nothing is copied or
 *     synchronised by the library; all work is enqueued on `stream` (a hipStream_t, 0 = null stream)
 *   - no allocation, no host sync: every entry point is legal inside hipStreamBeginCapture
 *   - return value: 0 on success; >0 = hipError_t of the failed launch; <0 = MMAMD_E_* argument error.
 *     mmamd_last_error() returns a thread-local description of the last non-zero return.
 *   - row-major everywhere; "ld*" are leading dimensions in ELEMENTS
 *   - dtype codes: MMAMD_F32 = 0, MMAMD_BF16 = 1
 */
#ifndef MMAMD_H_
#define MMAMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMAMD_ABI_VERSION 1

#define MMAMD_F32 0
#define MMAMD_BF16 1

#define MMAMD_ACT_NONE 0
#define MMAMD_ACT_QUICKGELU 1 /* x * sigmoid(1.702 x): modules/layers/activation.py:24-25 */
#define MMAMD_ACT_GELU_ERF 2  /* nn.GELU (FLAVA: models/flava/model.py:79) */
/* backward of the MLP: C = (A.W^T) * act'(residual) with `residual` = the saved bf16 pre-activation (bf16 output only) */
#define MMAMD_ACT_MUL_QUICKGELU_GRAD 3
#define MMAMD_ACT_MUL_GELU_GRAD 4

#define MMAMD_E_BADARG (-1)
#define MMAMD_E_UNSUPPORTED (-2)
#define MMAMD_E_ALIGN (-3)

#define MMAMD_REDUCE_MEAN 0
#define MMAMD_REDUCE_SUM 1

typedef void* mmamd_stream_t; /* hipStream_t */

int mmamd_abi_version(void);
const char* mmamd_last_error(void);
/* Reset the calling thread's sticky HIP error (returns the value it held).  The entry points report launch failures through
 * hipGetLastError(), which also returns statuses left behind by unrelated runtime calls of the same thread (an event query's
 * hipErrorNotReady, a device probe's hipErrorNoDevice): bindings call this immediately before an entry point. */
int mmamd_clear_last_hip_error(void);

/* Bench / experiment hooks (kernel-variant selectors, ablation switches, CU-mask streams, stream timers) are NOT part of this surface:
 * include/mmamd_debug.h declares them (same library, same ABI version). */


/* --- K2: row LayerNorm ----------------------------------------------------------------------
 * y[r,:] = (x[r,:]-mean)/sqrt(var+eps)*gamma+beta, statistics in fp32 (biased variance).
 * Replaces nn.LayerNorm norm1/norm2 inside torch's TransformerEncoderLayer created at
 * models/clip/image_encoder.py:65-73 / text_encoder.py:58-65, and Fp32LayerNorm
 * (modules/layers/normalizations.py:13-25) at image_encoder.py:106, text_encoder.py:125. */
int mmamd_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                    int y_dtype, int rows, int d, float eps, mmamd_stream_t stream);

/* The LayerNorms of BOTH towers of a dual-encoder layer in ONE launch, optionally with the residual add in front of them: per problem
 *   x[r,:] += delta[r,:]   (delta bf16 [rows,d] or NULL; x fp32 [rows,d], updated in place only when delta is given)
 *   y[r,:]  = LayerNorm(x[r,:]; gamma, beta, eps)   (bf16 [rows,d]; y NULL = add only)
 * With delta = the bf16 output of the out-projection / MLP-down GEMM (bias included) this is `x = x + sa_block(norm1(x))` followed by
 * `norm2(x)` of torch's pre-norm TransformerEncoderLayer (models/clip/image_encoder.py:65-73, text_encoder.py:58-65) with the fp32
 * read-modify-write of the residual stream taken out of the GEMM epilogue. */
typedef struct {
  float* x;
  const void* delta;
  const float* gamma;
  const float* beta;
  void* y;
  int rows, d;
  float eps;
} mmamd_ln_problem;
int mmamd_add_layernorm_grouped(const mmamd_ln_problem* probs, int nprob, mmamd_stream_t stream);

/* --- K3/K5/K6: C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]) -------------------
 * A, W bf16; fp32 accumulate on MFMA; bias fp32 or NULL; residual (dtype = out_dtype) or NULL, may
 * alias C.  Requires K % 64 == 0, lda/ldw % 8 == 0, ldc/ldr % 4 == 0, 16-byte aligned bases.
 * Replaces F.linear in-projection (torch functional `_in_projection_packed`), out_proj + residual,
 * linear1 + SiLU, linear2 + residual of the encoder layers (call sites image_encoder.py:108,
 * text_encoder.py:121), and the patch-embedding conv as a GEMM (image_encoder.py:91). */
int mmamd_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias,
                    const void* residual, int ldr, void* C, int ldc, int out_dtype, int M, int N,
                    int K, int act, mmamd_stream_t stream);

/* GROUPED form of mmamd_gemm_bf16: up to two problems that share out_dtype and the activation but not the shapes — the same projection
 * of the two towers of the dual encoder (models/clip/model.py:63-75 runs encoder_a then encoder_b; the layers of image_encoder.py:108
 * and text_encoder.py:121 are independent until the loss) — in ONE persistent launch whose workgroups walk the concatenated tile list.
 * Each problem: C = act(A W^T + bias) (+ R, dtype = out_dtype, may alias C).  Results are bit-identical to one mmamd_gemm_bf16 call per
 * problem; problems the persistent kernel cannot take (K % 128 != 0, too few tiles, a non-default GEMM variant) run as exactly those calls. */
typedef struct {
  const void* A;      /* bf16 [M, lda] */
  const void* W;      /* bf16 [N, ldw] */
  const float* bias;  /* fp32 [N] or NULL */
  const void* R;      /* residual [M, ldr] or NULL */
  void* C;            /* [M, ldc] */
  int M, N, K;
  int lda, ldw, ldr, ldc;
} mmamd_gemm_problem;
int mmamd_gemm_bf16_grouped(const mmamd_gemm_problem* probs, int nprob, int out_dtype, int act, mmamd_stream_t stream);

/* Out-projection + residual + the LayerNorm behind it, up to two problems (the two towers) in ONE persistent launch whose workgroups own whole
 * rows (64 x N tiles):  X (fp32, in place) = A W^T + bias + X;  Y (bf16) = LayerNorm(X; gamma, beta, eps).  X is bit-identical to
 * mmamd_gemm_bf16(..., residual = X, C = X, out fp32); Y is the LayerNorm of mmamd_layernorm up to the summation order of the two row
 * statistics.  N must be 512 or 768 (mmamd_gemm_bf16_residual_ln_supported); callers fall back to the two separate launches otherwise.
 * Replaces `x = x + self_attn.out_proj(...)` followed by `norm2(x)` of nn.TransformerEncoderLayer(norm_first=True) (the layers of
 * models/clip/image_encoder.py:108 and models/clip/text_encoder.py:121). */
typedef struct {
  const void* A;      /* bf16 [M, K] */
  const void* W;      /* bf16 [K / 32][N][32]: W [N, K] repacked by mmamd_pack_w_ksteps (once per weight) */
  const float* bias;  /* fp32 [N] */
  float* X;           /* fp32 [M, N]: residual in, updated stream out */
  const float* gamma; /* fp32 [N] */
  const float* beta;  /* fp32 [N] */
  void* Y;            /* bf16 [M, N] */
  int M, N, K;
  float eps;
} mmamd_gemm_ln_problem;
int mmamd_gemm_bf16_residual_ln_grouped(const mmamd_gemm_ln_problem* probs, int nprob, mmamd_stream_t stream);
int mmamd_gemm_bf16_residual_ln_supported(int M, int N, int K); /* 1 / 0 */
/* W [N, K] bf16 row-major -> [K / 32][N][32] (same size), the operand layout of mmamd_gemm_bf16_residual_ln_grouped; K % 32 == 0 */
int mmamd_pack_w_ksteps(const void* W, int N, int K, void* out, mmamd_stream_t stream);

/* Training forward of an MLP's first linear (linear1 of the encoder layers, modules/layers/mlp.py:60-79 under autograd): ONE pass
 * writes the pre-activation U = A W^T + bias (bf16 [M, ldu], kept for the backward) and G = act(U) (bf16 [M, ldg], the input of the
 * second linear), act = MMAMD_ACT_QUICKGELU or MMAMD_ACT_GELU_ERF applied to the bf16-rounded U (exactly what mmamd_act_fwd on U
 * gives).  Operand constraints as mmamd_gemm_bf16. */
int mmamd_gemm_bf16_dual(const void* A, int lda, const void* W, int ldw, const float* bias, void* U, int ldu, void* G, int ldg,
                         int M, int N, int K, int act, mmamd_stream_t stream);

/* --- K4: multi-head self-attention forward ---------------------------------------------------
 * qkv: bf16 [B*S, 3*H*64] rows = tokens, columns = [q | k | v], each H heads of 64;  out: bf16
 * [B*S, H*64] = softmax(q k^T * scale (+causal)) v, heads merged.  Head dim is 64 for every
 * model on the path.  Replaces F.scaled_dot_product_attention reached from nn.MultiheadAttention
 * (image_encoder.py:108 non-causal; text_encoder.py:121 is_causal=True).  S <= 288: whole K/V of a head resident in LDS (persistent
 * kernel); longer sequences stream K/V through LDS in 128-key chunks (two passes, no log-sum-exp output). */
int mmamd_attention_fwd(const void* qkv, void* out, int B, int S, int H, int causal, float scale,
                        mmamd_stream_t stream);

/* The attention of BOTH towers of a dual encoder layer (image_encoder.py:108 + text_encoder.py:121) in ONE persistent launch: the
 * workgroups walk problem 0's (batch, head) items, then problem 1's.  Operands per problem as mmamd_attention_fwd; lse optional
 * (NULL, or fp32 [B,H,S] as mmamd_attention_fwd_lse).  Bit-identical to one mmamd_attention_fwd call per problem.  Problems the
 * LDS-DMA ring kernel does not take (S > 224) run as the separate launches this call stands for. */
typedef struct {
  const void* qkv;
  void* out;
  float* lse;
  int B, S, H, causal;
} mmamd_attn_problem;
int mmamd_attention_fwd_grouped(const mmamd_attn_problem* probs, int nprob, float scale, mmamd_stream_t stream);

/* Backward of mmamd_attention_x_fwd (same operand description; out / dout bf16 [B*Sq, ldo], lse from the forward): writes
 * dq [B, Sq, lddq] (ALWAYS per sample — with batch-shared queries the caller sums over the batch), dk / dv [B*Sk, lddk = lddv]
 * (bf16; they may be column slices of one buffer).  head_dim 96 needs Sk <= 256 (LDS). */
int mmamd_attention_x_bwd(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                          int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                          int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo, const float* lse,
                          void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq, int Sk, int H, int head_dim,
                          float scale, mmamd_stream_t stream);
/* mmamd_attention_x_bwd for a forward that ran with `head_mask` (mmamd_attention_x_fwd_head_mask below: same mask, same element strides; the mask is a
 * constant, it multiplies P in dV = P'^T dO and dP in dS = P (dP m - D)): training with the reference's head_mask (modules/layers/attention.py:236-237). */
int mmamd_attention_x_bwd_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                    int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                    int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo, const float* lse,
                                    void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq, int Sk, int H, int head_dim,
                                    float scale, const float* head_mask, int64_t hm_stride_b, int64_t hm_stride_h, int64_t hm_stride_q,
                                    int64_t hm_stride_k, mmamd_stream_t stream);
/* mmamd_attention_x_fwd with the reference's `head_mask` (modules/layers/attention.py:190,236-237: `attn = attn * head_mask` after softmax and
 * dropout; what is returned as the attention weights and what multiplies V): fp32, any tensor that broadcasts to [b, h, q, k], given by its element
 * strides over those four dimensions (0 = broadcast). */
int mmamd_attention_x_fwd_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                    int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                    int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                                    float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, const float* head_mask,
                                    int64_t hm_stride_b, int64_t hm_stride_h, int64_t hm_stride_q, int64_t hm_stride_k,
                                    mmamd_stream_t stream);

/* The same two entries with TRAINING-TIME DROPOUT on the normalised probabilities (reference modules/layers/attention.py:234-239: F.dropout on
 * the softmax output, before the product with V; models/flava/transformer.py:87 builds SelfAttention(dropout)).  keep(b, h, q, key) = word key & 3
 * of the Philox4x32-10 block with counter ((b H + h) Sq + q) * ceil(Sk / 4) + key / 4 and (site, 0), keyed by `seed` (oracle/philox.py);
 * survivors are scaled by 1 / (1 - drop_p).  The returned probabilities are the dropped ones, like the reference's; lse is unaffected.  The
 * backward regenerates the mask from (seed, site). */
int mmamd_attention_x_fwd_dropout(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                          int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                          int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                          float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed,
                                  uint32_t site, mmamd_stream_t stream);
int mmamd_attention_x_bwd_dropout(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                          int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                          int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo, const float* lse,
                          void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq, int Sk, int H, int head_dim,
                          float scale, float drop_p, uint64_t seed,
                                  uint32_t site, mmamd_stream_t stream);

/* Both at once -- training-time dropout on the probabilities AND the reference's head_mask (modules/layers/attention.py:232-237 applies F.dropout and
 * then `attn = attn * head_mask`): P' = P keep / (1 - p) m.  Same conventions as the two pairs above. */
int mmamd_attention_x_fwd_dropout_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                            int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                            int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                                            float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, float drop_p, uint64_t seed,
                                            uint32_t site, const float* head_mask, int64_t hm_stride_b, int64_t hm_stride_h,
                                            int64_t hm_stride_q, int64_t hm_stride_k, mmamd_stream_t stream);
int mmamd_attention_x_bwd_dropout_head_mask(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                                            int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                                            int64_t full_mask_batch_stride, int causal, const void* out, const void* dout, int ldo,
                                            const float* lse, void* dq, int lddq, void* dk, void* dv, int lddk, int lddv, int B, int Sq, int Sk,
                                            int H, int head_dim, float scale, float drop_p, uint64_t seed, uint32_t site,
                                            const float* head_mask, int64_t hm_stride_b, int64_t hm_stride_h, int64_t hm_stride_q,
                                            int64_t hm_stride_k, mmamd_stream_t stream);

/* mmamd_attention_fwd that also saves the log2-domain log-sum-exp [B,H,S] (fp32) of the scaled scores for mmamd_attention_bwd. */
int mmamd_attention_fwd_lse(const void* qkv, void* out, float* lse, int B, int S, int H, int causal, float scale,
                            mmamd_stream_t stream);

/* Split-K form of the same GEMM for weight gradients: C[M,N] (fp32, ldc = N) = A[M,K] . W[N,K]^T with a long contraction
 * (K = tokens, a multiple of 128) and few output tiles: the K range is cut into `splits` chunks (one grid row each), partial
 * outputs go to ws (splits * M * N floats) and are summed by a second kernel.  dW = dY^T X of every nn.Linear on the path. */
int mmamd_gemm_bf16_splitk(const void* A, int lda, const void* W, int ldw, float* C, float* ws, int M, int N, int K, int splits,
                           mmamd_stream_t stream);
/* The same weight-gradient GEMM straight from the ROW-major operands: C[M,N] (fp32) = sum_t A[t, m] W[t, n], A = dY [K, lda] and
 * W = X [K, ldw] as the forward / backward kernels left them (bf16, K = tokens, a multiple of 128; M, N multiples of 8).  No
 * transposed copies: the kernel reads its MFMA operands from LDS with ds_read_b64_tr_b16. */
int mmamd_gemm_bf16_tn_splitk(const void* A, int lda, const void* W, int ldw, float* C, float* ws, int M, int N, int K, int splits,
                              mmamd_stream_t stream);
/* The same weight-gradient GEMM with the bias gradient db[M] = column sums of A (= dY) produced by the SAME pass over A (the torch autograd
 * it replaces computes grad_bias = grad_output.sum(0) as its own reduction: torch/csrc/autograd/FunctionsManual.cpp, linear backward;
 * reference caller: every nn.Linear of modules/layers/{attention,mlp,multi_head_attention}.py under loss.backward(),
 * examples/flava/native/train.py:312-322).  ws: (splits + 1) * M * N + splits * M floats. */
int mmamd_gemm_bf16_tn_splitk_colsum(const void* A, int lda, const void* W, int ldw, float* C, float* db, float* ws, int M, int N, int K,
                                     int splits, mmamd_stream_t stream);
/* Up to 8 weight gradients in ONE launch (the four dW = dY^T X of a transformer layer) + one reduce launch: job i is dw[M,N] = dy[K,M]^T x[K,N]
 * (row-major bf16 operands over the K tokens, K % 128 == 0, M and N multiples of 8), db[M] (optional) = the column sums of dy from the same pass.
 * `splits`: K is cut into that many parts per problem (every (tile, split) pair is one workgroup: pick it so that the launch fills the CUs a whole
 * number of times).  ws: mmamd_gemm_bf16_tn_splitk_group_ws(jobs, njobs, splits) floats.  Each result equals mmamd_gemm_bf16_tn_splitk(_colsum) of
 * that problem with the same `splits`, bit for bit.  Replaces the per-Linear weight / bias gradients of torch autograd. */
typedef struct {
  const void* dy;
  int lddy;
  const void* x;
  int ldx;
  float* dw;
  float* db;
  int M, N, K;
} mmamd_wgrad_job;
long long mmamd_gemm_bf16_tn_splitk_group_ws(const mmamd_wgrad_job* jobs, int njobs, int splits);
int mmamd_gemm_bf16_tn_splitk_group(const mmamd_wgrad_job* jobs, int njobs, int splits, float* ws, mmamd_stream_t stream);

/* Same, additionally returning the attention probabilities (normalised, [B,H,S,S], probs_dtype F32 or BF16) and honouring a
 * key-padding mask (uint8 [B,S], 0 = masked key, NULL = none).  Non-causal.  Replaces scaled_dot_product_attention of
 * modules/layers/attention.py:185-241 as FLAVA's encoders call it (models/flava/transformer.py:155-176 with
 * return_attn_weights; padding mask from modules/encoders/bert_text_encoder.py:86-91).  probs may be NULL (mask only). */
int mmamd_attention_probs_fwd(const void* qkv, const uint8_t* key_mask, void* out, void* probs, int probs_dtype, int B,
                              int S, int H, float scale, mmamd_stream_t stream);

/* The same probabilities [B,H,S,S] fp32 (no key mask) from the packed projections and the log2-domain log-sum-exp [B,H,S] a training
 * forward saved (mmamd_attention_fwd_lse): P = exp2(scale log2(e) q.k - lse), one pass, stored as whole cache lines
 * (csrc/attention_probs_lse.hip).  64 <= S <= 288 and S % 8 != 0, else MMAMD_E_UNSUPPORTED.  FLAVA's training forwards hand out `attentions` this way
 * (models/flava/transformer.py:254-259 of the reference returns them in training too) without running the attention a second time. */
int mmamd_attention_probs_from_lse(const void* qkv, const float* lse, void* probs, int B, int S, int H, float scale,
                                   mmamd_stream_t stream);

/* General attention: separate strided q / k / v (bf16), Sq != Sk, head_dim 64 or 96, optional causal / key-padding /
 * full [B or 1, Sq, Sk] uint8 masks (0 = masked), optional probabilities.  q row (b,i) at q + b*q_batch_stride + i*ldq
 * (q_batch_stride = 0: queries shared by every sample), k/v row (b,j) at k/v + b*kv_batch_stride + j*ldk/ldv; out bf16
 * [B*Sq, ldo].  Replaces F.scaled_dot_product_attention in MultiHeadAttentionWithCache / MultiHeadSelfAttention
 * (modules/layers/multi_head_attention.py:69-71,165-167) as CoCa calls them: decoder self-attention with the
 * padding-aware causal mask (models/coca/text_decoder.py:178-194), cross-attention of the multimodal decoder
 * (modules/layers/transformer.py:367-385), AttentionPooler (modules/layers/attention_pooler.py:58-70).
 * lse (optional, [B,H,Sq] fp32): log2-domain log-sum-exp of the scaled masked scores, saved for mmamd_attention_bwd.  * `causal` is a 2-bit flag: bit 0 = causal (needs Sq == Sk); bit 1 = the key-padding mask binds the LAST query row only — CoCa's text decoder
 * (models/coca/text_decoder.py:176-194: every row is causal, the CLS query in the last row additionally hides padded tokens), which then needs no
 * [B, S, S] mask tensor.  The same flag in mmamd_attention_x_bwd. */
int mmamd_attention_x_fwd(const void* q, int ldq, int64_t q_batch_stride, const void* k, const void* v, int ldk, int ldv,
                          int64_t kv_batch_stride, const uint8_t* key_mask, const uint8_t* full_mask,
                          int64_t full_mask_batch_stride, int causal, void* out, int ldo, void* probs, int probs_dtype,
                          float* lse, int B, int Sq, int Sk, int H, int head_dim, float scale, mmamd_stream_t stream);

/* Backward of self-attention over a packed qkv (head dim 64): given the forward's out [B*S, D] (bf16), d(out), and the
 * log2-domain log-sum-exp `lse` [B,H,S] that mmamd_attention_x_fwd saved, writes dqkv [B*S, 3D] = [dQ | dK | dV] (bf16).
 * key_mask (optional, uint8 [B,S], 0 = masked key) as in the forward.  Two kernels (dQ; dK+dV), no atomics.  This is what torch autograd computes for F.scaled_dot_product_attention under
 * nn.MultiheadAttention (call sites models/clip/image_encoder.py:108, text_encoder.py:121). */
int mmamd_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const uint8_t* key_mask, void* dqkv,
                        int B, int S, int H, int causal, float scale, mmamd_stream_t stream);

/* --- K1 front end: non-overlapping patch extraction ("im2col" of a stride==kernel conv) -------
 * images [B,C,HW,HW] (f32 or bf16) -> patches bf16 [B*(HW/P)^2, Kpad], column k = (c*P+py)*P+px,
 * columns >= C*P*P zero-filled.  Replaces the gather half of nn.Conv2d at image_encoder.py:50-56,91. */
int mmamd_patchify(const void* images, int img_dtype, void* patches, int B, int C, int HW, int P,
                   int Kpad, mmamd_stream_t stream);

/* --- K1 back end: prepend CLS, add positional embedding, ln_pre -------------------------------
 * patch_emb [B*G2, d] (pe_dtype) ; cls [d], pos [(G2+1), d], gamma/beta [d] fp32 -> x fp32 [B*(G2+1), d].
 * Replaces torch.cat/+/ln_pre at image_encoder.py:98-106. */
int mmamd_vit_assemble_ln(const void* patch_emb, int pe_dtype, const float* cls, const float* pos,
                          const float* gamma, const float* beta, float eps, float* x, int B, int G2,
                          int d, mmamd_stream_t stream);

/* Fused ViT stem (image_encoder.py:91-106 without the im2col copy and without a second pass over the tokens):
 * mmamd_patch_embed_gemm: x[b*(g*g+1) + 1 + i, :] = conv(patch i of image b) + pos[1 + i, :]  (fp32), the patch rows gathered straight from
 *   the bf16 image [B,3,image_size,image_size] by the GEMM's LDS-DMA (16-byte pieces = 8 pixels of one image row; patch 16 or 32), W = conv.weight
 *   viewed [width, 3*patch*patch] (bf16, row pitch ldw), pos fp32 [g*g+1, width].  Row 0 of every image (the CLS token) is not touched.
 * mmamd_vit_cls_lnpre_ln: x[b,0,:] = cls + pos0 (pos0 = pos[0,:]); x = ln_pre(x) in place (fp32); hn (optional, bf16 [B*S, d]) =
 *   LayerNorm(x; gamma1, beta1, eps1) = norm1 of the first encoder layer, same pass. */
int mmamd_patch_embed_gemm(const void* image, const void* W, int ldw, const float* pos, float* x, int B, int patch, int image_size,
                           int width, mmamd_stream_t stream);
int mmamd_vit_cls_lnpre_ln(float* x, const float* cls, const float* pos0, const float* gamma, const float* beta, float eps,
                           const float* gamma1, const float* beta1, float eps1, void* hn, int B, int S, int d, mmamd_stream_t stream);

/* --- K8: token embedding gather + positional embedding ----------------------------------------
 * x[b,s,:] = table[ids[b,s],:] + pos[s,:]  (fp32 out).  Returns the launch status only; ids are
 * range-clamped on device (out-of-range ids are a caller error, as with nn.Embedding).
 * Replaces text_encoder.py:118-119. */
int mmamd_embed_tokens(const int64_t* ids, const void* table, int table_dtype, const float* pos,
                       float* x, int B, int S, int d, int vocab, mmamd_stream_t stream);

/* out[i] = 1 where key i may be attended, 0 where it is padding.  kind 0: src = int64 token ids, keep = (id != pad_id);
 * kind 1/2/3: src = float32 / int64 / uint8-bool mask, keep = (value != 0).  Replaces the mask construction of
 * modules/encoders/bert_text_encoder.py:84-91 (+ utils/attention.py:13-52). */
/* out[b][c] = mean over tokens first .. S-1 of x[b][s][c], x fp32 [B,S,d], d % 4 == 0 (GlobalAveragePooler: reference
 * modules/encoders/vision_transformer.py:117-127 averages the patch rows and skips the CLS row, first = 1). */
int mmamd_token_mean(const float* x, float* out, int B, int S, int d, int first, mmamd_stream_t stream);
int mmamd_key_mask(const void* src, int kind, int64_t pad_id, uint8_t* out, int64_t n, mmamd_stream_t stream);

/* CoCa text embeddings: x[b,s] = table[ids[b,s]] + pos[s] (s < S_ids); x[b,S_ids] = cls + pos[S_ids] when cls != NULL.
 * fp32 [B, S_ids(+1), d].  Replaces CoCaTextEmbeddings.forward, models/coca/text_decoder.py:67-88. */
int mmamd_coca_text_embed(const int64_t* ids, const float* table, const float* pos, const float* cls, float* x, int B,
                          int S_ids, int d, int vocab, mmamd_stream_t stream);

/* CoCaTextDecoder.build_mask (models/coca/text_decoder.py:178-194) as uint8 [B, S+1, S+1] (1 = attend): causal, and the CLS
 * query row masks padded tokens (shifted by one column, like the reference's F.pad).  src/kind/pad_id as mmamd_key_mask. */
int mmamd_coca_text_mask(const void* src, int kind, int64_t pad_id, uint8_t* out, int B, int S, mmamd_stream_t stream);

/* BERT embeddings: x[b,s,:] = LayerNorm(word[ids] + position[pos_ids or s] + token_type[type_ids or 0]) (fp32 out).
 * gamma = beta = NULL: the un-normalised sum (the training forward keeps it for the LayerNorm backward).
 * Replaces modules/layers/text_embedding.py:74-104 (three gathers, add, nn.LayerNorm). */
int mmamd_bert_embed_ln(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids, const float* word,
                        const float* pos, const float* type, const float* gamma, const float* beta, float eps, float* x,
                        int B, int S, int d, int vocab, int max_pos, int n_types, mmamd_stream_t stream);

/* FLAVA image embeddings: x[b,0] = cls + pos[0]; x[b,1+i] = blend(patch_emb[b,i], mask_token, patches_mask[b,i]) + pos[1+i]
 * (fp32, no LayerNorm; patches_mask int64 [B,G2] / mask_token may be NULL; cls NULL = no CLS row, x is [B,G2,d]).  Replaces
 * models/flava/image_encoder.py:139-177 and modules/layers/patch_embedding.py:98-152 (CoCa's ViT). */
int mmamd_flava_image_embed(const float* patch_emb, const float* cls, const float* pos, const int64_t* patches_mask,
                            const float* mask_token, float* x, int B, int G2, int d, mmamd_stream_t stream);

/* --- FLAVA image codebook: the DALL-E dVAE encoder (models/flava/model.py:583-744) -------------------------------------------
 * Activations are bf16 rows [B*(H+2)*(W+2), C]: NHWC with a one-pixel ZERO border per image (= the convolutions' padding) and at
 * least W+3 READABLE rows in front of and behind the buffer (contents irrelevant: only border positions, stored as zeros, read them).  mmamd_conv_gemm_bf16 is one convolution as an implicit GEMM:
 *   out[m, n] = bias[n] + sum_t sum_c A[m + tap_row_offsets[t], c] * W[n, t*Cin + c]  (+ residual[m, n], bf16)
 * (3x3: the 9 offsets dy*(W+2)+dx; 1x1: the single offset 0; replaces nn.functional.conv2d in DalleConv2d.forward :597-598 and the
 * `id_path(x) + post_gain * res_path(x)` of DalleEncoderBlock.forward :624-625 with post_gain folded into W / bias).  Rows on the
 * border of the grid_h x grid_w image grid are stored as zeros (grid_h = grid_w = 0: no masking).  Outputs: C (bf16 [M, ldc], or
 * fp32 when out_dtype = MMAMD_F32), optionally with ReLU applied (relu_c), and optionally C_relu = relu(C) (bf16): the nn.ReLU in
 * front of the next convolution.  Cin % 64 == 0, N % 8 == 0, 1 <= ntaps <= 9. */
int mmamd_conv_gemm_bf16(const void* A, int lda, const int64_t* tap_row_offsets, int ntaps, const void* W, int ldw, const float* bias,
                         const void* residual, int ldr, void* C, int ldc, int out_dtype, void* C_relu, int ldc_relu, int relu_c, int M,
                         int N, int Cin, int grid_h, int grid_w, mmamd_stream_t stream);
/* 7x7 stem: cols[(b, y, x) over the padded grid][(c, ky, kx) padded to kpad] (bf16) from fp32 NCHW images — the stem is then a one-tap
 * mmamd_conv_gemm_bf16 on cols with the weight in its own [n_out][n_in*kw*kw] order. */
int mmamd_dalle_stem_im2col(const float* images, void* cols, int B, int C, int H, int W, int kw, int kpad, mmamd_stream_t stream);
/* nn.MaxPool2d(2) (:677) on a padded-grid tensor: [B,H+2,W+2,C] -> y [B,H/2+2,W/2+2,C] (zero border) and/or y_relu = relu(y). */
int mmamd_dalle_maxpool2(const void* x, void* y, void* y_relu, int B, int H, int W, int C, mmamd_stream_t stream);
/* torch.argmax(z_logits, axis=1) (:733-735): ids[b, y, x] (int64) from fp32 padded-grid logits [B,H+2,W+2,V]; first maximum wins. */
int mmamd_dalle_argmax(const float* logits, int64_t* ids, int B, int H, int W, int V, mmamd_stream_t stream);
/* In-place fp32 softmax over the V columns of every row (get_codebook_probs :737-739: nn.Softmax(dim=1) of the NCHW logits). */
int mmamd_row_softmax_(float* x, int64_t rows, int V, mmamd_stream_t stream);
/* kernel-ready copy of a DalleConv2d parameter (w [n_out, n_in, kw, kw] fp32, taps = kw*kw; a bias is n_in = taps = 1): dst[o*ld_dst +
 * idx] = gain * w[o][c][t], idx = t*n_in + c (tap_major) or c*taps + t; the tail of each row is zero. */
int mmamd_dalle_pack(const float* src, void* dst, int dst_dtype, int n_out, int n_in, int taps, int ld_dst, float gain, int tap_major,
                     mmamd_stream_t stream);

/* FLAVA position-embedding interpolation (models/flava/image_encoder.py:102-137, interpolate_pos_encoding): pos [1 + n_side^2, d]
 * fp32 -> out [1 + h0*w0, d]: row 0 copied, the patch grid resampled like F.interpolate(mode="bicubic", align_corners=False,
 * scale_factor=(scale_h, scale_w)) (torch's upsample_bicubic2d: A = -0.75, clamped taps). */
int mmamd_bicubic_pos_embed(const float* pos, int n_side, int d, float* out, int h0, int w0, float scale_h, float scale_w,
                            mmamd_stream_t stream);
/* out[b,s] = pad_id + (ids[b,s] != pad_id ? number of non-padding tokens in ids[b, :s+1] : 0)  (int64): the RoBERTa-style position ids of
 * BERTTextEmbeddings.create_position_ids_from_input_ids (modules/layers/text_embedding.py:55-68), used when offset_pos_ids is set. */
int mmamd_offset_position_ids(const int64_t* ids, int64_t pad_id, int64_t* out, int B, int S, mmamd_stream_t stream);
/* labels[i] = keep[i] ? labels[i] : fill (in place; FLAVAForPreTraining's image_labels[~image_patches_mask] = -1, model.py:340-343). */
int mmamd_mask_labels(int64_t* labels, const uint8_t* keep, int64_t fill, int64_t n, mmamd_stream_t stream);
/* ReLU backward on fp32: dz = y > 0 ? dy : 0 (classifier MLP of FLAVAForClassification under autograd). */
int mmamd_relu_bwd(const float* y, const float* dy, float* dz, int64_t n, mmamd_stream_t stream);

/* out[B,E] = act(rows . W^T + bias), rows i at h + i*ldh (fp32, exact-f32 MFMA); act 0 none, 1 tanh, 2 ReLU.  Replaces Pooler
 * (modules/losses/flava.py:84-97), the CLS projections (models/flava/model.py:243-247,260-264) and the classifier MLP on the CLS row
 * (FLAVAForClassification, models/flava/model.py:380-422 with modules/layers/mlp.py:13-66, nn.ReLU between the layers). */
int mmamd_rows_linear_f32(const float* h, int64_t ldh, const float* W, const float* bias, int act, float* out, int B, int d,
                          int E, mmamd_stream_t stream);

/* --- K7/K9/K10: pooled row -> LayerNorm -> projection (-> L2 normalize) -----------------------
 * row(b) = x[b, idx(b), :] with idx(b) = 0 when ids == NULL (CLS, image_encoder.py:111) or
 * argmax_s ids[b,s] (first maximum; text_encoder.py:129-132).  out[b,e] = sum_k LN(row)[k] *
 * proj[k*proj_sk + e*proj_se]  (image: projection [d,E] -> sk=E,se=1; text: Linear weight [E,d] ->
 * sk=1,se=d).  normalize != 0 additionally applies F.normalize (models/clip/model.py:72-73).
 * ws: scratch of B*d floats (the normalised pooled rows). */
int mmamd_pool_ln_proj(const float* x, int S, int d, const int64_t* ids, const float* gamma,
                       const float* beta, float eps, const float* proj, int proj_sk, int proj_se,
                       float* out, int B, int E, int normalize, float* ws, mmamd_stream_t stream);

/* F.normalize(x, p=2, dim=1, eps) on [rows,d]  (models/clip/model.py:72-73). */
int mmamd_l2_normalize(const void* x, int x_dtype, void* y, int y_dtype, int rows, int d, float eps,
                       mmamd_stream_t stream);
/* Same with an output row stride ldy >= d (elements): CLIP.forward normalises both towers straight into the two halves of the packed
 * [B, 2E] block the loss all-gathers (modules/losses/contrastive_loss_with_temperature.py:35-36: gather of a, gather of b). */
int mmamd_l2_normalize_ld(const void* x, int x_dtype, void* y, int y_dtype, int ldy, int rows, int d, float eps,
                          mmamd_stream_t stream);

/* --- K13: in-place clamp of the 0-dim logit_scale parameter ----------------------------------
 * (modules/losses/contrastive_loss_with_temperature.py:193). */
int mmamd_clamp_scalar(float* p, int has_min, float lo, int has_max, float hi, mmamd_stream_t stream);

/* --- K11/K12: logits + cross entropy ----------------------------------------------------------
 * a,b fp32 [B,E] local features; a_all,b_all fp32 [WB,E] gathered features (row stride ld_all
 * elements; pass a/b and ld_all=E for the single-process case); logit_scale: device scalar (log T).
 *   logits_a[B,WB] = (a . b_all^T) * exp(logit_scale);  logits_b[B,WB] = (b . a_all^T) * exp(..)
 *   labels[i] = label_offset + i;  row_mask (uint8 [B], NULL = all rows) drops rows from the loss
 *   out3 = {loss, loss_a, loss_b};  ws = scratch of at least 2*B floats.
 * Replaces contrastive_loss_with_temperature.py:81,90-107 (matmul x2, cross_entropy x2, mean). */
int mmamd_contrastive_fwd(const float* a, const float* b, const float* a_all, const float* b_all,
                          int ld_all, const float* logit_scale, int B, int WB, int E,
                          int label_offset, const uint8_t* row_mask, float label_smoothing,
                          int reduction, float* logits_a, float* logits_b, float* out3, float* ws,
                          mmamd_stream_t stream);

/* Same with a row stride for the LOCAL features (ld_local >= E): a / b may be the two halves of the packed [B, 2E] block that
 * CLIP.forward normalised into and the all-gather sent (no unpacking copies anywhere between the towers and the loss). */
int mmamd_contrastive_fwd_ld(const float* a, const float* b, int ld_local, const float* a_all, const float* b_all, int ld_all,
                             const float* logit_scale, int B, int WB, int E, int label_offset, const uint8_t* row_mask,
                             float label_smoothing, int reduction, float* logits_a, float* logits_b, float* out3, float* ws,
                             mmamd_stream_t stream);

/* --- backward of the contrastive loss (autograd of modules/losses/contrastive_loss_with_temperature.py:81-107; the
 * reference relies on torch autograd through matmul / cross_entropy / exp and, for BackpropType.GLOBAL, on the all-gather's
 * backward = reduce-scatter, utils/distributed.py:47-48).  Inputs as mmamd_contrastive_fwd plus its logits outputs and
 * grad_out3 = d(out3) (loss, loss_a, loss_b) on the device.  Outputs:
 *   G_a, G_b [B,WB]        d logits (workspace the caller may keep)
 *   grad_a, grad_b [B,E]   T G_a b_all (+ add_a),  T G_b a_all (+ add_b); add_* (row stride ld_add) may be NULL — the caller
 *                          passes its reduce-scattered share of the gathered gradients there (GLOBAL, world > 1), or points them
 *                          into grad_a_all / grad_b_all (computed first) when only its own block counts (LOCAL, or world = 1)
 *   grad_a_all / grad_b_all  rows [all_row0, all_row0 + all_rows) of d a_all = T G_b^T b and d b_all = T G_a^T a, written at
 *                          row 0.. of the given buffers (row stride ld_grad_all); both NULL = not wanted (BackpropType.NONE)
 *   grad_logit_scale [1]   sum(G_a * logits_a) + sum(G_b * logits_b)
 * ws: 2*B floats. */
int mmamd_contrastive_bwd(const float* a, const float* b, const float* a_all, const float* b_all, int ld_all,
                          const float* logit_scale, const float* logits_a, const float* logits_b, int B, int WB, int E,
                          int label_offset, const uint8_t* row_mask, float label_smoothing, int reduction,
                          const float* grad_out3, float* G_a, float* G_b, float* grad_a, float* grad_b, const float* add_a,
                          const float* add_b, int ld_add, float* grad_a_all, float* grad_b_all, int ld_grad_all,
                          int all_row0, int all_rows, float* grad_logit_scale, float* ws, mmamd_stream_t stream);

/* --- row / elementwise kernels of the backward pass (torch autograd of the modules named in the forward entries above)
 * LayerNorm backward: dx[rows,d] (fp32) = LN'(x; gamma)(dy) (+ add), dgamma[d], dbeta[d].  dy fp32 or bf16.  dx_bf16 (optional):
 * the same dx rounded to bf16 (operand of the following gradient GEMMs).  dx_colsum (optional, [d]): column sums of dx — the bias
 * gradient of the Linear whose output fed this LayerNorm's residual stream.
 * ws: (G + 1) * 3 * d floats, G = mmamd_layernorm_bwd_groups(rows, d) (the number of workgroups: <= 1024). */
int mmamd_layernorm_bwd_groups(int rows, int d);
int mmamd_layernorm_bwd(const float* x, const float* gamma, const void* dy, int dy_dtype, const float* add, float* dx,
                        void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, float* ws, int rows, int d, float eps,
                        mmamd_stream_t stream);
/* mmamd_layernorm_bwd with dgamma == dbeta == NULL leaves its per-workgroup partials in ws ([G][ns][d], G = mmamd_layernorm_bwd_groups(rows, d), ns = 2 or 3 with
 * dx_colsum) and skips the reduction; mmamd_colsum_stage2_batched reduces any number of such jobs in one launch per 64 (out0 | out1 | out2 receive the
 * column segments [0, seg) | [seg, 2 seg) | [2 seg, n) of sum_g part[g][0..n); out1 == NULL: one array of n).  Same arithmetic as the immediate form. */
typedef struct {
  const float* part;
  float *out0, *out1, *out2;
  int G, n, seg;
} mmamd_colsum_job;
int mmamd_colsum_stage2_batched(const mmamd_colsum_job* jobs, int njobs, mmamd_stream_t stream);
/* out[n] = column sums of x[rows,n] (bias gradients).  ws: min(1024, rows) * n floats. */
int mmamd_colsum(const void* x, int dtype, int rows, int n, float* out, float* ws, mmamd_stream_t stream);
/* g = act(u) and du = dg * act'(u), bf16, n % 4 == 0 (MMAMD_ACT_QUICKGELU / MMAMD_ACT_GELU_ERF). */
/* Training-time dropout / stochastic depth (reference: nn.Dropout at modules/layers/mlp.py:59-60, modules/layers/transformer.py:69-70,86-93;
 * torchvision StochasticDepth(mode="row") at transformer.py:64-67):  out[i] = (residual ? residual[i] : 0) + x[i] * keep(i) / (1 - p).
 * keep() is Philox4x32-10 keyed by `seed` with counter (index group, site): a pure function of (seed, site, i), so the backward calls the same
 * entry on the incoming gradient (residual = NULL) instead of storing masks.  group = 0: one decision per element; group > 0: one decision per
 * sample of `group` consecutive elements (drop path).  x / out: fp32 or bf16 (may alias); residual: fp32; mask_out (optional, tests): uint8 [n].
 * n % 4 == 0.  oracle/philox.py restates the generator. */
int mmamd_dropout(const void* x, int x_dtype, const float* residual, void* out, int out_dtype, uint8_t* mask_out, int64_t n, int64_t group,
                  float p, uint64_t seed, uint32_t site, mmamd_stream_t stream);
int mmamd_act_fwd(const void* u, void* g, int64_t n, int act, mmamd_stream_t stream);
int mmamd_act_bwd(const void* u, const void* dg, void* du, int64_t n, int act, mmamd_stream_t stream);
/* The activation MODULE called on its own (reference: modules/layers/activation.py:24-25, SiLU.forward = x * sigmoid(1.702 x); KAT
 * tests/modules/layers/test_activation.py:12-16): out = act(x) when dy == NULL, else out = dy * act'(x).  Any n; x / dy / out all of
 * `dtype` (MMAMD_F32 or MMAMD_BF16).  On the hot path the activation never runs as its own launch (GEMM epilogue). */
int mmamd_activation(const void* x, const void* dy, void* out, int dtype, int64_t n, int act, mmamd_stream_t stream);
/* dst[c*ld_dst + r] = bf16(src[r*ld_src + c]) for r < rows, 0 for rows <= r < ld_dst: operands of the weight-gradient GEMM
 * dW[N,K] = dY^T X computed by mmamd_gemm_bf16 as (dY^T)[N,M] . (X^T)[K,M]^T with the token index M as contraction.
 * colsum (optional, [cols] fp32) = column sums of the bf16-rounded source = the bias gradient of the same dY, produced in the same
 * pass; ws: ceil(ld_dst/64) * cols floats when colsum is given. */
int mmamd_transpose_to_bf16(const void* src, int src_dtype, int64_t ld_src, void* dst, int rows, int cols, int ld_dst,
                            float* colsum, float* ws, mmamd_stream_t stream);
/* Weight pack of a training step (what autograd's saved bf16 operands are in the reference's autocast run: torch re-casts every nn.Linear weight per step,
 * modules/layers/mlp.py:60-79, nn.MultiheadAttention's in / out projections): up to 64 fp32 [rows, cols] matrices -> their bf16 copies (nt, [rows, cols]) and /
 * or their bf16 transposes (tr, [cols, ld_t], columns >= rows zero-filled: the operand of dX = dY W as an NT GEMM) in ONE launch.  Bit-identical to
 * mmamd_convert / mmamd_transpose_to_bf16 per tensor. */
typedef struct {
  const float* src;
  void* nt;   /* bf16 [rows, cols] or NULL */
  void* tr;   /* bf16 [cols, ld_t] or NULL */
  int rows, cols, ld_t;
} mmamd_pack_desc;
int mmamd_pack_weights(const mmamd_pack_desc* descs, int n, mmamd_stream_t stream);
/* F.normalize backward (fp32): dx = (dy - y (y.dy)) / max(|x|, eps). */
int mmamd_l2_normalize_bwd(const float* x, const float* dy, float* dx, int rows, int d, float eps, mmamd_stream_t stream);
/* dst[idx[i], :] += src[i, :] with fp32 atomics (embedding-table gradient; pooled-row gradient into the sequence). */
int mmamd_scatter_add_rows(const float* src, const int64_t* idx, int n, int d, float* dst, int64_t dst_rows, mmamd_stream_t stream);

/* Small exact-fp32 GEMM with element strides: C[m,n] = sum_k X[m*sxm + k*sxk] * Y[n*syn + k*syk] (+ R[m*ldr + n]).  Used for
 * the pooled projections and their gradients in the training step (x @ projection, n^T de, de P^T: image_encoder.py:111-112,
 * text_encoder.py:130-132 and their autograd). */
int mmamd_f32_gemm_strided(const float* X, int64_t sxm, int64_t sxk, const float* Y, int64_t syn, int64_t syk, const float* R, int ldr,
                           float* C, int ldc, int M, int N, int K, mmamd_stream_t stream);

/* --- FLAVA pre-training heads (modules/losses/flava.py:110-238, 391-469)
 * Compaction of the labelled positions (replaces the boolean indexing `hidden_states[masked_tokens, :]`,
 * `masked_labels[masked_tokens]` :212-215 and the ITM row filter `sequence[pos_mask]` :433-437): for labels [B,L], in
 * row-major order, every (b,l) with labels != ignore_index (and row_keep[b] != 0 when row_keep is given) appends
 * idx_out = b*seq_S + tok_offset + l and label_out = labels[b,l] (label_out may be NULL); count_out[0] = number kept. */
int mmamd_select_tokens(const int64_t* labels, const uint8_t* row_keep, int64_t ignore_index, int B, int L, int seq_S,
                        int tok_offset, int32_t* idx_out, int64_t* label_out, int32_t* count_out, mmamd_stream_t stream);

/* dst[i,:] = src[idx[i]*row_stride : +d] (fp32 source rows; dst fp32 or bf16, dense [n,d]); zero_rows (optional, int64 [n]):
 * output rows with a non-zero flag are written as zeros (masked patches in the image-embedding backward). */
int mmamd_gather_rows(const float* src, int64_t row_stride, const int32_t* idx, int n, int d, void* dst, int dst_dtype,
                      const int64_t* zero_rows, mmamd_stream_t stream);

/* nn.CrossEntropyLoss(ignore_index=...) with mean reduction: out_loss[0] = mean over kept rows of lse(logits[i]) -
 * logits[i, labels[i]] (NaN when no row is kept, like torch).  ws: 2*N floats.  Replaces :137-140, :225-228. */
int mmamd_cross_entropy(const float* logits, int64_t ld, const int64_t* labels, int N, int V, int64_t ignore_index,
                        float* out_loss, float* ws, mmamd_stream_t stream);

/* Backward of mmamd_cross_entropy (mean over kept rows): dlogits[N, ldd] (bf16 or fp32; columns [V, ldd) zeroed) =
 * grad_out[0] / n_kept * (softmax - onehot) on kept rows, 0 on ignored rows.  ws: 2*N + 1 floats. */
int mmamd_cross_entropy_bwd(const float* logits, int64_t ld, const int64_t* labels, int N, int V, int64_t ignore_index,
                            const float* grad_out, void* dlogits, int dlogits_dtype, int64_t ldd, float* ws, mmamd_stream_t stream);

/* --- input side: CLIPImageTransform / FLAVAImageTransform on the device ----------------------------
 * Resize + CenterCrop / resized crop + ToTensor + Normalize (+ im2col) for a ragged batch of decoded uint8 images, bit for bit what
 * torchmultimodal/transforms/clip_transform.py:326-352 (and flava_transform.py:262-297) compute per image on the host through
 * torchvision + Pillow: Pillow's two 8-bit resampling passes (Resample.c; 22-bit fixed-point coefficients built by the host for
 * the filter in use -- bicubic or Lanczos --, uint8 intermediate), then a per-channel value table for the float conversion.
 * desc: int64 [B,16] in device memory, per image
 *   [0] address of the source view's first pixel   [1] source row stride (bytes)   [2] view height  [3] view width
 *   [4] row0: first view row the vertical pass reads   [5] nrows: how many   [13] bytes per source pixel (3 = RGB, 4 = RGBX)
 *   [6] / [7] / [8]  horizontal pass: index into `tables` of the coefficients [crop_w][ksize_h], of the (first column, taps)
 *                    pairs [crop_w][2], and ksize_h -- for the crop_w resized columns the crop keeps
 *   [9] / [10] / [11] vertical pass: coefficients [crop_h][ksize_v], (first row - row0, taps) pairs [crop_h][2], ksize_v
 *   [12] byte offset of this image's intermediate [nrows][crop_w][3] in tmp.        ([14], [15] reserved)
 * tables: int32, device.  max_rows = max nrows over the batch; max_seg_bytes = max over the batch of the source bytes per row the
 * horizontal pass reads, (last column's first + taps - first column's first) * bytes per pixel (0 = unknown: takes the untiled
 * kernels); max_coef_ints = max over the batch of crop_w * ksize_h (sizes the LDS copy of the horizontal
 * coefficient table; 0 = none).  lut: float [3][256], device: the value of byte v in channel c (ToTensor + Normalize: ((v/255) - mean[c]) / std[c]
 * evaluated in fp32; may be NULL when only out_u8 is requested).  Outputs (any subset, NULL = skip):
 * out_f32 [B,3,crop_h,crop_w]; patches bf16 [B*(crop_h/P)*(crop_w/P), kpad], column (c*P+py)*P+px (columns >= 3*P*P are NOT
 * written: zero the buffer once when kpad > 3*P*P); out_u8 [B,crop_h,crop_w,3], the resized crop itself. */
int mmamd_image_resample(const int64_t* desc, const int32_t* tables, uint8_t* tmp, int B, int crop_h, int crop_w, int max_rows,
                         int max_seg_bytes, int max_coef_ints, const float* lut, float* out_f32, void* patches, int P, int kpad, uint8_t* out_u8,
                         mmamd_stream_t stream);

/* --- zero-shot classification / retrieval read-outs -------------------------------------------------
 * The torch expressions of examples/flava/native/utils.py:100-160 and examples/flava/coco_zero_shot.py:24-31,78-90 on fp32 rows.
 * group_mean_normalize: out[g,:] = n(mean_t n(x[g*T+t,:])), n(v) = v/|v| -- one class of the zero-shot classifier from its T
 *   prompt embeddings (utils.py:108-111); x [G*T, d], out [G, d].
 * scale_normalize: y = scale * x/|x| per row (utils.py:141-142, the 100.0 * normalised image features).
 * target_rank: rank[r] = number of entries of scores[r, :C] that beat scores[r, target[r]] (ties: lower index first; target NULL =
 *   the row index, the diagonal of a retrieval matrix; a target outside [0, C) ranks C).  rank < k  <=>  the target is in torch.topk(k):
 *   replaces topk + eq + sum of `_accuracy` (utils.py:117-123) and `compute_recall` (coco_zero_shot.py:24-31). */
int mmamd_group_mean_normalize(const float* x, int G, int T, int d, float* out, mmamd_stream_t stream);
int mmamd_scale_normalize(const float* x, float* y, int rows, int d, float scale, mmamd_stream_t stream);
int mmamd_target_rank(const float* scores, int64_t ld, const int64_t* target, int R, int C, int32_t* rank, mmamd_stream_t stream);

/* Elementwise dtype conversion helper (fp32 <-> bf16), n elements. Used for weight packing. */
int mmamd_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n,
                  mmamd_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* MMAMD_H_ */
