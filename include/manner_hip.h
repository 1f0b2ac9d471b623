/*
 * manner_hip.h — C ABI of the MI355X (gfx950) MANNeR news-encoding + candidate-scoring hot path.
 *
 * The reference has no FFI: its "plugin boundary" for this path is a set of torch.nn.Module
 * classes imported by name.  Each entry point below states which reference
 * interface it replaces (paths relative to the reference repository root).  The Python mirror of
 * those classes (manner_amd/models/components/) binds these symbols through ctypes; see
 * INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - every pointer is CALLER-OWNED DEVICE memory unless marked "host"; nothing is retained past
 *     the call except by an encoder handle, which owns private packed copies of the weights;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and is asynchronous with respect to the host; no entry point allocates, frees or
 *     synchronises except *_create / *_destroy / manner_hip_encoder_status;
 *   - return value: 0 = ok, non-zero = MANNER_HIP_E_*; manner_hip_last_error() returns a
 *     thread-local message for the most recent failure.  No C++ exception crosses the ABI.
 */
#ifndef MANNER_HIP_H
#define MANNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MANNER_HIP_ABI_VERSION 8

enum {
  MANNER_HIP_OK = 0,
  MANNER_HIP_E_INVALID = 1,     /* bad argument (shape, alignment, unsupported architecture) */
  MANNER_HIP_E_WORKSPACE = 2,   /* workspace too small */
  MANNER_HIP_E_RUNTIME = 3,     /* HIP runtime error */
  MANNER_HIP_E_INPUT = 4        /* device-side input validation failed (see encoder_status) */
};

/* Device-side input validation.  Kernels never fault on bad inputs: they raise one of these bits in a status word
 * (the encoder handle's own, or the caller-owned int32 `status` argument of the scoring entry points, which may be
 * NULL) and continue on a safe substitute.  The reference raises a Python exception in each of these cases. */
enum {
  MANNER_HIP_STATUS_MASK = 1,     /* attention_mask is not a right-padded 0/1 prefix with 1..MAX_LEN tokens */
  MANNER_HIP_STATUS_TOKEN = 2,    /* input id / position outside the embedding tables (IndexError in the reference) */
  MANNER_HIP_STATUS_FUSED = 4,    /* reserved (bounded-wait overflow of a fused kernel) */
  MANNER_HIP_STATUS_INDEX = 8,    /* news / entity index outside the table (IndexError in the reference) */
  MANNER_HIP_STATUS_LENGTHS = 16  /* host_lengths disagree with attention_mask (tokens beyond the chunk bound dropped) */
};

enum { MANNER_HIP_ARCH_BERT = 0, MANNER_HIP_ARCH_ROBERTA = 1 };

/* Arithmetic of the encoder GEMMs/attention.  F32: f32 operands on the f32 MFMA (exact f32 FMA
 * chains; the 1e-4 parity mode).  BF16: bf16 operands, f32 accumulation, f32 LayerNorm/softmax.
 * BF16X3: f32 activations and f32 attention / LayerNorm / erf-GeLU as in F32, but every GEMM runs on the bf16 MFMA
 * over split operands — x = hi + lo, W = hi + lo (bf16 each), x W^T ~ hi.hi + hi.lo + lo.hi as ONE bf16 GEMM of
 * depth 3K: 16-bit operand mantissas, f32 accumulation.  2.3x the F32 mode's speed at 3e-5 .. 1.1e-4 of the reference
 * (stated tolerance 2.5e-4; F32 remains THE 1e-4 parity mode).  Needs H and I multiples of 256.
 * F16: the BF16 schedule and kernels on IEEE half operands (v_mfma_f32_*_f16: same rate, same bytes): 11 mantissa
 * bits instead of 8 (~8x smaller error), f16's exponent range (max 65504) — the arithmetic the reference's own GPU
 * setting computes in (`precision: 16-mixed`, configs/trainer/default.yaml:12).
 * F16X3: BF16X3 with IEEE half splits — hi and lo carry 11 bits each, so the three products reproduce ~21 operand
 * bits (lo parts below 6e-5 fall into f16's subnormals and keep 3e-8 absolute): within 1e-4 of the reference like F32,
 * at the speed of BF16X3.  Needs H and I multiples of 256. */
enum { MANNER_HIP_PREC_F32 = 0, MANNER_HIP_PREC_BF16 = 1, MANNER_HIP_PREC_BF16X3 = 2, MANNER_HIP_PREC_F16 = 3,
       MANNER_HIP_PREC_F16X3 = 4 };

typedef void* manner_hip_stream_t;
typedef struct manner_hip_encoder* manner_hip_encoder_t;

typedef struct manner_hip_encoder_config {
  int32_t arch;         /* MANNER_HIP_ARCH_* : position-id rule */
  int32_t hidden;       /* H, multiple of 128 */
  int32_t layers;
  int32_t heads;        /* head_dim = H / heads must be 64 */
  int32_t intermediate; /* I, multiple of 128 */
  int32_t vocab;
  int32_t max_pos;
  int32_t type_vocab;
  int32_t pad_id;       /* RoBERTa: positions start at pad_id + 1 */
  float ln_eps;
} manner_hip_encoder_config;

/* Order of the fp32 device pointers in the `weights` table of manner_hip_encoder_create: the HF
 * BertModel/RobertaModel state_dict tensors that sit under
 * "news_encoder.text_encoder.plm_model." in a reference checkpoint, nn.Linear
 * layout [out, in].  5 embedding tensors, then 16 per layer. */
enum {
  MANNER_HIP_W_WORD_EMB = 0, MANNER_HIP_W_POS_EMB, MANNER_HIP_W_TYPE_EMB,
  MANNER_HIP_W_EMB_LN_G, MANNER_HIP_W_EMB_LN_B,
  MANNER_HIP_W_EMB_COUNT
};
enum {
  MANNER_HIP_WL_Q_W = 0, MANNER_HIP_WL_Q_B, MANNER_HIP_WL_K_W, MANNER_HIP_WL_K_B,
  MANNER_HIP_WL_V_W, MANNER_HIP_WL_V_B, MANNER_HIP_WL_AO_W, MANNER_HIP_WL_AO_B,
  MANNER_HIP_WL_ALN_G, MANNER_HIP_WL_ALN_B, MANNER_HIP_WL_FF1_W, MANNER_HIP_WL_FF1_B,
  MANNER_HIP_WL_FF2_W, MANNER_HIP_WL_FF2_B, MANNER_HIP_WL_OLN_G, MANNER_HIP_WL_OLN_B,
  MANNER_HIP_WL_COUNT
};

int manner_hip_abi_version(void);
const char* manner_hip_last_error(void);

/* ---------------------------------------------------------------- text encoder (K1-K7)
 * Replaces MannerTextEncoder.__init__/forward — manner/models/components/news_encoder.py:11-37 —
 * i.e. `AutoModel.from_pretrained(plm)(**tokenized).last_hidden_state[:, 0, :]`, eval mode. */

/* Pack the PLM weights (fp32 device pointers, table order above, MANNER_HIP_W_EMB_COUNT +
 * layers*MANNER_HIP_WL_COUNT entries) into the handle's private bf16 and/or fp32 GEMM layouts.
 * `precisions` is a bit mask (1 << MANNER_HIP_PREC_*) of the modes the handle must serve.
 * Synchronises `stream` before returning; the source tensors may be freed afterwards. */
int manner_hip_encoder_create(const manner_hip_encoder_config* cfg, const float* const* weights /*host*/,
                              int32_t n_weights, uint32_t precisions, manner_hip_stream_t stream,
                              manner_hip_encoder_t* out /*host*/);
int manner_hip_encoder_destroy(manner_hip_encoder_t enc);

/* Bytes of scratch manner_hip_encode_cls needs to process `max_tokens` packed tokens and
 * `max_news` news per internal chunk. */
size_t manner_hip_encoder_workspace_bytes(manner_hip_encoder_t enc, int64_t max_news, int64_t max_tokens,
                                          int32_t precision);

/* CLS embeddings of `n_news` tokenised news.
 *   ids, mask : int64 [n_news, padded_len] row-major, exactly the tensors of the reference's
 *               BatchEncoding (manner/data/components/mind_rec_dataset.py:134-137); `mask` must be
 *               a right-padded 0/1 prefix mask with >= 1 real token per news (validated on device,
 *               see manner_hip_encoder_status);
 *   host_lengths : optional host int32 [n_news] copy of the row sums of `mask`.  With it the
 *               launch grids are exact; without it (NULL) grids cover n_news*padded_len tokens
 *               and surplus workgroups exit early.  No host synchronisation either way.  The lengths only
 *               size the chunks: the device derives its own from `mask`, and if the two disagree it raises
 *               MANNER_HIP_STATUS_LENGTHS and drops the tokens beyond the chunk's bound instead of writing
 *               past the workspace;
 *   out       : float32 [n_news, H].
 * News are processed in internal chunks that fit the workspace; padding tokens cost no FLOPs
 * (tokens are packed; SURVEY.md Q5 makes this equivalent to the padded reference computation). */
int manner_hip_encode_cls(manner_hip_encoder_t enc, const int64_t* ids, const int64_t* mask,
                          const int32_t* host_lengths /*host, nullable*/, int64_t n_news, int64_t padded_len,
                          int32_t precision, float* out, void* workspace, size_t workspace_bytes,
                          manner_hip_stream_t stream);

/* Hidden states after the first n_layers encoder layers — HF's hidden_states[n_layers] of the model built at
 * manner/models/components/news_encoder.py:20 (n_layers = 0: embedding output; = layers: last_hidden_state).
 * n_layers = 8 is the frozen / trainable boundary of the shipped configs (frozen_layers [0..7],
 * configs/model/cr_module.yaml:10; news_encoder.py:24-27): these activations do not change across training epochs and can
 * be cached per news — provided the embedding tables, which that name test does not freeze, are
 * not trained either.  Arguments as manner_hip_encode_cls; out [n_news, padded_len, H] of
 * out_dtype (0 = f32, 1 = bf16).  Rows of padded positions are written as zeros: HF computes throw-away values there
 * which never reach a real token (keys are masked) — feed the tensor with the same attention_mask. */
int manner_hip_encode_hidden(manner_hip_encoder_t enc, const int64_t* input_ids, const int64_t* attention_mask,
                             const int32_t* host_lengths, int64_t n_news, int64_t padded_len, int32_t precision,
                             int32_t n_layers, int32_t out_dtype, void* out, void* workspace, size_t workspace_bytes,
                             manner_hip_stream_t stream);

/* Blocking: synchronises `stream` and returns MANNER_HIP_E_INPUT if any encode_cls call on this
 * handle since the last status call saw an invalid mask (non-prefix, empty or longer than
 * MANNER_HIP_MAX_LEN); clears the flag. */
int manner_hip_encoder_status(manner_hip_encoder_t enc, manner_hip_stream_t stream);
/* Non-blocking form: enqueues a copy of the flag word into `host_flag` (pinned host int32) followed by its reset,
 * both on `stream`; the caller reads *host_flag once an event recorded after this call has completed.  This is how
 * the module mirror surfaces bad inputs of call k at call k+1 without a host synchronisation per forward. */
int manner_hip_encoder_status_async(manner_hip_encoder_t enc, int32_t* host_flag /*pinned host*/, manner_hip_stream_t stream);
#define MANNER_HIP_MAX_LEN 128

/* ABI v8 — sampled fingerprint of a set of tensors, for hosts that cache copies of caller-owned parameters (the module mirror's
 * inference handle packs the PLM weights once; torch's version counters do not see a write through `p.data`).  out[i] = a 32-bit
 * hash of min(counts[i], samples) evenly strided 32-bit words of tensors[i] (first and last word included) mixed with counts[i]:
 * a bulk rewrite of a tensor (scaling, a copy, an optimiser step through .data) changes its word; a poke into single elements
 * between the samples does not — this is a tripwire, not a checksum.  tensors / counts / out are DEVICE arrays of n entries;
 * one launch, no host synchronisation. */
int manner_hip_fingerprint(const void* const* tensors /*device*/, const int64_t* counts /*device*/, int32_t n, int32_t samples,
                           uint32_t* out /*device*/, manner_hip_stream_t stream);

/* Optional per-kernel timing (the reference delegates profiling to Lightning's `profiler: simple`,
 * configs/trainer/default.yaml:21; this is the build's equivalent for the roofline report).
 * While enabled, every launch inside encode_cls is bracketed by hipEvents on the launch stream.
 * profile_read synchronises `stream`, adds the elapsed times into ms[MANNER_HIP_PROF_COUNT] and the
 * launch counts into launches[...] (host arrays, accumulated since the last read) and resets. */
enum {
  MANNER_HIP_PROF_LENGTHS = 0, MANNER_HIP_PROF_EMBED, MANNER_HIP_PROF_GEMM_QKV, MANNER_HIP_PROF_ATTENTION,
  MANNER_HIP_PROF_GEMM_OUT, MANNER_HIP_PROF_LAYERNORM, MANNER_HIP_PROF_GEMM_FFN1, MANNER_HIP_PROF_GEMM_FFN2,
  MANNER_HIP_PROF_GATHER, MANNER_HIP_PROF_CLS_TAIL /* last layer's [CLS]-row-only launches */,
  MANNER_HIP_PROF_COUNT
};
int manner_hip_encoder_profile(manner_hip_encoder_t enc, int32_t enable);
int manner_hip_encoder_profile_read(manner_hip_encoder_t enc, manner_hip_stream_t stream, double* ms /*host*/,
                                    int64_t* launches /*host*/);

/* ---------------------------------------------------------------- pooler (K11)
 * Replaces AdditiveAttention.forward — manner/models/components/attention.py:12-29 — and through
 * it NAMLUserEncoder.forward — manner/models/components/user_encoder.py:17-21.
 *   x [B,S,D] f32 contiguous; lin_w [Q,D]; lin_b [Q]; query [Q]; out [B,D] f32.
 * No padding mask, as in the reference.  scratch: B*S floats. */
int manner_hip_additive_pool(const float* x, const float* lin_w, const float* lin_b, const float* query,
                             int64_t B, int64_t S, int32_t D, int32_t Q, float* out, float* scratch,
                             manner_hip_stream_t stream);
/* ABI v5 — the same operator with ONE pass over x (csrc/pool.hip): the rows of x stay on the CU (registers, IEEE-half hi/lo pairs
 * under a power-of-two row scale = 22 significant bits) between the logits — split (x3) products on the f16 matrix pipe: W.hi x.hi +
 * W.hi x.lo + W.lo x.hi, f32 accumulation; tanh by exp / rcp — and the softmax-weighted sum; x is read from HBM once.  Pooled vectors
 * within 1e-4 of the reference (measured 2e-7 on tests/golden/additive_attention.npz incl. the zero-padded Q2 case, 1.1e-5 on a
 * peaked-softmax stress input).  Runs when D = 768 (S <= 128) or D = 1024 (S <= 64),
 * Q <= 320 and x / out are 16-byte aligned; otherwise — or with strict != 0, or MANNER_HIP_POOL_STRICT=1 in the environment — the
 * exact-f32 two-pass path of manner_hip_additive_pool runs (logits on the f32 matrix pipe, x read twice).
 *   workspace: manner_hip_additive_pool_workspace_bytes(B, S, D, Q) bytes, 256-byte aligned (packed W fragments / the logits). */
size_t manner_hip_additive_pool_workspace_bytes(int64_t B, int64_t S, int32_t D, int32_t Q);
int manner_hip_additive_pool_fused(const float* x, const float* lin_w, const float* lin_b, const float* query,
                                   int64_t B, int64_t S, int32_t D, int32_t Q, float* out, void* workspace,
                                   size_t workspace_bytes, int32_t strict, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- entity branch (K8)
 * manner_hip_entity_encode replaces MannerEntityEncoder.forward — manner/models/components/news_encoder.py:60-72:
 * Embedding -> nn.MultiheadAttention(D, heads) -> AdditiveAttention(D -> Q), eval mode.  BATCH-FAITHFUL
 * to the reference: the MHA is batch_first=False but receives [N, E, D], so attention
 * runs across the N news of the call at each entity slot, without key_padding_mask; the result for
 * a news depends on the other news of the call, exactly as in the reference.
 *   entity_ids int64 [N, E] (0 = padding slot, looked up like any other row); table f32 [n_entities, D];
 *   in_proj_w [3D, D], in_proj_b [3D], out_proj_w [D, D], out_proj_b [D] (nn.MultiheadAttention keys);
 *   pool_w [Q, D], pool_b [Q], pool_q [Q] (AdditiveAttention keys); out f32 [N, D].
 * manner_hip_linear replaces the nn.Linear(H + D -> H) on cat[text, entity] — news_encoder.py:109-113,
 * 122-124: y[R, O] = x[R, K] weight[O, K]^T + bias[O] (bias may be NULL). */
size_t manner_hip_entity_workspace_bytes(int64_t N, int64_t E, int32_t D);
int manner_hip_entity_encode(const int64_t* entity_ids, int64_t N, int64_t E, const float* table, int64_t n_entities,
                             int32_t D, int32_t heads, const float* in_proj_w, const float* in_proj_b,
                             const float* out_proj_w, const float* out_proj_b, const float* pool_w,
                             const float* pool_b, const float* pool_q, int32_t Q, float* out, void* workspace,
                             size_t workspace_bytes, int32_t* status /*device, nullable*/, manner_hip_stream_t stream);
int manner_hip_linear(const float* x, const float* weight, const float* bias, int64_t R, int32_t K, int32_t O,
                      float* y, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- scorer (K12, K9+K10+K12)
 * Replaces DotProduct.forward — manner/models/components/click_predictors.py:9-12:
 * out[b,c] = <user[b,:], cand[b,:,c]>.  `cand` is addressed with element strides so that the
 * reference call site's permuted view (manner/models/cr_module.py:127-129) is read in place. */
int manner_hip_dot(const float* user, const float* cand, int64_t B, int64_t C, int32_t D,
                   int64_t cand_stride_b, int64_t cand_stride_d, int64_t cand_stride_c,
                   float* out /*[B,C]*/, manner_hip_stream_t stream);

/* Fused late-fusion scorer over a news-embedding table: replaces the tail of CRModule.forward —
 * manner/models/cr_module.py:108-131 with late_fusion=True — and of
 * EnsembleModule._submodel_forward — manner/models/ensemble_module.py:116-135:
 * user_i = sum(table[hist_idx[hist_off[i]:hist_off[i+1]]]) / h_i ;
 * out[j] = <user_i, table[cand_idx[j]]> for j in [cand_off[i], cand_off[i+1]).
 * table f32 [n_rows, D] (D % 4 == 0, D <= 3072); idx int32; off int64 [B+1]; out f32 [cand_off[B]] (ragged order).
 * An index outside [0, n_rows) raises MANNER_HIP_STATUS_INDEX in *status (device int32, nullable) where the
 * reference's gather raises IndexError. */
int manner_hip_score_late_fusion(const float* table, int64_t n_rows, int32_t D,
                                 const int32_t* hist_idx, const int64_t* hist_off,
                                 const int32_t* cand_idx, const int64_t* cand_off, int64_t B,
                                 float* out, int32_t* status, manner_hip_stream_t stream);

/* The same scorer (same reference lines: cr_module.py:108-131, ensemble_module.py:116-135) over an IEEE-half copy of the
 * table, table16 [n_rows, D] (D % 8 == 0): half the bytes per gathered row, and a MIND-large table (161 013 x 768 = 247 MB)
 * that stays resident in the 256 MiB Infinity Cache.  Accumulation, the user vector and the scores are f32; only the
 * stored rows are rounded.  mean == NULL: table16 = half(T).  mean != NULL (f32 [D]): table16 = half(T - mean), the rows
 * centred on the table's column mean — the tables of one encoder are nearly collinear, so the deviations are an order of
 * magnitude smaller than the entries and the rounding error of the scores shrinks alike; the scores are reassembled
 * exactly (<mean + u', mean + c'> = <w, mean> + <w, c'>, w = mean + u').  For tables produced by the 16-bit encoder
 * modes; the fp32 parity mode keeps the f32 table.
 * manner_hip_table_to_f16 makes the copy; with mean != NULL it first computes the column mean into `mean` (workspace:
 * manner_hip_table_to_f16_workspace_bytes(D) bytes). */
int manner_hip_score_late_fusion_f16(const void* table16, const float* mean, int64_t n_rows, int32_t D,
                                     const int32_t* hist_idx, const int64_t* hist_off,
                                     const int32_t* cand_idx, const int64_t* cand_off, int64_t B,
                                     float* out, int32_t* status, manner_hip_stream_t stream);
size_t manner_hip_table_to_f16_workspace_bytes(int32_t D);
int manner_hip_table_to_f16(const float* table, int64_t n_rows, int32_t D, float* mean /*nullable*/, void* table16,
                            void* workspace, size_t workspace_bytes, manner_hip_stream_t stream);

/* Same scorer with the user vectors given: the early-fusion tail of CRModule.forward —
 * manner/models/cr_module.py:125-129 (user_vector = user_encoder(...), then the click predictor) — on ragged
 * candidates: out[j] = <user[i, :], table[cand_idx[j]]> for j in [cand_off[i], cand_off[i+1]).  user f32 [B, D]. */
int manner_hip_score_user(const float* table, int64_t n_rows, int32_t D, const float* user,
                          const int32_t* cand_idx, const int64_t* cand_off, int64_t B, float* out,
                          int32_t* status, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- ragged -> dense (K9)
 * Replaces torch_geometric.utils.to_dense_batch at its call sites — manner/models/cr_module.py:108-110,114,142;
 * manner/models/ensemble_module.py:116-124,155-163: dense[b, j, :] = x[off[b] + j, :] for j < off[b+1] - off[b],
 * every other slot of row b = fill[b] (fill == NULL: 0, what to_dense_batch writes); mask[b, j] = 1 on real slots.
 * x f32 [off[B], D]; dense f32 [B, width, D]; mask uint8 [B, width] (nullable).  `width` comes from the caller
 * (the collate knows the batch maximum), so nothing is read back to the host; items beyond `width` are dropped.
 * `fill` carries the value the reference's dense ensemble matrix holds in padded slots (see zscore_fuse). */
int manner_hip_to_dense(const float* x, const int64_t* off, int64_t B, int64_t width, int32_t D, const float* fill,
                        float* dense, uint8_t* mask, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- ensemble (K13+K14)
 * Replaces the z-normalisation of EnsembleModule._submodel_forward —
 * manner/models/ensemble_module.py:138-149 — and the fusion of EnsembleModule.forward — :95-109:
 * out = z(scores[0]) + sum_{k>=1, w[k-1] != 0} w[k-1] * z(scores[k]), z per impression with the
 * unbiased std (c_i == 1 gives NaN, as torch.std does).
 * scores: K planes of f32 [total] at stride `plane_stride` elements; weights host f32 [K-1].
 * pad_value f32 [B] (nullable): the value the reference's dense [B, Cmax] result holds in the PADDED slots of row i —
 * its z-score runs over the whole zero-padded row (:145-149), so they become sum_k w_k (0 - mean_ik) / std_ik, not 0.
 * Feed it to manner_hip_to_dense as `fill` to reproduce the reference's matrix slot for slot. */
int manner_hip_zscore_fuse(const float* scores, int64_t plane_stride, int32_t K, const float* weights /*host*/,
                           const int64_t* cand_off, int64_t B, float* out, float* pad_value,
                           manner_hip_stream_t stream);

/* ---------------------------------------------------------------- ranking / nDCG (K15)
 * Replaces what RetrievalNormalizedDCG(top_k=k) computes per impression (constructed at
 * manner/models/cr_module.py:83-84, fed at :267-273): stable descending rank of each candidate.
 * topk_idx int32 [B,k]: position (within the impression) of the r-th ranked candidate, -1 padded.
 * ndcg f32 [B]: DCG@k/IDCG@k, 0 for impressions without a positive label.
 * mrr  f32 [B]: 1/(1 + rank of the best-ranked positive) over ALL candidates (RetrievalMRR, constructed at
 *               manner/models/cr_module.py:82), 0 without a positive.  Any output may be NULL. */
int manner_hip_rank_ndcg(const float* scores, const float* labels, const int64_t* cand_off, int64_t B,
                         int32_t k, int32_t* topk_idx, float* ndcg, float* mrr, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- Phase C in one launch (ABI v5)
 * EnsembleModule.forward over K per-module tables — manner/models/ensemble_module.py:95-151 — and the ranking consumer of
 * cr_module.py:267-273 in ONE kernel: for every impression gather-mean-dot against each active module's table (weights[k-1] == 0
 * skips module k as the reference does), per-impression z-score and weighted fusion (K == 1: the CR-Module's raw late-fusion scores,
 * cr_module.py:105-131), stable ranking, top-k, nDCG@k, MRR.  The K score planes stay in LDS.  BIT-IDENTICAL to
 * manner_hip_score_late_fusion x K -> manner_hip_zscore_fuse -> manner_hip_rank_ndcg (same code, same order of operations).
 *   tables: K HOST pointers to device tables f32 [n_rows, D] (D = 768 or 1024); weights: K-1 host floats; scores f32 [total_candidates]
 *   out; pad_value [B] or NULL; topk_idx int32 [B, k] or NULL; ndcg / mrr f32 [B] or NULL (need labels); workspace
 *   manner_hip_score_fuse_rank_workspace_bytes(K, total_candidates) bytes (touched only by impressions with more than 320 candidates). */
size_t manner_hip_score_fuse_rank_workspace_bytes(int32_t K, int64_t total_candidates);
int manner_hip_score_fuse_rank(const float* const* tables, int32_t K, const float* weights, int64_t n_rows, int32_t D,
                               const int32_t* hist_idx, const int64_t* hist_off, const int32_t* cand_idx, const int64_t* cand_off,
                               int64_t B, int64_t total_candidates, const float* labels, int32_t k, float* scores, float* pad_value,
                               int32_t* topk_idx, float* ndcg, float* mrr, void* workspace, size_t workspace_bytes, int32_t* status,
                               manner_hip_stream_t stream);

/* ---------------------------------------------------------------- aspect metrics
 * Replaces Diversity / Personalization @k — manner/metrics/functional.py:8-28, 31-62, 65-70 grouped per
 * impression as manner/metrics/base.py:92-129 and torchmetrics RetrievalMetric.compute do (constructed at
 * manner/models/ensemble_module.py:56-84).  topk_idx is the output of manner_hip_rank_ndcg for the same k.
 *   cand_aspect int32 [total candidates], hist_aspect int32 [total history] : class ids in [0, num_classes)
 *   diversity f32 [B]      : entropy of the class distribution of the top-k / ln(num_classes)
 *   personalization f32 [B]: generalised Jaccard of the top-k class counts vs the history class counts
 * Both are 0 for an impression whose candidate class ids sum to 0 (reference behaviour). Either may be NULL. */
int manner_hip_aspect_metrics(const int32_t* topk_idx, const int32_t* cand_aspect, const int32_t* hist_aspect,
                              const int64_t* cand_off, const int64_t* hist_off, int64_t B, int32_t k,
                              int32_t num_classes, float* diversity, float* personalization,
                              manner_hip_stream_t stream);

/* ---------------------------------------------------------------- global AUC
 * Replaces AUROC(task="binary") — constructed at manner/models/cr_module.py:81 (and ensemble_module.py), fed the
 * ragged preds / targets of every impression at cr_module.py:267-273; the arithmetic is torchmetrics'
 * (>=0.11.4, requirements.txt:4): ONE curve over all pairs of the run, not a per-impression mean.
 *   scores f32 [n], labels f32 [n] (positive iff > 0.5), n < 2^31.
 *   sigmoid_rule != 0 reproduces torchmetrics' format step: if any score lies outside [0, 1] every score is
 *   passed through the logistic function (f32) before thresholds are formed, so saturated scores tie.
 *   auc f64 [1] (device): (2U) / (2 P N), U = #(pos > neg) + #(pos == neg)/2, i.e. the trapezoid area through the
 *   distinct thresholds; 0 when there is no positive or no negative.
 *   counts int64 [3] (device, may be NULL): {2U, P, N} — exact integers, so ranks can be combined across
 *   shards only by re-running on the gathered scores (AUC does not decompose over impressions).
 * Either of auc / counts may be NULL, not both. */
size_t manner_hip_auc_workspace_bytes(int64_t n);
int manner_hip_auc(const float* scores, const float* labels, int64_t n, int32_t sigmoid_rule, void* workspace,
                   size_t workspace_bytes, double* auc, int64_t* counts, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- device-side collate
 * Replaces the per-step host work of MINDCollate.__call__ — manner/data/components/mind_rec_dataset.py:114-137
 * (pd.concat of the impressions' news rows + tokenizer call + padding) — and of MINDRecDatasetTest.__getitem__
 * (:87-99, DataFrame.loc per impression).  The news are tokenised ONCE into a device-resident store:
 *   store_ids int32 [n_news, Ls] (tokenizer output truncated to tokenizer_max_length), store_len int32 [n_news];
 *   store_ent int32 [n_news, Es] entity indices, store_cnt int32 [n_news]; category / sentiment int32 [n_news],
 *   sentiment_score f32 [n_news]  (columns of the parsed news frame, mind_dataframe.py:245-246).
 * `rows` int32 [M] are the store rows of the batch's concatenated history (or candidate) news.
 *   collate_segments: _make_batch_assignees (:171-174) — seg int64 [total] = repeat_interleave(arange(B), sizes)
 *                     from the int64 offsets [B+1] (this is MINDRecBatch.batch_hist / batch_cand).
 *   collate_text    : BatchEncoding of tokenizer(padding=True, truncation=True) (:134-137): ids / mask int64 [M, Lp],
 *                     Lp = longest news of the batch (the host knows the lengths), pad_id outside, mask 1/0.
 *   collate_entities: _tokenize_entities (:139-144): int64 [M, E], right-padded with 0 to the batch max E.
 *   collate_aspects : category / sentiment int64 [M], sentiment_score f32 [M] (:164-168); any output may be NULL. */
int manner_hip_collate_segments(const int64_t* off, int64_t B, int64_t total, int64_t* seg, manner_hip_stream_t stream);
int manner_hip_collate_text(const int32_t* store_ids, const int32_t* store_len, int64_t n_news, int32_t Ls,
                            const int32_t* rows, int64_t M, int32_t Lp, int32_t pad_id, int64_t* ids, int64_t* mask,
                            manner_hip_stream_t stream);
int manner_hip_collate_entities(const int32_t* store_ent, const int32_t* store_cnt, int64_t n_news, int32_t Es,
                                const int32_t* rows, int64_t M, int32_t E, int64_t* out, manner_hip_stream_t stream);
int manner_hip_collate_aspects(const int32_t* category, const int32_t* sentiment, const float* sentiment_score,
                               int64_t n_news, const int32_t* rows, int64_t M, int64_t* out_category,
                               int64_t* out_sentiment, float* out_score, manner_hip_stream_t stream);

/* ---------------------------------------------------------------- evaluation loss (val/loss, test/loss)
 * Replaces the loss of CRModule.model_step — manner/models/cr_module.py:140-171, run by validation_step and test_step
 * (:211-262) on every batch — per impression, on the ragged scores (no dense [B, Cmax] matrix, no host loops).
 *   mode 0 (supcon_loss: True, configs/model/cr_module.yaml:4): the reference's SupConLoss on the score matrix
 *           (manner/models/components/losses.py:12-40): losses[i] = -sum_{j positive}(s_j/T - logsumexp_{j real} s_j/T)
 *           / (n_pos + FLT_MIN); 0 for an impression without a positive.  The batch value is the reducer's job
 *           (pytorch_metric_learning default for SupConLoss: mean over the non-zero entries).
 *   mode 1 (supcon_loss: False): nn.CrossEntropyLoss(scores [B, c_max], y_true [B, c_max]) with probability targets —
 *           the c_max - c_i zero-padded scores of the dense row take part in the softmax, as in the reference;
 *           losses[i] = -sum_j y_ij log_softmax(row_i)_j, batch value = mean.  temperature is ignored (pass 1).
 * scores / labels f32 [cand_off[B]], cand_off int64 [B+1], losses f32 [B]. */
int manner_hip_eval_loss(const float* scores, const float* labels, const int64_t* cand_off, int64_t B, int32_t mode,
                         float temperature, int64_t c_max, float* losses, manner_hip_stream_t stream);

/* ---- the training path of MannerTextEncoder (train.hip) -------------------------------------------------
 * Replaces, for the CR-/A-Module training step (manner/models/cr_module.py:140-171 -> news_encoder.py:29-37 in train()
 * mode, then loss.backward()): the HF BertModel / RobertaModel forward WITH its dropouts (embeddings, attention
 * probabilities, attention output, FFN output: modeling_bert.py:107,140,183,351), MannerTextEncoder's own dropout on the
 * [CLS] vector (news_encoder.py:35), and the autograd backward down to the parameter gradients.
 *
 *   weights : HOST array of n_weights device pointers (f32, the order of manner_hip_encoder_create) — the live master
 *             parameters, read at every call (nothing is cached between optimiser steps);
 *   ids / mask / n_news / padded_len : as manner_hip_encode_cls (padded_len <= MANNER_HIP_MAX_LEN);
 *   m_bound : rows every activation buffer holds: a multiple of 256, >= the number of real tokens (n_news * padded_len
 *             rounded up always works; a device-side check raises MANNER_HIP_STATUS_LENGTHS otherwise);
 *   precision : MANNER_HIP_PREC_F32 (f32 MFMA GEMMs) or _F16 / _BF16 (GEMM operands rounded to 16 bits, f32 accumulation,
 *             f32 activations — "16-mixed"); activations, gradients and everything that is not a GEMM are f32;
 *   start_layer / prefix_hidden : 0 / NULL runs embeddings + all layers in training mode.  start_layer = f > 0 starts
 *             from prefix_hidden = hidden_states[f] ([n_news, padded_len, H] f32, e.g. manner_hip_encode_hidden of the
 *             frozen layers) — valid when no tensor below layer f receives a gradient;
 *   p_hidden / p_attn / p_out : hidden_dropout_prob, attention_probs_dropout_prob, MannerTextEncoder.dropout.p; 0 = off;
 *   seed    : dropout masks are a pure function of (seed, site, element index); the backward call must repeat the
 *             forward call's seed and probabilities.  manner_hip_dropout_mask returns the keep-bits of one site
 *             (site 0: embeddings over [m, H]; 1: [CLS] output over [n_news, H]; 8*(layer+1)+0: attention
 *             probabilities over [(m*heads + head)*256 + key]; +1: attention output; +2: FFN output, over [m, H]);
 *   saved   : manner_hip_train_saved_bytes(cfg, n_news, m_bound, start_layer) bytes that carry the activations from the
 *             forward to the backward call;  workspace: manner_hip_train_workspace_bytes(cfg, m_bound) bytes of scratch;
 *   cls_out : f32 [n_news, H].
 * Backward: grad_cls f32 [n_news, H]; grads = HOST array of n_weights device pointers, NULL where no gradient is
 * wanted (requires_grad = False; a LayerNorm's weight and bias come together); each non-NULL tensor is OVERWRITTEN with
 * d loss / d parameter.  The activation gradient travels as far down as a requested gradient needs — through frozen
 * layers into the embedding tables when an embedding tensor is trainable, which is the reference's default
 * (news_encoder.py:24-27 freezes "layer.N." parameters only).  grad_prefix (optional, start_layer > 0): f32
 * [n_news, padded_len, H] gradient of prefix_hidden.  Embedding-table gradients use f32 atomics (summation order, hence
 * the last bits, vary between runs — as torch's CUDA embedding backward). */
size_t manner_hip_train_saved_bytes(const manner_hip_encoder_config* cfg, int64_t n_news, int64_t m_bound, int32_t start_layer);
size_t manner_hip_train_workspace_bytes(const manner_hip_encoder_config* cfg, int64_t m_bound);
/* ABI v7 — the bytes manner_hip_train_forward needs for `saved` IN THE GIVEN PRECISION (never more than the function above, which
 * stays valid for every mode).  Round 5: in the 16-bit modes the tensors that only GEMMs and the attention consume — Q | K | V, the
 * attention output, the FFN pre-activation and its gelu — are saved in the 16-bit type alone (what the reference's
 * `precision: 16-mixed` autocast keeps: configs/trainer/default.yaml:12), 31 instead of 49 KB per token and layer for bert-base.
 * Evaluate it right before the forward call: the layout follows the same rule the forward applies (matrix-pipe attention and fused
 * GeLU epilogues available for the shape; MANNER_HIP_TRAIN_SAVE16=0 keeps the f32 layout); a forward whose rule asks for more than
 * it was given fails with MANNER_HIP_E_WORKSPACE, it never overruns. */
size_t manner_hip_train_saved_bytes_for(const manner_hip_encoder_config* cfg, int64_t n_news, int64_t m_bound, int32_t start_layer,
                                        int32_t precision);
/* ABI v5 — optional cache of the 16-bit weight copies the 16-bit training modes make on every call (round 4).  Registers, for the NEXT
 * manner_hip_train_forward / _backward call of THIS thread (consumed by it; ignored in fp32 mode), 2 * n_weights caller-owned device
 * buffers (host array `slots`, entries may be NULL = not cached) and as many host flags `valid` (in / out):
 *   slot 2 i     the mode's 16-bit copy of weight i of the table, same layout          (read by the forward GEMMs)
 *   slot 2 i + 1 the 16-bit copy of its transpose                                      (read by the data-gradient GEMMs)
 *   at a layer's query weight: the packed [3H, H] Q | K | V copy / its [H, 3H] transpose; at its query bias, slot 2 i: f32 [3H] biases.
 * valid[s] == 0: the library fills slot s on first use and sets the flag; != 0: it reads the slot and launches nothing.  Meant for
 * FROZEN weights (reference news_encoder.py:24-27 freezes layers by name): the caller clears a flag when the weight's contents change
 * and keeps the buffers alive until the stream has passed the call.  slots == NULL: no cache (the default). */
int manner_hip_train_weight_cache(void* const* slots /*host*/, int32_t* valid /*host, in/out*/, int32_t n_slots);
int manner_hip_train_forward(const manner_hip_encoder_config* cfg, const float* const* weights /*host*/, int32_t n_weights,
                             const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, int64_t m_bound,
                             int32_t precision, int32_t start_layer, const float* prefix_hidden, float p_hidden, float p_attn,
                             float p_out, uint64_t seed, float* cls_out, void* saved, size_t saved_bytes, void* workspace,
                             size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream);
int manner_hip_train_backward(const manner_hip_encoder_config* cfg, const float* const* weights /*host*/, int32_t n_weights,
                              const int64_t* ids, int64_t n_news, int64_t padded_len, int64_t m_bound, int32_t precision,
                              int32_t start_layer, float p_hidden, float p_attn, float p_out, uint64_t seed,
                              const float* grad_cls, void* saved, size_t saved_bytes, float* const* grads /*host*/,
                              float* grad_prefix, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream);
/* ABI v8 — the layout of `saved` travels WITH the call pair instead of being guessed.  manner_hip_train_layout_last() returns the
 * layout word of the most recent successful manner_hip_train_forward / _full_forward of THIS thread (bit 0: attention on the matrix
 * pipe with 16-bit Q | K | V, bit 1: 16-bit-only saved activations; -1: no forward yet); manner_hip_train_layout_next(word) hands it
 * to the NEXT manner_hip_train_backward / _full_backward of this thread (consumed by it, like manner_hip_train_weight_cache).  A host
 * keeps the word next to the saved buffer (the mirror: in the autograd ctx).  A backward that gets no word falls back to the
 * per-address record its forward left in this process image; one that finds neither FAILS with MANNER_HIP_E_INVALID rather than
 * deriving a layout from the environment of the moment (which may differ from the forward's and would plan other slot offsets). */
int32_t manner_hip_train_layout_last(void);
int manner_hip_train_layout_next(int32_t word);
int manner_hip_dropout_mask(uint64_t seed, uint32_t site, float p, int64_t n, uint8_t* keep, manner_hip_stream_t stream);

/* The same training path with "full rows" — the PLM inside PLMTextEncoder in train() mode (manner/models/components/
 * news_encoder.py:132-171: `self.plm_model(**tokenized_text)[0]` = HF last_hidden_state, whose PADDED positions the
 * reference's un-masked MultiheadAttention / AdditiveAttention mix into the result; trained by baselines/
 * nrms_plm_module.py:119-135): every position of the padded batch is a row (m = n_news * padded_len), the real tokens
 * of a news are its attention keys, padded positions embed the pad token (RoBERTa: at position pad_id) and still
 * produce outputs, no layer is pruned to the [CLS] rows.  hidden / grad_hidden: f32 [n_news * padded_len, H] (row
 * n * padded_len + t).  saved: manner_hip_train_saved_bytes(cfg, n_news, M, 0), workspace:
 * manner_hip_train_workspace_bytes(cfg, M), M = n_news * padded_len rounded up to 256.  Dropout sites and the grads
 * table as in manner_hip_train_forward / _backward (there is no [CLS] dropout here: the caller owns what follows). */
int manner_hip_train_full_forward(const manner_hip_encoder_config* cfg, const float* const* weights /*host*/, int32_t n_weights,
                                  const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t precision,
                                  float p_hidden, float p_attn, uint64_t seed, float* hidden, void* saved, size_t saved_bytes,
                                  void* workspace, size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream);
int manner_hip_train_full_backward(const manner_hip_encoder_config* cfg, const float* const* weights /*host*/, int32_t n_weights,
                                   const int64_t* ids, int64_t n_news, int64_t padded_len, int32_t precision, float p_hidden,
                                   float p_attn, uint64_t seed, const float* grad_hidden, void* saved, size_t saved_bytes,
                                   float* const* grads /*host*/, void* workspace, size_t workspace_bytes,
                                   manner_hip_stream_t stream);

/* The rest of the training step (cr_module.py:105-171), f32, ragged order:
 *  - late-fusion scorer on the vectors of every occurrence (training encodes x_hist / x_cand per impression, :107-113):
 *    user_i = mean(hist rows of i) (:116-123), scores_j = <user_i, cand_j> (:127-129); backward: d hist, d cand.
 *    hist f32 [hist_off[B], D], cand f32 [cand_off[B], D], user f32 [B, D] (kept for the backward), scores f32 [cand_off[B]];
 *  - DotProduct backward for the drop-in click predictor (click_predictors.py:9-12): grad_user [B, D], grad_cand
 *    CONTIGUOUS [B, D, C] (the shape of the permuted view the reference passes);
 *  - the loss of model_step (:140-171) with its gradient: mode / temperature / c_max as manner_hip_eval_loss;
 *    losses f32 [B]; loss_and_scale f32 [2] = {batch loss, reducer factor} — SupCon: mean over the non-zero
 *    per-impression losses (pytorch_metric_learning AvgNonZeroReducer); CE: mean; grad_scores f32 [cand_off[B]] =
 *    d loss / d scores (ragged order). */
int manner_hip_late_fusion_train_forward(const float* hist, const int64_t* hist_off, const float* cand, const int64_t* cand_off,
                                         int64_t B, int32_t D, float* user, float* scores, manner_hip_stream_t stream);
int manner_hip_late_fusion_train_backward(const float* grad_scores, const float* user, const int64_t* hist_off, const float* cand,
                                          const int64_t* cand_off, int64_t B, int32_t D, float* grad_hist, float* grad_cand,
                                          manner_hip_stream_t stream);
int manner_hip_dot_backward(const float* grad_out, const float* user, const float* cand, int64_t B, int64_t C, int32_t D,
                            int64_t cand_stride_b, int64_t cand_stride_d, int64_t cand_stride_c, float* grad_user,
                            float* grad_cand, manner_hip_stream_t stream);
int manner_hip_train_loss(const float* scores, const float* labels, const int64_t* cand_off, int64_t B, int32_t mode,
                          float temperature, int64_t c_max, float* losses, float* loss_and_scale, float* grad_scores,
                          manner_hip_stream_t stream);

/* The A-Module's loss (manner/models/a_module.py:73-75,102-108): pytorch_metric_learning (>= 2.1.1, requirements.txt:6 — not
 * vendored by the reference; restated) SupConLoss(temperature, distance=DotProductSimilarity(normalize_embeddings=False)) on the
 * news embeddings emb f32 [N, D] and their aspect labels int64 [N]: mat = emb emb^T / T; positives of anchor i = other news
 * with its label, negatives = news with another label; losses[i] = -sum_pos(m_ij - logsumexp_{j != i} m_ij) / (n_pos + FLT_MIN);
 * loss_and_scale = {mean over the losses > 0 (AvgNonZeroReducer), its factor}; all zero when the batch holds no positive or no
 * negative pair.  grad_emb f32 [N, D] = d loss / d emb.  workspace: manner_hip_supcon_embeddings_workspace_bytes(N). */
size_t manner_hip_supcon_embeddings_workspace_bytes(int64_t N);
int manner_hip_supcon_embeddings(const float* emb, const int64_t* labels, int64_t N, int32_t D, float temperature, float* losses,
                                 float* loss_and_scale, float* grad_emb, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream);

/* The small operators of the training step (train_small.hip; f32): what autograd needs below and beside the text encoder
 * for the reference's default `use_entities: True` (configs/model/cr_module.yaml:13) and for early fusion —
 *  - manner_hip_linear_backward: nn.Linear (news_encoder.py:110-113 `linear` on cat[text, entity]; the projections of
 *    nn.MultiheadAttention; AdditiveAttention.linear): y = x W^T + b, x [R, K], W [O, K]; any of grad_x / grad_w /
 *    grad_b may be NULL; add_to_dx (nullable) [R, K] is added into grad_x;
 *  - manner_hip_additive_pool_backward: AdditiveAttention.forward (attention.py:21-27) — grad_x [B, S, D], grad_w [Q, D],
 *    grad_b [Q], grad_q [Q] from grad_out [B, D]; workspace manner_hip_additive_pool_backward_workspace_bytes;
 *  - manner_hip_axis0_attention(_backward): the attention core of nn.MultiheadAttention(batch_first=False) as the reference
 *    calls it (quirk Q1: along axis 0 of [L0, B1, E], per position B1 and head, no mask, no attention dropout) on projected
 *    qkv [L0, B1, 3E] = [q | k | v]; backward: grad_qkv [L0, B1, 3E] from grad_out [L0, B1, E], stats = L0*B1*heads*3 floats of scratch;
 *  - manner_hip_embedding(_backward): nn.Embedding.from_pretrained(..., freeze=False, padding_idx=0) (news_encoder.py:99-103):
 *    lookup, and grad_table [n_rows, D] (overwritten; row padding_idx receives no gradient; f32 atomics);
 *  - manner_hip_dropout: out = keep ? x / (1 - p) : 0 over a flat tensor with the training path's generator (its own
 *    backward: apply it to the gradient with the same seed / site). */
int manner_hip_linear_backward(const float* x, const float* weight, const float* grad_y, int64_t R, int32_t K, int32_t O,
                               const float* add_to_dx, float* grad_x, float* grad_w, float* grad_b, manner_hip_stream_t stream);
size_t manner_hip_additive_pool_backward_workspace_bytes(int64_t B, int64_t S, int32_t D, int32_t Q);
int manner_hip_additive_pool_backward(const float* x, const float* lin_w, const float* lin_b, const float* query, const float* grad_out,
                                      int64_t B, int64_t S, int32_t D, int32_t Q, float* grad_x, float* grad_w, float* grad_b,
                                      float* grad_q, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream);
int manner_hip_axis0_attention(const float* qkv, int64_t L0, int64_t B1, int32_t E, int32_t heads, float* out, manner_hip_stream_t stream);
int manner_hip_axis0_attention_backward(const float* qkv, const float* grad_out, int64_t L0, int64_t B1, int32_t E, int32_t heads,
                                        float* grad_qkv, float* stats, manner_hip_stream_t stream);
int manner_hip_embedding(const int64_t* ids, int64_t R, const float* table, int64_t n_rows, int32_t D, float* out, int32_t* status,
                         manner_hip_stream_t stream);
int manner_hip_embedding_backward(const int64_t* ids, int64_t R, const float* grad_out, int64_t n_rows, int32_t D, int64_t padding_idx,
                                  float* grad_table, manner_hip_stream_t stream);
int manner_hip_dropout(const float* x, float* out, int64_t n, uint64_t seed, uint32_t site, float p, manner_hip_stream_t stream);

/* ---- the PLM baseline encoders (f32 activations; not on the throughput path) ------------------------------
 * manner_hip_encode_full replaces `self.plm_model(**tokenized_text)[0]` of PLMTextEncoder.forward
 * (manner/models/components/news_encoder.py:158-160): HF last_hidden_state f32 [n_news, padded_len, H] INCLUDING the padded
 * positions — they embed the pad token (RoBERTa: at position pad_id), attend over the real keys only, and the consumer
 * below mixes them into real tokens.  weights / precision / status as manner_hip_train_forward.
 * manner_hip_mha_axis0 replaces nn.MultiheadAttention(embed_dim=E, num_heads=heads) as the reference calls it —
 * batch_first=False on a [batch, seq, E] tensor, no masks (news_encoder.py:163-165; user_encoder.py:35-37): attention along
 * AXIS 0 of x [L0, B1, E] (across the news / users of the call, independently per position B1), in/out projections included.
 * in_proj_w [3E, E], in_proj_b [3E], out_proj_w [E, E], out_proj_b [E]; out [L0, B1, E]; eval mode (no dropout). */
size_t manner_hip_encode_full_workspace_bytes(const manner_hip_encoder_config* cfg, int64_t n_news, int64_t padded_len);
int manner_hip_encode_full(const manner_hip_encoder_config* cfg, const float* const* weights /*host*/, int32_t n_weights,
                           const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t precision,
                           float* hidden, void* workspace, size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream);
size_t manner_hip_mha_axis0_workspace_bytes(int64_t L0, int64_t B1, int32_t E);
int manner_hip_mha_axis0(const float* x, int64_t L0, int64_t B1, int32_t E, int32_t heads, const float* in_proj_w,
                         const float* in_proj_b, const float* out_proj_w, const float* out_proj_b, float* out, void* workspace,
                         size_t workspace_bytes, manner_hip_stream_t stream);

/* ---- content-addressed news-embedding cache (ABI v6, round 4; csrc/cache.hip) ----------------------------------------------
 * SURVEY.md §8(d) mode T ("each unique news encoded once per module") behind the unchanged drop-in call pattern: the reference
 * re-encodes every occurrence — manner/models/cr_module.py:107,113 call manner/models/components/news_encoder.py:29-37 per batch —
 * while in eval() the text encoder is a pure function of a row's real tokens and the weights.  The caller (MannerTextEncoder.forward,
 * opt-in) keys every row, looks the keys up and encodes only the rows whose state is not 0.
 * manner_hip_news_key128: keys uint64 [n_news, 2] of the tokens at mask != 0 (independent of padded_len; keys[n][0] != 0).
 * manner_hip_news_cache_lookup: open-addressing table in CALLER-OWNED device memory — slot_keys uint64 [2, n_slots] (zero-filled =
 *   empty; n_slots a power of two), slot_rows int32 [n_slots], row_count int32 [1] (rows handed out so far; may run past
 *   capacity_rows once the table is full), scratch int32 [2 n_news].  Per input row: rows_out int32 (row of the embedding table, -1 if
 *   none) and state_out int32: 0 = cached (or a duplicate of a key that is new in this call — its row is valid once the state-1
 *   occurrence has been encoded and stored), 1 = new: encode and store at rows_out, 2 = encode without storing (table full).
 *   Exactly one occurrence of a new key gets state 1.  One stream at a time per table. */
int manner_hip_news_key128(const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, uint64_t* keys,
                           manner_hip_stream_t stream);
int manner_hip_news_cache_lookup(const uint64_t* keys, int64_t n_news, uint64_t* slot_keys, int32_t* slot_rows, int64_t n_slots,
                                 int32_t* row_count, int32_t capacity_rows, int32_t* rows_out, int32_t* state_out, int32_t* scratch,
                                 manner_hip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MANNER_HIP_H */
