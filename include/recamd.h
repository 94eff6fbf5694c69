/*
 * recamd.h — C ABI of the MI355X (gfx950) embedding-lookup + feature-interaction library.
 *
 * This is the drop-in boundary for the one hot path of littlemesie/recommend-tf2.0:
 *     sparse id -> per-field Embedding gather -> concat -> interaction layer
 * The reference has no FFI/operator registry; its "plugin API" is the tf.keras Layer.call
 * surface, whose arithmetic lives in TensorFlow ops.  Each entry point below replaces the TF op
 * sequence of one reference call site (cited as file:line relative to the reference repo root).
 *
 * Conventions
 *   - plain C types only; every pointer except `rec_table_desc*` arrays and the ones marked
 *     "host" is a DEVICE pointer owned by the caller; the library never allocates on the hot path.
 *   - all tensors are dense row-major fp32 unless stated; ids are int32 (or fp32 with
 *     ids_dtype = REC_IDS_F32, truncated toward zero exactly like Keras' Embedding cast).
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and returns immediately; it is re-entrant and keeps no global mutable state except
 *     the thread-local last-error string.
 *   - return value: REC_OK or a negative rec_status; nothing throws across this ABI.
 *   - out-of-range ids never fault: the row reads as zeros (TF-GPU gather semantics) and, if
 *     `oob_flag` is non-NULL, *oob_flag is set to 1 so the host can raise (TF-CPU semantics).
 */
#ifndef RECAMD_H_
#define RECAMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REC_VERSION 100 /* 0.1.0 */

typedef enum rec_status {
  REC_OK = 0,
  REC_EINVAL = -1,   /* NULL pointer / bad enum / bad alignment */
  REC_ESHAPE = -2,   /* unsupported or inconsistent shape */
  REC_EOOB = -3,     /* reserved: host-visible out-of-range id (reported through oob_flag) */
  REC_EHIP = -4,     /* a HIP runtime call failed; see rec_last_error */
  REC_ENOTIMPL = -5
} rec_status;

typedef enum rec_ids_dtype { REC_IDS_I32 = 0, REC_IDS_F32 = 1 } rec_ids_dtype;

/* activation applied by Dense-like entry points (Keras activation strings of the reference) */
typedef enum rec_act {
  REC_ACT_NONE = 0,
  REC_ACT_RELU = 1,
  REC_ACT_SIGMOID = 2,
  REC_ACT_TANH = 3,
  REC_ACT_PRELU = 4 /* per-output-channel alpha vector, Keras PReLU (zero-init == relu) */
} rec_act;

/* One embedding table (one tf.keras.layers.Embedding of the reference).
 * `base` is a device pointer to a (vocab, dim) row-major fp32 matrix; `out_col` is the first
 * output column of this field inside the concatenated row (tf.concat(axis=-1) offset). */
typedef struct rec_table_desc {
  const float* base;
  int64_t vocab;
  int32_t dim;
  int32_t out_col;
} rec_table_desc;

#define REC_MAX_TABLES 64 /* per call; callers chunk wider models */

int rec_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
int rec_last_error(char* buf, int n);
/* Kernel selection is by shape only and the library reads NO environment variable.  Tests and A/B scripts may force a
 * kernel variant that a shape would not select: key / value pairs listed in csrc/common.h ("dense" "b"|"f"|"s"|"t",
 * "mha" "f"|"v", "din" "l"|"s", "pairdot" "v", "topk" "f", "cross" "l", ...); value NULL or "" clears the key.
 * Process-global, not thread-safe, never set by the product. */
int rec_debug_force(const char* key, const char* value);

/* ---- a1 / K1: per-field Embedding gather + concat -------------------------------------------
 * Replaces  tf.concat([Embedding_f(sparse_inputs[:, f]) for f], axis=-1)
 *   src/ctr/deep_fm/model.py:53, dcn/model.py:47, dlrm/model.py:45, autoint/model.py:46,
 *   din/model.py:62,66,71-72, match/sasrec/model.py:75-79, match/youtube_dnn/model.py:47,53
 * out[b, tables[f].out_col + c] = tables[f].base[ids[b*ids_stride + f] * dim_f + c]
 * tables: HOST array of F descriptors (copied into the launch).  ids: (B, F) with row stride
 * ids_stride (elements).  out: (B, *) with row stride out_stride (floats). */
int rec_gather_concat_f32(const rec_table_desc* tables, int32_t F,
                          const void* ids, int32_t ids_dtype, int64_t ids_stride,
                          int64_t B, float* out, int64_t out_stride,
                          int32_t* oob_flag, void* stream);

/* Fused K1 + nv dot products per sample: emb_out gets the gathered concat row (as rec_gather_concat_f32) and
 * out_dots[b, v] = <row b, Wd[v, 0:width]> for nv <= 8 weight vectors Wd (nv, width) row-major, width % 4 == 0 and
 * >= the concat width (columns that no table writes must be zero in Wd).  DCN's cross tower in closed form:
 * x_l = alpha_l x0 + sum_{j<l} b_j (src/ctr/layers/modules.py:105-112) only needs d_l = x0 . w_l, and since
 * cross_x only meets Dense(1) afterwards (src/ctr/dcn/model.py:55-56) it is never materialised:
 * rec_dcn_logit_f32 evaluates, per sample, alpha = 1; alpha += alpha d_l + G_l (l < L);
 * out = sigmoid(alpha d_L + c + extra[b])  with the weight-only constants G_l = sum_{j<l} b_j . w_l,
 * c = (sum_j b_j) . w_c + bias, d_L = x0 . w_c and extra = dnn_x . w_d. */
int rec_gather_dots_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                        int64_t ids_stride, const float* Wd, int32_t nv, int32_t width, int64_t B,
                        float* emb_out, int64_t emb_stride, float* out_dots, int32_t* oob_flag, void* stream);
/* The same, also delivering row_absmax[b] = max |element| of sample b's gathered row (B floats, may be NULL) for
 * rec_dense_prep_rs_f32: the DNN that consumes emb_out (src/ctr/dcn/model.py:53) then needs no pass over it. */
int rec_gather_dots_absmax_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                               int64_t ids_stride, const float* Wd, int32_t nv, int32_t width, int64_t B,
                               float* emb_out, int64_t emb_stride, float* out_dots, int32_t* oob_flag,
                               float* row_absmax, void* stream);
int rec_dcn_logit_f32(const float* dots, int32_t L, const float* G, float c, const float* extra, int64_t B,
                      float* out, void* stream);

/* ---- a5 / K5: DLRM pairwise-dot interaction ---------------------------------------------------
 * The reference's DLRM.call (src/ctr/dlrm/model.py:42-54) has no interaction op; this is the
 * interaction of the paper the file cites (src/ctr/dlrm/model.py:7):
 *   X = (B, n, D);  Z = X X^T;  out[b, i*(i-1)/2 + j] = Z[b,i,j]  for n > i > j >= 0.
 * P = n*(n-1)/2 outputs per sample, row stride out_stride. */
int rec_pairwise_dot_f32(const float* x, int64_t B, int32_t n, int32_t D,
                         float* out, int64_t out_stride, void* stream);

/* Fused K1+K5: X rows 0..F-1 are gathered embedding rows (all tables must share dim D), and, when
 * `dense` != NULL, row F is dense[b, 0:D] (the bottom-MLP output; n = F+1).  Writes
 * out[b, 0:P] = strictly-lower-triangle dots and, if append_dense, out[b, P:P+D] = dense[b]
 * (mirrors tf.concat([sparse_part, dense_fea]) of src/ctr/dlrm/model.py:48).
 * PAD COLUMNS: with width = P (+ D), when `out` is 16-byte aligned AND out_stride % 4 == 0 the kernels store whole
 * 16-byte groups, i.e. they also write ZEROS to out[b, width .. roundup4(width) - 1] (e.g. column 479 of a 479-wide
 * result; out_stride >= roundup4(width) is then required).  A caller that places the result inside a wider row whose
 * next column is live data must either start that data at a multiple of 4 columns, or pass an `out` / out_stride that
 * does not satisfy the alignment condition (the tail group is then written with scalar stores, nothing past `width`).
 * rec_pairwise_dot_f32 follows the same rule for its P columns. */
int rec_gather_pairwise_dot_f32(const rec_table_desc* tables, int32_t F,
                                const void* ids, int32_t ids_dtype, int64_t ids_stride,
                                const float* dense, int64_t dense_stride,
                                int64_t B, float* out, int64_t out_stride,
                                int32_t append_dense, int32_t* oob_flag, void* stream);

/* ---- a3 / K3: FM layer (DeepFM wide part), src/ctr/layers/modules.py:57-72 -----------------
 * first: (B, L1) stride first_stride; w: (L1); second: (B, M) stride second_stride.
 * first_order = sum over THE WHOLE BATCH of first @ w (one scalar, modules.py:65)
 * out[b] = first_order + 0.5 * ((sum_j second[b,j])^2 - sum_j second[b,j]^2)
 * workspace: >= rec_fm_layer_workspace_floats(B) floats, caller-owned scratch. */
int64_t rec_fm_layer_workspace_floats(int64_t B);
int rec_fm_layer_f32(const float* first, int64_t first_stride, int32_t L1, const float* w,
                     const float* second, int64_t second_stride, int32_t M,
                     int64_t B, float* out, float* workspace, void* stream);

/* Fused K1+K3 (DeepFM, src/ctr/deep_fm/model.py:53-59): gathers the F rows of every sample into the
 * concat buffer `emb_out` (tables[f].out_col = column of field f; the DNN consumes it) and, in the
 * same pass over the rows, accumulates the FM layer's three sums, so sparse_embed is not re-read:
 *   fm_out[b] = sum_batch(dense @ w[:nd] + sparse_embed @ w[nd:]) + 0.5((sum x)^2 - sum x^2)
 * All tables share one dim D (D/4 a power of two <= 64).  w: (nd + F*D) in concat order
 * [dense, field 0, field 1, ...].  workspace: rec_fm_layer_workspace_floats(B) floats. */
int rec_gather_fm_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                      int64_t ids_stride, const float* dense, int64_t dense_stride, int32_t nd,
                      const float* w, int64_t B, float* emb_out, int64_t emb_stride, float* fm_out,
                      float* workspace, int32_t* oob_flag, void* stream);
/* The same, also delivering row_absmax[b] = max |element| of sample b's dense block and gathered rows (B floats, may be
 * NULL): the row maxima rec_dense_prep_rs_f32 scales by, so that the DNN's first layer (src/ctr/deep_fm/model.py:61) takes
 * the f16x2 kernel without a pass over the concat buffer. */
int rec_gather_fm_absmax_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                             int64_t ids_stride, const float* dense, int64_t dense_stride, int32_t nd,
                             const float* w, int64_t B, float* emb_out, int64_t emb_stride, float* fm_out,
                             float* workspace, int32_t* oob_flag, float* row_absmax, void* stream);

/* ---- a4 / K4: DCN CrossNetwork, src/ctr/layers/modules.py:105-112 --------------------------
 * x_{l+1} = x0 * (x_l . w_l) + b_l + x_l,  l = 0..L-1;  x: (B, dim), w,b: (L, dim) */
int rec_cross_f32(const float* x, int64_t x_stride, int32_t dim, const float* w, const float* b,
                  int32_t L, int64_t B, float* out, int64_t out_stride, void* stream);

/* ---- a2 / K2: ctr FM model in gather form, src/ctr/fm/model.py:34-53 ------------------------
 * The reference builds a (B, nd + sum V_f) one-hot stack and multiplies by w (L,1) and V^T (L,k).
 * Equivalent gather form (never materialises the one-hot; OOB id -> zero one-hot row):
 *   lin  = w0 + sum_d dense[b,d] w[d] + sum_f w[off_f + id_f]
 *   s_k  = sum_d dense[b,d] V[k,d] + sum_f V[k, off_f + id_f]
 *   q_k  = sum_d dense[b,d]^2 V[k,d]^2 + sum_f V[k, off_f + id_f]^2
 *   out  = sigmoid(lin + 0.5 * sum_k (s_k^2 - q_k))
 * V is stored (k, L) row-major exactly like the reference (fm/model.py:29), L = nd + sum vocab.
 * vocab: HOST int64 array of F field cardinalities. */
int rec_fm_onehot_f32(const float* dense, int64_t dense_stride, int32_t nd,
                      const int32_t* ids, int64_t ids_stride, int32_t F, const int64_t* vocab,
                      const float* w0, const float* w, const float* V, int32_t k,
                      int64_t B, float* out, void* stream);

/* ---- K11: Dense (+ folded BatchNorm) -----------------------------------------------------------
 * out = act(x @ W + bias); x: (M, K) stride x_stride, W: (K, N) row-major (Keras kernel layout),
 * bias: (N) or NULL, alpha: (N) PReLU slopes or NULL.  Used for DNN towers
 * (src/ctr/layers/modules.py:129-135, src/match/layers/modules.py:21-26), Conv1D(k=1)
 * (src/match/layers/modules.py:146-149) and the QKV projections. */
int rec_dense_f32(const float* x, int64_t x_stride, const float* W, const float* bias,
                  const float* alpha, int32_t act, int64_t M, int32_t K, int32_t N,
                  float* out, int64_t out_stride, void* stream);

/* Prepared weights for the large-layer path: the bf16x3 kernel splits every fp32 operand into three bf16 planes;
 * for W (constant between optimiser steps) that split can be done once.  `prepared` is an opaque device buffer of
 * rec_dense_prepared_bytes(K, N) bytes filled by rec_dense_prepare_f32; rec_dense_prep_f32 is rec_dense_f32 with
 * that buffer as a hint (NULL allowed; results are identical either way, layers routed to other kernels ignore
 * it).  Re-prepare whenever W changes. */
int64_t rec_dense_prepared_bytes(int32_t K, int32_t N);
int rec_dense_prepare_f32(const float* W, int32_t K, int32_t N, void* prepared, void* stream);
int rec_dense_prep_f32(const float* x, int64_t x_stride, const float* W, const void* prepared,
                       const float* bias, const float* alpha, int32_t act, int64_t M, int32_t K, int32_t N,
                       float* out, int64_t out_stride, void* stream);

/* The same with one float of device workspace per row (row_absmax, M floats): large layers with 16-B aligned x rows and
 * K % 32 == 0 then run on the f16 matrix cores with THREE MFMAs per fp32 product — rows of x and columns of W scaled by
 * exact powers of two into f16's range, two f16 terms each, scaled back in the epilogue; as accurate against fp64 as the
 * other kernels (csrc/dense_f16x2.hip) — instead of the six bf16 MFMAs of the bf16x3 kernels.  absmax_valid != 0: the caller
 * (a producer kernel) already wrote max_k |x[r][k]| into row_absmax; otherwise one pass over x fills it — which pays from
 * N = 384, narrower layers keep the bf16x3 kernels unless the maxima are valid.  out_absmax (optional, M floats): receives
 * max_c |out[r][c]|, i.e. the row_absmax of a layer that consumes `out` (filled by whichever kernel answers; the caller
 * ZEROES it first: the epilogue accumulates with atomic maxima).  row_absmax NULL or a shape the kernel does not cover:
 * rec_dense_prep_f32.  Non-finite inputs poison their own row, non-finite weights their own column.
 * Same reference call sites as rec_dense_f32. */
int rec_dense_prep_rs_f32(const float* x, int64_t x_stride, const float* W, const void* prepared,
                          const float* bias, const float* alpha, int32_t act, int64_t M, int32_t K, int32_t N,
                          float* out, int64_t out_stride, float* row_absmax, int32_t absmax_valid, float* out_absmax,
                          void* stream);

/* ---- a7 / K6: ctr MultiHeadAttention (AutoInt interacting layer) ----------------------------
 * src/ctr/layers/modules.py:285-325.  q = act(Xq Wq), k = act(Xk Wk), v = act(Xv Wv) (no bias),
 * heads (B,H,N,S); P = softmax(q k^T * sqrt(S)) (the reference DIVIDES by S^-0.5, :235-237);
 * out = merge(P v) (B,N,H*S); if W0 != NULL (use_res): out = relu(out + act(Xv W0)).
 * Xq/Xk/Xv: (B, N, din);  W*: (din, H*S). */
int rec_mha_ctr_f32(const float* xq, const float* xk, const float* xv, int64_t B, int32_t N,
                    int32_t din, const float* Wq, const float* Wk, const float* Wv,
                    const float* W0, int32_t H, int32_t S, int32_t act,
                    float* out, void* stream);
/* L stacked interacting layers in ONE launch (AutoInt's `for layer in attention_layers`, BASELINE configs[2]): the
 * output of layer l is the input of layer l + 1 and never leaves the registers; x (B, N, din) -> out (B, N, H*S).
 * Wq/Wk/Wv/W0: HOST arrays of L device pointers (layer 0: (din, H*S), later layers: (H*S, H*S); W0 may be NULL or hold
 * NULL entries = no residual).  Covered: S = 16, din in {16, 32}, H in {1, 2}, N <= 64, L <= 4; otherwise
 * REC_ENOTIMPL (run rec_mha_ctr_f32 per layer).  fp32 MFMA: exact fp32 arithmetic. */
int rec_mha_ctr_stack_f32(const float* x, int64_t B, int32_t N, int32_t din, const float* const* Wq,
                          const float* const* Wk, const float* const* Wv, const float* const* W0, int32_t L, int32_t H,
                          int32_t S, int32_t act, float* out, void* stream);
/* AutoInt.call in ONE launch (src/ctr/autoint/model.py:46-55, the `intended` 3-D form): fields of a sample =
 * [n_sparse embedding rows fetched by id (:46) | n_dense dense values times their embedding rows (:47-50)], all D wide;
 * the L interacting layers (rec_mha_ctr_stack_f32) run on them in registers and the flattened (N * H * S) result meets
 * head_w / head_b (the final Dense(1), :54) and the sigmoid (:55): out_prob (B).  Neither the (B, N, D) field tensor nor
 * the (B, N, H*S) interaction output is written unless out_fields != NULL (then it receives the latter).  Out-of-range
 * ids read as zero rows and raise *oob_flag.  Coverage as rec_mha_ctr_stack_f32 (S = 16, D in {16, 32}, H in {1, 2},
 * N <= 64, L <= 4); otherwise REC_ENOTIMPL. */
int rec_autoint_forward_f32(const rec_table_desc* tables, int32_t n_sparse, const int32_t* ids, int64_t ids_stride,
                            const float* dense, int64_t dense_stride, int32_t n_dense, const float* dense_embed, int32_t D,
                            const float* const* Wq, const float* const* Wk, const float* const* Wv, const float* const* W0,
                            int32_t L, int32_t H, int32_t S, int32_t act, const float* head_w, const float* head_b, int64_t B,
                            float* out_prob, float* out_fields, int32_t* oob_flag, void* stream);

/* ---- a9 / K7: DIN AttentionLayer pooling, src/ctr/layers/modules.py:144-175 ----------------
 * score[b,t] = act([q, k_t, q-k_t, q*k_t] . W + bias)  (Dense(hidden_unit=1)),
 * masked (mask[b,t]==0 -> -4294967296.0; mask==NULL -> ALL scores replaced -> uniform),
 * softmax over t, out[b] = sum_t P[b,t] v[b,t].  q: (B,d); k,v: (B,T,d); mask: (B,T) fp32;
 * W: (4d); bias: (1); alpha: (1) for PReLU or NULL. */
int rec_din_attn_pool_f32(const float* q, const float* k, const float* v, const float* mask,
                          const float* W, const float* bias, const float* alpha, int32_t act,
                          int64_t B, int32_t T, int32_t d, float* out, void* stream);

/* Fused history gather + DIN pooling: k = v = concat_t(tables[t][ids[b, j, t]]) is never materialised
 * (the (B,T,d) history of src/ctr/din/model.py:71-74 costs as much HBM traffic as the pooling itself).
 * tables: n_tab descriptors sharing one dim Dt (d = n_tab * Dt <= 256); ids: (B, T, n_tab) int32 or
 * fp32 (truncated); mask as above, or mask == NULL with mask_from_ids != 0: slot j is real iff
 * ids[b, j, 0] != 0 (pad id 0). */
int rec_gather_din_attn_pool_f32(const float* q, const rec_table_desc* tables, int32_t n_tab,
                                 const void* ids, int32_t ids_dtype, const float* mask,
                                 int32_t mask_from_ids, const float* W, const float* bias,
                                 const float* alpha, int32_t act, int64_t B, int32_t T, float* out,
                                 int32_t* oob_flag, void* stream);

/* ---- a12 / K8: match MultiHeadAttention (row-masked, non-causal, no out-proj) ---------------
 * src/match/layers/modules.py:115-131 with scaled_dot_product_attention :76-96.
 * q: (B, Sq, dm), k/v: (B, Sk, dm) already-projected tensors (projection = rec_dense_f32 with
 * bias); logits = q k^T / sqrt(dm/H); query rows with mask[b,i]==0 get EVERY logit =
 * -4294967296.0 (=> uniform 1/Sk); keys are never masked, not causal; softmax over keys;
 * out = P v merged to (B,Sq,dm).  mask: (B,Sq) fp32.  Sq < Sk serves SASRec's last block, where
 * only the final query row is consumed (src/match/sasrec/model.py:88). */
/* Same with explicit row strides (floats, multiples of 4, >= dm) for q / k / v: views of a wider buffer, e.g. the
 * K and V halves of one fused [Wk | Wv] projection.  Batch stride = S * row stride.  Strided operands are served
 * by the default kernels (Sq <= 8, or Sq >= 16 with dk in {32, 64}); other shapes need contiguous tensors. */
int rec_mha_rowmask_strided_f32(const float* q, int64_t q_stride, const float* k, int64_t k_stride,
                                const float* v, int64_t v_stride, const float* mask, int64_t B,
                                int32_t Sq, int32_t Sk, int32_t dm, int32_t H, float* out, void* stream);
/* Few query rows (Sq <= 8) against keys = values = rows of ONE embedding table addressed by ids (B, Sk): the fused
 * form of `seq_embed = Embedding(seq) * mask` followed by the attention of SASRec's last block
 * (src/match/sasrec/model.py:75,81-88) — the (B, Sk, dm) sequence tensor is never written or re-read.  ids outside
 * [0, vocab) (e.g. pad ids remapped to -1) read as zero rows; keys are never masked, query rows by `mask`. */
int rec_gather_mha_fewq_f32(const float* q, int64_t q_stride, const float* table, int32_t vocab, const void* ids,
                            int32_t ids_dtype, const float* mask, int64_t B, int32_t Sq, int32_t Sk, int32_t dm,
                            int32_t H, float* out, void* stream);
int rec_mha_rowmask_f32(const float* q, const float* k, const float* v, const float* mask,
                        int64_t B, int32_t Sq, int32_t Sk, int32_t dm, int32_t H, float* out,
                        void* stream);

/* ---- SASRec forward with ONE encoder block, one head, last position only — in one launch -----------------
 * src/match/sasrec/model.py:72-96 (mask :72, seq lookup * mask :75,:81-82, the encoder block of
 * src/match/layers/modules.py:152-185, seq_info = att_outputs[:, -1] :88, pos / neg lookups and logits :77-79,:88-96).
 * Only x[:, -1] of the block is consumed, so only that query row is encoded (exact), against all S keys.
 * rec_sasrec_block: the block's weights in their Keras layouts — wq / wk / wv (d, d) (in, out) with biases (bk never
 * influences the softmax and is not read), LayerNormalization gamma / beta, the two k=1 Conv1D kernels as (d, ffn)
 * and (ffn, d).
 * seq_ids (B, S) int32: a slot whose id equals pad_id, or lies outside [0, seq_vocab), is a ZERO row (the reference
 * multiplies pad rows by mask = 0); out-of-range ids other than pad_id also raise *oob_flag.  The query / output mask
 * of sample b is mask_ids[b * mask_stride] != 0 (pass seq_ids + S - 1 with stride seq_ids_stride for the reference's
 * `seq != 0`).  Candidates: n_pos ids from pos_table then n_neg ids from neg_table (the reference's three DIFFERENT
 * tables); logits[b, j] = candidate_j . seq_info[b], j < n_pos + n_neg.  seq_info (B, d) may be NULL.
 * This kernel: d = 64, ffn_hidden in {64, 128}, and S / candidate counts whose id buffers fit the LDS next to the
 * weights (S <= 256 with <= 128 candidates at ffn 128; rec_sasrec_last_row_supported answers 1 / 0); other shapes
 * return REC_ENOTIMPL and the caller composes rec_gather_mha_fewq_f32 / rec_dense_f32 / rec_layernorm_residual_f32 / rec_gather_dot_scores_f32. */
typedef struct rec_sasrec_block {
  const float *wq, *bq, *wk, *wv, *bv;
  const float *ln1_gamma, *ln1_beta;
  const float *w1, *b1, *w2, *b2;
  const float *ln2_gamma, *ln2_beta;
  float ln1_eps, ln2_eps;
  int32_t ffn_hidden;
} rec_sasrec_block;
int rec_sasrec_last_row_supported(int32_t d, int32_t ffn_hidden, int32_t S, int32_t n_cand);
int rec_sasrec_last_row_f32(const rec_sasrec_block* blk, const float* seq_table, int32_t seq_vocab,
                            const int32_t* seq_ids, int64_t seq_ids_stride, int32_t S, int32_t pad_id,
                            const int32_t* mask_ids, int64_t mask_stride, const float* pos_table, int32_t pos_vocab,
                            const int32_t* pos_ids, int64_t pos_ids_stride, int32_t n_pos, const float* neg_table,
                            int32_t neg_vocab, const int32_t* neg_ids, int64_t neg_ids_stride, int32_t n_neg, int64_t B,
                            int32_t d, float* seq_info, float* logits, int64_t logits_stride, int32_t* oob_flag,
                            void* stream);

/* ---- a13 / K9: LayerNormalization(x + r) [* mask], src/match/layers/modules.py:173-185 -----
 * y = LN(x + r) * gamma + beta over the last axis (biased variance, eps); r may be NULL;
 * if row_mask != NULL each output row is multiplied by row_mask[row]
 * (src/match/sasrec/model.py:86). */
int rec_layernorm_residual_f32(const float* x, const float* r, const float* gamma,
                               const float* beta, float eps, const float* row_mask,
                               int64_t rows, int32_t d, float* out, void* stream);

/* ---- a14 / K10: SASRec last-position scores, src/match/sasrec/model.py:88-96 ----------------
 * out[b, j] = seq_info[b,:] . table[ids[b,j], :]   (fused gather + dot), ids: (B, n) */
int rec_gather_dot_scores_f32(const float* seq_info, int64_t seq_stride,
                              const rec_table_desc* table, const int32_t* ids,
                              int64_t ids_stride, int32_t n, int64_t B,
                              float* out, int64_t out_stride, int32_t* oob_flag, void* stream);

/* ---- small fused epilogues ------------------------------------------------------------------------
 * out[i] = sigmoid(a[i] + b[i]) (b may be NULL): tf.nn.sigmoid(tf.add(fm_outputs, deep_outputs))
 * src/ctr/deep_fm/model.py:64 and the final sigmoids of dcn/model.py:56, dlrm/model.py:53. */
int rec_add_sigmoid_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
/* out[i] = act(alpha * a[i] + beta * b[i]), act in {none, relu, sigmoid, tanh}:  `relu(x + inputs)` of
 * Residual_Units (src/ctr/layers/modules.py:33) and `sigmoid(0.5 * wide_out + 0.5 * deep_out)` of
 * src/ctr/wide_deep/model.py:78. */
int rec_axpby_act_f32(const float* a, float alpha, const float* b, float beta, int64_t n, int32_t act,
                      float* out, void* stream);
/* out[i] = act(a[i] * b[i]):  sigmoid(tf.multiply(gmf_user_embed, gmf_pos_embed)) of src/match/ncf/model.py:53-54,
 * tf.multiply(ctr_pred, cvr_pred) of src/ctr/esmm/model.py:37. */
int rec_mul_act_f32(const float* a, const float* b, int64_t n, int32_t act, float* out, void* stream);
/* Dssm.cosine_similarity (src/match/dssm/model.py:49-62): both tensors flattened to ONE vector (the reshape
 * (1, -1) makes it a single scalar for the whole batch), out[0] = <a,b> / (|a| |b|), optionally through the
 * sigmoid of :80.  fp64 accumulation, fixed reduction order (deterministic). */
int64_t rec_cosine_flat_workspace_bytes(int64_t n);
int rec_cosine_flat_f32(const float* a, const float* b, int64_t n, int32_t apply_sigmoid, float* out,
                        void* workspace, void* stream);
/* out[r, :] = x[r, :] * row_scale[r]:  `att_outputs *= mask`, src/match/sasrec/model.py:82 */
int rec_scale_rows_f32(const float* x, const float* row_scale, int64_t rows, int32_t d, float* out,
                       void* stream);
/* tf.concat([... , t, ...], axis=-1) for a tensor t no kernel of ours produced (raw dense inputs, ids fed as floats:
 * src/ctr/din/model.py:64,68,81): dst[m, 0:N] = src[m, 0:N] at a column offset of a wider buffer; src_is_f32 = 0
 * converts int32 values (the reference concatenates `item_sparse_input`, i.e. ids, as floats). */
int rec_copy2d_f32(const void* src, int64_t src_stride, int32_t src_is_f32, int64_t M, int64_t N, float* dst,
                   int64_t dst_stride, void* stream);
/* AutoInt (intended form, SURVEY config 3): dense feature j joins the fields as x[b, j] * E[j, :]:
 * out[b, j*D : (j+1)*D] = x[b, j] * E[j, :], written behind the gathered sparse fields of the same buffer. */
int rec_scale_embed_f32(const float* x, int64_t x_stride, const float* E, int64_t B, int32_t nd, int32_t D, float* out,
                        int64_t out_stride, void* stream);
/* Dice, src/ctr/layers/modules.py:333-337 (inference): p = sigmoid((x - mean) * rsqrt(var + eps));
 * out = alpha * (1 - p) * x + p * x.  mean/var: (d) moving statistics or NULL (0 / 1);
 * alpha: device scalar. */
int rec_dice_f32(const float* x, const float* alpha, const float* mean, const float* var, float eps,
                 int64_t rows, int32_t d, float* out, void* stream);

/* ---- §8f-1 (next row after the path): embedding backward + optimiser step ---------------------------
 * TF computes the gradient of tf.gather as IndexedSlices and Keras' Adam applies it as
 *   m = b1 m + (1-b1) g,  v = b2 v + (1-b2) g^2  over ALL rows (g = 0 for untouched rows),
 *   var -= lr_t m / (sqrt(v) + eps),  lr_t = lr sqrt(1-b2^t) / (1-b1^t)
 * (every train script, e.g. src/ctr/fm/train.py:49; Adam defaults b1 .9, b2 .999, eps 1e-7), and the
 * `embeddings_regularizer=l2(c)` of every model (e.g. src/ctr/dlrm/model.py:35) adds the DENSE
 * gradient 2 c var.  So the exact step is: scatter-add the sparse row gradients (duplicates summed) into a
 * dense (V, D) accumulator, then one dense Adam pass that also adds the L2 term.
 *
 * rec_embedding_grad_f32: grads[f].base[ids[b, f], :] += dy[b, grads[f].out_col : +dim_f]  (fp32 atomics:
 *   the sum over duplicate ids is order-dependent in the last bits).  Out-of-range ids are skipped. */
int rec_embedding_grad_f32(const rec_table_desc* grads, int32_t F, const void* ids, int32_t ids_dtype,
                           int64_t ids_stride, const float* dy, int64_t dy_stride, int64_t B,
                           void* stream);
/* Dense Adam step over n contiguous elements (one table or a whole arena); l2 = regulariser
 * coefficient c (0 = none); step = t >= 1.  grad is read-only (the caller zeroes it for the next step). */
int rec_adam_f32(float* var, float* m, float* v, const float* grad, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, float l2, void* stream);

/* ---- T2: backward of the path's layers (what Keras' fit() differentiates: src/ctr/deep_fm/train.py:58-65) --------
 * GEMM-shaped parts of a Dense backward reuse rec_dense_f32:  dX = dY W^T = dense(dY, transpose(W)),
 * dW = X^T dY = dense(transpose(X), dY);  the rest are row / column passes. */
int rec_transpose_f32(const float* x, int64_t M, int64_t N, int64_t x_stride, float* out /* (N, M) */, void* stream);
/* dy <- dy * act'(.) evaluated from the layer OUTPUT y (relu / sigmoid / tanh; REC_ACT_NONE is a no-op) */
int rec_act_grad_f32(float* dy, int64_t dy_stride, const float* y, int64_t y_stride, int64_t M, int64_t N, int32_t act,
                     void* stream);
/* out[n] = sum_m row_w[m] * a[m,n] * b[m,n]   (b and row_w may be NULL); deterministic: fixed summation order,
 * fp64 across 256-row chunks.  Bias gradients, BN statistics, dw of the cross layers. */
int64_t rec_colsum_workspace_bytes(int64_t M, int64_t N);
int rec_colsum_f32(const float* a, int64_t a_stride, const float* b, int64_t b_stride, const float* row_w, int64_t M,
                   int64_t N, float* out, void* workspace, void* stream);
/* tf.keras.layers.BatchNormalization(training=True) on (M, N) (src/ctr/layers/modules.py:131, din/model.py:83):
 * batch mean / biased variance, y = gamma (x - mean) rsqrt(var + eps) + beta, moving <- moving * momentum +
 * batch * (1 - momentum).  save_mean / save_inv (N floats each) feed the backward.  gamma/beta/moving_* may be NULL.
 * workspace: rec_colsum_workspace_bytes(M, N). */
int rec_bn_train_f32(const float* x, int64_t x_stride, int64_t M, int64_t N, const float* gamma, const float* beta,
                     float eps, float momentum, float* moving_mean, float* moving_var, float* y, int64_t y_stride,
                     float* save_mean, float* save_inv, void* workspace, void* stream);
int64_t rec_bn_train_grad_workspace_bytes(int64_t M, int64_t N);
int rec_bn_train_grad_f32(const float* x, int64_t x_stride, const float* dy, int64_t dy_stride, int64_t M, int64_t N,
                          const float* gamma, const float* save_mean, const float* save_inv, float* dx,
                          int64_t dx_stride, float* dgamma, float* dbeta, void* workspace, void* stream);
/* dlogit[i] = scale * d BCE_keras(y_i, sigmoid(z_i)) / dz_i given p = sigmoid(z) (the loss of rec_binary_crossentropy_f32;
 * scale = 1/n for the mean) */
int rec_bce_sigmoid_grad_f32(const float* y_true, const float* p, int64_t n, float scale, float* dlogit, void* stream);
/* Backward of rec_gather_pairwise_dot_f32 (int32 ids): dz (B, P [+ D]) -> embedding-row gradients atomically added into
 * grad_tables (same shapes as tables; duplicates sum, out-of-range ids contribute nothing) and d_dense (B, D). */
int rec_gather_pairwise_dot_grad_f32(const rec_table_desc* tables, const rec_table_desc* grad_tables, int32_t F,
                                     const int32_t* ids, int64_t ids_stride, const float* dense, int64_t dense_stride,
                                     int64_t B, const float* dz, int64_t dz_stride, int32_t append_dense, float* d_dense,
                                     int64_t d_dense_stride, void* stream);
/* Backward of rec_fm_layer_f32 (src/ctr/layers/modules.py:57-72): dout (B) -> d_first (B, L1; may be NULL),
 * d_second (B, M), dw (L1).  The first-order term is ONE scalar for the whole batch, so d_first rows are all equal. */
int64_t rec_fm_layer_grad_workspace_bytes(int64_t B, int64_t L1);
int rec_fm_layer_grad_f32(const float* first, int64_t first_stride, int64_t L1, const float* second,
                          int64_t second_stride, int64_t M, const float* w, const float* dout, int64_t B, float* d_first,
                          int64_t d_first_stride, float* d_second, int64_t d_second_stride, float* dw, void* workspace,
                          void* stream);
/* One cross layer backward (src/ctr/layers/modules.py:105-112; contiguous (B, dim) operands): given g = dL/dx_{l+1},
 * in place g <- dL/dx_l, dx0 += g s_l, ds[b] = g_b . x0_b  (then dw_l = rec_colsum_f32(xl, row_w = ds), db_l =
 * rec_colsum_f32 of the incoming g). */
int rec_cross_layer_grad_f32(const float* x0, const float* xl, const float* w, int64_t dim, int64_t B, float* g,
                             float* dx0, float* ds, void* stream);
/* "Lazy" row-wise Adam over the rows the batch touched (each exactly once: `stamp` holds one int32 per row, the last
 * step that updated it).  DEVIATION from the reference, which applies Adam + the dense l2 gradient to EVERY row each
 * step (rec_adam_f32 is that exact form): untouched rows keep var, m and v.  The touched rows' gradients are cleared. */
int rec_adam_rows_f32(const rec_table_desc* var, const rec_table_desc* m, const rec_table_desc* v,
                      const rec_table_desc* grad, const rec_table_desc* stamp, int32_t F, const int32_t* ids,
                      int64_t ids_stride, int64_t B, float lr, float beta1, float beta2, float eps, int64_t step, float l2,
                      void* stream);

/* ---- T3: backward of the attention-shaped layers and of the remaining heads (csrc/train_attn.hip) -----------------
 * Attention core on PROJECTED operands laid out as the Dense layers write them: q (B, Nq, H*S), k / v (B, Nk, H*S), head
 * h = columns [h*S, (h+1)*S), explicit row strides.  out[b,i,h] = softmax_j(scale * q_i . k_j) v_j.  row_mask (B, Nq)
 * or NULL: a query row whose mask is 0 has EVERY logit replaced by -4294967296.0 (src/match/layers/modules.py:90-91) =>
 * uniform attention; no gradient reaches that q row or the keys through it.  ctr MultiHeadAttention
 * (src/ctr/layers/modules.py:221-283): scale = sqrt(S), no mask; match: scale = 1/sqrt(S).  Nk <= 512.
 * _grad: dout (B, Nq, H*S) -> dq, dk, dv; workspace = rec_attn_core_grad_workspace_bytes (the probabilities and the
 * logit gradients of every (b, h) pair).  Deterministic. */
int rec_attn_core_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                      const float* row_mask, int64_t B, int32_t Nq, int32_t Nk, int32_t H, int32_t S, float scale,
                      float* out, int64_t ldo, void* stream);
int64_t rec_attn_core_grad_workspace_bytes(int64_t B, int32_t Nq, int32_t Nk, int32_t H);
int rec_attn_core_grad_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                           const float* row_mask, const float* dout, int64_t lddo, int64_t B, int32_t Nq, int32_t Nk,
                           int32_t H, int32_t S, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                           int64_t lddv, void* workspace, void* stream);
/* Backward of rec_din_attn_pool_f32 (AttentionLayer, src/ctr/layers/modules.py:144-175; contiguous q (B, d), k / v
 * (B, T, d), mask (B, T) or NULL / mask_is_none = 1 for the all-padding branch): dout (B, d) -> dq (B, d), dk, dv
 * (B, T, d) and `partials` (B, 4d + 2): per-sample [dW (4d) | dbias | dalpha]; the parameter gradients are their column
 * sums (rec_colsum_f32).  T <= 256. */
int rec_din_attn_pool_grad_f32(const float* q, const float* k, const float* v, const float* mask, int32_t mask_is_none,
                               const float* W, const float* bias, int32_t act, const float* alpha, const float* dout,
                               int64_t B, int32_t T, int32_t d, float* dq, float* dk, float* dv, float* partials,
                               void* stream);
/* tf.keras.layers.PReLU as a Dense activation (src/ctr/din/model.py:52): y = z >= 0 ? z : alpha[n] z on (M, N);
 * _grad: dz (M, N contiguous) and neg_part = min(z, 0) (dalpha[n] = rec_colsum_f32(dy, neg_part)). */
int rec_prelu_f32(const float* z, int64_t z_stride, const float* alpha, int64_t M, int64_t N, float* y, int64_t y_stride,
                  void* stream);
int rec_prelu_grad_f32(const float* z, int64_t z_stride, const float* alpha, const float* dy, int64_t dy_stride, int64_t M,
                       int64_t N, float* dz, float* neg_part, void* stream);
/* Dice (src/ctr/layers/modules.py:327-337) given xn = BatchNormalization(center=False, scale=False)(x) (training: batch
 * statistics, rec_bn_train_f32): y = alpha (1 - p) x + p x, p = sigmoid(xn); n contiguous elements, alpha one float.
 * _grad: dx (direct path), dxn (to be taken through rec_bn_train_grad_f32) and dalpha_elem (dalpha = its sum). */
int rec_dice_train_f32(const float* x, const float* xn, const float* alpha, int64_t n, float* y, void* stream);
int rec_dice_train_grad_f32(const float* x, const float* xn, const float* alpha, const float* dy, int64_t n, float* dx,
                      float* dxn, float* dalpha_elem, void* stream);
/* Backward of rec_layernorm_residual_f32 (y = LN(x + residual) gamma + beta [* row_mask]; contiguous (M, d)): ds =
 * d(x + residual), xhat and dy_masked (M, d) for dgamma = colsum(dy_masked o xhat), dbeta = colsum(dy_masked). */
int rec_layernorm_residual_grad_f32(const float* x, const float* residual, const float* gamma, const float* row_mask,
                                    const float* dy, int64_t M, int32_t d, float eps, float* ds, float* xhat,
                                    float* dy_masked, void* stream);
/* dlogits = scale * d rec_pairwise_rank_loss_f32 / d logits  (src/match/sasrec/model.py:93-95) */
int rec_pairwise_rank_loss_grad_f32(const float* logits, int64_t logits_stride, int64_t B, int32_t n_neg, float scale,
                                    float* dlogits, int64_t dlogits_stride, void* stream);
/* Backward of rec_gather_dot_scores_f32 for one table (int32 ids (B, n)): dseq (B, d) [accumulate = 1: +=] and the
 * looked-up rows' gradients atomically added into grad_table (vocab, d); out-of-range ids contribute nothing. */
int rec_gather_dot_scores_grad_f32(const float* seq, const float* table, float* grad_table, int64_t vocab, int32_t d,
                                   const int32_t* ids, int64_t ids_stride, int32_t n, const float* dlogits,
                                   int64_t dlogits_stride, int64_t B, float* dseq, int32_t accumulate, void* stream);
/* Backward of rec_fm_onehot_f32 (src/ctr/fm/model.py:34-53) given dlogit (B) = dL/d(pre-sigmoid output): dw (L) and dV
 * (k, L) receive the gradients of the touched columns by fp32 atomics (zero them first); dw0 = sum(dlogit).
 * vocab: HOST array of F field sizes; n_dense + F <= 64. */
int rec_fm_onehot_grad_f32(const float* dense, int64_t dense_stride, int32_t n_dense, const int32_t* ids,
                           int64_t ids_stride, int32_t F, const int32_t* vocab, const float* V, int32_t k,
                           const float* dlogit, int64_t B, float* dw, float* dV, void* stream);
/* out (M, N contiguous) = x W for x (M, K) with row stride x_stride and W (K, N), no bias / activation, when M x N is a
 * handful of 128 x 128 tiles and K is long (the weight gradient dW = X^T dY of a Dense layer: M = its input width, K =
 * the batch): the reduction is split over gridDim.y slices on the bf16x3 kernel (fp32-accurate) and the slices are summed
 * in a fixed order in fp64.  workspace: rec_dense_splitk_workspace_bytes(M, K, N). */
int64_t rec_dense_splitk_workspace_bytes(int64_t M, int32_t K, int32_t N);
int rec_dense_splitk_f32(const float* x, int64_t x_stride, const float* W, int64_t M, int32_t K, int32_t N, float* out,
                         void* workspace, void* stream);
/* Weight gradient of a Dense layer with a SMALL kernel and a LONG batch axis: out (K, N) = x^T dy for x (M, K), dy
 * (M, N), N <= 256 and K <= 64 * (256 / N) — the rows are split over workgroups and the partials summed in a fixed order (deterministic),
 * where rec_dense_f32 on the transposed operand would walk all M rows in one or two workgroups (the projections of
 * src/ctr/layers/modules.py:255-269 and src/match/layers/modules.py:110-112 see M = batch x positions).
 * workspace: rec_wgrad_small_workspace_bytes(M, K, N). */
int64_t rec_wgrad_small_workspace_bytes(int64_t M, int32_t K, int32_t N);
int rec_wgrad_small_f32(const float* x, int64_t x_stride, const float* dy, int64_t dy_stride, int64_t M, int32_t K,
                        int32_t N, float* out, void* workspace, void* stream);
/* dp[i] = scale * d BCE_keras(y_i, p_i) / dp_i for a probability that is not itself a sigmoid output (ESMM's
 * pCTCVR = pCTR * pCVR, src/ctr/esmm/model.py:44, trained with loss=["binary_crossentropy", "binary_crossentropy"],
 * src/ctr/esmm/train.py:101); the clip of rec_binary_crossentropy_f32 zeroes the gradient outside [eps, 1 - eps]. */
int rec_bce_prob_grad_f32(const float* y_true, const float* p, int64_t n, float scale, float* dp, void* stream);
/* tf.keras.layers.Dropout(rate) in training mode: y[e] = keep(seed, e) ? x[e] / (1 - rate) : 0 with a counter-based
 * mask (splitmix64 of seed and the element index; TensorFlow's own stream cannot be reproduced — parity unpinned by
 * construction).  The backward pass is the same call on dy.  In place (y == x) allowed. */
int rec_dropout_f32(const float* x, int64_t n, float rate, uint64_t seed, float* y, void* stream);

/* ---- P1: the step before the path, on the device --------------------------------------------------
 * Raw columns arrive over PCIe; these kernels turn them into what the models take.
 * rec_label_encode_u32: sklearn LabelEncoder.transform of src/ctr/utils/data_process.py:66-68.  tokens (B, F) uint32
 *   (Criteo's 8-digit hex categories; REC_TOKEN_MISSING = the "-1" of fillna('-1'), :63, which sorts first);
 *   vocabs[f]: DEVICE pointer to column f's sorted (by that order) unique tokens, vocab_sizes[f] entries (the HOST
 *   arrays `vocabs` / `vocab_sizes` are copied into the launch).  ids[b, f] = rank, or -1 and *unseen_flag = 1.
 * rec_hash_ids_u32: id = mix32(token ^ mix32(seed, f)) mod vocab_sizes[f] (no vocabulary kept; not in the reference).
 * rec_minmax_fit_f32 / rec_minmax_scale_f32: MinMaxScaler of :76-78, intended per-column form on astype(int) values
 *   (truncate_to_int = 1): out = (trunc(x) - min) / (max - min), a constant column maps to 0.
 * rec_pad_sequences_i32: tf.keras pad_sequences(maxlen) of src/match/utils/data_process.py:138 on a ragged batch
 *   (values, offsets[B+1] int64): defaults pre_padding = 1, pre_truncating = 1, pad_value = 0. */
#define REC_TOKEN_MISSING 0xffffffffu
int rec_label_encode_u32(const uint32_t* const* vocabs /* host array of device ptrs */, const int32_t* vocab_sizes /* host */,
                         int32_t F, const uint32_t* tokens, int64_t tok_stride, int64_t B, int32_t* ids, int64_t ids_stride,
                         int32_t* unseen_flag, void* stream);
int rec_hash_ids_u32(const uint32_t* tokens, int64_t tok_stride, const int32_t* vocab_sizes /* host */, int32_t F, int64_t B,
                     uint32_t seed, int32_t* ids, int64_t ids_stride, void* stream);
int64_t rec_minmax_workspace_bytes(int64_t M, int32_t N);
int rec_minmax_fit_f32(const float* x, int64_t x_stride, int64_t M, int32_t N, int32_t truncate_to_int, float* col_min,
                       float* col_max, void* workspace, void* stream);
int rec_minmax_scale_f32(const float* x, int64_t x_stride, int64_t M, int32_t N, const float* col_min, const float* col_max,
                         int32_t truncate_to_int, float* out, int64_t out_stride, void* stream);
int rec_pad_sequences_i32(const int32_t* values, const int64_t* offsets, int64_t B, int32_t maxlen, int32_t pad_value,
                          int32_t pre_padding, int32_t pre_truncating, int32_t* out, int64_t out_stride, void* stream);

/* ---- C2: row-sharded lookup helpers (exchange itself = RCCL all-to-all issued by the host) --
 * Bucket a flat id list by owner rank for cyclic row sharding (owner = id % G, local = id / G):
 *   counts[g]      = number of ids owned by g             (device int32[G], zeroed by the call)
 *   perm[i]        = position of id i in the owner-sorted send buffer (stable within owner)
 *   send_local[p]  = id / G  at sorted position p
 * n ids (int32).  Deterministic (stable) so results are reproducible.  A negative id goes to owner 0 with local
 * row -1 (zero row on the owner); INT32_MIN marks an id that is not to be sent at all (perm = -1). */
int64_t rec_shard_bucket_workspace_bytes(int64_t n, int32_t G);
int rec_shard_bucket_i32(const int32_t* ids, int64_t n, int32_t G, int32_t* counts,
                         int32_t* perm, int32_t* send_local, void* workspace, void* stream);
/* The device part of a sharded lookup on its own (rec_shard_plan_ids = this + the count exchange): exact
 * de-duplication of n virtual row ids + stable bucketing of the unique ones by owner.
 *   rep_table  one int32 per virtual row, INT32_MAX between calls (NULL = no de-duplication)
 *   first[i]   index of the first lookup that asks for the same row (-1 for a negative id)
 *   uniq[i]    vids[i] if i is that first lookup, else INT32_MIN      perm[i]  its position in the send list, else -1
 *   uidx[i]    position in the send list (= row of the returned buffer) that lookup i reads, -1 = zero row
 *   send_local[p], counts[g]  as rec_shard_bucket_i32;  workspace: rec_shard_bucket_workspace_bytes(n, G) */
int rec_shard_dedup_bucket_i32(const int32_t* vids, int64_t n, int32_t G, int32_t* rep_table, int32_t* first,
                               int32_t* uniq, int32_t* perm, int32_t* uidx, int32_t* send_local, int32_t* counts,
                               void* workspace, void* stream);
/* The general form (rec_shard_dedup_bucket_i32 = me -1, no cache, bases 0): where does each lookup read its row?
 * The consumer kernels address ONE row space  [this rank's shard | hot-row replica cache | rows returned by the
 * exchange]  (recamd/dist.py keeps it in one allocation so that a single table descriptor covers it):
 *   vids[i] < 0                    uidx[i] = -1 (zero row)
 *   me >= 0 and vids[i] % G == me  uidx[i] = vids[i] / G: read in place from this rank's shard, never sent
 *   cache_slot[v] >= 0             uidx[i] = cache_base + cache_slot[v]: a replica of a hot remote row
 *                                  (cache_slot: one int32 per virtual row, -1 = not cached; NULL = no cache)
 *   otherwise                      uidx[i] = recv_base + position in the send list (de-duplicated as above)
 * hot_count (one int32 per virtual row, or NULL) is incremented for every lookup of a remote row — the statistic a
 * caller ranks rows by when it refills the cache.  stat (2 x uint64, or NULL) accumulates the number of lookups
 * answered from the local shard and from the cache.  first[i] <= -2 encodes a direct row (-2 - row). */
int rec_shard_resolve_i32(const int32_t* vids, int64_t n, int32_t G, int32_t me, int32_t* rep_table,
                          const int32_t* cache_slot, int32_t* hot_count, int32_t cache_base, int32_t recv_base,
                          uint64_t* stat, int32_t* first, int32_t* uniq, int32_t* perm, int32_t* uidx,
                          int32_t* send_local, int32_t* counts, void* workspace, void* stream);
/* out[i, :] = rows[perm[i], :]  (un-permute the returned rows), D floats per row */
int rec_unpermute_rows_f32(const float* rows, const int32_t* perm, int64_t n, int32_t D,
                           float* out, int64_t out_stride, void* stream);

/* ---- C2 / C1: the exchange of the row-sharded lookup and the gradient merge, on RCCL over xGMI ----------
 * The reference distributes with tf.distribute.MirroredStrategy only (replicated variables + NCCL gradient
 * all-reduce inside fit(): src/ctr/fm/train.py:43-45 and nine more train scripts); these entry points are the
 * north-star replacement: tables row-sharded cyclically over the `world` GPUs of one node (virtual row v =
 * f * Vpad + id, owner = v % world, local row = v / world), one process per GPU.
 *
 * rec_comm: an opaque communicator.  rec_comm_init_rank creates an RCCL communicator (rank 0 obtains the 128-byte
 * id with rec_comm_unique_id and hands it to the other ranks by any means); rec_comm_from_nccl borrows an existing
 * ncclComm_t; rec_comm_create_with_transport takes caller-supplied collectives (another fabric, a test double);
 * rec_comm_create_local builds `world` in-process communicators on one device whose collectives complete when the
 * LAST simulated rank has entered them (tests: run each phase for every rank before the next phase). */
typedef struct rec_comm rec_comm;
typedef struct rec_transport {
  void* ctx;
  /* all-gather of `world` int32 send counts per rank into a (world x world) matrix, row p = rank p's counts */
  int (*allgather_counts)(void* ctx, const int32_t* counts_dev, int32_t* matrix_dev, void* stream);
  /* all-to-all(v) of elements of elem_bytes; counts and displacements in elements, HOST arrays of `world` entries */
  int (*alltoallv)(void* ctx, int32_t rank, const void* send, const int64_t* send_counts, const int64_t* send_displ,
                   void* recv, const int64_t* recv_counts, const int64_t* recv_displ, int32_t elem_bytes, void* stream);
  /* in-place sum over ranks (may be NULL if rec_comm_allreduce_sum_f32 is never called) */
  int (*allreduce_sum_f32)(void* ctx, int32_t rank, float* buf, int64_t n, void* stream);
  int32_t deferred; /* 1: a collective's data movement is enqueued only when the last rank has entered it */
} rec_transport;
int rec_comm_unique_id(void* id128 /* host, 128 bytes out */);
int rec_comm_init_rank(rec_comm** out, const void* id128, int32_t world, int32_t rank);
int rec_comm_from_nccl(rec_comm** out, void* nccl_comm, int32_t world, int32_t rank);
int rec_comm_create_with_transport(rec_comm** out, const rec_transport* t, int32_t world, int32_t rank);
int rec_comm_create_local(int32_t world, rec_comm** comms_out /* host array of `world` handles */);
int rec_comm_destroy(rec_comm* comm);
int32_t rec_comm_world(const rec_comm* comm);
int32_t rec_comm_rank(const rec_comm* comm);
/* "rccl" | "rccl (borrowed ncclComm_t)" | "in-process" | "caller-supplied": what a benchmark line reports as having run */
const char* rec_comm_transport_name(const rec_comm* comm);
/* gradient merge of replicated (dense) parameters: buf <- sum over ranks, in place (MirroredStrategy's all-reduce) */
int rec_comm_allreduce_sum_f32(rec_comm* comm, float* buf, int64_t n, void* stream);

/* One sharded lookup of n virtual row ids (device int32; negative = out of range, answered with a zero row):
 *   rec_shard_plan_ids     device: exact de-duplication (rep_table: one int32 per virtual row holding INT32_MAX
 *                          between calls, or NULL = no de-duplication), stable bucketing of the unique ids by owner,
 *                          all-gather of the send counts, asynchronous copy of the count matrix to pinned host
 *                          memory behind an event.  A pipelined caller issues this for batch i+1 before it runs
 *                          batch i, so rec_shard_plan_finish never waits.
 *   rec_shard_plan_finish  host: waits for that event, derives the split sizes; returns the number of unique rows
 *                          this rank will receive back (n_unique) and the number it must serve (n_recv).
 *   rec_shard_exchange_ids all-to-all #1: local rows of the unique ids -> recv_local (n_recv int32)
 *   rec_shard_serve_f32    served[i, :] = arena[recv_local[i], :]   (this rank's (arena_rows, D) shard)
 *   rec_shard_exchange_rows_f32  reverse = 0: all-to-all #2, served (n_recv, D) -> dst (n_unique, D) in send order;
 *                          reverse = 1: the backward direction, one gradient row per unique lookup (n_unique, D) ->
 *                          dst (n_recv, D) aligned with recv_local for the owner's scatter-add.
 *   rec_shard_lookup_f32   = finish + exchange_ids + serve + exchange_rows(forward).
 * Lookup i then reads row rec_shard_plan_uidx()[i] of the returned rows (-1 = zero row): pass the buffer as the
 * table and uidx as the ids of rec_gather_concat_f32 / rec_gather_pairwise_dot_f32 — no un-permute pass.
 * workspace: rec_shard_plan_workspace_bytes(max_ids, world) bytes of device memory owned by the caller; it holds
 * uidx and the send list and must stay untouched until the lookup's last use of them. */
typedef struct rec_shard_plan rec_shard_plan;
int64_t rec_shard_plan_workspace_bytes(int64_t max_ids, int32_t world);
int rec_shard_plan_create(rec_comm* comm, int64_t max_ids, rec_shard_plan** out);
int rec_shard_plan_destroy(rec_shard_plan* plan);
int rec_shard_plan_ids(rec_shard_plan* plan, const int32_t* vids, int64_t n, int32_t* rep_table, void* workspace,
                       void* stream);
/* rec_shard_plan_ids with the row space of rec_shard_resolve_i32: bypass_local = rows this rank owns are read in
 * place (they are not sent to itself: the all-to-alls then carry remote rows only); NULL opts = rec_shard_plan_ids.
 * Every call takes the stream it runs on: a pipelined caller plans and exchanges batch i+1 on a communication stream
 * while batch i's consumer kernel runs on the compute stream (recamd/dist.py: events order the two). */
typedef struct rec_shard_resolve_opts {
  int32_t bypass_local;
  const int32_t* cache_slot; /* device, or NULL */
  int32_t* hot_count;        /* device, or NULL */
  int32_t cache_base, recv_base;
  uint64_t* stat;            /* device, 2 counters, or NULL */
} rec_shard_resolve_opts;
int rec_shard_plan_ids_ex(rec_shard_plan* plan, const int32_t* vids, int64_t n, int32_t* rep_table,
                          const rec_shard_resolve_opts* opts, void* workspace, void* stream);
int rec_shard_plan_finish(rec_shard_plan* plan, int64_t* n_unique /* host out */, int64_t* n_recv /* host out */);
const int32_t* rec_shard_plan_uidx(const rec_shard_plan* plan);
int rec_shard_exchange_ids(rec_shard_plan* plan, int32_t* recv_local, void* stream);
int rec_shard_serve_f32(rec_shard_plan* plan, const float* arena, int64_t arena_rows, int32_t D,
                        const int32_t* recv_local, float* served, int32_t* oob_flag, void* stream);
int rec_shard_exchange_rows_f32(rec_shard_plan* plan, const float* src, int32_t D, float* dst, int32_t reverse,
                                void* stream);
int rec_shard_lookup_f32(rec_shard_plan* plan, const float* arena, int64_t arena_rows, int32_t D, int32_t* recv_local,
                         int64_t recv_cap, float* served, float* rows_out, int64_t rows_cap, int32_t* oob_flag,
                         void* stream);

/* ---- §8f-2 (first slice): the loss and the metric of every ctr train script --------------------------
 * model.compile(loss=binary_crossentropy, metrics=[AUC()]) / model.evaluate(...)[1]
 * (src/ctr/deep_fm/train.py:50-51,68).  y_true, y_pred: n fp32 values (labels 0/1, probabilities).
 * BCE: Keras epsilon clipping 1e-7, mean over n.  AUC: Keras defaults (200 thresholds, ROC, trapezoid).
 * workspace: rec_metrics_workspace_bytes(n) bytes of device memory; out: one float. */
int64_t rec_metrics_workspace_bytes(int64_t n);
int rec_binary_crossentropy_f32(const float* y_true, const float* y_pred, int64_t n, float* out,
                                void* workspace, void* stream);
int rec_auc_f32(const float* y_true, const float* y_pred, int64_t n, float* out, void* workspace,
                void* stream);
/* add_loss of the match models (src/match/sasrec/model.py:93-95, src/match/ncf/model.py:75-77): logits (B, 1 + n_neg)
 * with column 0 = positive score; out[0] = mean over (b, j) of [-log sigmoid(pos_b) - log(1 - sigmoid(neg_bj))] / 2.
 * workspace: rec_metrics_workspace_bytes(B * n_neg) bytes. */
int rec_pairwise_rank_loss_f32(const float* logits, int64_t logits_stride, int64_t B, int32_t n_neg, float* out,
                               void* workspace, void* stream);

/* ---- §8f-4: retrieval after the towers — exact inner-product top-k ---------------------------------
 * Replaces faiss.IndexFlatIP(d).add(items).search(queries, k) of src/match/dssm/dssm_train.py:74-78 and
 * src/match/fm/train.py:71-75.  queries (Q, d), items (N, d) row-major with the given strides, d <= 128,
 * k <= 32.  out_scores (Q, k) descending, out_idx (Q, k) int64 row numbers into `items` (faiss labels); equal
 * scores order by smaller index; with N < k the tail is (-inf, -1). */
int rec_topk_ip_f32(const float* queries, int64_t q_stride, int64_t Q, const float* items,
                    int64_t items_stride, int64_t N, int32_t d, int32_t k, float* out_scores,
                    int64_t* out_idx, void* stream);
/* Same with a caller-owned workspace of rec_topk_ip_workspace_bytes(Q, N, k) bytes (0 = none needed): with few queries
 * (a MovieLens-sized evaluation: 6 040 users x 3 706 items) the item range is split over several workgroups per query
 * tile and the partial lists are merged by a second kernel; results are identical (same tie rule). */
int64_t rec_topk_ip_workspace_bytes(int64_t Q, int64_t N, int32_t k);
int rec_topk_ip_ws_f32(const float* queries, int64_t q_stride, int64_t Q, const float* items,
                       int64_t items_stride, int64_t N, int32_t d, int32_t k, float* out_scores,
                       int64_t* out_idx, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECAMD_H_ */
