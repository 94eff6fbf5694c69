"""CPU oracle — a numpy restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``recommend-tf2.0_amd/``) never does.

PARITY UNPINNED: the reference (littlemesie/recommend-tf2.0) is pure Python on TensorFlow 2.x,
TensorFlow is not installed here (``import tensorflow`` -> ModuleNotFoundError, no wheel, no network)
and the reference ships no tests, golden vectors or fixtures.  Every function below is therefore a
restatement written from the reference SOURCE TEXT plus the documented semantics of the TF/Keras ops
it calls; it is cross-checked against an independent torch-CPU restatement
(``oracle/ref_torch.py``) and against closed-form known-answer tests (``tests/test_oracle_kat.py``),
not against an execution of the reference.

Citations are file:line relative to the reference repository root.
All functions take/return numpy arrays and compute in ``dtype`` (float64 for golden vectors,
float32 to mirror the reference's arithmetic type).
"""
from __future__ import annotations

import numpy as np

# ``-2 ** 32 + 1`` (src/ctr/layers/modules.py:161, src/match/layers/modules.py:90) is the Python int
# -4294967295; multiplied into an fp32 tensor it rounds to -4294967296.0.
NEG_PAD = float(np.float32(-2 ** 32 + 1))

BN_EPS = 1e-3  # tf.keras.layers.BatchNormalization default epsilon


# --------------------------------------------------------------------------------------------
# activations (Keras activation strings used by the reference)
# --------------------------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def activation(x, act, alpha=None):
    if act in (None, "linear", "none"):
        return x
    if act == "relu":
        return np.maximum(x, 0)
    if act == "sigmoid":
        return sigmoid(x)
    if act == "tanh":
        return np.tanh(x)
    if act == "prelu":  # keras PReLU: max(0,x) + alpha*min(0,x); alpha zero-initialised
        a = 0.0 if alpha is None else alpha
        return np.maximum(x, 0) + a * np.minimum(x, 0)
    raise ValueError(f"unknown activation {act!r}")


def softmax(x, axis=-1):
    """tf.nn.softmax: max-subtracted, last axis."""
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


# --------------------------------------------------------------------------------------------
# a1 — Embedding gather + concat
# --------------------------------------------------------------------------------------------
def cast_ids(ids):
    """Keras Embedding casts non-int inputs to int32 (tf.cast truncates toward zero).
    DIN / YoutubeDNN declare float32 Inputs (src/ctr/din/model.py:96-100,
    src/match/youtube_dnn/model.py:65-68)."""
    ids = np.asarray(ids)
    if np.issubdtype(ids.dtype, np.floating):
        return np.trunc(ids).astype(np.int64)
    return ids.astype(np.int64)


def embedding_lookup(table, ids, oob="zero"):
    """tf.gather(table, ids) along axis 0 -> ids.shape + (D,).
    oob='raise' = TF-CPU (InvalidArgumentError), oob='zero' = TF-GPU (zeros)."""
    ids = cast_ids(ids)
    V = table.shape[0]
    bad = (ids < 0) | (ids >= V)
    if bad.any() and oob == "raise":
        raise IndexError("embedding id out of range")
    out = table[np.where(bad, 0, ids)]
    if bad.any():
        out = np.where(bad[..., None], np.zeros((), table.dtype), out)
    return out


def gather_concat(tables, ids, oob="zero"):
    """tf.concat([Embedding_f(sparse_inputs[:, f]) for f], axis=-1)
    src/ctr/deep_fm/model.py:53, dcn/model.py:47, dlrm/model.py:45, autoint/model.py:46.
    tables: list of (V_f, D_f); ids: (B, F) -> (B, sum D_f).  Bit-exact copy."""
    ids = np.asarray(ids)
    assert ids.ndim == 2 and ids.shape[1] == len(tables)
    return np.concatenate([embedding_lookup(t, ids[:, f], oob) for f, t in enumerate(tables)], axis=-1)


# --------------------------------------------------------------------------------------------
# a2 — ctr FM model (one-hot form, as written) and its gather form
# --------------------------------------------------------------------------------------------
def one_hot(ids, depth, dtype):
    """tf.one_hot: out-of-range (incl. negative) index -> all-zero row."""
    ids = cast_ids(ids)
    out = np.zeros(ids.shape + (depth,), dtype)
    ok = (ids >= 0) & (ids < depth)
    rows = np.nonzero(ok)[0]
    out[rows, ids[ok]] = 1
    return out


def fm_model_onehot(dense, ids, vocab, w0, w, V, dtype=np.float64):
    """src/ctr/fm/model.py:34-53 exactly as written (materialises the one-hot stack).
    dense (B,nd); ids (B,F); w0 (1,); w (L,1); V (k,L) with L = nd + sum(vocab)."""
    dense = np.asarray(dense, dtype)
    w0, w, V = (np.asarray(a, dtype) for a in (w0, w, V))
    sparse = np.concatenate([one_hot(ids[:, i], vocab[i], dtype) for i in range(ids.shape[1])], axis=1)
    stack = np.concatenate([dense, sparse], axis=1)                        # :43
    first = w0 + stack @ w                                                # :45
    second = 0.5 * np.sum((stack @ V.T) ** 2 - (stack ** 2) @ (V.T ** 2), axis=1, keepdims=True)  # :47-49
    return sigmoid(first + second)                                        # :51-52


def fm_model_gather(dense, ids, vocab, w0, w, V, dtype=np.float64):
    """Gather form of fm_model_onehot: rows off_f+id of w and of V^T plus the dense part."""
    dense = np.asarray(dense, dtype)
    w0, w, V = (np.asarray(a, dtype) for a in (w0, w, V))
    B, nd = dense.shape
    ids = cast_ids(ids)
    lin = w0 + dense @ w[:nd]
    s = dense @ V[:, :nd].T
    q = (dense ** 2) @ (V[:, :nd].T ** 2)
    off = nd
    for f, vf in enumerate(vocab):
        i = ids[:, f]
        ok = (i >= 0) & (i < vf)
        col = off + np.where(ok, i, 0)
        lin = lin + np.where(ok[:, None], w[col], 0)
        vv = np.where(ok[:, None], V[:, col].T, 0)
        s = s + vv
        q = q + vv ** 2
        off += vf
    return sigmoid(lin + 0.5 * np.sum(s ** 2 - q, axis=1, keepdims=True))


# --------------------------------------------------------------------------------------------
# a3 — FM layer (DeepFM wide part)
# --------------------------------------------------------------------------------------------
def fm_layer(first, second, w, dtype=np.float64):
    """src/ctr/layers/modules.py:57-72.  first (B,L1), second (B,M) [2-D, the only live form:
    src/ctr/deep_fm/model.py:58-59], w (L1,1).
    first_order = reduce_sum(first @ w) -> ONE scalar over the whole batch (:65)."""
    first, second, w = (np.asarray(a, dtype) for a in (first, second, w))
    first_order = np.sum(first @ w.reshape(-1, 1))                                # :65
    square_sum = np.sum(second, axis=1, keepdims=True) ** 2                      # :67
    sum_square = np.sum(second ** 2, axis=1, keepdims=True)                      # :68
    second_order = 0.5 * np.sum(square_sum - sum_square, axis=1)                 # :69
    return (first_order + second_order).reshape(-1, 1)                           # :70-71


# --------------------------------------------------------------------------------------------
# a4 — CrossNetwork
# --------------------------------------------------------------------------------------------
def cross_network(x, W, Bv, dtype=np.float64):
    """src/ctr/layers/modules.py:105-112.  x (B,dim); W, Bv (L,dim) (each w_i/b_i is (dim,1)).
    x_l1 = tensordot(x_l, w_i, axes=[1,0]) -> (B,1,1);  x_l = x_0 @ x_l1 + b_i + x_l."""
    x, W, Bv = (np.asarray(a, dtype) for a in (x, W, Bv))
    x0 = x[:, :, None]
    xl = x0
    for i in range(W.shape[0]):
        xl1 = np.tensordot(xl, W[i][:, None], axes=[1, 0])      # (B,1,1)
        xl = np.matmul(x0, xl1) + Bv[i][:, None] + xl
    return xl[:, :, 0]


# --------------------------------------------------------------------------------------------
# a6 / a15 — Dense, BatchNormalization, DNN towers
# --------------------------------------------------------------------------------------------
def dense(x, W, b=None, act=None, alpha=None):
    """keras Dense: tensordot over the last axis, + bias, activation."""
    y = np.tensordot(x, W, axes=[[x.ndim - 1], [0]])
    if b is not None:
        y = y + b
    return activation(y, act, alpha)


def batch_norm_inference(x, gamma=None, beta=None, mean=None, var=None, eps=BN_EPS):
    """keras BatchNormalization(training=False): y = (x-mean)*gamma/sqrt(var+eps)+beta.
    Fresh layer (src/ctr/layers/modules.py:131): gamma=1, beta=0, mean=0, var=1
    => y = x / sqrt(1 + 1e-3)."""
    d = x.shape[-1]
    gamma = np.ones(d, x.dtype) if gamma is None else gamma
    beta = np.zeros(d, x.dtype) if beta is None else beta
    mean = np.zeros(d, x.dtype) if mean is None else mean
    var = np.ones(d, x.dtype) if var is None else var
    inv = gamma / np.sqrt(var + eps)
    return x * inv + (beta - mean * inv)


def dnn_ctr(x, layers, act="relu", bn=None, dtype=np.float64):
    """src/ctr/layers/modules.py:129-135: BatchNormalization()(x) -> Dense stack -> Dropout(id).
    layers: list of (W, b); bn: dict(gamma,beta,mean,var) or None for a fresh BN."""
    x = np.asarray(x, dtype)
    x = batch_norm_inference(x, **({} if bn is None else {k: np.asarray(v, dtype) for k, v in bn.items()}))
    for W, b in layers:
        x = dense(x, np.asarray(W, dtype), np.asarray(b, dtype), act)
    return x


def dnn_match(x, layers, act="relu", dtype=np.float64):
    """src/match/layers/modules.py:21-26: Dense stack, no BN."""
    x = np.asarray(x, dtype)
    for W, b in layers:
        x = dense(x, np.asarray(W, dtype), np.asarray(b, dtype), act)
    return x


def dice(x, alpha, mean=None, var=None, eps=BN_EPS):
    """src/ctr/layers/modules.py:333-337: p = sigmoid(BN_noaffine(x)); alpha*(1-p)*x + p*x."""
    xn = batch_norm_inference(x, None, None, mean, var, eps)
    p = sigmoid(xn)
    return alpha * (1.0 - p) * x + p * x


# --------------------------------------------------------------------------------------------
# a5 — DLRM
# --------------------------------------------------------------------------------------------
def pairwise_dot(X, dtype=np.float64):
    """DLRM dot interaction of the paper cited at src/ctr/dlrm/model.py:7 (the reference file has
    no interaction op).  X (B,n,D) -> (B, n(n-1)/2): Z = X X^T, strictly-lower triangle,
    order (i,j), i>j, row-major: (1,0),(2,0),(2,1),(3,0)...  [our definition, SURVEY §8c-10]."""
    X = np.asarray(X, dtype)
    Z = np.matmul(X, np.swapaxes(X, 1, 2))
    n = X.shape[1]
    li, lj = zip(*[(i, j) for i in range(n) for j in range(i)]) if n > 1 else ((), ())
    return Z[:, list(li), list(lj)]


def dlrm_forward(dense_in, ids, tables, bot, top, final, interaction="cat", dtype=np.float64):
    """src/ctr/dlrm/model.py:42-54 in its INTENDED form (as written it raises AttributeError at
    :44 `self.dense_inputs` and :50 `self.dnn_network`):
      dense_fea = bot_dnn(dense_inputs); sparse_embed = gather_concat; x = concat[sparse_embed,
      dense_fea] (:48); top_dnn; final_dense; sigmoid.
    interaction='dot' replaces sparse_embed by the pairwise dots of [emb_0..emb_{F-1}, dense_fea]
    (needs bot output width == D).  bot/top: dict(layers=[(W,b)...], bn=None|dict)."""
    dense_fea = dnn_ctr(dense_in, bot["layers"], "relu", bot.get("bn"), dtype)
    emb = gather_concat([np.asarray(t, dtype) for t in tables], ids)
    if interaction == "cat":
        x = np.concatenate([emb, dense_fea], axis=-1)
    else:
        B, F = np.asarray(ids).shape
        D = tables[0].shape[1]
        X = np.concatenate([emb.reshape(B, F, D), dense_fea[:, None, :]], axis=1)
        x = np.concatenate([pairwise_dot(X, dtype), dense_fea], axis=-1)
    h = dnn_ctr(x, top["layers"], "relu", top.get("bn"), dtype)
    return sigmoid(dense(h, np.asarray(final[0], dtype), np.asarray(final[1], dtype)))


# --------------------------------------------------------------------------------------------
# DeepFM / DCN forwards
# --------------------------------------------------------------------------------------------
def deepfm_forward(dense_in, ids, tables, fm_w, dnn, final, act="relu", dtype=np.float64):
    """src/ctr/deep_fm/model.py:50-65."""
    dense_in = np.asarray(dense_in, dtype)
    sparse_embed = gather_concat([np.asarray(t, dtype) for t in tables], ids)      # :53
    embeds = np.concatenate([dense_in, sparse_embed], axis=-1)                     # :56
    fm_out = fm_layer(embeds, sparse_embed, fm_w, dtype)                           # :59
    deep = dnn_ctr(embeds, dnn["layers"], act, dnn.get("bn"), dtype)               # :61
    deep = dense(deep, np.asarray(final[0], dtype), np.asarray(final[1], dtype))   # :62
    return sigmoid(fm_out + deep)                                                  # :64


def dcn_forward(ids, tables, cross_W, cross_B, dnn, final, act="relu", dtype=np.float64):
    """src/ctr/dcn/model.py:45-57."""
    x = gather_concat([np.asarray(t, dtype) for t in tables], ids)
    cross_x = cross_network(x, cross_W, cross_B, dtype)
    dnn_x = dnn_ctr(x, dnn["layers"], act, dnn.get("bn"), dtype)
    total = np.concatenate([cross_x, dnn_x], axis=-1)
    return sigmoid(dense(total, np.asarray(final[0], dtype), np.asarray(final[1], dtype)))


# --------------------------------------------------------------------------------------------
# a7 / a8 — ctr MultiHeadAttention, AutoInt
# --------------------------------------------------------------------------------------------
def mha_ctr(xq, xk, xv, Wq, Wk, Wv, W0=None, head_num=1, head_size=None, act="relu", dtype=np.float64):
    """src/ctr/layers/modules.py:285-325 on 3-D inputs (B, N, d_model).
    q,k,v = act(X W) without bias (:255-269); heads (B,H,N,S) (:211-219);
    product = q k^T / (S ** -0.5)  i.e. TIMES sqrt(S) (:235-237); softmax; out = P v (:238-239);
    merge to (B,N,H*S) (:281-283); use_res (W0 given): relu(out + act(Xv W0)) (:316-323)."""
    xq, xk, xv, Wq, Wk, Wv = (np.asarray(a, dtype) for a in (xq, xk, xv, Wq, Wk, Wv))
    H = head_num
    S = head_size if head_size is not None else Wq.shape[1] // H
    q = activation(dense(xq, Wq), act)
    k = activation(dense(xk, Wk), act)
    v = activation(dense(xv, Wv), act)

    def split(t):
        return np.transpose(t.reshape(-1, t.shape[1], H, S), (0, 2, 1, 3))

    q, k, v = split(q), split(k), split(v)
    product = np.matmul(q, np.swapaxes(k, -1, -2)) / (S ** -0.5)
    out = np.matmul(softmax(product), v)
    out = np.transpose(out, (0, 2, 1, 3))
    out = out.reshape(-1, out.shape[1], out.shape[2] * out.shape[3])
    if W0 is not None:
        res = activation(dense(xv, np.asarray(W0, dtype)), act)
        out = np.maximum(out + res, 0)
    return out


def autoint_forward_intended(x3, att_layers, final, H, S, act="relu", use_res=False, dtype=np.float64):
    """AutoInt on the INTENDED 3-D field tensor x3 (B, fields, d): stacked interacting layers ->
    flatten -> Dense(1) -> sigmoid (src/ctr/autoint/model.py:50-55 with a 3-D input).
    att_layers: list of dict(Wq,Wk,Wv[,W0])."""
    h = np.asarray(x3, dtype)
    for L in att_layers:
        h = mha_ctr(h, h, h, L["Wq"], L["Wk"], L["Wv"], L.get("W0") if use_res else None, H, S, act, dtype)
    flat = h.reshape(h.shape[0], -1)
    return sigmoid(dense(flat, np.asarray(final[0], dtype), np.asarray(final[1], dtype)))


def autoint_forward_as_written(dense_in, ids, tables, L, final, S, act="relu", dtype=np.float64):
    """src/ctr/autoint/model.py:44-55 exactly as written: the 2-D (B, F*D+nd) tensor goes into a
    layer written for 3-D input, so `reshape([-1, q.shape[1], H, S])` (modules.py:211-212) mixes
    samples: B/(H*S) pseudo-batches of (H*S) pseudo-fields.  head_num=1 (:40).  Needs B % S == 0.
    Returns (B/S, 1)."""
    emb = gather_concat([np.asarray(t, dtype) for t in tables], ids)
    x = np.concatenate([emb, np.asarray(dense_in, dtype)], axis=-1)              # :48
    q = activation(dense(x, np.asarray(L["Wq"], dtype)), act)                    # (B, S)
    k = activation(dense(x, np.asarray(L["Wk"], dtype)), act)
    v = activation(dense(x, np.asarray(L["Wv"], dtype)), act)

    def split(t):  # reshape [-1, t.shape[1], H=1, S]
        return np.transpose(t.reshape(-1, t.shape[1], 1, S), (0, 2, 1, 3))

    q, k, v = split(q), split(k), split(v)
    product = np.matmul(q, np.swapaxes(k, -1, -2)) / (S ** -0.5)
    out = np.matmul(softmax(product), v)
    out = np.transpose(out, (0, 2, 1, 3))
    out = out.reshape(-1, out.shape[1], out.shape[2] * out.shape[3])
    flat = out.reshape(-1, out.shape[1] * out.shape[2])                          # :52
    return sigmoid(dense(flat, np.asarray(final[0], dtype), np.asarray(final[1], dtype)))


# --------------------------------------------------------------------------------------------
# a9 / a10 — DIN AttentionLayer pooling
# --------------------------------------------------------------------------------------------
def din_attention_layer(q, k, v, mask, W, b, act="sigmoid", alpha=None, dtype=np.float64):
    """src/ctr/layers/modules.py:144-175 with Dense(hidden_unit=1).
    q (B,d); k,v (B,T,d); mask (B,T) array or None (non-tensor mask => ALL scores replaced by the
    padding value => uniform softmax, :162-165); W (4d,1); b (1,)."""
    q, k, v, W, b = (np.asarray(a, dtype) for a in (q, k, v, W, b))
    B, T, d = k.shape
    qt = np.tile(q, (1, T)).reshape(-1, T, d)                                   # :150-151
    info = np.concatenate([qt, k, qt - k, qt * k], axis=-1)                      # :154
    outputs = dense(info, W.reshape(4 * d, 1), b, act, alpha)                    # :157
    outputs = outputs.reshape(-1, T)                                             # :159
    paddings = np.ones_like(outputs) * NEG_PAD                                   # :161
    if mask is not None:
        outputs = np.where(np.asarray(mask) == 0, paddings, outputs)             # :163
    else:
        outputs = paddings                                                       # :165
    p = softmax(outputs)[:, None, :]                                             # :169-170
    return np.matmul(p, v)[:, 0, :]                                              # :172-173


# --------------------------------------------------------------------------------------------
# a12 / a13 / a14 — match MultiHeadAttention, TransformerEncoder, SASRec
# --------------------------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps):
    """keras LayerNormalization over the last axis, biased variance."""
    mu = np.mean(x, axis=-1, keepdims=True)
    var = np.mean((x - mu) ** 2, axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def sdpa_match(q, k, v, mask):
    """src/match/layers/modules.py:76-96.  q,k,v (B,H,S,dk); mask (B,H,S,1).
    where(mask==0, pad, logits) broadcasts the (…,S,1) mask along the KEY axis => whole QUERY
    rows are replaced (padded queries attend uniformly); keys are never masked; not causal."""
    dk = k.shape[-1]
    logits = np.matmul(q, np.swapaxes(k, -1, -2)) / np.sqrt(np.asarray(dk, q.dtype))     # :85-88
    paddings = np.ones_like(logits) * NEG_PAD
    outputs = np.where(mask == 0, paddings, logits)                                       # :91
    return np.matmul(softmax(outputs), v)                                                 # :93-94


def mha_match(q, k, v, mask, P, num_heads, dtype=np.float64):
    """src/match/layers/modules.py:115-131.  q,k,v (B,S,d_in); mask (B,S,1);
    P: dict(Wq,bq,Wk,bk,Wv,bv) Dense WITH bias, no activation; no output projection."""
    q, k, v = (np.asarray(a, dtype) for a in (q, k, v))
    q = dense(q, np.asarray(P["Wq"], dtype), np.asarray(P["bq"], dtype))
    k = dense(k, np.asarray(P["Wk"], dtype), np.asarray(P["bk"], dtype))
    v = dense(v, np.asarray(P["Wv"], dtype), np.asarray(P["bv"], dtype))
    B, S, dm = q.shape
    H = num_heads

    def split(t):
        return np.transpose(t.reshape(-1, S, H, dm // H), (0, 2, 1, 3))

    m = np.tile(np.asarray(mask, dtype)[:, None, :, :], (1, H, 1, 1))            # :126
    att = sdpa_match(split(q), split(k), split(v), m)
    return np.transpose(att, (0, 2, 1, 3)).reshape(-1, S, dm)                    # :130


def ffn_match(x, P, dtype=np.float64):
    """src/match/layers/modules.py:146-149: Conv1D(k=1, relu) -> Conv1D(k=1) == Dense on last axis."""
    h = dense(x, np.asarray(P["W1"], dtype), np.asarray(P["b1"], dtype), "relu")
    return dense(h, np.asarray(P["W2"], dtype), np.asarray(P["b2"], dtype))


def transformer_encoder(x, mask, P, num_heads=1, eps=1e-6, dtype=np.float64):
    """src/match/layers/modules.py:173-185."""
    x = np.asarray(x, dtype)
    att = mha_match(x, x, x, mask, P, num_heads, dtype)
    out1 = layer_norm(x + att, np.asarray(P["ln1_g"], dtype), np.asarray(P["ln1_b"], dtype), eps)
    f = ffn_match(out1, P, dtype)
    return layer_norm(out1 + f, np.asarray(P["ln2_g"], dtype), np.asarray(P["ln2_b"], dtype), eps)


def sasrec_forward(seq, pos, neg, T_seq, T_pos, T_neg, blocks, num_heads=1, eps=1e-6, dtype=np.float64):
    """src/match/sasrec/model.py:60-97.  seq (B,S), pos (B,1), neg (B,n) int ids; three DIFFERENT
    tables (:75-79); no positional embedding (:74).  Returns (logits (B,1+n), loss scalar)."""
    seq = cast_ids(seq)
    mask = (seq != 0).astype(dtype)[..., None]                                   # :72
    x = embedding_lookup(np.asarray(T_seq, dtype), seq)                          # :75
    pos_e = embedding_lookup(np.asarray(T_pos, dtype), pos)                      # :77
    neg_e = embedding_lookup(np.asarray(T_neg, dtype), neg)                      # :79
    x = x * mask                                                                 # :82
    for P in blocks:
        x = transformer_encoder(x, mask, P, num_heads, eps, dtype)               # :85
        x = x * mask                                                             # :86
    seq_info = x[:, -1][:, None, :]                                              # :88
    pos_scores = np.sum(seq_info * pos_e, axis=-1)                               # :90
    neg_scores = np.sum(seq_info * neg_e, axis=-1)                               # :91
    loss = np.mean(-np.log(sigmoid(pos_scores)) - np.log(1 - sigmoid(neg_scores))) / 2   # :93-94
    return np.concatenate([pos_scores, neg_scores], axis=-1), loss               # :96


def youtube_dnn_towers(user_ids, user_tables, item_ids, item_tables, user_layers, item_layers,
                       act="relu", dtype=np.float64):
    """src/match/youtube_dnn/model.py:47-56: per-feature (B,1) inputs -> gather -> concat ->
    DNN towers; returns (user_dnn_out, item_dnn_out) each (B,1,units).  The SampledSoftmaxLayer
    (:59) is stochastic (unseeded log-uniform sampler) and excluded from parity."""
    def tower(ids_list, tables, layers):
        emb = np.concatenate([embedding_lookup(np.asarray(t, dtype), i) for i, t in zip(ids_list, tables)], axis=-1)
        return dnn_match(emb, layers, act, dtype)
    return tower(user_ids, user_tables, user_layers), tower(item_ids, item_tables, item_layers)


# --------------------------------------------------------------------------------------------
# §8f-1 — embedding backward and the Keras Adam step (the step after the hot path)
# --------------------------------------------------------------------------------------------
def embedding_grad(ids, dy, vocabs, dims, dtype=np.float64):
    """Gradient of gather_concat w.r.t. the tables: IndexedSlices of tf.gather densified — rows of dy are
    scatter-ADDED at their ids (duplicates sum), out-of-range ids contribute nothing."""
    ids = cast_ids(ids)
    dy = np.asarray(dy, dtype)
    grads, col = [], 0
    for f, (V, D) in enumerate(zip(vocabs, dims)):
        g = np.zeros((V, D), dtype)
        ok = (ids[:, f] >= 0) & (ids[:, f] < V)
        np.add.at(g, ids[ok, f], dy[ok, col:col + D])
        grads.append(g)
        col += D
    return grads


def adam_step(var, m, v, grad, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, l2=0.0, dtype=np.float64):
    """tf.keras.optimizers.Adam (TF2.x, amsgrad=False) dense update; `embeddings_regularizer=l2(c)` adds
    c * sum(w^2) to the loss, i.e. 2 c w to the gradient.  Returns (var, m, v)."""
    var, m, v, grad = (np.asarray(a, dtype) for a in (var, m, v, grad))
    g = grad + 2.0 * l2 * var
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    var = var - lr_t * m / (np.sqrt(v) + eps)
    return var, m, v


# --------------------------------------------------------------------------------------------
# §8f-4 — zoo models that reuse the a1 lookup: Deep&Crossing, Wide&Deep, ESMM
# --------------------------------------------------------------------------------------------
def residual_unit(x, W1, b1, W2, b2, dtype=np.float64):
    """src/ctr/layers/modules.py:29-34: relu(Dense2(relu(Dense1(x))) + x)."""
    x = np.asarray(x, dtype)
    h = dense(dense(x, np.asarray(W1, dtype), np.asarray(b1, dtype), "relu"), np.asarray(W2, dtype), np.asarray(b2, dtype))
    return np.maximum(h + x, 0)


def deep_crossing_forward(ids, tables, res_units, final, dtype=np.float64):
    """src/ctr/deep_crossing/model.py:42-51.  res_units: list of (W1, b1, W2, b2)."""
    r = gather_concat([np.asarray(t, dtype) for t in tables], ids)
    for W1, b1, W2, b2 in res_units:
        r = residual_unit(r, W1, b1, W2, b2, dtype)
    return sigmoid(dense(r, np.asarray(final[0], dtype), np.asarray(final[1], dtype)))


def wide_deep_forward(dense_in, ids, tables, linear, dnn_layers, final, act="relu", dtype=np.float64):
    """src/ctr/wide_deep/model.py:66-79: sigmoid(0.5 * Linear(dense) + 0.5 * Dense(DNN([emb, dense])))."""
    dense_in = np.asarray(dense_in, dtype)
    emb = gather_concat([np.asarray(t, dtype) for t in tables], ids)
    x = np.concatenate([emb, dense_in], axis=-1)
    wide = dense(dense_in, np.asarray(linear[0], dtype), np.asarray(linear[1], dtype))
    deep = dense(dnn_match(x, dnn_layers, act, dtype), np.asarray(final[0], dtype), np.asarray(final[1], dtype))
    return sigmoid(0.5 * wide + 0.5 * deep)


def esmm_tower(user_num, user_cate, item_num, item_cate, user_tables, user_cols, item_tables, item_cols,
               user_dnn, item_dnn, head, act="relu", dtype=np.float64):
    """src/ctr/esmm/model.py:42-73 (one tower).  *_cate are float id matrices (Keras cast);
    head = dict(bn=..., dense=(W, b), out=(W, b))."""
    ue = gather_concat([np.asarray(t, dtype) for t in user_tables], np.asarray(user_cate)[:, user_cols])
    ie = gather_concat([np.asarray(t, dtype) for t in item_tables], np.asarray(item_cate)[:, item_cols])
    uf = dnn_ctr(np.concatenate([np.asarray(user_num, dtype), ue], axis=-1), user_dnn["layers"], act, user_dnn.get("bn"), dtype)
    itf = dnn_ctr(np.concatenate([np.asarray(item_num, dtype), ie], axis=-1), item_dnn["layers"], act, item_dnn.get("bn"), dtype)
    x = np.concatenate([uf, itf], axis=-1)
    x = batch_norm_inference(x, **{k: np.asarray(v, dtype) for k, v in head["bn"].items()})
    x = dense(x, np.asarray(head["dense"][0], dtype), np.asarray(head["dense"][1], dtype), "relu")
    return sigmoid(dense(x, np.asarray(head["out"][0], dtype), np.asarray(head["out"][1], dtype)))


def topk_inner_product(queries, items, k, dtype=np.float64):
    """faiss.IndexFlatIP(d).add(items).search(queries, k) (src/match/dssm/dssm_train.py:74-78): exact inner
    products, k best per query, scores descending; ties -> smaller index (our definition; faiss leaves it open);
    fewer than k items -> (-inf, -1) padding."""
    q, it = np.asarray(queries, dtype), np.asarray(items, dtype)
    Q, N = q.shape[0], it.shape[0]
    scores = q @ it.T if N else np.zeros((Q, 0), dtype)
    order = np.argsort(-scores, axis=1, kind="stable")[:, :k]
    D = np.take_along_axis(scores, order, axis=1)
    I = order.astype(np.int64)
    if N < k:
        D = np.concatenate([D, np.full((Q, k - N), -np.inf, dtype)], axis=1)
        I = np.concatenate([I, np.full((Q, k - N), -1, np.int64)], axis=1)
    return D, I


def cosine_flat(a, b, dtype=np.float64):
    """Dssm.cosine_similarity (src/match/dssm/model.py:49-62): both tensors reshaped to (1, -1) -> one scalar."""
    a, b = np.asarray(a, dtype).reshape(-1), np.asarray(b, dtype).reshape(-1)
    return np.sum(a * b) / (np.sqrt(np.sum(a * a)) * np.sqrt(np.sum(b * b)))


def dssm_forward(user_ids, item_ids, user_tables, item_tables, user_dnn, item_dnn, act="relu", dtype=np.float64):
    """src/match/dssm/model.py:64-82.  *_ids (B, n_feat) float/int; returns (out (1,1), user_out (B,1,U), item_out)."""
    u = dnn_match(gather_concat([np.asarray(t, dtype) for t in user_tables], user_ids), user_dnn, act, dtype)[:, None, :]
    i = dnn_match(gather_concat([np.asarray(t, dtype) for t in item_tables], item_ids), item_dnn, act, dtype)[:, None, :]
    return sigmoid(np.array([[cosine_flat(i, u, dtype)]])), u, i


def ncf_forward(user, pos, neg, user_table, item_table, neg_table, dnn_layers, final, act="relu", dtype=np.float64):
    """src/match/ncf/model.py:47-80: logits (B, 1 + neg_num)."""
    ut, it, nt = (np.asarray(t, dtype) for t in (user_table, item_table, neg_table))
    ue = embedding_lookup(ut, user)                      # (B, 1, dim)
    W, b = np.asarray(final[0], dtype), np.asarray(final[1], dtype)

    def branch(items):                                   # items (B, T, dim)
        T = items.shape[1]
        gmf = sigmoid(ue * items)
        mlp = dnn_match(np.concatenate([np.tile(ue, (1, T, 1)), items], axis=-1), dnn_layers, act, dtype)
        return dense(np.concatenate([gmf, mlp], axis=-1), W, b)[..., 0]

    return np.concatenate([branch(embedding_lookup(it, pos)), branch(embedding_lookup(nt, neg))], axis=-1)


# --------------------------------------------------------------------------------------------
# §8f-2 — loss and metric of the ctr train scripts
# --------------------------------------------------------------------------------------------
def binary_crossentropy(y_true, y_pred, eps=1e-7, dtype=np.float64):
    """tf.keras.losses.binary_crossentropy on probabilities, averaged over all samples
    (src/ctr/deep_fm/train.py:50).  tf.keras.backend.binary_crossentropy (from_logits=False): the prediction is
    clipped to [eps, 1-eps] and eps is added again INSIDE both logs: -(y log(p + eps) + (1-y) log(1 - p + eps)).
    (Graph-mode TF may route a Sigmoid-op output through sigmoid_cross_entropy_with_logits instead — no clip;
    `binary_crossentropy_from_logits` restates that branch.)  Parity unpinned: no fixtures, TF not importable."""
    y = np.asarray(y_true, dtype).reshape(-1)
    p = np.clip(np.asarray(y_pred, dtype).reshape(-1), eps, 1.0 - eps)
    return float(np.mean(-(y * np.log(p + eps) + (1.0 - y) * np.log(1.0 - p + eps))))


def binary_crossentropy_from_logits(y_true, logits, dtype=np.float64):
    """tf.nn.sigmoid_cross_entropy_with_logits averaged: max(x,0) - x z + log(1 + exp(-|x|))."""
    z = np.asarray(y_true, dtype).reshape(-1)
    x = np.asarray(logits, dtype).reshape(-1)
    return float(np.mean(np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))))


def keras_auc(y_true, y_pred, num_thresholds=200):
    """tf.keras.metrics.AUC() defaults (src/ctr/deep_fm/train.py:51): thresholds {0-1e-7, i/(T-1), 1+1e-7} in fp32,
    `pred > threshold`, ROC, trapezoidal ('interpolation') summation, rates via div_no_nan in fp32."""
    y = np.asarray(y_true).reshape(-1) != 0
    p = np.asarray(y_pred, np.float32).reshape(-1)
    thr = np.array([0.0 - 1e-7] + [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + 1e-7],
                   np.float32)
    gt = p[None, :] > thr[:, None]
    tp = np.sum(gt & y[None, :], axis=1).astype(np.float32)
    fp = np.sum(gt & ~y[None, :], axis=1).astype(np.float32)
    P, N = np.float32(np.sum(y)), np.float32(np.sum(~y))
    tpr = tp / P if P > 0 else np.zeros_like(tp)
    fpr = fp / N if N > 0 else np.zeros_like(fp)
    return float(np.sum((fpr[:-1] - fpr[1:]).astype(np.float32) * ((tpr[:-1] + tpr[1:]) * np.float32(0.5))))


def pairwise_rank_loss(logits, dtype=np.float64):
    """add_loss of src/match/sasrec/model.py:93-95 and src/match/ncf/model.py:75-77 on logits (B, 1+n), column 0 = pos:
    reduce_mean(-log(sigmoid(pos)) - log(1 - sigmoid(neg))) / 2 with (B,1) + (B,n) broadcasting."""
    lg = np.asarray(logits, dtype)
    pos, neg = lg[:, :1], lg[:, 1:]
    return np.mean(-np.log(sigmoid(pos)) - np.log(1 - sigmoid(neg))) / 2


def match_fm_forward(user_ids, item_ids, user_tables, item_tables, w0, w, V, dtype=np.float64):
    """src/match/fm/model.py:62-83: stack = [user embeddings, item embeddings]; sigmoid(w0 + stack w +
    0.5 sum_k((stack V^T)_k^2 - (stack^2 (V^T)^2)_k)).  Returns (out (B,1), user_embeds, item_embeds)."""
    ue = gather_concat([np.asarray(t, dtype) for t in user_tables], user_ids)
    ie = gather_concat([np.asarray(t, dtype) for t in item_tables], item_ids)
    stack = np.concatenate([ue, ie], axis=-1)
    w0, w, V = (np.asarray(a, dtype) for a in (w0, w, V))
    first = w0 + stack @ w
    second = 0.5 * np.sum((stack @ V.T) ** 2 - (stack ** 2) @ (V.T ** 2), axis=1, keepdims=True)
    return sigmoid(first + second), ue, ie
