"""CPU oracle of the TRAINING step (TEST INFRASTRUCTURE ONLY; parity unpinned like the rest of oracle/: TensorFlow is
not importable here and the reference ships no fixtures).

Restates, in fp64 torch with autograd, what `model.fit` does for one batch of the ctr models whose mirrors have a
training-mode forward — src/ctr/dlrm/model.py:42-54 (intended form + the cited paper's dot interaction),
src/ctr/deep_fm/model.py:50-65, src/ctr/dcn/model.py:45-57 — with BatchNormalization in TRAINING mode
(src/ctr/layers/modules.py:131: batch mean / biased variance, eps 1e-3, moving averages with momentum 0.99), the Keras
binary cross-entropy on probabilities (clip + eps inside the logs) and tf.keras.optimizers.Adam (TF2 defaults), the
models' l2 regularisers added to the loss exactly as Keras does (embeddings_regularizer=l2(c): c * sum(w^2)).

The attention-shaped models of recamd/train_attn.py (classic FM, AutoInt, DIN, SASRec) are restated with the functions of
oracle/ref_torch.py under autograd.

Weights travel as {name: ndarray} with the names of recamd.train.named_weights."""
import numpy as np
import torch

EPS_BCE = 1e-7
BN_EPS, BN_MOM = 1e-3, 0.99


def T(a, grad=False):
    t = torch.as_tensor(np.asarray(a), dtype=torch.float64).clone()
    t.requires_grad_(grad)
    return t


def act(x, name):
    if name in (None, "linear", "none"):
        return x
    return {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[name](x)


def bn_train(x, P, prefix, new_moving):
    mu = x.mean(dim=0)
    var = x.var(dim=0, unbiased=False)
    new_moving[prefix + "/moving_mean"] = (P[prefix + "/moving_mean"] * BN_MOM + mu * (1 - BN_MOM)).detach()
    new_moving[prefix + "/moving_variance"] = (P[prefix + "/moving_variance"] * BN_MOM + var * (1 - BN_MOM)).detach()
    return (x - mu) / torch.sqrt(var + BN_EPS) * P[prefix + "/gamma"] + P[prefix + "/beta"]


def bn_infer(x, P, prefix):
    return (x - P[prefix + "/moving_mean"]) / torch.sqrt(P[prefix + "/moving_variance"] + BN_EPS) * P[prefix + "/gamma"] \
        + P[prefix + "/beta"]


def dnn(x, P, prefix, n_layers, activation, training, new_moving):
    h = bn_train(x, P, prefix + "/bn", new_moving) if training else bn_infer(x, P, prefix + "/bn")
    for i in range(n_layers):
        h = act(h @ P[f"{prefix}/dense_{i}/kernel"] + P[f"{prefix}/dense_{i}/bias"], activation)
    return h


def embed(P, i, ids):
    tab = P[f"embed_{i}/embeddings"]
    idx = torch.as_tensor(np.asarray(ids), dtype=torch.int64)
    ok = (idx >= 0) & (idx < tab.shape[0])
    return tab[idx.clamp(0, tab.shape[0] - 1)] * ok[:, None].to(torch.float64)


def gather_concat(P, ids):
    return torch.cat([embed(P, f, ids[:, f]) for f in range(ids.shape[1])], dim=-1)


def n_dense(P, prefix):
    return len([k for k in P if k.startswith(prefix + "/dense_") and k.endswith("/kernel")])


def dlrm_forward(P, dense_in, ids, interaction, activation="relu", training=True, new_moving=None):
    new_moving = {} if new_moving is None else new_moving
    dense_fea = dnn(T(dense_in), P, "bot_dnn", n_dense(P, "bot_dnn"), activation, training, new_moving)
    emb = gather_concat(P, ids)
    if interaction == "dot":
        B, F = ids.shape
        X = torch.cat([emb.view(B, F, -1), dense_fea[:, None, :]], dim=1)
        Z = X @ X.transpose(1, 2)
        n = F + 1
        li, lj = zip(*[(i, j) for i in range(n) for j in range(i)])
        x = torch.cat([Z[:, list(li), list(lj)], dense_fea], dim=-1)
    else:
        x = torch.cat([emb, dense_fea], dim=-1)
    top = dnn(x, P, "top_dnn", n_dense(P, "top_dnn"), activation, training, new_moving)
    return torch.sigmoid(top @ P["final_dense/kernel"] + P["final_dense/bias"]).reshape(-1)


def deepfm_forward(P, dense_in, ids, activation="relu", training=True, new_moving=None):
    new_moving = {} if new_moving is None else new_moving
    sparse_embed = gather_concat(P, ids)
    embeds = torch.cat([T(dense_in), sparse_embed], dim=-1)
    first = torch.sum(embeds @ P["fm/w"])                                          # ONE scalar (modules.py:65)
    second = 0.5 * (sparse_embed.sum(dim=1) ** 2 - (sparse_embed ** 2).sum(dim=1))
    fm_out = (first + second).reshape(-1, 1)
    deep = dnn(embeds, P, "dnn", n_dense(P, "dnn"), activation, training, new_moving) @ P["dense/kernel"] + P["dense/bias"]
    return torch.sigmoid(fm_out + deep).reshape(-1)


def dcn_forward(P, ids, activation="relu", training=True, new_moving=None):
    new_moving = {} if new_moving is None else new_moving
    x0 = gather_concat(P, ids)
    xl = x0
    W, Bv = P["cross_network/cross_weights"], P["cross_network/cross_bias"]
    for l in range(W.shape[0]):
        xl = x0 * (xl @ W[l])[:, None] + Bv[l] + xl
    d = dnn(x0, P, "dnn_network", n_dense(P, "dnn_network"), activation, training, new_moving)
    return torch.sigmoid(torch.cat([xl, d], dim=-1) @ P["dense_final/kernel"] + P["dense_final/bias"]).reshape(-1)


def keras_bce(p, y):
    pc = torch.clamp(p, EPS_BCE, 1 - EPS_BCE)
    return torch.mean(-(y * torch.log(pc + EPS_BCE) + (1 - y) * torch.log(1 - pc + EPS_BCE)))


def l2_of(name, l2):
    if name in l2:
        return l2[name]
    best = 0.0
    for k, c in l2.items():
        if len(k) > 1 and (name.startswith(k) or name.endswith(k)):
            best = c
    return best


def reg_loss(P, l2):
    tot = 0.0
    for k, v in P.items():
        c = l2_of(k, l2)
        if c:
            tot = tot + c * torch.sum(v.detach() ** 2 if not v.requires_grad else v ** 2)
    return tot


def is_moving(k):
    return k.endswith("moving_mean") or k.endswith("moving_variance")


class AdamOracle:
    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, decay=0.0):
        self.lr, self.b1, self.b2, self.eps, self.t, self.m, self.v, self.decay = lr, b1, b2, eps, 0, {}, {}, decay

    def apply(self, W, grads):
        """W: {name: fp64 ndarray} updated in place; grads include the regularisers' gradients.  decay: the legacy Keras
        schedule lr / (1 + decay * iterations) with `iterations` counted before this step (src/ctr/esmm/train.py:95)"""
        lr = self.lr / (1.0 + self.decay * self.t)
        self.t += 1
        lr_t = lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            m = self.m.setdefault(k, np.zeros_like(W[k]))
            v = self.v.setdefault(k, np.zeros_like(W[k]))
            m[...] = self.b1 * m + (1 - self.b1) * g
            v[...] = self.b2 * v + (1 - self.b2) * g * g
            W[k] = W[k] - lr_t * m / (np.sqrt(v) + self.eps)


# ---- the attention-shaped models (recamd/train_attn.py): forwards restated with oracle/ref_torch.py's functions, which are
# differentiable when they are handed fp64 tensors that require grad ---------------------------------------------------
def fm_forward(P, dense_in, ids, vocab):
    """src/ctr/fm/model.py:34-53 (explicit one-hot form)"""
    from . import ref_torch
    return ref_torch.fm_model(T(dense_in), ids, vocab, P["w0"], P["w"], P["V"]).reshape(-1)


def autoint_forward(P, dense_in, ids, H, S, activation="relu", use_res=False):
    """AutoInt, intended (B, fields, D) form: src/ctr/autoint/model.py:44-55 + src/ctr/layers/modules.py:285-325"""
    from . import ref_torch
    ids = np.asarray(ids)
    B, F = ids.shape
    emb = gather_concat(P, ids)
    D = emb.shape[1] // F
    h = emb.view(B, F, D)
    if "dense_embed" in P:
        h = torch.cat([h, T(dense_in)[:, :, None] * P["dense_embed"][None]], dim=1)
    li = 0
    while f"attention_{li}/Wq" in P:
        pre = f"attention_{li}/"
        h = ref_torch.mha_ctr(h, h, h, P[pre + "Wq"], P[pre + "Wk"], P[pre + "Wv"], P.get(pre + "W0") if use_res else None,
                              H, S, activation)
        li += 1
    return torch.sigmoid(h.reshape(B, -1) @ P["final_dense/kernel"] + P["final_dense/bias"]).reshape(-1)


def _table(P, name, ids):
    tab = P[name]
    idx = torch.as_tensor(np.trunc(np.asarray(ids, np.float64)).astype(np.int64))
    ok = (idx >= 0) & (idx < tab.shape[0])
    return tab[idx.clamp(0, tab.shape[0] - 1)] * ok[..., None].to(torch.float64)


def din_forward(P, inputs, user_keys, item_keys, maxlen, att_activation, ffn_activation, n_ffn, training=True,
                new_moving=None):
    """DIN, canonical form: src/ctr/din/model.py:57-93 with AttentionLayer (src/ctr/layers/modules.py:144-175) as the
    pooling; ffn Dense(PReLU() | Dice()); BatchNormalization on batch statistics when training"""
    from . import ref_torch
    new_moving = {} if new_moving is None else new_moving
    user_dense, user_sparse, item_dense, item_sparse, behavior = [np.asarray(a, np.float64) for a in inputs]
    ue = torch.cat([_table(P, f"embed_{k}/embeddings", user_sparse[:, i]) for i, k in enumerate(user_keys)], dim=-1)
    ie = torch.cat([_table(P, f"embed_{k}/embeddings", item_sparse[:, i]) for i, k in enumerate(item_keys)], dim=-1)
    n_item = len(item_keys)
    beh = torch.stack([torch.cat([_table(P, f"embed_{k}/embeddings", behavior[:, ml * n_item + i])
                                  for i, k in enumerate(item_keys)], dim=-1) for ml in range(maxlen)], dim=1)
    mask = torch.as_tensor((np.trunc(behavior[:, ::n_item]) != 0).astype(np.float64))
    att = ref_torch.din_attention(ie, beh, beh, mask, P["attention_layer/kernel"], P["attention_layer/bias"], att_activation,
                                  P.get("attention_layer/alpha"))
    x = torch.cat([T(user_dense), ue, T(item_sparse), ie, att], dim=-1)
    x = bn_train(x, P, "bn", new_moving) if training else bn_infer(x, P, "bn")
    for i in range(n_ffn):
        z = x @ P[f"ffn_{i}/kernel"] + P[f"ffn_{i}/bias"]
        if ffn_activation == "prelu":
            x = torch.where(z >= 0, z, P[f"ffn_{i}/prelu/alpha"] * z)
        else:
            pre = f"ffn_{i}/dice/bn"
            if training:
                mu, var = z.mean(dim=0), z.var(dim=0, unbiased=False)
                new_moving[pre + "/moving_mean"] = (P[pre + "/moving_mean"] * BN_MOM + mu * (1 - BN_MOM)).detach()
                new_moving[pre + "/moving_variance"] = (P[pre + "/moving_variance"] * BN_MOM + var * (1 - BN_MOM)).detach()
            else:
                mu, var = P[pre + "/moving_mean"], P[pre + "/moving_variance"]
            pz = torch.sigmoid((z - mu) / torch.sqrt(var + BN_EPS))
            a = P[f"ffn_{i}/dice/alpha"]
            x = a * (1.0 - pz) * z + pz * z
    return torch.sigmoid(x @ P["final_output/kernel"] + P["final_output/bias"]).reshape(-1)


def sasrec_forward(P, seq, pos, neg, n_blocks, H, eps=1e-6):
    """src/match/sasrec/model.py:60-97 -> (logits, add_loss)"""
    from . import ref_torch
    blocks = []
    for b in range(n_blocks):
        e = f"encoder_{b}/"
        blocks.append(dict(Wq=P[e + "mha/wq/kernel"], bq=P[e + "mha/wq/bias"], Wk=P[e + "mha/wk/kernel"], bk=P[e + "mha/wk/bias"],
                           Wv=P[e + "mha/wv/kernel"], bv=P[e + "mha/wv/bias"], ln1_g=P[e + "layernorm1/gamma"],
                           ln1_b=P[e + "layernorm1/beta"], W1=P[e + "ffn/conv1/kernel"], b1=P[e + "ffn/conv1/bias"],
                           W2=P[e + "ffn/conv2/kernel"], b2=P[e + "ffn/conv2/bias"], ln2_g=P[e + "layernorm2/gamma"],
                           ln2_b=P[e + "layernorm2/beta"]))
    return ref_torch.sasrec(seq, pos, neg, P["user_embed_seq_item/embeddings"], P["user_embed_pos_item/embeddings"],
                            P["user_embed_neg_item/embeddings"], blocks, H, eps)


def dropout_mask(n, rate, seed):
    """keep mask of rec_dropout_f32 (csrc/train_attn.hip): splitmix64 finaliser of seed * K + e, upper 32 bits >= rate * 2^32"""
    M = (1 << 64) - 1
    e = np.arange(n, dtype=np.uint64)
    x = (np.uint64((seed * 0xD1342543DE82EF95) & M) + e)          # wraps mod 2^64
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    r = (x >> np.uint64(32)).astype(np.uint64)
    t = float(rate) * 4294967296.0
    thresh = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    return r >= np.uint64(thresh)


def dense_stack(x, P, prefix, activation="relu"):
    i = 0
    while f"{prefix}/dense_{i}/kernel" in P:
        x = act(x @ P[f"{prefix}/dense_{i}/kernel"] + P[f"{prefix}/dense_{i}/bias"], activation)
        i += 1
    return x


def wide_deep_forward(P, dense_in, ids, activation="relu"):
    """src/ctr/wide_deep/model.py:66-79"""
    d = T(dense_in)
    x = torch.cat([gather_concat(P, np.asarray(ids)), d], dim=-1)
    wide = d @ P["linear/dense/kernel"] + P["linear/dense/bias"]
    deep = dense_stack(x, P, "dnn_network", activation) @ P["final_dense/kernel"] + P["final_dense/bias"]
    return torch.sigmoid(0.5 * wide + 0.5 * deep).reshape(-1)


def deep_crossing_forward(P, ids):
    """src/ctr/deep_crossing/model.py:42-51 with the residual unit of src/ctr/layers/modules.py:29-34"""
    r = gather_concat(P, np.asarray(ids))
    i = 0
    while f"res_{i}/layer1/kernel" in P:
        h = torch.relu(r @ P[f"res_{i}/layer1/kernel"] + P[f"res_{i}/layer1/bias"]) @ P[f"res_{i}/layer2/kernel"] \
            + P[f"res_{i}/layer2/bias"]
        r = torch.relu(h + r)
        i += 1
    return torch.sigmoid(r @ P["dense/kernel"] + P["dense/bias"]).reshape(-1)


def ncf_forward(P, user, pos, neg, activation="relu"):
    """src/match/ncf/model.py:47-80 -> (logits (B, 1 + n), add_loss)"""
    ue = _table(P, "user_embedding/embeddings", user)                 # (B, 1, dim)

    def branch(items):                                                # (B, T, dim)
        Tn = items.shape[1]
        u = ue.expand(-1, Tn, -1)
        gmf = torch.sigmoid(u * items)
        mlp = dense_stack(torch.cat([u, items], dim=-1), P, "dnn", activation)
        return (torch.cat([gmf, mlp], dim=-1) @ P["dense/kernel"] + P["dense/bias"])[..., 0]
    pl = branch(_table(P, "item_embedding/embeddings", pos))
    nl = branch(_table(P, "neg_item_embedding/embeddings", neg))
    loss = torch.mean(-torch.log(torch.sigmoid(pl)) - torch.log(1 - torch.sigmoid(nl))) / 2
    return torch.cat([pl, nl], dim=-1), loss



def _tower_embed(P, inputs, prefix):
    """tf.concat([embed_layers['embed_<k>'](v) for k, v in inputs.items()], axis=-1) squeezed to (B, sum of dims):
    src/match/fm/model.py:63-64,68-69 and src/match/dssm/model.py:68-69,74-75 (dict order; float ids truncate)"""
    return torch.cat([_table(P, f"{prefix}_embed_{k}/embeddings", np.asarray(v).reshape(-1)) for k, v in inputs.items()], dim=-1)


def match_fm_forward(P, user_in, item_in):
    """src/match/fm/model.py:61-82: FM over the concatenated user / item embeddings -> p (B,)"""
    stack = torch.cat([_tower_embed(P, user_in, "user"), _tower_embed(P, item_in, "item")], dim=-1)          # :73-75
    first = P["w0"] + stack @ P["w"]                                                                         # :76
    Vt = P["V"].T
    second = 0.5 * torch.sum((stack @ Vt) ** 2 - (stack ** 2) @ (Vt ** 2), dim=1, keepdim=True)              # :77-79
    return torch.sigmoid(first + second).reshape(-1)                                                         # :80-82


def dssm_forward(P, user_in, item_in, activation="relu"):
    """src/match/dssm/model.py:64-82 -> (y_pred (1,), loss): ONE value for the batch — cosine_similarity (:49-62) flattens both
    towers' outputs of all samples into one vector each — and the script's objective mean(y_pred)
    (src/match/dssm/dssm_train.py:47, src/match/utils/loss_util.py:11-13), which ignores the labels"""
    uo = dense_stack(_tower_embed(P, user_in, "user"), P, "user_dnn", activation)                            # :68-72
    io = dense_stack(_tower_embed(P, item_in, "item"), P, "item_dnn", activation)                            # :74-77
    a, b = io.reshape(-1), uo.reshape(-1)
    cos = torch.sum(a * b) / (torch.sqrt(torch.sum(a * a)) * torch.sqrt(torch.sum(b * b)))                  # :52-60
    p = torch.sigmoid(cos).reshape(1)                                                                        # :80
    return p, torch.mean(p)


def esmm_forward(P, inputs, user_keys, user_cols, item_keys, item_cols, activation="relu", training=True, new_moving=None):
    """src/ctr/esmm/model.py:37-92 -> (pCTR, pCTCVR).  The shared DNNs' BatchNormalization layers are called once per
    tower: in training mode the second call starts from the moving statistics the first one left (Keras updates them
    call by call)."""
    new_moving = {} if new_moving is None else new_moving
    xs = [np.asarray(a, np.float64) for a in inputs]

    def tower(head, un, uc, inum, ic):
        Q = dict(P)
        Q.update(new_moving)                                  # what the previous tower's calls left behind
        ue = torch.cat([_table(P, f"embed_{k}/embeddings", uc[:, c]) for k, c in zip(user_keys, user_cols)], dim=-1)
        ie = torch.cat([_table(P, f"embed_{k}/embeddings", ic[:, c]) for k, c in zip(item_keys, item_cols)], dim=-1)
        uf = dnn(torch.cat([T(un), ue], dim=-1), Q, "user_dnn", n_dense(P, "user_dnn"), activation, training, new_moving)
        itf = dnn(torch.cat([T(inum), ie], dim=-1), Q, "item_dnn", n_dense(P, "item_dnn"), activation, training, new_moving)
        x = torch.cat([uf, itf], dim=-1)
        x = bn_train(x, P, head + "/bn", new_moving) if training else bn_infer(x, P, head + "/bn")
        x = torch.relu(x @ P[head + "/dense/kernel"] + P[head + "/dense/bias"])
        return torch.sigmoid(x @ P[head + "/out/kernel"] + P[head + "/out/bias"]).reshape(-1)
    ctr = tower("ctr_head", *xs[:4])
    cvr = tower("cvr_head", *xs[4:])
    return ctr, ctr * cvr


def _forward(kind, P, inputs, training, new_moving, kw):
    if kind == "wide_deep":
        return wide_deep_forward(P, inputs[0], inputs[1]), None
    if kind == "deep_crossing":
        return deep_crossing_forward(P, inputs), None
    if kind == "ncf":
        return ncf_forward(P, inputs[0], inputs[1], inputs[2])
    if kind == "match_fm":
        return match_fm_forward(P, inputs[0], inputs[1]), None
    if kind == "dssm":
        return dssm_forward(P, inputs[0], inputs[1])
    if kind == "dlrm":
        return dlrm_forward(P, inputs[0], np.asarray(inputs[1]), kw.get("interaction", "dot"), training=training,
                            new_moving=new_moving), None
    if kind == "deepfm":
        return deepfm_forward(P, inputs[0], np.asarray(inputs[1]), training=training, new_moving=new_moving), None
    if kind == "dcn":
        return dcn_forward(P, np.asarray(inputs), training=training, new_moving=new_moving), None
    if kind == "fm":
        return fm_forward(P, inputs[0], np.asarray(inputs[1]), kw["vocab"]), None
    if kind == "autoint":
        return autoint_forward(P, inputs[0], inputs[1], kw["H"], kw["S"], kw.get("activation", "relu"), kw.get("use_res", False)), None
    if kind == "din":
        return din_forward(P, inputs, kw["user_keys"], kw["item_keys"], kw["maxlen"], kw["att_activation"], kw["ffn_activation"],
                           kw["n_ffn"], training=training, new_moving=new_moving), None
    if kind == "sasrec":
        return sasrec_forward(P, inputs[0], inputs[1], inputs[2], kw["n_blocks"], kw.get("H", 1), kw.get("eps", 1e-6))
    raise ValueError(kind)


def train_step(kind, W, opt, inputs, y, l2, **kw):
    """One Keras training step on {name: ndarray} weights W (updated in place).  Returns (predictions, loss, reg loss);
    loss = mean Keras BCE against y, or the model's add_loss for 'sasrec' (y = None, predictions = logits)."""
    P = {k: T(v, grad=not is_moving(k)) for k, v in W.items()}
    new_moving = {}
    if kind == "esmm":
        ctr, ctcvr = esmm_forward(P, inputs, kw["user_keys"], kw["user_cols"], kw["item_keys"], kw["item_cols"], training=True,
                                  new_moving=new_moving)
        p = torch.stack([ctr, ctcvr])
        own_loss = keras_bce(ctr, T(y[0]).reshape(-1)) + keras_bce(ctcvr, T(y[1]).reshape(-1))
    else:
        p, own_loss = _forward(kind, P, inputs, True, new_moving, kw)
    loss = own_loss if own_loss is not None else keras_bce(p, T(y).reshape(-1))
    reg = reg_loss(P, l2)
    (loss + reg).backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in P.items() if v.requires_grad}
    opt.apply(W, {k: g for k, g in grads.items() if g is not None})
    for k, v in new_moving.items():
        W[k] = v.numpy()
    return p.detach().numpy(), float(loss.detach()), float(reg.detach()) if hasattr(reg, "detach") else float(reg)


def gradients(kind, W, inputs, y, **kw):
    """{name: dLoss/dW} of one batch without the regularisers or the optimiser (kernel-level backward checks)"""
    P = {k: T(v, grad=not is_moving(k)) for k, v in W.items()}
    p, own_loss = _forward(kind, P, inputs, True, {}, kw)
    loss = own_loss if own_loss is not None else keras_bce(p, T(y).reshape(-1))
    loss.backward()
    return {k: v.grad.numpy() for k, v in P.items() if v.requires_grad and v.grad is not None}, float(loss.detach())


def predict(kind, W, inputs, **kw):
    P = {k: T(v) for k, v in W.items()}
    with torch.no_grad():
        p, _ = _forward(kind, P, inputs, False, {}, kw)
        return p.numpy()
