"""CPU oracle of the TRAINING step (TEST INFRASTRUCTURE ONLY; parity unpinned like the rest of oracle/: TensorFlow is
not importable here and the reference ships no fixtures).

Restates, in fp64 torch with autograd, what `model.fit` does for one batch of the ctr models whose mirrors have a
training-mode forward — src/ctr/dlrm/model.py:42-54 (intended form + the cited paper's dot interaction),
src/ctr/deep_fm/model.py:50-65, src/ctr/dcn/model.py:45-57 — with BatchNormalization in TRAINING mode
(src/ctr/layers/modules.py:131: batch mean / biased variance, eps 1e-3, moving averages with momentum 0.99), the Keras
binary cross-entropy on probabilities (clip + eps inside the logs) and tf.keras.optimizers.Adam (TF2 defaults), the
models' l2 regularisers added to the loss exactly as Keras does (embeddings_regularizer=l2(c): c * sum(w^2)).

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
    best = 0.0
    for k, c in l2.items():
        if name == k or name.startswith(k) or name.endswith(k):
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
    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t, self.m, self.v = lr, b1, b2, eps, 0, {}, {}

    def apply(self, W, grads):
        """W: {name: fp64 ndarray} updated in place; grads include the regularisers' gradients"""
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            m = self.m.setdefault(k, np.zeros_like(W[k]))
            v = self.v.setdefault(k, np.zeros_like(W[k]))
            m[...] = self.b1 * m + (1 - self.b1) * g
            v[...] = self.b2 * v + (1 - self.b2) * g * g
            W[k] = W[k] - lr_t * m / (np.sqrt(v) + self.eps)


def train_step(kind, W, opt, inputs, y, l2, **kw):
    """One Keras training step on {name: ndarray} weights W (updated in place).  Returns (predictions, BCE, reg loss)."""
    P = {k: T(v, grad=not is_moving(k)) for k, v in W.items()}
    new_moving = {}
    if kind == "dlrm":
        p = dlrm_forward(P, inputs[0], np.asarray(inputs[1]), kw.get("interaction", "dot"), training=True, new_moving=new_moving)
    elif kind == "deepfm":
        p = deepfm_forward(P, inputs[0], np.asarray(inputs[1]), training=True, new_moving=new_moving)
    else:
        p = dcn_forward(P, np.asarray(inputs), training=True, new_moving=new_moving)
    bce = keras_bce(p, T(y).reshape(-1))
    reg = reg_loss(P, l2)
    (bce + reg).backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in P.items() if v.requires_grad}
    opt.apply(W, {k: g for k, g in grads.items() if g is not None})
    for k, v in new_moving.items():
        W[k] = v.numpy()
    return p.detach().numpy(), float(bce), float(reg)


def predict(kind, W, inputs, **kw):
    P = {k: T(v) for k, v in W.items()}
    with torch.no_grad():
        if kind == "dlrm":
            return dlrm_forward(P, inputs[0], np.asarray(inputs[1]), kw.get("interaction", "dot"), training=False).numpy()
        if kind == "deepfm":
            return deepfm_forward(P, inputs[0], np.asarray(inputs[1]), training=False).numpy()
        return dcn_forward(P, np.asarray(inputs), training=False).numpy()
