"""Independent torch-CPU restatement of the hot path (TEST INFRASTRUCTURE ONLY).

Second, separately written restatement of the reference semantics, used only to cross-check
oracle/ref_numpy.py (tests/test_oracle_kat.py): two restatements that agree to 1e-12 in fp64 are
the defence against transcription mistakes, because the reference itself cannot be executed here
(PARITY UNPINNED, see ref_numpy.py).  Deliberately written with different primitives (einsum,
index_select, F.softmax, F.layer_norm, F.embedding) than the numpy version.
Citations: file:line relative to the reference repository root."""
import torch
import torch.nn.functional as F

NEG = float(torch.tensor(-2 ** 32 + 1, dtype=torch.float32))  # -4294967296.0


def t64(a):
    return torch.as_tensor(a).to(torch.float64)


def ids64(a):
    a = torch.as_tensor(a)
    return torch.trunc(a).to(torch.int64) if a.is_floating_point() else a.to(torch.int64)


def embed(table, ids):
    """Embedding lookup with TF-GPU out-of-range semantics (zeros)."""
    table = t64(table)
    ids = ids64(ids)
    bad = (ids < 0) | (ids >= table.shape[0])
    out = F.embedding(ids.clamp(0, table.shape[0] - 1), table)
    return out.masked_fill(bad.unsqueeze(-1), 0.0)


def gather_concat(tables, ids):
    ids = torch.as_tensor(ids)
    return torch.cat([embed(t, ids[:, f]) for f, t in enumerate(tables)], dim=-1)


def act(x, name, alpha=None):
    if name in (None, "linear", "none"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "prelu":
        a = 0.0 if alpha is None else t64(alpha)
        return torch.where(x >= 0, x, a * x)
    raise ValueError(name)


def fm_model(dense, ids, vocab, w0, w, V):
    """src/ctr/fm/model.py:34-53 via an explicit one-hot built with scatter."""
    dense, w0, w, V = t64(dense), t64(w0), t64(w), t64(V)
    ids = ids64(ids)
    B = dense.shape[0]
    parts = [dense]
    for f, vf in enumerate(vocab):
        oh = torch.zeros(B, vf, dtype=torch.float64)
        ok = (ids[:, f] >= 0) & (ids[:, f] < vf)
        oh[torch.nonzero(ok).squeeze(1), ids[ok, f]] = 1.0
        parts.append(oh)
    stack = torch.cat(parts, dim=1)
    first = w0 + stack @ w
    sv = stack @ V.t()
    second = 0.5 * ((sv ** 2) - (stack ** 2) @ (V.t() ** 2)).sum(dim=1, keepdim=True)
    return torch.sigmoid(first + second)


def fm_layer(first, second, w):
    """src/ctr/layers/modules.py:57-72."""
    first, second, w = t64(first), t64(second), t64(w).reshape(-1)
    first_order = torch.einsum("bj,j->", first, w)
    s = second.sum(dim=1)
    q = (second * second).sum(dim=1)
    return (first_order + 0.5 * (s * s - q)).reshape(-1, 1)


def cross_network(x, W, Bv):
    """src/ctr/layers/modules.py:105-112 in the scalar form x0 * <x_l, w_l> + b_l + x_l."""
    x0 = t64(x)
    W, Bv = t64(W), t64(Bv)
    xl = x0
    for l in range(W.shape[0]):
        s = torch.einsum("bd,d->b", xl, W[l])
        xl = x0 * s[:, None] + Bv[l][None, :] + xl
    return xl


def pairwise_dot(X):
    X = t64(X)
    Z = torch.einsum("bik,bjk->bij", X, X)
    n = X.shape[1]
    idx = torch.tril_indices(n, n, offset=-1)  # row-major (i, j), i > j
    return Z[:, idx[0], idx[1]]


def bn_infer(x, gamma, beta, mean, var, eps=1e-3):
    return F.batch_norm(x, t64(mean), t64(var), t64(gamma), t64(beta), False, 0.0, eps)


def dnn_ctr(x, layers, activation="relu", bn=None):
    x = t64(x)
    d = x.shape[-1]
    if bn is None:
        bn = dict(gamma=torch.ones(d), beta=torch.zeros(d), mean=torch.zeros(d), var=torch.ones(d))
    x = bn_infer(x, bn["gamma"], bn["beta"], bn["mean"], bn["var"])
    for W, b in layers:
        x = act(F.linear(x, t64(W).t(), t64(b)), activation)
    return x


def mha_ctr(xq, xk, xv, Wq, Wk, Wv, W0=None, H=1, S=None, activation="relu"):
    """src/ctr/layers/modules.py:285-325."""
    xq, xk, xv = t64(xq), t64(xk), t64(xv)
    S = S if S is not None else Wq.shape[1] // H
    q = act(xq @ t64(Wq), activation)
    k = act(xk @ t64(Wk), activation)
    v = act(xv @ t64(Wv), activation)
    B, N, _ = q.shape
    qh = q.view(B, N, H, S)
    kh = k.view(B, N, H, S)
    vh = v.view(B, N, H, S)
    logits = torch.einsum("bihs,bjhs->bhij", qh, kh) * (float(S) ** 0.5)   # "/ (S ** -0.5)"
    p = F.softmax(logits, dim=-1)
    out = torch.einsum("bhij,bjhs->bihs", p, vh).reshape(B, N, H * S)
    if W0 is not None:
        out = torch.relu(out + act(xv @ t64(W0), activation))
    return out


def din_attention(q, k, v, mask, W, b, activation="sigmoid", alpha=None):
    """src/ctr/layers/modules.py:144-175."""
    q, k, v, W, b = t64(q), t64(k), t64(v), t64(W).reshape(-1, 1), t64(b)
    B, T, d = k.shape
    qq = q[:, None, :].expand(B, T, d)
    info = torch.cat([qq, k, qq - k, qq * k], dim=-1)
    s = act(info @ W + b, activation, alpha).reshape(B, T)
    if mask is None:
        s = torch.full_like(s, NEG)
    else:
        s = torch.where(torch.as_tensor(mask) == 0, torch.full_like(s, NEG), s)
    return torch.einsum("bt,btd->bd", F.softmax(s, dim=-1), v)


def mha_match(x, mask, P, H):
    """src/match/layers/modules.py:115-131 (self-attention form)."""
    x = t64(x)
    q = x @ t64(P["Wq"]) + t64(P["bq"])
    k = x @ t64(P["Wk"]) + t64(P["bk"])
    v = x @ t64(P["Wv"]) + t64(P["bv"])
    B, S, dm = q.shape
    dk = dm // H
    logits = torch.einsum("bihd,bjhd->bhij", q.view(B, S, H, dk), k.view(B, S, H, dk)) / (float(dk) ** 0.5)
    m = torch.as_tensor(mask).reshape(B, 1, S, 1)
    logits = torch.where(m == 0, torch.full_like(logits, NEG), logits)
    p = F.softmax(logits, dim=-1)
    return torch.einsum("bhij,bjhd->bihd", p, v.view(B, S, H, dk)).reshape(B, S, dm)


def encoder(x, mask, P, H, eps=1e-6):
    """src/match/layers/modules.py:173-185."""
    x = t64(x)
    d = x.shape[-1]
    att = mha_match(x, mask, P, H)
    o1 = F.layer_norm(x + att, (d,), t64(P["ln1_g"]), t64(P["ln1_b"]), eps)
    f = torch.relu(o1 @ t64(P["W1"]) + t64(P["b1"])) @ t64(P["W2"]) + t64(P["b2"])
    return F.layer_norm(o1 + f, (d,), t64(P["ln2_g"]), t64(P["ln2_b"]), eps)


def sasrec(seq, pos, neg, T_seq, T_pos, T_neg, blocks, H=1, eps=1e-6):
    """src/match/sasrec/model.py:60-97."""
    seq = ids64(seq)
    mask = (seq != 0).to(torch.float64)
    x = embed(T_seq, seq) * mask[..., None]
    for P in blocks:
        x = encoder(x, mask, P, H, eps) * mask[..., None]
    info = x[:, -1, :]
    pos_s = torch.einsum("bd,bjd->bj", info, embed(T_pos, pos))
    neg_s = torch.einsum("bd,bjd->bj", info, embed(T_neg, neg))
    loss = torch.mean(-torch.log(torch.sigmoid(pos_s)) - torch.log(1 - torch.sigmoid(neg_s))) / 2
    return torch.cat([pos_s, neg_s], dim=-1), loss
