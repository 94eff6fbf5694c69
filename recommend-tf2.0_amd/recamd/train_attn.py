"""Training-mode forwards of the attention-shaped models: classic FM (src/ctr/fm/model.py:34-53, trained by
src/ctr/fm/train.py:43-67), AutoInt in its (B, fields, D) form (src/ctr/autoint/model.py:44-55 + the interacting layer of
src/ctr/layers/modules.py:285-325), DIN in its canonical form (src/ctr/din/model.py:57-93 with the AttentionLayer of
src/ctr/layers/modules.py:137-175; trained by src/ctr/din/train.py:95-114) and SASRec (src/match/sasrec/model.py:60-97, whose
loss is the add_loss of :93-95).

Same construction as recamd/train.py: a tape of backward closures over explicit HIP kernels (csrc/train_attn.hip for
the attention cores, the DIN pooling, PReLU / Dice, LayerNormalization, the rank loss, the dot scores, classic FM and
dropout; the projections / FFN reuse the Dense backward).  No autograd engine, nothing on the CPU."""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

from . import nn, ops
from ._lib import C
from .train import (TRAIN_FORWARDS, Tape, TrainState, Var, _s, _ws, bn_fwd, colsum, concat_cols, dense_fwd, sigmoid_bce, sum_squares, transpose,
                    weight_grad)


# ---- small tape ops ----------------------------------------------------------------------------------------------
def view_fwd(tape: Tape, x: Var, shape) -> Var:
    """reshape of a contiguous activation (no copy); the gradient flows back through the inverse view"""
    src = x.v if x.v.is_contiguous() else x.v.contiguous()
    y = Var(src.view(*shape))

    def bwd():
        if y.g is not None:
            x.acc(y.g.contiguous().view(src.shape))
    tape.ops.append(bwd)
    return y


def matmul_act_fwd(tape: Tape, x: Var, W: torch.Tensor, gname: str, act) -> Var:
    """act(x W) for a bare kernel (the ctr MultiHeadAttention builds bias-free Dense layers inside call(),
    src/ctr/layers/modules.py:255-269): x (M, K) contiguous"""
    y = Var(ops.dense(x.v, W, None, act))

    def bwd():
        dy = y.g if y.g.is_contiguous() else y.g.contiguous()
        if act not in (None, "linear", "none"):
            dy = dy.clone()
            C.act_grad_f32(dy.data_ptr(), dy.stride(0), y.v.data_ptr(), y.v.stride(0), dy.shape[0], dy.shape[1],
                           ops._act_id(act), _s())
        tape.add_grad(gname, weight_grad(x.v, dy))
        x.acc(ops.dense(dy, transpose(W)))
    tape.ops.append(bwd)
    return y


def dropout_fwd(tape: Tape, x: Var, rate: float) -> Var:
    """tf.keras.layers.Dropout(rate)(x, training=True) with the library's counter-based mask (rec_dropout_f32); the
    seed is (tape.seed, position of the layer on the tape), so a step is reproducible and the backward reuses it"""
    if not rate:
        return x
    seed = (int(getattr(tape, "seed", 0)) << 20) + len(tape.ops)
    xin = x.v if x.v.is_contiguous() else x.v.contiguous()
    y = Var(torch.empty_like(xin))
    C.dropout_f32(xin.data_ptr(), xin.numel(), float(rate), seed, y.v.data_ptr(), _s())

    def bwd():
        g = y.g if y.g.is_contiguous() else y.g.contiguous()
        dx = torch.empty_like(g)
        C.dropout_f32(g.data_ptr(), g.numel(), float(rate), seed, dx.data_ptr(), _s())
        x.acc(dx)
    tape.ops.append(bwd)
    return y


def attn_core_fwd(tape: Tape, q: Var, k: Var, v: Var, row_mask: Optional[torch.Tensor], H: int, S: int, scale: float) -> Var:
    """softmax(scale q k^T [query rows with mask 0 -> uniform]) v per (sample, head); q (B, Nq, H*S), k / v (B, Nk, H*S)"""
    qv, kv, vv = (t.v if t.v.is_contiguous() else t.v.contiguous() for t in (q, k, v))
    B, Nq, hs = qv.shape
    Nk = kv.shape[1]
    out = Var(torch.empty((B, Nq, hs), dtype=torch.float32, device=qv.device))
    mptr = 0 if row_mask is None else row_mask.data_ptr()
    C.attn_core_f32(qv.data_ptr(), hs, kv.data_ptr(), hs, vv.data_ptr(), hs, mptr, B, Nq, Nk, H, S, float(scale),
                    out.v.data_ptr(), hs, _s())

    def bwd():
        do = out.g if out.g.is_contiguous() else out.g.contiguous()
        dq, dk, dv = torch.empty_like(qv), torch.empty_like(kv), torch.empty_like(vv)
        ws = _ws(C.attn_core_grad_workspace_bytes(B, Nq, Nk, H), qv.device)
        C.attn_core_grad_f32(qv.data_ptr(), hs, kv.data_ptr(), hs, vv.data_ptr(), hs, mptr, do.data_ptr(), hs, B, Nq, Nk, H, S,
                             float(scale), dq.data_ptr(), hs, dk.data_ptr(), hs, dv.data_ptr(), hs, ws.data_ptr(), _s())
        q.acc(dq)
        k.acc(dk)
        v.acc(dv)
    tape.ops.append(bwd)
    return out


def add_act_fwd(tape: Tape, a: Var, b: Var, act=None) -> Var:
    """act(a + b), elementwise (relu / None)"""
    y = Var(ops.axpby_act(a.v, b.v, 1.0, 1.0, act))

    def bwd():
        g = y.g if y.g.is_contiguous() else y.g.contiguous()
        if act not in (None, "linear", "none"):
            g = g.clone()
            g2, y2 = g.view(-1, g.shape[-1]), y.v.view(-1, g.shape[-1])
            C.act_grad_f32(g2.data_ptr(), g2.stride(0), y2.data_ptr(), y2.stride(0), g2.shape[0], g2.shape[1],
                           ops._act_id(act), _s())
        a.acc(g)
        b.acc(g)
    tape.ops.append(bwd)
    return y


def layernorm_fwd(tape: Tape, ln: nn.LayerNormalization, name: str, x: Var, r: Optional[Var], row_mask=None) -> Var:
    """LayerNormalization(x + r) [* row_mask] (src/match/layers/modules.py:175,183-185)"""
    xv = x.v if x.v.is_contiguous() else x.v.contiguous()
    rv = None if r is None else (r.v if r.v.is_contiguous() else r.v.contiguous())
    d = xv.shape[-1]
    if not ln.built:
        ln.build(d)
    y = Var(ops.layernorm_residual(xv, rv, ln._w["gamma"], ln._w["beta"], ln.epsilon, row_mask))

    def bwd():
        dy = y.g if y.g.is_contiguous() else y.g.contiguous()
        M = xv.numel() // d
        ds, xhat, dym = torch.empty_like(xv), torch.empty_like(xv), torch.empty_like(xv)
        C.layernorm_residual_grad_f32(xv.data_ptr(), 0 if rv is None else rv.data_ptr(), ln._w["gamma"].data_ptr(),
                                      0 if row_mask is None else row_mask.data_ptr(), dy.data_ptr(), M, d,
                                      float(ln.epsilon), ds.data_ptr(), xhat.data_ptr(), dym.data_ptr(), _s())
        tape.add_grad(name + "/gamma", colsum(dym.view(M, d), xhat.view(M, d)))
        tape.add_grad(name + "/beta", colsum(dym.view(M, d)))
        x.acc(ds)
        if r is not None:
            r.acc(ds)
    tape.ops.append(bwd)
    return y


def din_pool_fwd(tape: Tape, layer, name: str, q: Var, kv: Var, mask: Optional[torch.Tensor]) -> Var:
    """AttentionLayer([q, k, v, mask]) with k = v = the behaviour embeddings (src/ctr/din/model.py intended form)"""
    kt = kv.v if kv.v.is_contiguous() else kv.v.contiguous()
    qt = q.v if q.v.is_contiguous() else q.v.contiguous()
    B, T, d = kt.shape
    if not layer.built:
        layer.build(d)
    W, bias, alpha = layer._w["kernel"], layer._w["bias"], layer._w.get("alpha")
    act = layer.activation
    out = Var(ops.din_attention_pool(qt, kt, kt, mask, W, bias, act, alpha))

    def bwd():
        do = out.g if out.g.is_contiguous() else out.g.contiguous()
        dq, dk, dv = torch.empty_like(qt), torch.empty_like(kt), torch.empty_like(kt)
        part = torch.empty((B, 4 * d + 2), dtype=torch.float32, device=qt.device)
        C.din_attn_pool_grad_f32(qt.data_ptr(), kt.data_ptr(), kt.data_ptr(), 0 if mask is None else mask.data_ptr(),
                                 1 if mask is None else 0, W.data_ptr(), bias.data_ptr(), ops._act_id(act), ops._ptr(alpha),
                                 do.data_ptr(), B, T, d, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), part.data_ptr(), _s())
        g = colsum(part)
        tape.add_grad(name + "/kernel", g[:4 * d].reshape(4 * d, 1))
        tape.add_grad(name + "/bias", g[4 * d:4 * d + 1])
        if alpha is not None:
            tape.add_grad(name + "/alpha", g[4 * d + 1:4 * d + 2])
        q.acc(dq)
        kv.acc(ops.axpby_act(dk, dv, 1.0, 1.0, None))
    tape.ops.append(bwd)
    return out


# ---- classic FM ----------------------------------------------------------------------------------------------------
def fm_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/fm/model.py:34-53, training mode (no BatchNormalization / Dropout in this model)."""
    dense_inputs, sparse_inputs = inputs
    dense_inputs = nn.to_device_f32(dense_inputs, m.device)
    ids = nn.to_device_ids(sparse_inputs, m.device)
    if ids.dtype != torch.int32:
        ids = ids.to(torch.int32)
    vocab = [int(feat['feat_num']) for feat in m.sparse_feature_columns]
    w0, w, V = m._w['w0'], m._w['w'], m._w['V']
    p = ops.fm_onehot(dense_inputs, ids, vocab, w0, w, V)
    yt = y_true.reshape(-1).contiguous()
    loss = ops.binary_crossentropy(yt, p.reshape(-1))

    def bwd():
        n = yt.numel()
        dz = torch.empty(n, dtype=torch.float32, device=p.device)
        C.bce_sigmoid_grad_f32(yt.data_ptr(), p.reshape(-1).data_ptr(), n, grad_scale / n, dz.data_ptr(), _s())
        dw, dV = torch.zeros_like(w), torch.zeros_like(V)
        C.fm_onehot_grad_f32(dense_inputs.data_ptr(), dense_inputs.stride(0), dense_inputs.shape[1], ids.data_ptr(),
                             ids.stride(0), vocab, V.data_ptr(), V.shape[0], dz.data_ptr(), n, dw.data_ptr(), dV.data_ptr(), _s())
        tape.add_grad("w0", colsum(dz.view(n, 1)))
        tape.add_grad("w", dw)
        tape.add_grad("V", dV)
    tape.ops.append(bwd)
    return p, loss


# ---- AutoInt ---------------------------------------------------------------------------------------------------------
def _embed_names(n: int):
    return [f"embed_{i}/embeddings" for i in range(n)]


def autoint_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """AutoInt(mode='intended'), training mode: fields (B, F + nd, D) -> stacked interacting layers -> Dense(1) -> sigmoid."""
    from .train import gather_concat_fwd
    if m.mode != 'intended':
        raise NotImplementedError("training: AutoInt(mode='as_written') mixes samples (see ctr/autoint/model.py); train the intended form")
    dense_inputs, sparse_inputs = inputs
    dense_inputs = nn.to_device_f32(dense_inputs, m.device)
    ids = nn.to_device_ids(sparse_inputs, m.device)
    B, F = ids.shape
    D = m._group.dims[0]
    emb = gather_concat_fwd(tape, state, ops.TableGroup(m._group.tables), _embed_names(F), ids)        # (B, F*D), :46
    if m.embed_dense:
        E = m._w['dense_embed']
        nd = E.shape[0]
        dpart = torch.empty((B, nd * D), dtype=torch.float32, device=m.device)
        ops.scale_embed(dense_inputs, E, dpart)
        h = h0 = Var(concat_cols([emb.v, dpart]).view(B, F + nd, D))

        def bwd_cat():          # `h` is rebound by the layer loop below: the closure keeps its own name
            g = h0.g.contiguous().view(B, (F + nd) * D)
            emb.acc(g[:, :F * D].contiguous())
            gd = g[:, F * D:].contiguous().view(B, nd, D)
            dE = torch.empty_like(E)
            for n in range(nd):     # dE[n] = sum_b dense[b, n] g[b, n, :]
                dE[n] = colsum(gd[:, n, :], row_w=dense_inputs[:, n].contiguous())
            tape.add_grad("dense_embed", dE)
        tape.ops.append(bwd_cat)
        N = F + nd
    else:
        h = view_fwd(tape, emb, (B, F, D))
        N = F
    for li, L in enumerate(m.attention_layers):
        din = h.v.shape[-1]
        if not L.built:
            L.build(din)
        H, S, act = L._head_num, L._head_size, L._activation
        x2 = view_fwd(tape, h, (B * N, din))
        pre = f"attention_{li}/"
        q = view_fwd(tape, matmul_act_fwd(tape, x2, L._w['Wq'], pre + "Wq", act), (B, N, H * S))
        k = view_fwd(tape, matmul_act_fwd(tape, x2, L._w['Wk'], pre + "Wk", act), (B, N, H * S))
        v = view_fwd(tape, matmul_act_fwd(tape, x2, L._w['Wv'], pre + "Wv", act), (B, N, H * S))
        att = attn_core_fwd(tape, q, k, v, None, H, S, math.sqrt(float(S)))            # "/ (S ** -0.5)", modules.py:235
        if L._use_res:
            r = view_fwd(tape, matmul_act_fwd(tape, x2, L._w['W0'], pre + "W0", act), (B, N, H * S))
            att = add_act_fwd(tape, att, r, 'relu')                                     # :316-323
        h = att
    flat = view_fwd(tape, h, (B, N * h.v.shape[-1]))
    top = dense_fwd(tape, m.final_dense, "final_dense", flat)
    return sigmoid_bce(tape, [top], y_true, grad_scale)


# ---- DIN ---------------------------------------------------------------------------------------------------------------
def din_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """DIN(mode='intended'), training mode (src/ctr/din/model.py:57-93 with the AttentionLayer pooling): BatchNormalization on
    batch statistics (:83), Dense(PReLU() | Dice()) stack (:52,:85-87), Dropout (:89), Dense(1), sigmoid."""
    from .train import gather_concat_fwd
    if m.mode != 'intended':
        raise NotImplementedError("training: DIN as written raises for maxlen > 1; train mode='intended'")
    user_dense, user_sparse, item_dense, item_sparse, behavior = [nn.to_device_f32(t, m.device) for t in inputs]
    B = user_sparse.shape[0]
    to_i32 = lambda t: t.to(torch.int32).contiguous()    # the Keras Embedding cast (truncation toward zero)  # noqa: E731
    uids = to_i32(m._cols(user_sparse, m._user_cols))
    iids = to_i32(m._cols(item_sparse, m._item_cols))
    bids = to_i32(m._cols(behavior, m._beh_cols))
    if not m._beh_regular:
        raise NotImplementedError("training: behaviour columns must be maxlen repeats of the item tables")
    unames = ['embed_' + k + '/embeddings' for k in m.user_sparse_feature_index]
    inames = ['embed_' + k + '/embeddings' for k in m.item_sparse_feature_index]
    n_item, T = len(inames), m.maxlen
    user_emb = gather_concat_fwd(tape, state, ops.TableGroup(m._user_group.tables), unames, uids)       # :62-64
    item_emb = gather_concat_fwd(tape, state, ops.TableGroup(m._item_group.tables), inames, iids)       # :66-68
    beh_flat = gather_concat_fwd(tape, state, ops.TableGroup(m._item_group.tables), inames, bids.view(B * T, n_item))  # :71-74
    d = item_emb.v.shape[1]
    beh = view_fwd(tape, beh_flat, (B, T, d))
    mask = (bids.view(B, T, n_item)[:, :, 0] != 0).to(torch.float32).contiguous()
    att = din_pool_fwd(tape, m.attention_layer, "attention_layer", item_emb, beh, mask)
    parts = [Var(user_dense), user_emb, Var(item_sparse.contiguous()), item_emb, att]                  # :62-68, :81
    widths = [p.v.shape[1] for p in parts]
    allv = Var(concat_cols([p.v for p in parts]))

    def bwd_cat():
        off = 0
        for p, wdt in zip(parts, widths):
            if p in (user_emb, item_emb, att):
                p.acc(allv.g[:, off:off + wdt].contiguous())
            off += wdt
    tape.ops.append(bwd_cat)
    x = bn_fwd(tape, m.bn, "bn", allv)                                                                 # :83
    for i, dense in enumerate(m.ffn):                                                                  # :85-87
        x = dense_fwd(tape, dense, f"ffn_{i}", x)
    x = dropout_fwd(tape, x, getattr(m.dropout, "rate", 0.0))                                           # :89
    top = dense_fwd(tape, m.final_output, "final_output", x)
    return sigmoid_bce(tape, [top], y_true, grad_scale)                                               # :91


# ---- SASRec ------------------------------------------------------------------------------------------------------------
def _rows_fwd(tape: Tape, state: TrainState, table: torch.Tensor, name: str, ids: torch.Tensor) -> Var:
    """Embedding.call on one table for (B, n) int32 ids (out-of-range / -1 -> zero rows, no gradient) -> (B*n, d)"""
    from .train import gather_concat_fwd
    return gather_concat_fwd(tape, state, ops.TableGroup([table]), [name], ids.reshape(-1, 1).contiguous())


def rows_slice_fwd(tape: Tape, x: Var, lo: int, hi: int) -> Var:
    """rows lo..hi-1 of a (n, d) activation (a view); the gradient is added into the same rows of x.g"""
    y = Var(x.v[lo:hi])

    def bwd():
        if y.g is None:
            return
        if x.g is None:
            x.g = torch.zeros_like(x.v)
        x.g[lo:hi] += y.g
    tape.ops.append(bwd)
    return y


def _sharded_rows_fwd(tape: Tape, state: TrainState, st, names, vids: torch.Tensor) -> Var:
    """All lookups of a model through ONE sharded exchange (recamd.dist.ShardedTables): (n_lookups, d) rows, zero for
    vids = -1.  Backward: the gradient of every lookup travels to the owner of its row (local rows directly, remote
    ones by the reverse all-to-all) into the sharded gradient arena (state.sharded_grad)."""
    space, uidx, plan = st.lookup_rows(vids, keep_plan=True)
    E = Var(ops.gather_concat(ops.TableGroup([space]), uidx.reshape(-1, 1).contiguous()))

    def bwd():
        ga = state.sharded_grad(st, names)
        g = E.g if E.g is not None else torch.zeros_like(E.v)
        if plan is None:                       # one rank: the arena is the row space, uidx the arena row
            ops.embedding_grad(ops.TableGroup([ga]), uidx.reshape(-1, 1).contiguous(), g)
        else:
            st.backward(plan, g, ga)
    tape.ops.append(bwd)
    return E


def sasrec_train_forward(tape: Tape, state: TrainState, m, inputs, y_true=None, grad_scale: float = 1.0):
    """src/match/sasrec/model.py:60-97, training mode: every block encodes all positions (the gradients of the K / V
    projections need them); the loss is the model's add_loss (:93-95).  Returns (logits, loss).
    Row-sharded tables (SASRec(sharded=...), BASELINE configs[4]): the seq / pos / neg lookups of :75-79 travel in one
    exchange and their gradients return to the owners through ShardedTables.backward; every rank scales its loss by
    grad_scale = 1 / world, the dense parameters merge by all-reduce (train_step), the tables never do."""
    seq, pos, neg = [nn.to_device_ids(t, m.device) for t in inputs]
    seq, pos, neg = [t if t.dtype == torch.int32 else t.to(torch.int32) for t in (seq, pos, neg)]
    B, S = seq.shape
    n_neg = neg.shape[1]
    d = m.d_model
    tb = m.user_embed_layers
    names = {k: f"user_embed_{k}/embeddings" for k in ("seq_item", "pos_item", "neg_item")}
    mask = (seq != 0).to(torch.float32).contiguous()                                                    # :72
    st = m._sharded
    if st is not None:
        vids = torch.cat([st.virtual_ids(0, seq, pad_id=0).reshape(-1), st.virtual_ids(1, pos).reshape(-1),
                          st.virtual_ids(2, neg).reshape(-1)])
        E = _sharded_rows_fwd(tape, state, st, [names["seq_item"], names["pos_item"], names["neg_item"]], vids)
        x = rows_slice_fwd(tape, E, 0, B * S)                                                           # (B*S, d), :75, :81-82
        cand = rows_slice_fwd(tape, E, B * S, B * S + B * (1 + n_neg))    # pos rows (B), then neg rows (B * n_neg)
    else:
        seq_m = torch.where(seq == 0, torch.full_like(seq, -1), seq)        # `seq_embed * mask` (:81-82): pad rows are zero
        x = _rows_fwd(tape, state, tb['embed_seq_item'].table, names["seq_item"], seq_m)              # (B*S, d), :75
    rate = float(getattr(m.dropout, "rate", 0.0) or 0.0)
    mflat = mask.reshape(-1)
    for bi, enc in enumerate(m.encoder_layer):                                                          # :84-86
        pre = f"encoder_{bi}/"
        mha = enc.mha
        H = mha.num_heads
        depth = d // H
        q = view_fwd(tape, dense_fwd(tape, mha.wq, pre + "mha/wq", x), (B, S, d))
        k = view_fwd(tape, dense_fwd(tape, mha.wk, pre + "mha/wk", x), (B, S, d))
        v = view_fwd(tape, dense_fwd(tape, mha.wv, pre + "mha/wv", x), (B, S, d))
        att = view_fwd(tape, attn_core_fwd(tape, q, k, v, mask, H, depth, 1.0 / math.sqrt(float(depth))), (B * S, d))
        att = dropout_fwd(tape, att, getattr(enc.dropout1, "rate", 0.0))
        out1 = layernorm_fwd(tape, enc.layernorm1, pre + "layernorm1", x, att)                          # modules.py:175
        f = dense_fwd(tape, enc.ffn.conv2, pre + "ffn/conv2", dense_fwd(tape, enc.ffn.conv1, pre + "ffn/conv1", out1))
        f = dropout_fwd(tape, f, getattr(enc.dropout2, "rate", 0.0))
        x = layernorm_fwd(tape, enc.layernorm2, pre + "layernorm2", out1, f, row_mask=mflat)            # :183 + `*= mask`
    xs = x
    seq_info = Var(xs.v.view(B, S, d)[:, -1, :].contiguous())                                           # :88

    def bwd_last():
        g = torch.zeros((B, S, d), dtype=torch.float32, device=m.device)
        g[:, -1, :] = seq_info.g
        xs.acc(g.view(B * S, d))
    tape.ops.append(bwd_last)
    logits = torch.empty((B, 1 + n_neg), dtype=torch.float32, device=m.device)
    if st is not None:
        # the candidate rows arrived with the exchange: "table" = those rows, "ids" = their positions
        pos_t = neg_t = cand.v
        pos_i = torch.arange(B, dtype=torch.int32, device=m.device).view(B, 1)
        neg_i = (B + torch.arange(B * n_neg, dtype=torch.int32, device=m.device)).view(B, n_neg)
    else:
        pos_t, neg_t = tb['embed_pos_item'].table, tb['embed_neg_item'].table
        pos_i, neg_i = pos.contiguous(), neg.contiguous()
    ops.gather_dot_scores(seq_info.v, pos_t, pos_i, out=logits[:, :1])                                   # :77, :90
    ops.gather_dot_scores(seq_info.v, neg_t, neg_i, out=logits[:, 1:])                                   # :79, :91
    loss = ops.pairwise_rank_loss(logits)                                                               # :93-95

    def bwd_loss():
        dl = torch.empty_like(logits)
        C.pairwise_rank_loss_grad_f32(logits.data_ptr(), logits.stride(0), B, n_neg, float(grad_scale), dl.data_ptr(),
                                      dl.stride(0), _s())
        dseq = torch.empty((B, d), dtype=torch.float32, device=m.device)
        if st is not None:
            gp = gn = torch.zeros_like(cand.v)
        else:
            gp, gn = state.grad(names["pos_item"]), state.grad(names["neg_item"])
        C.gather_dot_scores_grad_f32(seq_info.v.data_ptr(), pos_t.data_ptr(), gp.data_ptr(), pos_t.shape[0], d, pos_i.data_ptr(),
                                     pos_i.stride(0), pos_i.shape[1], dl.data_ptr(), dl.stride(0), B, dseq.data_ptr(), 0, _s())
        C.gather_dot_scores_grad_f32(seq_info.v.data_ptr(), neg_t.data_ptr(), gn.data_ptr(), neg_t.shape[0], d, neg_i.data_ptr(),
                                     neg_i.stride(0), n_neg, dl[:, 1:].data_ptr(), dl.stride(0), B, dseq.data_ptr(), 1, _s())
        if st is not None:
            cand.acc(gp)
        seq_info.acc(dseq)
    tape.ops.append(bwd_loss)
    m._logits = logits
    return logits, loss


TRAIN_FORWARDS.update({"FM": fm_train_forward, "AutoInt": autoint_train_forward, "DIN": din_train_forward,
                       "SASRec": sasrec_train_forward})


# ---- zoo models that reuse the path: Wide&Deep, Deep&Crossing, NCF --------------------------------------
def scale_fwd(tape: Tape, x: Var, c: float) -> Var:
    y = Var(ops.axpby_act(x.v, x.v, c, 0.0, None))

    def bwd():
        x.acc(ops.axpby_act(y.g, y.g, c, 0.0, None))
    tape.ops.append(bwd)
    return y


def concat_fwd(tape: Tape, parts: Sequence[Var]) -> Var:
    """tf.concat(parts, axis=-1) of 2-D activations; every part receives its column slice of the gradient"""
    y = Var(concat_cols([p.v for p in parts]))
    widths = [p.v.shape[1] for p in parts]

    def bwd():
        off = 0
        for p, w in zip(parts, widths):
            p.acc(y.g[:, off:off + w].contiguous())
            off += w
    tape.ops.append(bwd)
    return y


def mul_sigmoid_fwd(tape: Tape, a: Var, b: Var) -> Var:
    """sigmoid(a * b), elementwise (NCF's GMF vector, src/match/ncf/model.py:53-54)"""
    y = Var(ops.mul_act(a.v, b.v, 'sigmoid'))

    def bwd():
        g = y.g.contiguous().clone()
        C.act_grad_f32(g.data_ptr(), g.stride(0), y.v.data_ptr(), y.v.stride(0), g.shape[0], g.shape[1],
                       ops._act_id('sigmoid'), _s())
        a.acc(ops.mul_act(g, b.v, None))
        b.acc(ops.mul_act(g, a.v, None))
    tape.ops.append(bwd)
    return y


def tile_rows_fwd(tape: Tape, u: Var, T: int) -> Var:
    """tf.tile(u[:, None, :], [1, T, 1]) flattened to (B * T, dim); the backward sums the T copies — as a product with
    T stacked identity matrices on the Dense kernel (0 / 1 weights: exact)"""
    B, dim = u.v.shape
    if T == 1:
        return u
    y = Var(u.v[:, None, :].expand(B, T, dim).reshape(B * T, dim).contiguous())

    def bwd():
        eye = torch.eye(dim, dtype=torch.float32, device=u.v.device).repeat(T, 1)        # (T * dim, dim)
        u.acc(ops.dense(y.g.contiguous().view(B, T * dim), eye))
    tape.ops.append(bwd)
    return y


def dense_stack_fwd(tape: Tape, layers, name: str, x: Var, dropout_rate: float = 0.0) -> Var:
    """a BatchNorm-free Dense stack + Dropout (wide_deep DNN, match DNN)"""
    for i, layer in enumerate(layers):
        x = dense_fwd(tape, layer, f"{name}/dense_{i}", x)
    return dropout_fwd(tape, x, dropout_rate)


def wide_deep_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/wide_deep/model.py:66-79, training mode (trained by src/ctr/wide_deep/train.py)"""
    from .train import gather_concat_fwd
    dense_inputs, sparse_inputs = inputs
    dense_inputs = nn.to_device_f32(dense_inputs, m.device)
    ids = nn.to_device_ids(sparse_inputs, m.device)
    emb = gather_concat_fwd(tape, state, ops.TableGroup(m._group.tables), _embed_names(ids.shape[1]), ids)     # :68-69
    dv = Var(dense_inputs)
    x = concat_fwd(tape, [emb, dv])                                                                          # :70
    wide = dense_fwd(tape, m.linear.dense, "linear/dense", dv)                                               # :73
    deep = dense_stack_fwd(tape, m.dnn_network.dnn_network, "dnn_network", x, getattr(m.dnn_network.dropout, "rate", 0.0))
    deep = dense_fwd(tape, m.final_dense, "final_dense", deep)                                               # :75-76
    return sigmoid_bce(tape, [scale_fwd(tape, wide, 0.5), scale_fwd(tape, deep, 0.5)], y_true, grad_scale)   # :78


def deep_crossing_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/deep_crossing/model.py:42-51, training mode"""
    from .train import gather_concat_fwd
    ids = nn.to_device_ids(inputs, m.device)
    r = gather_concat_fwd(tape, state, ops.TableGroup(m._group.tables), _embed_names(ids.shape[1]), ids)       # :44-45
    for i, res in enumerate(m.res_network):                                                                  # :47-48
        h = dense_fwd(tape, res.layer2, f"res_{i}/layer2", dense_fwd(tape, res.layer1, f"res_{i}/layer1", r))
        r = add_act_fwd(tape, h, r, 'relu')                                          # src/ctr/layers/modules.py:33
    r = dropout_fwd(tape, r, getattr(m.res_dropout, "rate", 0.0))
    return sigmoid_bce(tape, [dense_fwd(tape, m.dense, "dense", r)], y_true, grad_scale)                      # :50


def ncf_train_forward(tape: Tape, state: TrainState, m, inputs, y_true=None, grad_scale: float = 1.0):
    """src/match/ncf/model.py:47-80, training mode; the objective is the model's add_loss (:75-77).  Returns (logits, loss)."""
    user_in, pos_in, neg_in = [nn.to_device_ids(t, m.device) for t in inputs]
    user_in, pos_in, neg_in = [t if t.dtype == torch.int32 else t.to(torch.int32) for t in (user_in, pos_in, neg_in)]
    B = user_in.shape[0]
    n_neg = neg_in.shape[1]
    user = _rows_fwd(tape, state, m.user_embedding.table, "user_embedding/embeddings", user_in)      # (B, dim)
    pos = _rows_fwd(tape, state, m.item_embedding.table, "item_embedding/embeddings", pos_in)        # (B, dim)
    neg = _rows_fwd(tape, state, m.neg_item_embedding.table, "neg_item_embedding/embeddings", neg_in)  # (B * n, dim)
    rate = getattr(m.dnn.dropout, "rate", 0.0)

    def branch(item: Var, T: int) -> Var:
        u = tile_rows_fwd(tape, user, T)                                                               # :63
        gmf = mul_sigmoid_fwd(tape, u, item)                                                           # :53-54
        mlp = dense_stack_fwd(tape, m.dnn.dnn_network, "dnn", concat_fwd(tape, [u, item]), rate)       # :61-66
        return dense_fwd(tape, m.dense, "dense", concat_fwd(tape, [gmf, mlp]))                         # :69-73, (B * T, 1)
    pos_l = branch(pos, 1)
    neg_l = view_fwd(tape, branch(neg, n_neg), (B, n_neg))
    logits = concat_fwd(tape, [pos_l, neg_l])                                                          # :79
    lg = logits.v
    loss = ops.pairwise_rank_loss(lg)                                                                  # :75-77

    def bwd_loss():
        dl = torch.empty_like(lg)
        C.pairwise_rank_loss_grad_f32(lg.data_ptr(), lg.stride(0), B, n_neg, float(grad_scale), dl.data_ptr(), dl.stride(0), _s())
        logits.acc(dl)
    tape.ops.append(bwd_loss)
    m._logits = lg
    return lg, loss


def mul_fwd(tape: Tape, a: Var, b: Var) -> Var:
    y = Var(ops.mul_act(a.v, b.v, None))

    def bwd():
        g = y.g if y.g.is_contiguous() else y.g.contiguous()
        a.acc(ops.mul_act(g, b.v, None))
        b.acc(ops.mul_act(g, a.v, None))
    tape.ops.append(bwd)
    return y


def bce_prob(tape: Tape, p: Var, y_true: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
    """mean Keras binary cross-entropy of probabilities p (B, 1) against y; the backward hands dL/dp to p"""
    yt = y_true.reshape(-1).contiguous()
    pv = p.v.reshape(-1).contiguous()
    loss = ops.binary_crossentropy(yt, pv)

    def bwd():
        n = yt.numel()
        dp = torch.empty(n, dtype=torch.float32, device=pv.device)
        C.bce_prob_grad_f32(yt.data_ptr(), pv.data_ptr(), n, grad_scale / n, dp.data_ptr(), _s())
        p.acc(dp.view(p.v.shape))
    tape.ops.append(bwd)
    return loss


def esmm_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/esmm/model.py:37-92 in training mode with the two-target loss of src/ctr/esmm/train.py:99-103
    (loss=["binary_crossentropy", "binary_crossentropy"], loss_weights [1, 1] on [pCTR, pCTCVR]).  The user / item DNNs and
    the embeddings are shared by the two towers: their gradients add up, and their BatchNormalization layers see two
    batches per step (two moving-average updates, as in Keras).  Returns ([pCTR, pCTCVR], loss_ctr + loss_ctcvr)."""
    from .train import dnn_fwd, gather_concat_fwd
    xs = [nn.to_device_f32(t, m.device) for t in inputs]
    y_ctr, y_cvr = y_true
    unames = ['embed_' + k + '/embeddings' for k in m.user_cate_feature_dict]
    inames = ['embed_' + k + '/embeddings' for k in m.item_cate_feature_dict]
    rate = getattr(m.user_dnn.dropout, "rate", 0.0)

    def tower(head, hname, un, uc, inum, ic):
        uids = uc[:, m._user_cols].to(torch.int32).contiguous()            # cate_input[:, v[0]] + the Embedding cast (:44-48)
        iids = ic[:, m._item_cols].to(torch.int32).contiguous()
        ue = gather_concat_fwd(tape, state, ops.TableGroup(m._user_group.tables), unames, uids)
        ie = gather_concat_fwd(tape, state, ops.TableGroup(m._item_group.tables), inames, iids)
        uf = dnn_fwd(tape, m.user_dnn, "user_dnn", concat_fwd(tape, [Var(un), ue]))        # :50-53
        itf = dnn_fwd(tape, m.item_dnn, "item_dnn", concat_fwd(tape, [Var(inum), ie]))     # :51-54
        x = dropout_fwd(tape, concat_fwd(tape, [uf, itf]), rate)                            # :56-57
        x = bn_fwd(tape, head.bn, hname + "/bn", x)                                         # :58
        x = dense_fwd(tape, head.dense, hname + "/dense", x)                                # :59
        return dense_fwd(tape, head.out, hname + "/out", x)                                 # :60 (sigmoid)
    ctr = tower(m.ctr_head, "ctr_head", *xs[:4])
    cvr = tower(m.cvr_head, "cvr_head", *xs[4:])
    ctcvr = mul_fwd(tape, ctr, cvr)                                                         # :44
    l1 = bce_prob(tape, ctr, y_ctr, grad_scale)
    l2 = bce_prob(tape, ctcvr, y_cvr, grad_scale)
    return [ctr.v, ctcvr.v], ops.axpby_act(l1.reshape(1), l2.reshape(1), 1.0, 1.0, None)


TRAIN_FORWARDS.update({"WideDeep": wide_deep_train_forward, "Deep_Crossing": deep_crossing_train_forward,
                       "NCF": ncf_train_forward, "ESMM": esmm_train_forward})


# ---- two-tower match models trained by their own scripts: match FM (src/match/fm/train.py), DSSM (src/match/dssm/dssm_train.py) ----
def _dict_ids(m, inputs: dict) -> torch.Tensor:
    """{feat: (B, 1)} -> (B, n) int32 ids in dict order (the reference walks `.items()`); float ids truncate like the
    Keras Embedding cast"""
    cols = [nn.to_device_f32(v, m.device).reshape(-1, 1) for v in inputs.values()]
    return torch.cat(cols, dim=1).to(torch.int32).contiguous()


def _tower_embed_fwd(tape: Tape, state: TrainState, m, inputs: dict, layers: dict, prefix: str) -> Var:
    from .train import gather_concat_fwd
    keys = list(inputs.keys())
    group = ops.TableGroup([layers['embed_{}'.format(k)].table for k in keys])
    names = [f"{prefix}_embed_{k}/embeddings" for k in keys]
    return gather_concat_fwd(tape, state, group, names, _dict_ids(m, inputs))


def match_fm_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/match/fm/model.py:61-82 in training mode (compiled with binary_crossentropy + Adam, src/match/fm/train.py:47).
    With a = stack V^T:  second = 0.5 (sum_k a_k^2 - stack^2 . vsq), vsq_l = sum_k V_kl^2, so
    d second / d stack = a V - stack * vsq  and  d second / d V = a^T (dz * stack) - V * colsum(dz * stack^2)."""
    user_in, item_in = inputs
    u = _tower_embed_fwd(tape, state, m, user_in, m.user_embed_layers, "user")               # :63-64
    it = _tower_embed_fwd(tape, state, m, item_in, m.item_embed_layers, "item")              # :68-69
    stack = concat_fwd(tape, [u, it])                                                       # :73-75
    w0, w, V = m._w['w0'], m._w['w'], m._w['V']
    x = stack.v
    vsq = colsum(V, V)                                                                      # (L,)
    a = ops.dense(x, transpose(V))                                                          # (B, k)
    ones = torch.ones((V.shape[0], 1), dtype=torch.float32, device=x.device)
    first = Var(ops.dense(x, w, w0))                                                        # :76
    second = Var(ops.axpby_act(ops.dense(ops.mul_act(a, a), ones), ops.dense(ops.mul_act(x, x), vsq.reshape(-1, 1)),
                               0.5, -0.5, None))                                            # :77-79

    def bwd():
        dz = first.g.contiguous()                 # = second.g: both logit parts receive the same dL/dz (B, 1)
        tape.add_grad("w0", colsum(dz))
        tape.add_grad("w", weight_grad(x, dz))
        xz = ops.scale_rows(x, dz.reshape(-1))                                               # dz_b * stack_b
        s = colsum(xz, x)                                                                    # sum_b dz_b stack_bl^2
        tape.add_grad("V", ops.axpby_act(weight_grad(a, xz), ops.mul_act(V, s.reshape(1, -1).expand_as(V).contiguous()),
                                         1.0, -1.0, None))
        dx = ops.axpby_act(ops.dense(dz, transpose(w)), ops.scale_rows(ops.dense(a, V), dz.reshape(-1)), 1.0, 1.0, None)
        stack.acc(ops.axpby_act(dx, ops.mul_act(xz, vsq.reshape(1, -1).expand_as(x).contiguous()), 1.0, -1.0, None))
    tape.ops.append(bwd)          # the tape runs backwards: sigmoid_bce's op (appended next) has filled first.g by then
    return sigmoid_bce(tape, [first, second], y_true, grad_scale)                            # :80-82


def dssm_train_forward(tape: Tape, state: TrainState, m, inputs, y_true=None, grad_scale: float = 1.0):
    """src/match/dssm/model.py:64-82 in training mode with the script's objective, loss = mean(y_pred)
    (src/match/dssm/dssm_train.py:47, src/match/utils/loss_util.py:11-13): y_pred is the ONE value sigmoid(cos(vec(item
    tower), vec(user tower))) the model emits for a batch, labels are ignored.  With c = <a, b> / (|a| |b|):
    dc/da = b / (|a| |b|) - c a / |a|^2 (and symmetrically for b).  Returns (y_pred (1, 1), loss)."""
    user_in, item_in = inputs
    ue = _tower_embed_fwd(tape, state, m, user_in, m.user_embed_layers, "user")              # :68-69
    ie = _tower_embed_fwd(tape, state, m, item_in, m.item_embed_layers, "item")              # :74-75
    uo = dense_stack_fwd(tape, m.user_dnn.dnn_network, "user_dnn", ue, getattr(m.user_dnn.dropout, "rate", 0.0))   # :72
    io = dense_stack_fwd(tape, m.item_dnn.dnn_network, "item_dnn", ie, getattr(m.item_dnn.dropout, "rate", 0.0))   # :77
    a, b = io.v.contiguous(), uo.v.contiguous()
    p = ops.cosine_flat(a, b, sigmoid=True)                                                  # :79-80
    loss = p.reshape(())

    def bwd():
        ab = colsum(colsum(a, b).view(-1, 1))
        sab, sa2, sb2, pv = [float(v) for v in torch.cat([ab, sum_squares(a), sum_squares(b), p.reshape(1)]).cpu()]
        inv = 1.0 / ((sa2 ** 0.5) * (sb2 ** 0.5))
        c = sab * inv
        g = grad_scale * pv * (1.0 - pv)                                                     # d mean(y_pred) / dc
        io.acc(ops.axpby_act(b, a, g * inv, -g * c / sa2, None))
        uo.acc(ops.axpby_act(a, b, g * inv, -g * c / sb2, None))
    tape.ops.append(bwd)
    return p.reshape(1, 1), loss


dssm_train_forward.ignores_labels = True

# keyed by module-qualified class name where a bare name is taken (ctr FM / match FM)
TRAIN_FORWARDS.update({"match.fm.model.FM": match_fm_train_forward, "Dssm": dssm_train_forward})
