"""§8f-1/-2, attention-shaped part: the backward kernels of csrc/train_attn.hip and the training steps of classic FM, AutoInt,
DIN and SASRec (recamd/train_attn.py) against fp64 torch autograd over oracle/ref_torch.py's restatements
(oracle/ref_train.py).  Tolerances as in tests/test_training_gpu.py: |a-b| <= 1e-5 * max(|b|, floor), floor = the
magnitude the value was accumulated from, stated per check."""
import numpy as np
import pytest
import torch

from oracle import ref_torch
from oracle import ref_train as rt
from tests.test_training_gpu import G, close, tr_weights

pytestmark = pytest.mark.gpu


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---- kernels -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Nq,Nk,H,S,scale,masked", [
    (3, 5, 5, 2, 4, 2.0, False),
    (2, 39, 39, 2, 16, 4.0, False),            # AutoInt configs[2] layer: x sqrt(S)
    (2, 200, 200, 1, 64, 0.125, True),         # SASRec configs[4] block: / sqrt(depth), masked query rows
    (1, 70, 130, 2, 8, 0.35, True),            # Nq != Nk, keys beyond one per lane
    (4, 1, 300, 1, 32, 0.2, False),
])
def test_attention_core_forward_backward(dev, B, Nq, Nk, H, S, scale, masked):
    from recamd import train as tr, train_attn as ta
    rng = np.random.default_rng(B * 1000 + Nq + Nk)
    hs = H * S
    # logits of magnitude L carry an fp32 rounding error of L * 6e-8, which the softmax turns into the same RELATIVE error
    # of every probability: q and k are scaled so that L stays within a few units, and the gradient floors below are the
    # magnitude of the terms dq / dk are summed from (|dS| <= scale * |dP - delta| ~ 4 * scale)
    q, k, v = (rng.normal(size=(B, n, hs)).astype(np.float32) * f for n, f in ((Nq, 0.5), (Nk, 0.5), (Nk, 1.0)))
    do = rng.normal(size=(B, Nq, hs)).astype(np.float32)
    mask = (rng.random((B, Nq)) < 0.7).astype(np.float32) if masked else None
    if masked:
        mask[0, 0] = 0.0
    tape = tr.Tape()
    qv, kv, vv = tr.Var(G(q, dev)), tr.Var(G(k, dev)), tr.Var(G(v, dev))
    out = ta.attn_core_fwd(tape, qv, kv, vv, None if mask is None else G(mask, dev), H, S, scale)
    out.g = G(do, dev)
    tape.backward()
    qt, kt, vt = rt.T(q, True), rt.T(k, True), rt.T(v, True)
    logits = torch.einsum("bihs,bjhs->bhij", qt.view(B, Nq, H, S), kt.view(B, Nk, H, S)) * scale
    if masked:
        m = torch.as_tensor(mask).reshape(B, 1, Nq, 1)
        logits = torch.where(m == 0, torch.full_like(logits, ref_torch.NEG), logits)
    p = torch.softmax(logits, dim=-1)
    ot = torch.einsum("bhij,bjhs->bihs", p, vt.view(B, Nk, H, S)).reshape(B, Nq, hs)
    ot.backward(rt.T(do))
    # outputs are convex combinations of O(1) values; gradients are sums over up to Nq / Nk O(1) terms
    assert close(out.v.cpu().numpy(), ot.detach().numpy(), floor=1.0)
    assert close(qv.g.cpu().numpy(), qt.grad.numpy(), floor=max(1.0, 4 * scale))
    assert close(kv.g.cpu().numpy(), kt.grad.numpy(), floor=max(1.0, 4 * scale))
    assert close(vv.g.cpu().numpy(), vt.grad.numpy(), floor=1.0)
    if masked:   # a masked query row: uniform attention, and no gradient to that q row
        o0 = out.v.cpu().numpy()[0, 0].reshape(H, S)
        assert close(o0, v[0].reshape(Nk, H, S).astype(np.float64).mean(0), floor=1.0)
        assert np.all(qv.g.cpu().numpy()[0, 0] == 0.0)


def test_attention_core_matches_the_fused_forward_kernels(dev):
    """the training-side core agrees with the inference kernels it stands in for (rec_mha_rowmask_f32)"""
    from recamd import ops, train as tr, train_attn as ta
    rng = np.random.default_rng(3)
    B, S_, H, dk = 3, 50, 2, 16
    q, k, v = (G(rng.normal(size=(B, S_, H * dk)), dev) for _ in range(3))
    mask = G((rng.random((B, S_)) < 0.8), dev)
    out = ta.attn_core_fwd(tr.Tape(), tr.Var(q), tr.Var(k), tr.Var(v), mask, H, dk, 1.0 / np.sqrt(dk))
    exp = ops.mha_rowmask(q, k, v, mask, H)
    assert close(out.v.cpu().numpy(), exp.cpu().numpy(), floor=1.0)


@pytest.mark.parametrize("act", ["sigmoid", "relu", None, "prelu", "tanh"])
@pytest.mark.parametrize("mask_kind", ["mixed", "none"])
def test_din_attention_pool_backward(dev, act, mask_kind):
    from ctr.layers.modules import AttentionLayer
    from recamd import train as tr, train_attn as ta
    rng = np.random.default_rng(5)
    B, T, d = 9, 70, 24
    q, k = rng.normal(size=(B, d)).astype(np.float32), rng.normal(size=(B, T, d)).astype(np.float32)
    W, b = (rng.normal(size=(4 * d, 1)) * 0.3).astype(np.float32), np.float32([0.1])
    do = rng.normal(size=(B, d)).astype(np.float32)
    mask = (rng.random((B, T)) < 0.6).astype(np.float32)
    mask[0] = 0.0                                            # a sample whose history is all padding: uniform
    layer = AttentionLayer(1, activation=act)
    layer.build(d)
    w = {"kernel": W, "bias": b}
    if act == "prelu":
        w["alpha"] = np.float32([0.25])
    layer.set_weights(w)
    tape = tr.Tape()
    qv, kv = tr.Var(G(q, dev)), tr.Var(G(k, dev))
    out = ta.din_pool_fwd(tape, layer, "att", qv, kv, G(mask, dev) if mask_kind == "mixed" else None)
    out.g = G(do, dev)
    tape.backward()
    qt, kt, Wt, bt = rt.T(q, True), rt.T(k, True), rt.T(W, True), rt.T(b, True)
    at = rt.T(w["alpha"], True) if act == "prelu" else None
    ot = ref_torch.din_attention(qt, kt, kt, mask if mask_kind == "mixed" else None, Wt, bt, act, at)
    ot.backward(rt.T(do))
    assert close(out.v.cpu().numpy(), ot.detach().numpy(), floor=1.0)
    assert close(qv.g.cpu().numpy(), np.zeros(q.shape) if qt.grad is None else qt.grad.numpy(), floor=1.0)
    assert close(kv.g.cpu().numpy(), kt.grad.numpy(), floor=1.0)
    # parameter gradients are sums over B * T slots of O(1) terms
    # (mask = None replaces every score by the padding constant: autograd then reports no gradient at all for W, b)
    zero = lambda t: np.zeros(t.shape) if t.grad is None else t.grad.numpy()   # noqa: E731
    assert close(tape.grads["att/kernel"].cpu().numpy(), zero(Wt), floor=float(np.sqrt(B * T)))
    assert close(tape.grads["att/bias"].cpu().numpy(), zero(bt), floor=float(np.sqrt(B * T)))
    if act == "prelu":
        assert close(tape.grads["att/alpha"].cpu().numpy(), zero(at), floor=float(np.sqrt(B * T)))


@pytest.mark.parametrize("kind", ["prelu", "dice"])
def test_dense_with_layer_activation_backward(dev, kind):
    from recamd import nn, train as tr
    rng = np.random.default_rng(8)
    M, K, N = 200, 30, 20
    x, W, b = rng.normal(size=(M, K)).astype(np.float32), (rng.normal(size=(K, N)) * 0.3).astype(np.float32), \
        (rng.normal(size=N) * 0.1).astype(np.float32)
    dy = rng.normal(size=(M, N)).astype(np.float32)
    layer = nn.Dense(N, activation=nn.PReLU() if kind == "prelu" else nn.Dice())
    layer.build(K)
    layer.set_weights({"kernel": W, "bias": b})
    if kind == "prelu":
        alpha = (rng.normal(size=N) * 0.3).astype(np.float32)
        layer.set_weights({"prelu/alpha": alpha})
    else:
        alpha = np.float32(0.3)
        layer.activation.bn.build(N)
        layer.set_weights({"dice/alpha": alpha})
    tape = tr.Tape()
    xv = tr.Var(G(x, dev))
    y = tr.dense_fwd(tape, layer, "d", xv)
    y.g = G(dy, dev)
    tape.backward()
    xt, Wt, bt, at = rt.T(x, True), rt.T(W, True), rt.T(b, True), rt.T(alpha, True)
    z = xt @ Wt + bt
    if kind == "prelu":
        yt = torch.where(z >= 0, z, at * z)
    else:
        pz = torch.sigmoid((z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-3))
        yt = at * (1 - pz) * z + pz * z
    yt.backward(rt.T(dy))
    assert close(y.v.cpu().numpy(), yt.detach().numpy(), floor=1.0)
    assert close(xv.g.cpu().numpy(), xt.grad.numpy(), floor=1.0)
    assert close(tape.grads["d/kernel"].cpu().numpy(), Wt.grad.numpy(), floor=float(np.sqrt(M)))
    assert close(tape.grads["d/bias"].cpu().numpy(), bt.grad.numpy(), floor=float(np.sqrt(M)))
    assert close(tape.grads[f"d/{kind}/alpha"].cpu().numpy(), at.grad.numpy(), floor=float(np.sqrt(M)))
    if kind == "dice":   # the Dice BatchNormalization updated its moving statistics from the batch (momentum 0.99)
        zz = z.detach().numpy()
        got = layer.get_weights()
        assert close(got["dice/bn/moving_mean"], 0.01 * zz.mean(0), floor=1e-2)
        assert close(got["dice/bn/moving_variance"], 0.99 + 0.01 * zz.var(0), floor=1.0)


@pytest.mark.parametrize("with_r,with_mask,M,d", [(True, True, 37, 64), (False, False, 5, 7), (True, False, 300, 130)])
def test_layernorm_residual_backward(dev, with_r, with_mask, M, d):
    from recamd import nn, train as tr, train_attn as ta
    rng = np.random.default_rng(M + d)
    x, r = rng.normal(size=(M, d)).astype(np.float32), rng.normal(size=(M, d)).astype(np.float32)
    g, bta = (1 + 0.2 * rng.normal(size=d)).astype(np.float32), (0.1 * rng.normal(size=d)).astype(np.float32)
    dy = rng.normal(size=(M, d)).astype(np.float32)
    mask = (rng.random(M) < 0.7).astype(np.float32)
    ln = nn.LayerNormalization(epsilon=1e-6)
    ln.build(d)
    ln.set_weights({"gamma": g, "beta": bta})
    tape = tr.Tape()
    xv, rv = tr.Var(G(x, dev)), tr.Var(G(r, dev))
    y = ta.layernorm_fwd(tape, ln, "ln", xv, rv if with_r else None, G(mask, dev) if with_mask else None)
    y.g = G(dy, dev)
    tape.backward()
    xt, rt_, gt, bt = rt.T(x, True), rt.T(r, True), rt.T(g, True), rt.T(bta, True)
    yt = torch.nn.functional.layer_norm(xt + rt_ if with_r else xt, (d,), gt, bt, 1e-6)
    if with_mask:
        yt = yt * torch.as_tensor(mask, dtype=torch.float64)[:, None]
    yt.backward(rt.T(dy))
    # 2e-5 on rows of d = 7: the normalisation divides by a standard deviation of seven values (tests/test_rowwise_gpu.py)
    tol = 2e-5 if d < 16 else 1e-5
    assert close(y.v.cpu().numpy(), yt.detach().numpy(), tol=tol, floor=1.0)
    assert close(xv.g.cpu().numpy(), xt.grad.numpy(), tol=tol, floor=1.0)
    if with_r:
        assert close(rv.g.cpu().numpy(), rt_.grad.numpy(), tol=tol, floor=1.0)
    assert close(tape.grads["ln/gamma"].cpu().numpy(), gt.grad.numpy(), tol=tol, floor=float(np.sqrt(M)))
    assert close(tape.grads["ln/beta"].cpu().numpy(), bt.grad.numpy(), tol=tol, floor=float(np.sqrt(M)))


def test_rank_loss_and_dot_scores_backward(dev):
    from recamd import ops
    from recamd._lib import C
    rng = np.random.default_rng(12)
    B, n, d, V = 33, 20, 64, 50
    seq = rng.normal(size=(B, d)).astype(np.float32)
    table = (rng.normal(size=(V, d)) * 0.3).astype(np.float32)
    ids = rng.integers(-1, V + 1, size=(B, 1 + n)).astype(np.int32)         # duplicates, out-of-range ids
    t_seq, t_tab, t_ids = G(seq, dev), G(table, dev), torch.from_numpy(ids).to(dev)
    logits = ops.gather_dot_scores(t_seq, t_tab, t_ids)
    loss = ops.pairwise_rank_loss(logits)
    dl = torch.empty_like(logits)
    C.pairwise_rank_loss_grad_f32(logits.data_ptr(), logits.stride(0), B, n, 1.0, dl.data_ptr(), dl.stride(0), stream())
    gt = torch.zeros_like(t_tab)
    dseq = torch.full((B, d), 7.0, device=dev)                                 # accumulate = 0 must overwrite
    C.gather_dot_scores_grad_f32(t_seq.data_ptr(), t_tab.data_ptr(), gt.data_ptr(), V, d, t_ids.data_ptr(), t_ids.stride(0),
                                 1 + n, dl.data_ptr(), dl.stride(0), B, dseq.data_ptr(), 0, stream())
    st, tt = rt.T(seq, True), rt.T(table, True)
    rows = ref_torch.embed(tt, ids)
    lg = torch.einsum("bd,bjd->bj", st, rows)
    ls = torch.mean(-torch.log(torch.sigmoid(lg[:, :1])) - torch.log(1 - torch.sigmoid(lg[:, 1:]))) / 2
    ls.backward()
    assert abs(float(loss.item()) - float(ls)) <= 1e-5 * max(1.0, abs(float(ls)))
    assert close(dseq.cpu().numpy(), st.grad.numpy(), floor=1.0 / B)
    assert close(gt.cpu().numpy(), tt.grad.numpy(), floor=1.0 / B)


@pytest.mark.parametrize("rate", [0.0, 0.3, 0.5])
def test_dropout_mask_is_the_documented_function(dev, rate):
    """TensorFlow's dropout stream cannot be reproduced (parity unpinned by construction): the contract is the
    documented counter-based mask, kept values scaled by 1 / (1 - rate), and a backward that reuses the mask."""
    from recamd._lib import C
    n, seed = 100003, (7 << 20) + 5
    x = G(np.random.default_rng(1).normal(size=n), dev)
    y = torch.empty_like(x)
    C.dropout_f32(x.data_ptr(), n, rate, seed, y.data_ptr(), stream())
    keep = rt.dropout_mask(n, rate, seed)
    exp = np.where(keep, x.cpu().numpy() * np.float32(1.0 / (1.0 - rate)), np.float32(0))
    assert np.array_equal(y.cpu().numpy(), exp.astype(np.float32))
    if rate:
        assert abs(keep.mean() - (1 - rate)) < 0.01
    C.dropout_f32(y.data_ptr(), n, rate, seed, y.data_ptr(), stream())          # in place
    assert np.array_equal(y.cpu().numpy(), np.where(keep, exp * np.float32(1.0 / (1.0 - rate)), 0).astype(np.float32))


# ---- one / two optimiser steps of the models ------------------------------------------------------------------------
def check_weights(m, W, lr):
    got = tr_weights(m)
    assert set(got) == set(W)
    for k, e in W.items():
        d = np.abs(got[k] - e)
        band = 1e-5 * np.maximum(np.abs(e), 1e-2) + 1e-3 * lr      # as tests/test_training_gpu.py: Adam amplifies |g| ~ eps
        assert (d > band).mean() <= 2e-3 and d.max() <= 1e-2 * lr * 2, (k, float(d.max()), int((d > band).sum()), d.size)


def run_steps(m, okind, kw, inputs, y, steps=2, lr=1e-2):
    from recamd import train as tr
    l2 = tr.default_l2(m)
    W = {k: v.astype(np.float64) for k, v in tr_weights(m).items()}
    opt, state, oo = tr.Adam(m, lr, l2=l2), tr.TrainState(m), rt.AdamOracle(lr=lr)
    for _ in range(steps):
        p, loss = tr.train_step(m, opt, state, inputs, y)
        ep, eloss, _ = rt.train_step(okind, W, oo, inputs, y, l2, **kw)
        assert close(p.cpu().numpy().reshape(-1), ep.reshape(-1), floor=1.0)
        assert abs(float(loss.item()) - eloss) <= 1e-5 * max(1.0, abs(eloss))
    check_weights(m, W, lr)
    return l2


def test_fm_training_step(dev):
    """classic FM (BASELINE configs[0], trained by src/ctr/fm/train.py:43-67): w0, w and V after two Adam steps"""
    from ctr.fm.model import FM
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(31)
    B, nd, vocab = 64, 5, [7, 11, 4, 30]
    sparse = [{'feat': f'C{i}', 'feat_num': v, 'embed_dim': 8} for i, v in enumerate(vocab)]
    m = FM([[{'feat': f'I{i}'} for i in range(nd)], sparse], k=6, w_reg=1e-3, v_reg=1e-3)
    randomize(m, rng, 0.3)
    dense = rng.random((B, nd)).astype(np.float32)
    ids = np.stack([rng.integers(-1, v + 1, size=B) for v in vocab], axis=1).astype(np.int32)    # incl. out-of-range ids
    y = (rng.random(B) < 0.4).astype(np.float32)
    l2 = run_steps(m, "fm", {"vocab": vocab}, [dense, ids], y)
    assert l2 == {"w": 1e-3, "V": 1e-3}


@pytest.mark.parametrize("use_res,layers,H", [(True, 2, 2), (False, 1, 1)])
def test_autoint_training_step(dev, use_res, layers, H):
    from ctr.autoint.model import AutoInt
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(32)
    B, F, V, D, nd, S = 48, 6, 19, 8, 3, 8
    sparse = [{'feat': f'C{i}', 'feat_num': V + i, 'embed_dim': D} for i in range(F)]
    m = AutoInt([[{'feat': f'I{i}'} for i in range(nd)], sparse], att_hidden_units=S, head_num=H, att_layer_num=layers,
                use_res=use_res, embed_reg=1e-4)
    dense = rng.random((B, nd)).astype(np.float32)
    ids = np.stack([rng.integers(0, V + i, size=B) for i in range(F)], axis=1).astype(np.int32)
    y = (rng.random(B) < 0.4).astype(np.float32)
    m.fused = False
    m([dense, ids])                            # builds the lazily created layers
    m.fused = True
    randomize(m, rng, 0.3)
    l2 = run_steps(m, "autoint", {"H": H, "S": S, "use_res": use_res}, [dense, ids], y)
    assert l2["attention_0/Wq"] == 1e-4 and l2["/embeddings"] == 1e-4


@pytest.mark.parametrize("ffn_act", ["prelu", "dice"])
def test_din_training_step(dev, ffn_act):
    """the model of src/ctr/din/train.py (canonical pooling form), dropout 0: BatchNormalization on batch statistics,
    Dense(PReLU() | Dice()), AttentionLayer pooling, shared item / behaviour tables (their gradients add up)"""
    from ctr.din.model import DIN
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(33)
    B, maxlen, D = 40, 6, 8
    ukeys, ikeys = ['user_sparse_0', 'user_sparse_1'], ['item_sparse_0', 'item_sparse_1', 'item_sparse_2']
    sfd = {k: (10, D) for k in ukeys + ikeys}
    idx = [{k: i for i, k in enumerate(ukeys)}, {k: i for i, k in enumerate(ikeys)},
           {f'item_sparse_{ml}_{i}': ml * 3 + i for ml in range(maxlen) for i in range(3)}]
    m = DIN(sfd, idx, att_hidden_units=8, ffn_hidden_units=(16, 8), att_activation='sigmoid', ffn_activation=ffn_act,
            maxlen=maxlen, dnn_dropout=0.0, embed_reg=1e-4)
    beh = rng.integers(1, 10, size=(B, maxlen, 3)).astype(np.float32)
    for b in range(B):                                     # pre-padded histories of varying length
        beh[b, :rng.integers(0, maxlen)] = 0
    inputs = [rng.random((B, 5)).astype(np.float32), rng.integers(0, 10, size=(B, 2)).astype(np.float32),
              rng.random((B, 5)).astype(np.float32), rng.integers(0, 10, size=(B, 3)).astype(np.float32), beh.reshape(B, -1)]
    y = (rng.random(B) < 0.4).astype(np.float32)
    m(inputs)
    if ffn_act == "dice":
        for d_ in m.ffn:
            if not d_.activation.bn.built:
                d_.activation.bn.build(d_.units)
    randomize(m, rng, 0.3)
    for k, v in m.get_weights().items():
        if k.endswith("gamma"):
            m.set_weights({k: (1 + 0.1 * rng.normal(size=v.shape)).astype(np.float32)})
    kw = dict(user_keys=ukeys, item_keys=ikeys, maxlen=maxlen, att_activation='sigmoid', ffn_activation=ffn_act, n_ffn=2)
    run_steps(m, "din", kw, inputs, y)


def test_din_training_with_dropout_is_reproducible(dev):
    """dnn_dropout = 0.5 as in src/ctr/din/train.py:29: the masks are a function of (optimiser step, tape position), so two
    runs from the same weights agree (to the rounding of the embedding-gradient atomics, whose order is not fixed), and
    they differ from the dropout-free run"""
    from ctr.din.model import DIN
    from recamd import train as tr
    rng = np.random.default_rng(34)
    B, maxlen, D = 32, 4, 8
    ikeys = ['item_sparse_0']
    sfd = {'user_sparse_0': (10, D), 'item_sparse_0': (10, D)}
    idx = [{'user_sparse_0': 0}, {'item_sparse_0': 0}, {f'item_sparse_{ml}_0': ml for ml in range(maxlen)}]
    inputs = [rng.random((B, 2)).astype(np.float32), rng.integers(0, 10, size=(B, 1)).astype(np.float32),
              rng.random((B, 2)).astype(np.float32), rng.integers(0, 10, size=(B, 1)).astype(np.float32),
              rng.integers(0, 10, size=(B, maxlen)).astype(np.float32)]
    y = (rng.random(B) < 0.5).astype(np.float32)
    outs = []
    for rate in (0.5, 0.5, 0.0):
        m = DIN(sfd, idx, ffn_hidden_units=(16, 8), att_activation='sigmoid', ffn_activation='prelu', maxlen=maxlen,
                dnn_dropout=rate)
        m(inputs)
        if outs:
            m.set_weights(outs[0][0])
        w_init = m.get_weights()
        tr.Trainer(m).compile(learning_rate=1e-2).fit(inputs, y, batch_size=16, epochs=1, shuffle=False)
        outs.append((w_init, m.get_weights()))
    a, b, c = outs[0][1], outs[1][1], outs[2][1]
    assert all(np.abs(a[k] - b[k]).max() <= 1e-5 for k in a), max((float(np.abs(a[k] - b[k]).max()), k) for k in a)
    assert max(float(np.abs(a[k] - c[k]).max()) for k in a if k.startswith("ffn_")) > 1e-3


@pytest.mark.parametrize("blocks,H,S", [(1, 1, 12), (2, 2, 20)])
def test_sasrec_training_step(dev, blocks, H, S):
    """SASRec's add_loss (src/match/sasrec/model.py:93-95) minimised by Adam: every weight of every block after two steps"""
    from match.sasrec.model import SASRec
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(35 + blocks)
    B, d, n_neg, V = 24, 16, 7, 40
    cols = [{'feat': k, 'feat_num': V, 'feat_len': n, 'embed_dim': d} for k, n in
            (('seq_item', S), ('pos_item', 1), ('neg_item', n_neg))]
    m = SASRec(cols, [], blocks=blocks, num_heads=H, att_hidden_unit=d, ffn_hidden_unit=24, seq_len=S, neg_len=n_neg,
               embed_reg=1e-4, last_row_only=False)
    seq = rng.integers(1, V, size=(B, S)).astype(np.int32)
    for b in range(B):
        seq[b, :rng.integers(0, S)] = 0                  # pre-padding; some samples end up all-pad except the last slot
    seq[0, :] = 0                                        # a fully padded sequence: logits 0, loss ln 2 for that sample
    pos, neg = rng.integers(1, V, size=(B, 1)).astype(np.int32), rng.integers(1, V, size=(B, n_neg)).astype(np.int32)
    m([seq, pos, neg])
    randomize(m, rng, 0.3)
    for k, v in m.get_weights().items():
        if k.endswith("gamma"):
            m.set_weights({k: (1 + 0.1 * rng.normal(size=v.shape)).astype(np.float32)})
    run_steps(m, "sasrec", {"n_blocks": blocks, "H": H}, [seq, pos, neg], None)


def test_sasrec_row_sharded_training_matches_unsharded(dev):
    """BASELINE configs[4], trained: G = 2 ranks (simulated in one process, tests/shard_oracle.py::PeersShardedTables),
    tables row-sharded, every rank its own batch.  The lookups travel through the sharded exchange, their gradients
    return to the owners (ShardedTables.backward), dense gradients are all-reduced, tables are not.  After two Adam
    steps every rank's dense weights equal the unsharded model's trained on the concatenated batch (itself pinned to
    the fp64 autograd oracle by test_sasrec_training_step), and its table shards equal rows r::G of the full tables."""
    from match.sasrec.model import SASRec
    from recamd import train as tr
    from recamd.dist import shard_table
    from tests.shard_oracle import PeersShardedTables
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(77)
    G, B, S, d, n_neg, V = 2, 16, 6, 16, 5, 30
    cols = [{'feat': k, 'feat_num': V, 'feat_len': n, 'embed_dim': d} for k, n in
            (('seq_item', S), ('pos_item', 1), ('neg_item', n_neg))]
    kw = dict(blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=24, seq_len=S, neg_len=n_neg, embed_reg=1e-4,
              last_row_only=False)

    def batch(r):
        g = np.random.default_rng(500 + r)
        seq = g.integers(1, V, size=(B, S)).astype(np.int32)
        for b in range(B):
            seq[b, :g.integers(0, S)] = 0
        return [seq, g.integers(1, V, size=(B, 1)).astype(np.int32), g.integers(1, V, size=(B, n_neg)).astype(np.int32)]
    batches = [batch(r) for r in range(G)]
    glob = [np.concatenate([batches[r][i] for r in range(G)]) for i in range(3)]
    full = SASRec(cols, [], **kw)
    full(glob)
    randomize(full, rng, 0.3)
    w0 = full.get_weights()
    names = [f"user_embed_{k}/embeddings" for k in ("seq_item", "pos_item", "neg_item")]
    ranks = [SASRec(cols, [], sharded=(r, G), shard_factory=PeersShardedTables, **kw) for r in range(G)]
    for m in ranks:
        m._sharded.link_peers([x._sharded for x in ranks])
    for r, m in enumerate(ranks):
        m(batches[r])                                        # builds the lazily created layers (and sizes the receive slots)
        ws = {k: v for k, v in w0.items() if k not in names}
        for n in names:
            ws[n] = shard_table(torch.from_numpy(w0[n]), r, G).numpy()
        m.set_weights(ws)
    lr = 1e-2
    opt_f, st_f = tr.Adam(full, lr, l2=tr.default_l2(full)), tr.TrainState(full)
    opts = [tr.Adam(m, lr, l2=tr.default_l2(m)) for m in ranks]
    states = [tr.TrainState(m) for m in ranks]
    for r, m in enumerate(ranks):                            # the owners' gradient arenas exist before anyone sends into them
        m._sharded.peer_grad_arena = states[r].sharded_grad(m._sharded, names)
    for _step in range(2):
        tr.train_step(full, opt_f, st_f, glob, None)
        outs = [tr.compute_gradients(m, st, batches[r], None, 1.0 / G) for r, (m, st) in enumerate(zip(ranks, states))]
        for k in sorted(outs[0][2]):                         # the all-reduce of the dense gradients, by hand
            tot = outs[0][2][k] + outs[1][2][k]
            for o in outs:
                o[2][k] = tot.clone()
        for r in range(G):
            assert states[r].owner_summed == set(names)      # the tables' gradients arrived summed: no all-reduce for them
            opts[r].apply(outs[r][2], states[r])
    wf = {k: v.detach().cpu().numpy() for k, v in tr.named_weights(full).items()}
    for r, m in enumerate(ranks):
        wr = {k: v.detach().cpu().numpy() for k, v in tr.named_weights(m).items()}
        for k, v in wr.items():
            e = wf[k][r::G] if k in names else wf[k]
            assert v.shape == e.shape, k
            dd = np.abs(v - e)       # as check_weights: Adam amplifies rounding where |g| ~ eps
            band = 1e-5 * np.maximum(np.abs(e), 1e-2) + 1e-3 * lr
            assert (dd > band).mean() <= 2e-3 and dd.max() <= 1e-2 * lr * 2, (k, r, float(dd.max()), int((dd > band).sum()))
        assert all(not s.busy for s in m._sharded._slots)    # every kept plan released its receive slot


def test_sasrec_fit_without_labels(dev):
    """Trainer.fit(x, None): the loss is the model's add_loss; it goes down on a learnable synthetic task and
    evaluate() reports the same quantity with the inference path (one-launch kernel where it applies)"""
    from match.sasrec.model import SASRec
    from recamd import train as tr
    rng = np.random.default_rng(40)
    B, S, d, n_neg, V = 256, 10, 64, 5, 30
    cols = [{'feat': k, 'feat_num': V, 'feat_len': n, 'embed_dim': d} for k, n in
            (('seq_item', S), ('pos_item', 1), ('neg_item', n_neg))]
    m = SASRec(cols, [], att_hidden_unit=d, ffn_hidden_unit=64, seq_len=S, neg_len=n_neg)
    seq = rng.integers(1, V, size=(B, S)).astype(np.int32)
    pos = seq[:, -1:].copy()                                   # the positive item is the last item of the sequence
    neg = ((pos + rng.integers(1, V - 1, size=(B, n_neg)) - 1) % (V - 1) + 1).astype(np.int32)
    trainer = tr.Trainer(m).compile(learning_rate=5e-3)
    before = trainer.evaluate([seq, pos, neg])[0]
    hist = trainer.fit([seq, pos, neg], None, batch_size=64, epochs=4, validation_split=0.25, seed=1)
    after = trainer.evaluate([seq, pos, neg])[0]
    assert set(hist) == {"loss", "val_loss"} and len(hist["loss"]) == 4
    assert hist["loss"][-1] < hist["loss"][0] and after < before


# ---- zoo models: Wide&Deep (src/ctr/wide_deep/train.py), Deep&Crossing, NCF (src/match/ncf/train.py) ----
def test_wide_deep_and_deep_crossing_training_steps(dev):
    from ctr.deep_crossing.model import Deep_Crossing
    from ctr.wide_deep.model import WideDeep
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(50)
    B, F, V, D, nd = 64, 4, 17, 8, 5
    sparse = [{'feat': f'C{i}', 'feat_num': V + i, 'embed_dim': D} for i in range(F)]
    dense = rng.random((B, nd)).astype(np.float32)
    ids = np.stack([rng.integers(0, V + i, size=B) for i in range(F)], axis=1).astype(np.int32)
    y = (rng.random(B) < 0.4).astype(np.float32)
    m = WideDeep([[{'feat': f'I{i}'} for i in range(nd)], sparse], hidden_units=[24, 12], embed_reg=1e-4)
    m([dense, ids])
    randomize(m, rng, 0.3)
    run_steps(m, "wide_deep", {}, [dense, ids], y)
    m = Deep_Crossing(sparse, hidden_units=[16, 8], embed_reg=1e-4)
    m(ids)
    randomize(m, rng, 0.3)
    run_steps(m, "deep_crossing", {}, ids, y)


def test_ncf_training_step(dev):
    """NCF's add_loss minimised by Adam (dropout 0 for the parity check; src/match/ncf/train.py uses the default 0.2)"""
    from match.ncf.model import NCF
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(51)
    B, V, dim, n_neg = 32, 25, 8, 6
    m = NCF({'feat': 'user_id', 'feat_num': V, 'embed_dim': dim}, {'feat': 'item_id', 'feat_num': V, 'embed_dim': dim},
            hidden_units=[16, 8], dropout=0.0, neg_num=n_neg, embed_reg=1e-4)
    user, pos = rng.integers(0, V, size=(B, 1)).astype(np.int32), rng.integers(0, V, size=(B, 1)).astype(np.int32)
    neg = rng.integers(0, V, size=(B, n_neg)).astype(np.int32)
    m([user, pos, neg])
    randomize(m, rng, 0.3)
    run_steps(m, "ncf", {}, [user, pos, neg], None)


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (2048, 16, 32), (5000, 64, 64), (3001, 128, 128), (40000, 39, 7), (2500, 1248, 1),
                                   (2100, 5, 200), (3000, 64, 128)])
def test_weight_grad_small_kernel(dev, M, K, N):
    """rec_wgrad_small_f32 (rows split over workgroups, fixed-order fp64 finish) against x^T dy in fp64; and the
    dispatcher's two paths agree"""
    from recamd import train as tr
    from recamd._lib import C
    rng = np.random.default_rng(M + K)
    x, dy = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(M, N)).astype(np.float32)
    tx, tdy = G(x, dev), G(dy, dev)
    out = torch.empty((K, N), device=dev)
    ws = torch.empty(int(C.wgrad_small_workspace_bytes(M, K, N)), dtype=torch.uint8, device=dev)
    C.wgrad_small_f32(tx.data_ptr(), tx.stride(0), tdy.data_ptr(), tdy.stride(0), M, K, N, out.data_ptr(), ws.data_ptr(), stream())
    exp = x.astype(np.float64).T @ dy.astype(np.float64)
    assert close(out.cpu().numpy(), exp, floor=float(np.sqrt(M)))          # sums of M products of O(1) operands
    assert close(tr.weight_grad(tx, tdy).cpu().numpy(), exp, floor=float(np.sqrt(M)))
    again = torch.empty_like(out)
    C.wgrad_small_f32(tx.data_ptr(), tx.stride(0), tdy.data_ptr(), tdy.stride(0), M, K, N, again.data_ptr(), ws.data_ptr(), stream())
    assert torch.equal(out, again)                                           # deterministic


def test_esmm_training_step_two_targets_shared_towers(dev):
    """ESMM as src/ctr/esmm/train.py trains it: loss = BCE(pCTR, y_ctr) + BCE(pCTR * pCVR, y_cvr), Adam(lr, decay); the
    user / item DNNs and the embeddings are shared by the two towers (gradients add, their BatchNormalization layers see
    two batches per step)"""
    from ctr.esmm.model import ESMM
    from recamd import train as tr
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(60)
    user_feats = {f'u{i}': (int(rng.integers(5, 30)), 4) for i in range(3)}
    item_feats = {f'i{i}': (int(rng.integers(5, 30)), 6) for i in range(2)}
    user_dict = {k: (i,) for i, k in enumerate(user_feats)}
    item_dict = {k: (i,) for i, k in enumerate(item_feats)}
    m = ESMM({**user_feats, **item_feats}, [user_dict, item_dict], hidden_units=[24, 12], embed_reg=1e-4)
    B = 80

    def tower_inputs():
        return [rng.random((B, 5)).astype(np.float32),
                np.stack([rng.integers(0, user_feats[k][0], size=B) for k in user_feats], axis=1).astype(np.float32),
                rng.random((B, 4)).astype(np.float32),
                np.stack([rng.integers(0, item_feats[k][0], size=B) for k in item_feats], axis=1).astype(np.float32)]
    x = tower_inputs() + tower_inputs()
    y = [(rng.random(B) < 0.4).astype(np.float32), (rng.random(B) < 0.2).astype(np.float32)]
    m(x)
    randomize(m, rng, 0.3)
    for k, v in m.get_weights().items():
        if k.endswith("gamma"):
            m.set_weights({k: (1 + 0.1 * rng.normal(size=v.shape)).astype(np.float32)})
    lr, decay = 3e-3, 1e-2
    l2 = tr.default_l2(m)
    W = {k: v.astype(np.float64) for k, v in tr_weights(m).items()}
    opt, state, oo = tr.Adam(m, lr, l2=l2, decay=decay), tr.TrainState(m), rt.AdamOracle(lr=lr, decay=decay)
    kw = dict(user_keys=list(user_feats), user_cols=[0, 1, 2], item_keys=list(item_feats), item_cols=[0, 1])
    for _ in range(3):
        p, loss = tr.train_step(m, opt, state, x, y)
        ep, eloss, _ = rt.train_step("esmm", W, oo, x, y, l2, **kw)
        assert close(p[0].cpu().numpy().reshape(-1), ep[0], floor=1.0) and close(p[1].cpu().numpy().reshape(-1), ep[1], floor=1.0)
        assert abs(float(loss.item()) - eloss) <= 1e-5 * max(1.0, abs(eloss))
    check_weights(m, W, lr)


# ---- the two-tower match models with their own train scripts: match FM, DSSM (YoutubeDNN's objective is an unseeded sampler) ----
def _two_tower_inputs(rng, B):
    ufeat = [{'feat': 'user_id', 'feat_num': 40, 'feat_len': 1, 'embed_dim': 8}, {'feat': 'age', 'feat_num': 7, 'feat_len': 1, 'embed_dim': 4}]
    ifeat = [{'feat': 'movie_id', 'feat_num': 50, 'feat_len': 1, 'embed_dim': 8}, {'feat': 'genre', 'feat_num': 5, 'feat_len': 1, 'embed_dim': 4}]
    user = {f['feat']: rng.integers(0, f['feat_num'], size=(B, 1)).astype(np.float32) for f in ufeat}
    item = {f['feat']: rng.integers(0, f['feat_num'], size=(B, 1)).astype(np.float32) for f in ifeat}
    return ufeat, ifeat, user, item


def test_match_fm_training_step(dev):
    """src/match/fm/train.py:43-58: binary cross-entropy + Adam on the match FM — w0, w, V and the four tables after two
    steps against the fp64 autograd oracle (l2 on w, V and the tables as the model attaches them, src/match/fm/model.py:43-65)"""
    from match.fm.model import FM
    from recamd import train as tr
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(60)
    B = 96
    ufeat, ifeat, user, item = _two_tower_inputs(rng, B)
    y = (rng.random(B) < 0.5).astype(np.float32)
    m = FM(ufeat, ifeat, k=6, w_reg=1e-3, v_reg=1e-3, l2_reg_embedding=1e-4)
    m([user, item])
    randomize(m, rng, 0.3)
    l2 = run_steps(m, "match_fm", {}, [user, item], y)
    assert l2 == {"/embeddings": 1e-4, "w": 1e-3, "V": 1e-3}
    assert tr.train_forward_of(m).__name__ == "match_fm_train_forward"


def test_dssm_training_step_and_fit(dev):
    """src/match/dssm/dssm_train.py:43-60: loss = mean(y_pred), the ONE sigmoid(cosine) value the model emits per batch
    (dropout 0 for the parity check; the script uses 0.5).  Two Adam steps against the oracle, then fit() with the
    script's arguments on dict inputs: labels are accepted and ignored, the loss falls (it is being minimised)."""
    from match.dssm.model import Dssm
    from recamd import train as tr
    from tests.test_models_gpu import randomize
    rng = np.random.default_rng(61)
    B = 80
    ufeat, ifeat, user, item = _two_tower_inputs(rng, B)
    m = Dssm(ufeat, ifeat, user_dnn_hidden_units=(16, 8), item_dnn_hidden_units=(16, 8), l2_reg_embedding=1e-4)
    m([user, item])
    randomize(m, rng, 0.3)
    run_steps(m, "dssm", {}, [user, item], None)
    t = tr.Trainer(m).compile(learning_rate=1e-2)
    hist = t.fit([user, item], (rng.random(B) < 0.5).astype(np.float32), batch_size=32, epochs=4, validation_split=0.1)
    assert set(hist) == {"loss", "val_loss"} and hist["loss"][-1] < hist["loss"][0]
    assert np.isfinite(t.evaluate([user, item])[0])


def test_youtube_dnn_has_no_trainer(dev):
    from match.youtube_dnn.model import YoutubeDNN
    from recamd import train as tr
    ufeat, ifeat, _, _ = _two_tower_inputs(np.random.default_rng(0), 4)
    with pytest.raises(NotImplementedError, match="unseeded"):
        tr.Trainer(YoutubeDNN(ufeat, ifeat))
