"""Parity of the row-wise interaction kernels (K2, K3, K4, K9, K10) vs the fp64 numpy oracle.
Tolerance: |a-b| <= 1e-5 * max(1,|b|) (fp32 logits, BASELINE north_star)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close, close_scaled

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("B,L1,M", [(1, 5, 3), (7, 13 + 26 * 8, 26 * 8), (300, 41, 28), (5000, 3341, 3328), (20000, 29, 16)])
def test_fm_layer(dev, B, L1, M):
    from recamd import ops
    rng = np.random.default_rng(B + L1)
    first = rng.normal(size=(B, L1)).astype(np.float32) * 0.1
    second = rng.normal(size=(B, M)).astype(np.float32) * 0.1
    w = (rng.normal(size=(L1, 1)) * 0.05).astype(np.float32)
    out = ops.fm_layer(T(first, dev), T(second, dev), T(w, dev)).cpu().numpy()
    f64, s64 = first.astype(np.float64), second.astype(np.float64)
    scale = np.sum(np.abs(f64) * np.abs(w.astype(np.float64)).reshape(1, -1)) + \
        0.5 * (np.abs(s64).sum(1) ** 2 + (s64 ** 2).sum(1))[:, None]           # what each output is accumulated from
    assert close_scaled(out, ref.fm_layer(first, second, w), scale)


def test_fm_layer_kat_and_strided_concat(dev):
    """x=[1,2,3] => second = 0.5(36-14) = 11; the first-order term is ONE batch scalar.
    Also exercises the DeepFM layout: first = buf[:, 3:] (unaligned), second = buf[:, 16:]."""
    from recamd import ops
    second = np.array([[1, 2, 3], [0, 0, 0]], np.float32)
    first = np.array([[1, 1], [2, 2]], np.float32)
    w = np.array([[0.5], [0.25]], np.float32)
    out = ops.fm_layer(T(first, dev), T(second, dev), T(w, dev)).cpu().numpy()
    assert np.allclose(out, [[11 + 2.25], [0 + 2.25]])
    rng = np.random.default_rng(0)
    buf = rng.normal(size=(64, 16 + 40)).astype(np.float32)
    tb = T(buf, dev)
    w2 = rng.normal(size=(53, 1)).astype(np.float32)
    got = ops.fm_layer(tb[:, 3:], tb[:, 16:], T(w2, dev)).cpu().numpy()
    assert close(got, ref.fm_layer(buf[:, 3:], buf[:, 16:], w2))


@pytest.mark.parametrize("B,dim,L", [(1, 8, 1), (33, 26 * 8, 3), (257, 3328, 3), (100, 4096, 2), (19, 50, 4), (10, 6000, 2), (64, 208, 0)])
def test_cross_network(dev, B, dim, L):
    from recamd import ops
    rng = np.random.default_rng(dim + L)
    x = (rng.normal(size=(B, dim)) * 0.1).astype(np.float32)
    W = (rng.normal(size=(L, dim)) * 0.05).astype(np.float32)
    Bv = (rng.normal(size=(L, dim)) * 0.05).astype(np.float32)
    out = ops.cross_network(T(x, dev), T(W, dev), T(Bv, dev)).cpu().numpy()
    x64, scale = x.astype(np.float64), np.abs(x).astype(np.float64)
    xl = x64
    for l in range(L):   # |x0| (|x_l| . |w_l|) + |b_l| + the scale carried by x_l
        scale = np.abs(x64) * (np.abs(xl) @ np.abs(W[l].astype(np.float64)))[:, None] + np.abs(Bv[l]) + scale
        xl = x64 * (xl @ W[l].astype(np.float64))[:, None] + Bv[l] + xl
    assert close_scaled(out, ref.cross_network(x, W, Bv), scale)


def test_cross_kat_identity(dev):
    """w=0, b=0 => identity; 1 layer closed form x0 (x0.w) + b + x0."""
    from recamd import ops
    rng = np.random.default_rng(1)
    x = rng.normal(size=(5, 16)).astype(np.float32)
    z = np.zeros((2, 16), np.float32)
    assert np.array_equal(ops.cross_network(T(x, dev), T(z, dev), T(z, dev)).cpu().numpy(), x)
    w = rng.normal(size=(1, 16)).astype(np.float32)
    b = rng.normal(size=(1, 16)).astype(np.float32)
    exp = x * (x @ w[0])[:, None] + b + x
    assert close(ops.cross_network(T(x, dev), T(w, dev), T(b, dev)).cpu().numpy(), exp)


@pytest.mark.parametrize("B,nd,k", [(256, 13, 10), (3, 0, 4), (65, 13, 1), (100, 70, 8)])
def test_fm_onehot_vs_literal_onehot(dev, B, nd, k):
    """BASELINE config 1 (FM, batch 256, k=10): gather form on the GPU == the reference's literal
    one-hot matmul form (oracle), incl. out-of-range ids (zero one-hot row)."""
    from recamd import ops
    rng = np.random.default_rng(2020 + B)
    vocab = [int(v) for v in rng.integers(2, 300, size=26)]
    L = nd + sum(vocab)
    dense = rng.random((B, nd)).astype(np.float32)
    ids = np.stack([rng.integers(0, v, size=B) for v in vocab], axis=1).astype(np.int32)
    ids[0, 0] = -1
    ids[B - 1, 3] = vocab[3]
    w0 = np.array([0.1], np.float32)
    w = (rng.normal(size=(L, 1)) * 0.05).astype(np.float32)
    V = (rng.normal(size=(k, L)) * 0.05).astype(np.float32)
    out = ops.fm_onehot(T(dense, dev), T(ids, dev), vocab, T(w0, dev), T(w, dev), T(V, dev)).cpu().numpy()
    assert close(out, ref.fm_model_onehot(dense, ids, vocab, w0, w, V))
    assert close(out, ref.fm_model_gather(dense, ids, vocab, w0, w, V))


@pytest.mark.parametrize("rows,d", [(1, 64), (1000, 64), (77, 128), (5, 1000), (33, 7)])
@pytest.mark.parametrize("with_r,with_mask", [(True, True), (False, False)])
def test_layernorm_residual(dev, rows, d, with_r, with_mask):
    from recamd import ops
    rng = np.random.default_rng(rows + d)
    x = rng.normal(size=(rows, d)).astype(np.float32)
    r = rng.normal(size=(rows, d)).astype(np.float32) if with_r else None
    g = rng.normal(size=d).astype(np.float32)
    be = rng.normal(size=d).astype(np.float32)
    m = (rng.random(rows) > 0.3).astype(np.float32) if with_mask else None
    out = ops.layernorm_residual(T(x, dev), None if r is None else T(r, dev), T(g, dev), T(be, dev), 1e-6,
                                 None if m is None else T(m, dev)).cpu().numpy()
    exp = ref.layer_norm((x + (0 if r is None else r)).astype(np.float64), g, be, 1e-6)
    if m is not None:
        exp = exp * m[:, None]
    assert close(out, exp, 2e-5)  # rsqrt(var) amplifies fp32 rounding of the mean for d=7 rows


@pytest.mark.parametrize("B,n,d", [(1, 1, 64), (100, 101, 64), (33, 5, 16), (10, 7, 128), (9, 3, 20)])
def test_gather_dot_scores(dev, B, n, d):
    from recamd import ops
    rng = np.random.default_rng(B * n + d)
    V = 200
    table = rng.normal(size=(V, d)).astype(np.float32)
    ids = rng.integers(0, V, size=(B, n)).astype(np.int32)
    if B > 5:
        ids[2, 0] = V  # OOB -> score 0
    seq = rng.normal(size=(B, d)).astype(np.float32)
    flag = ops.new_oob_flag(dev)
    out = ops.gather_dot_scores(T(seq, dev), T(table, dev), T(ids, dev), oob_flag=flag).cpu().numpy()
    emb = ref.embedding_lookup(table.astype(np.float64), ids)
    exp = np.sum(seq[:, None, :].astype(np.float64) * emb, axis=-1)
    assert close(out, exp)
    assert int(flag.item()) == (1 if B > 5 else 0)


@pytest.mark.parametrize("B,F,D,nd", [(1, 3, 4, 0), (300, 26, 8, 13), (1000, 26, 128, 13), (77, 5, 16, 4), (4100, 26, 64, 13), (33, 2, 256, 1)])
def test_gather_fm_fused(dev, B, F, D, nd):
    """Fused K1+K3 == gather_concat followed by the FM layer (oracle), and the gathered rows are bit-exact."""
    from recamd import ops
    rng = np.random.default_rng(B + F + D)
    vocabs = [int(v) for v in rng.integers(3, 200, size=F)]
    tables = [(rng.normal(size=(v, D)) * 0.1).astype(np.float32) for v in vocabs]
    ids = np.stack([rng.integers(-1, v + 1, size=B) for v in vocabs], axis=1).astype(np.int32)
    dense = rng.random((B, nd)).astype(np.float32)
    w = (rng.normal(size=(nd + F * D, 1)) * 0.05).astype(np.float32)
    pad = (-nd) % 4
    cols = [pad + nd + f * D for f in range(F)]
    g = ops.TableGroup([T(t, dev) for t in tables], out_cols=cols)
    buf = torch.zeros((B, pad + nd + F * D), device=dev)
    buf[:, pad:pad + nd] = T(dense, dev)
    wp = np.zeros(pad + nd + F * D, np.float32)
    wp[pad:] = w[:, 0]
    am = torch.full((B,), -1.0, device=dev)
    out = ops.gather_fm(g, T(ids, dev), buf[:, :pad + nd], T(wp, dev), pad + nd, buf, row_absmax=am).cpu().numpy()
    emb = ref.gather_concat(tables, ids, oob="zero")
    assert np.array_equal(buf[:, pad + nd:].cpu().numpy(), emb)
    first = np.concatenate([dense, emb], axis=1)
    assert close(out, ref.fm_layer(first, emb, w))
    # the same pass delivers every concat row's largest magnitude (the first Dense's row scale): exact
    assert np.array_equal(am.cpu().numpy(), np.abs(buf.cpu().numpy()).max(axis=1))
    assert np.array_equal(out, ops.gather_fm(g, T(ids, dev), buf[:, :pad + nd], T(wp, dev), pad + nd, buf).cpu().numpy())


@pytest.mark.parametrize("B,F,D,nv", [(1, 3, 8, 1), (300, 26, 16, 4), (129, 5, 128, 8), (64, 26, 128, 4)])
def test_gather_dots_and_dcn_logit(dev, B, F, D, nv):
    """fused gather + per-sample dot products, and the closed-form cross/Dense(1) logit built on them, vs the oracle's
    literal CrossNetwork recurrence followed by the final Dense(1) over [cross_x, dnn_x]"""
    from recamd import ops
    rng = np.random.default_rng(B + F + D)
    V = 30
    tables = [rng.normal(size=(V, D)).astype(np.float32) * 0.3 for _ in range(F)]
    ids = rng.integers(-1, V + 1, size=(B, F)).astype(np.int32)
    dim = F * D
    Wd = (rng.normal(size=(nv, dim)) / np.sqrt(dim)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    g = ops.TableGroup([t(x) for x in tables])
    am = torch.full((B,), -1.0, device=dev)
    x, dots = ops.gather_dots(g, t(ids), t(Wd), row_absmax=am)
    ex = ref.gather_concat([a.astype(np.float64) for a in tables], ids)
    assert np.array_equal(x.cpu().numpy(), ex.astype(np.float32))
    assert np.array_equal(am.cpu().numpy(), np.abs(ex.astype(np.float32)).max(axis=1))     # the rows' largest magnitudes
    assert close(dots.cpu().numpy(), ex @ Wd.astype(np.float64).T)
    # DCN logit: L = nv - 1 cross layers, last vector = the cross half of the final Dense(1)
    L = nv - 1
    cb = (rng.normal(size=(L, dim)) * 0.1).astype(np.float32)
    H = 7
    dnn_x = rng.normal(size=(B, H)).astype(np.float32)
    w_d = rng.normal(size=(H, 1)).astype(np.float32)
    bias = np.float32(0.3)
    cross = ref.cross_network(ex, Wd[:L], cb) if L > 0 else ex
    logit = cross @ Wd[L].astype(np.float64) + (dnn_x.astype(np.float64) @ w_d.astype(np.float64))[:, 0] + bias
    csum = np.cumsum(cb.astype(np.float64), axis=0) if L > 0 else np.zeros((0, dim))
    G = np.zeros(L)
    for l in range(1, L):
        G[l] = csum[l - 1] @ Wd[l].astype(np.float64)
    c = (csum[-1] @ Wd[L].astype(np.float64) if L > 0 else 0.0) + bias
    extra = t(dnn_x) @ t(w_d)
    got = ops.dcn_logit(dots, t(G.astype(np.float32)), float(c), extra).cpu().numpy()
    assert close(got[:, 0], ref.sigmoid(logit))
