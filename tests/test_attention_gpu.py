"""Parity of the attention-family kernels (K6, K7, K8) vs the fp64 numpy oracle (1e-5 relative)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("B,N,din,H,S", [(1, 3, 4, 1, 4), (64, 39, 16, 2, 16), (7, 39, 32, 2, 16), (5, 26, 8, 1, 64), (3, 100, 192, 1, 64)])
@pytest.mark.parametrize("use_res,act", [(False, "relu"), (True, "relu"), (True, None)])
def test_mha_ctr(dev, B, N, din, H, S, use_res, act):
    from recamd import ops
    rng = np.random.default_rng(B * N + din + H)
    x = (rng.normal(size=(B, N, din)) * 0.5).astype(np.float32)
    Ws = [(rng.normal(size=(din, H * S)) / np.sqrt(din)).astype(np.float32) for _ in range(4)]
    tx = T(x, dev)  # one tensor as q, k and v (the layer's single-input form, modules.py:306-309)
    out = ops.mha_ctr(tx, tx, tx, T(Ws[0], dev), T(Ws[1], dev), T(Ws[2], dev),
                      T(Ws[3], dev) if use_res else None, H, S, act).cpu().numpy()
    exp = ref.mha_ctr(x, x, x, Ws[0], Ws[1], Ws[2], Ws[3] if use_res else None, H, S, act)
    assert close(out, exp)


def test_mha_ctr_distinct_qkv(dev):
    """list-of-3 input form (modules.py:297-300)."""
    from recamd import ops
    rng = np.random.default_rng(77)
    B, N, din, H, S = 9, 11, 8, 2, 4
    xq, xk, xv = [(rng.normal(size=(B, N, din)) * 0.5).astype(np.float32) for _ in range(3)]
    Ws = [(rng.normal(size=(din, H * S)) / np.sqrt(din)).astype(np.float32) for _ in range(4)]
    out = ops.mha_ctr(T(xq, dev), T(xk, dev), T(xv, dev), T(Ws[0], dev), T(Ws[1], dev), T(Ws[2], dev), T(Ws[3], dev),
                      H, S, "relu").cpu().numpy()
    assert close(out, ref.mha_ctr(xq, xk, xv, Ws[0], Ws[1], Ws[2], Ws[3], H, S, "relu"))


@pytest.mark.parametrize("B,N,din,H", [(9, 11, 8, 2), (130, 39, 16, 2), (5, 64, 64, 4), (3, 16, 4, 1), (6, 50, 20, 3)])
@pytest.mark.parametrize("use_res", [False, True])
def test_mha_ctr_mfma_shapes_distinct_qkv(dev, B, N, din, H, use_res):
    """S = 16 runs on the fp32 matrix cores; list-of-3 input form, odd field counts, several heads."""
    from recamd import ops
    rng = np.random.default_rng(B + N + din)
    xq, xk, xv = [(rng.normal(size=(B, N, din)) * 0.5).astype(np.float32) for _ in range(3)]
    Ws = [(rng.normal(size=(din, H * 16)) / np.sqrt(din)).astype(np.float32) for _ in range(4)]
    out = ops.mha_ctr(T(xq, dev), T(xk, dev), T(xv, dev), T(Ws[0], dev), T(Ws[1], dev), T(Ws[2], dev),
                      T(Ws[3], dev) if use_res else None, H, 16, "relu").cpu().numpy()
    assert close(out, ref.mha_ctr(xq, xk, xv, Ws[0], Ws[1], Ws[2], Ws[3] if use_res else None, H, 16, "relu"))


def test_mha_ctr_kat_uniform_and_scale(dev):
    """Wq = Wk = 0 => uniform attention => out = mean_n act(X Wv); scale is x sqrt(S) (S=16 -> x4):
    with q=k=one-hot rows the logits are exactly 4 on the diagonal."""
    from recamd import ops
    rng = np.random.default_rng(5)
    B, N, din, S = 2, 6, 16, 16
    x = rng.normal(size=(B, N, din)).astype(np.float32)
    Wv = rng.normal(size=(din, S)).astype(np.float32)
    Z = np.zeros((din, S), np.float32)
    out = ops.mha_ctr(T(x, dev), T(x, dev), T(x, dev), T(Z, dev), T(Z, dev), T(Wv, dev), None, 1, S, "relu").cpu().numpy()
    exp = np.maximum(x @ Wv, 0).mean(axis=1, keepdims=True).repeat(N, axis=1)
    assert close(out, exp)
    # scale: x = I (N = din = S = 16), Wq = Wk = I, no activation -> logits = 4 * I
    I = np.eye(16, dtype=np.float32)
    x2 = I[None]
    out2 = ops.mha_ctr(T(x2, dev), T(x2, dev), T(x2, dev), T(I, dev), T(I, dev), T(I, dev), None, 1, 16, None).cpu().numpy()
    p = np.exp(4.0) / (np.exp(4.0) + 15.0)
    exp2 = np.full((16, 16), (1 - p) / 15.0)
    np.fill_diagonal(exp2, p)
    assert close(out2[0], exp2)


@pytest.mark.parametrize("B,T_,d", [(1, 1, 4), (64, 100, 192), (9, 10, 16), (33, 37, 64), (5, 100, 256)])
@pytest.mark.parametrize("act", ["sigmoid", "relu", None, "prelu"])
@pytest.mark.parametrize("mask_mode", ["lengths", "none", "allzero"])
def test_din_attention_pool(dev, B, T_, d, act, mask_mode):
    from recamd import ops
    rng = np.random.default_rng(B + T_ + d)
    q = rng.normal(size=(B, d)).astype(np.float32)
    k = rng.normal(size=(B, T_, d)).astype(np.float32)
    W = (rng.normal(size=(4 * d, 1)) / np.sqrt(d)).astype(np.float32)
    b = rng.normal(size=1).astype(np.float32)
    alpha = np.array([0.25], np.float32) if act == "prelu" else None
    if mask_mode == "lengths":  # pre-padded histories: zeros first (create_sasrec_dataset-style padding)
        lens = rng.integers(0, T_ + 1, size=B)
        mask = (np.arange(T_)[None, :] >= (T_ - lens)[:, None]).astype(np.float32)
    elif mask_mode == "allzero":
        mask = np.zeros((B, T_), np.float32)
    else:
        mask = None
    tk = T(k, dev)
    out = ops.din_attention_pool(T(q, dev), tk, tk, None if mask is None else T(mask, dev), T(W, dev), T(b, dev), act,
                                 None if alpha is None else T(alpha, dev)).cpu().numpy()
    exp = ref.din_attention_layer(q, k, k, mask, W, b, act, None if alpha is None else float(alpha[0]))
    assert close(out, exp)
    if mask_mode in ("none", "allzero"):  # KAT: uniform attention => mean of v
        assert close(out, k.astype(np.float64).mean(axis=1))


def test_din_attention_pool_distinct_v(dev):
    from recamd import ops
    rng = np.random.default_rng(8)
    B, T_, d = 12, 20, 32
    q = rng.normal(size=(B, d)).astype(np.float32)
    k = rng.normal(size=(B, T_, d)).astype(np.float32)
    v = rng.normal(size=(B, T_, d)).astype(np.float32)
    W = rng.normal(size=(4 * d, 1)).astype(np.float32) * 0.2
    b = np.zeros(1, np.float32)
    mask = (rng.random((B, T_)) > 0.4).astype(np.float32)
    out = ops.din_attention_pool(T(q, dev), T(k, dev), T(v, dev), T(mask, dev), T(W, dev), T(b, dev), "sigmoid").cpu().numpy()
    assert close(out, ref.din_attention_layer(q, k, v, mask, W, b, "sigmoid"))


@pytest.mark.parametrize("B,Sq,Sk,dm,H", [(1, 1, 1, 8, 1), (8, 200, 200, 64, 1), (3, 10, 10, 64, 2), (5, 300, 300, 64, 4),
                                          (6, 1, 200, 64, 1), (2, 33, 33, 128, 2),
                                          (4, 3, 50, 64, 2), (7, 8, 33, 64, 4), (5, 2, 200, 128, 2),   # decode-style kernel
                                          (3, 64, 64, 64, 2), (2, 100, 40, 64, 1), (2, 17, 257, 32, 1),   # MFMA kernel
                                          (2, 600, 700, 128, 2), (1, 40, 1500, 32, 1),   # longer than the fp32 kernel's LDS
                                          (2, 290, 31, 64, 2), (3, 16, 32, 64, 1)])        # > 8 query tiles, 1 key tile
def test_mha_rowmask(dev, B, Sq, Sk, dm, H):
    from recamd import ops
    rng = np.random.default_rng(B + Sq + dm + H)
    q = rng.normal(size=(B, Sq, dm)).astype(np.float32)
    k = rng.normal(size=(B, Sk, dm)).astype(np.float32)
    v = rng.normal(size=(B, Sk, dm)).astype(np.float32)
    mask = (rng.random((B, Sq)) > 0.3).astype(np.float32)
    out = ops.mha_rowmask(T(q, dev), T(k, dev), T(v, dev), T(mask, dev), H).cpu().numpy()

    def split(t):
        return np.transpose(t.reshape(B, t.shape[1], H, dm // H), (0, 2, 1, 3)).astype(np.float64)

    m4 = np.tile(mask[:, None, :, None], (1, H, 1, 1))
    att = ref.sdpa_match(split(q), split(k), split(v), m4)
    exp = np.transpose(att, (0, 2, 1, 3)).reshape(B, Sq, dm)
    assert close(out, exp)
    # KAT: padded query rows attend uniformly over ALL keys (keys are never masked, not causal)
    padded = mask == 0
    if padded.any():
        mean_v = v.astype(np.float64).mean(axis=1)
        for b in range(B):
            for i in np.nonzero(padded[b])[0]:
                assert close(out[b, i], mean_v[b])


@pytest.mark.parametrize("B,T_,n_tab,Dt", [(33, 100, 3, 64), (7, 10, 1, 16), (20, 37, 4, 32), (5, 3, 2, 128),
                                           (41, 100, 3, 32), (19, 77, 2, 128), (64, 100, 4, 16), (9, 130, 1, 128),
                                           (6, 20, 3, 8)])      # widths 16 / 32 / 64 / 128: the lane-group kernel
@pytest.mark.parametrize("mask_mode", ["ids", "tensor", "none"])
def test_gather_din_attention_pool_fused(dev, B, T_, n_tab, Dt, mask_mode):
    """Fused history gather + pooling == oracle gather followed by the oracle AttentionLayer."""
    from recamd import ops
    rng = np.random.default_rng(B + T_ + Dt)
    V = 50
    d = n_tab * Dt
    tables = [rng.normal(size=(V, Dt)).astype(np.float32) for _ in range(n_tab)]
    lens = rng.integers(0, T_ + 1, size=B)
    ids = rng.integers(1, V, size=(B, T_, n_tab)).astype(np.int32)
    ids[np.arange(T_)[None, :] < (T_ - lens)[:, None]] = 0
    if B > 10:
        ids[1, T_ - 1, n_tab - 1] = V + 3  # OOB id in a real slot -> zero sub-row
    q = rng.normal(size=(B, d)).astype(np.float32)
    W = (rng.normal(size=(4 * d, 1)) / np.sqrt(d)).astype(np.float32)
    b = rng.normal(size=1).astype(np.float32)
    g = ops.TableGroup([T(t, dev) for t in tables])
    hist = np.concatenate([ref.embedding_lookup(tables[t].astype(np.float64), ids[:, :, t]) for t in range(n_tab)], axis=-1)
    if mask_mode == "ids":
        mask_np, mask_t, mfi = (ids[:, :, 0] != 0).astype(np.float64), None, True
    elif mask_mode == "tensor":
        mask_np = (rng.random((B, T_)) > 0.5).astype(np.float32)
        mask_t, mfi = T(mask_np, dev), False
    else:
        mask_np, mask_t, mfi = None, None, False
    flag = ops.new_oob_flag(dev)
    out = ops.gather_din_attention_pool(T(q, dev), g, T(ids, dev), mask_t, T(W, dev), T(b, dev), "sigmoid",
                                        mask_from_ids=mfi, oob_flag=flag).cpu().numpy()
    assert close(out, ref.din_attention_layer(q, hist, hist, mask_np, W, b, "sigmoid"))
    assert int(flag.item()) == (1 if B > 10 else 0)


@pytest.mark.parametrize("B,Sq,Sk,d,H", [(3, 200, 200, 64, 1), (5, 1, 200, 64, 1), (2, 40, 70, 64, 2), (4, 6, 33, 32, 1)])
def test_mha_rowmask_strided_views(dev, B, Sq, Sk, d, H):
    """q / k / v as column slices of wider buffers (the fused [Wq | Wk | Wv] projection output) == contiguous."""
    from recamd import ops
    rng = np.random.default_rng(B + Sq)
    kv = T(rng.normal(size=(B, Sk, 2 * d + 8)).astype(np.float32), dev)
    qb = T(rng.normal(size=(B, Sq, d + 4)).astype(np.float32), dev)
    mask = T((rng.random((B, Sq)) > 0.3).astype(np.float32), dev)
    q, k, v = qb[..., 4:], kv[..., :d], kv[..., d + 8:]
    got = ops.mha_rowmask(q, k, v, mask, H)
    ref_out = ops.mha_rowmask(q.contiguous(), k.contiguous(), v.contiguous(), mask, H)
    assert torch.equal(got, ref_out)


@pytest.mark.parametrize("B,Sq,Sk,d,H,fids", [(9, 1, 200, 64, 1, False), (4, 3, 50, 64, 2, False), (6, 8, 33, 32, 1, True),
                                              (3, 2, 17, 16, 1, False)])
def test_gather_mha_fewq_matches_unfused(dev, B, Sq, Sk, d, H, fids):
    """fused table lookup + few-query attention == gather (out-of-range ids -> zero rows) then the plain kernel"""
    from recamd import ops
    rng = np.random.default_rng(B + Sk)
    V = 40
    table = T(rng.normal(size=(V, d)).astype(np.float32), dev)
    ids_np = rng.integers(-1, V + 1, size=(B, Sk))                       # includes -1 (pad) and V (out of range)
    ids = T(ids_np.astype(np.float32 if fids else np.int32), dev)
    q = T(rng.normal(size=(B, Sq, d)).astype(np.float32), dev)
    mask = T((rng.random((B, Sq)) > 0.3).astype(np.float32), dev)
    x = ops.gather_concat(ops.TableGroup([table]), ids.reshape(B * Sk, 1).contiguous()).reshape(B, Sk, d)
    exp = ops.mha_rowmask(q, x, x, mask, H)
    got = ops.gather_mha_fewq(q, table, ids, mask, H)
    assert torch.equal(got, exp)


@pytest.mark.parametrize("N,din,H,L,use_res", [(39, 16, 2, 3, True), (39, 16, 2, 1, False), (16, 32, 1, 2, True),
                                               (64, 16, 2, 4, True), (7, 32, 2, 2, False), (23, 16, 1, 3, True)])
@pytest.mark.parametrize("B", [1, 5, 4100])
def test_mha_ctr_stack_matches_layerwise_oracle(dev, N, din, H, L, use_res, B):
    """L stacked interacting layers in one launch (activations stay in registers) == the oracle applied layer by layer;
    B = 4100 exceeds one sample per wave on a 256-CU part, so waves loop over several samples."""
    from recamd import ops
    rng = np.random.default_rng(N + din + L + B)
    S = 16
    hs = H * S
    x = (rng.normal(size=(B, N, din)) * 0.5).astype(np.float32)
    layers = []
    for l in range(L):
        kin = din if l == 0 else hs
        layers.append([(rng.normal(size=(kin, hs)) / np.sqrt(kin)).astype(np.float32) for _ in range(4 if use_res else 3)] +
                      ([] if use_res else [None]))
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)  # noqa: E731
    out = ops.mha_ctr_stack(t(x), [tuple(t(w) for w in lw) for lw in layers], H, S, "relu")
    assert out is not None
    nb = min(B, 64)                                        # the fp64 oracle on a slice of the batch (first and last)
    sel = np.r_[0:nb // 2 + 1, B - nb // 2:B] if B > nb else np.arange(B)
    exp = x.astype(np.float64)[sel]
    for lw in layers:
        exp = ref.mha_ctr(exp, exp, exp, lw[0], lw[1], lw[2], lw[3], H, S, "relu")
    assert close(out.cpu().numpy()[sel], exp)
