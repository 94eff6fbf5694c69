"""Non-finite and denormal inputs through every bf16x3 entry point (csrc/bf16x3.h: x = h + m + l, three bf16 terms, six
bf16 MFMAs per fp32-equivalent product) — Dense (large-layer kernel), ctr MultiHeadAttention, match MultiHeadAttention,
inner-product top-k.  The split cannot keep fp32's inf arithmetic (inf - inf in the split, inf * 0 against an exactly
representable weight), so the CONTRACT these tests pin is:
  * a +-inf / NaN input never turns into a wrong FINITE output: every output that depends on it is non-finite;
  * outputs that do not depend on it are bit-identical to the run without it (no contamination across rows / samples);
  * fp32 denormals are handled to within their own magnitude (low parts flush: |err| <= K * 2^-126), never as NaN;
  * the fp32 kernels (forced "dense" "f", "mha" "f") and the fused gather + pairwise-dot (fp32 MFMA, an fmaf chain)
    keep exact IEEE behaviour — tests/test_pairwise_dot_gpu.py::test_ring_kernel_order_kat_and_nonfinite."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan])
def test_dense_bf16x3_nonfinite_rows(dev, force, bad):
    from recamd import ops
    rng = np.random.default_rng(1)
    M, K, N = 2048, 256, 128                         # the prepared-weights bf16x3 path
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = (rng.normal(size=(K, N)) / 16).astype(np.float32)
    W[7, :] = 0.5                                     # exactly representable in bf16: its m / l terms are 0
    force("dense", "b")
    clean = ops.dense(G(x, dev), G(W, dev)).cpu().numpy()
    xb = x.copy()
    xb[100, 7] = bad
    xb[1500, 200] = bad
    got = ops.dense(G(xb, dev), G(W, dev)).cpu().numpy()
    assert not np.isfinite(got[100]).any() and not np.isfinite(got[1500]).any()
    rest = np.ones(M, bool)
    rest[[100, 1500]] = False
    assert np.array_equal(got[rest].view(np.uint32), clean[rest].view(np.uint32))
    force("dense", "f")         # the fp32-MFMA kernel: IEEE fp32 semantics
    gf = ops.dense(G(xb, dev), G(W, dev)).cpu().numpy()
    if np.isnan(bad):
        assert np.isnan(gf[100]).all()
    else:
        assert np.all(gf[100] == bad) and np.all(np.isinf(gf[1500]))


def test_dense_bf16x3_denormals(dev, force):
    from recamd import ops
    rng = np.random.default_rng(2)
    M, K, N = 1024, 128, 64
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = (rng.normal(size=(K, N)) / 8).astype(np.float32)
    x[:, :16] = (rng.random((M, 16)) * 1e-39).astype(np.float32)       # fp32 denormals
    W[:8, :] = (rng.random((8, N)) * 1e-40).astype(np.float32)
    force("dense", "b")
    got = ops.dense(G(x, dev), G(W, dev)).cpu().numpy()
    exp = x.astype(np.float64) @ W.astype(np.float64)
    assert np.isfinite(got).all()
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64)
    assert np.all(np.abs(got - exp) <= 1e-5 * np.maximum(np.abs(exp), 1e-3) + 2.5e-7 * scale)


@pytest.mark.parametrize("bad", [np.inf, np.nan])
def test_mha_ctr_and_rowmask_nonfinite_samples(dev, bad):
    """one poisoned sample: its outputs are non-finite, every other sample is bit-identical"""
    from recamd import ops
    rng = np.random.default_rng(3)
    B, Nf, d, H, S = 64, 39, 16, 2, 16
    x = (rng.normal(size=(B, Nf, d)) * 0.3).astype(np.float32)
    Ws = [G(rng.normal(size=(d, H * S)) * 0.2, dev) for _ in range(4)]
    tc = G(x, dev)                                    # q = k = v = ONE tensor: the bf16x3 self-attention kernel
    clean = ops.mha_ctr(tc, tc, tc, *Ws, H, S, "relu").cpu().numpy()
    xb = x.copy()
    xb[5, 3, 2] = bad
    t = G(xb, dev)
    got = ops.mha_ctr(t, t, t, *Ws, H, S, "relu").cpu().numpy()
    assert not np.isfinite(got[5]).all()
    keep = np.arange(B) != 5
    assert np.array_equal(got[keep].view(np.uint32), clean[keep].view(np.uint32))
    # match MHA (row mask), S = 200, d = 64
    B2, S2, dm = 8, 200, 64
    q = (rng.normal(size=(B2, S2, dm)) * 0.3).astype(np.float32)
    mask = np.ones((B2, S2), np.float32)
    tq = G(q, dev)
    clean2 = ops.mha_rowmask(tq, tq, tq, G(mask, dev), 1).cpu().numpy()
    qb = q.copy()
    qb[2, 17, 5] = bad
    t2 = G(qb, dev)
    got2 = ops.mha_rowmask(t2, t2, t2, G(mask, dev), 1).cpu().numpy()
    assert not np.isfinite(got2[2]).all()
    keep2 = np.arange(B2) != 2
    assert np.array_equal(got2[keep2].view(np.uint32), clean2[keep2].view(np.uint32))


def test_topk_nonfinite_items_and_queries(dev):
    """an item with an inf / NaN component scores NaN for d <= 64 (bf16x3) and is never selected; a NaN query row
    returns no finite neighbour score; other queries are unaffected"""
    from recamd import ops
    rng = np.random.default_rng(4)
    Q, N, d, k = 50, 3000, 32, 10
    q = rng.normal(size=(Q, d)).astype(np.float32)
    items = rng.normal(size=(N, d)).astype(np.float32)
    D0, I0 = ops.topk_inner_product(G(q, dev), G(items, dev), k)
    ib = items.copy()
    ib[123, 4] = np.inf
    ib[2000, 0] = np.nan
    D1, I1 = ops.topk_inner_product(G(q, dev), G(ib, dev), k)
    I1n = I1.cpu().numpy()
    assert not np.isin(I1n, [123, 2000]).any()
    assert np.isfinite(D1.cpu().numpy()).all()
    # queries whose clean top-k did not contain the poisoned items are unchanged
    same = ~np.isin(I0.cpu().numpy(), [123, 2000]).any(axis=1)
    assert np.array_equal(I1n[same], I0.cpu().numpy()[same])
    qb = q.copy()
    qb[7, 3] = np.nan
    D2, I2 = ops.topk_inner_product(G(qb, dev), G(items, dev), k)
    sel = I2.cpu().numpy()[7] >= 0                    # whatever is returned for a NaN query carries no finite score
    assert not np.isfinite(D2.cpu().numpy()[7][sel]).any()
    keepq = np.arange(Q) != 7
    assert np.array_equal(I2.cpu().numpy()[keepq], I0.cpu().numpy()[keepq])
