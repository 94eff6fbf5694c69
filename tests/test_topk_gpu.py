"""§8f-4: exact inner-product top-k (the faiss IndexFlatIP step of the match train scripts) vs the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref

pytestmark = pytest.mark.gpu


def check(D, I, eD, eI, q, items, tol=1e-5):
    assert D.shape == eD.shape and I.shape == eI.shape and I.dtype == np.int64
    fin = np.isfinite(eD)
    assert np.array_equal(np.isfinite(D), fin)
    assert np.all(np.abs(D[fin] - eD[fin]) <= tol * np.maximum(1.0, np.abs(eD[fin])))
    assert np.all(I[~fin] == -1)
    # indices: equal wherever the oracle's neighbouring scores are separated by more than the fp32 tolerance;
    # otherwise the returned index must still score like the expected one
    same = I == eI
    if not same.all():
        rows, cols = np.nonzero(~same)
        got_scores = np.einsum('nd,nd->n', q[rows].astype(np.float64), items[I[rows, cols]].astype(np.float64))
        assert np.all(np.abs(got_scores - eD[rows, cols]) <= 2 * tol * np.maximum(1.0, np.abs(eD[rows, cols])))
    for r in range(D.shape[0]):                                                       # descending
        v = D[r][np.isfinite(D[r])]
        assert np.all(np.diff(v) <= 0)
    for r in range(I.shape[0]):                                                       # no duplicates
        v = I[r][I[r] >= 0]
        assert len(set(v.tolist())) == len(v)


@pytest.mark.parametrize("Q,N,d,k", [(1, 1, 8, 1), (5, 7, 8, 10), (130, 129, 32, 10), (257, 1000, 32, 10),
                                     (64, 5000, 64, 32), (300, 300, 17, 5), (128, 2048, 128, 10),
                                     (100, 20000, 32, 10), (6040, 3706, 32, 10), (7, 70000, 64, 32)])   # split-N path
def test_topk_matches_oracle(dev, Q, N, d, k):
    from recamd import ops
    rng = np.random.default_rng(Q * 7 + N)
    q = rng.normal(size=(Q, d)).astype(np.float32)
    items = rng.normal(size=(N, d)).astype(np.float32)
    D, I = ops.topk_inner_product(torch.from_numpy(q).to(dev), torch.from_numpy(items).to(dev), k)
    eD, eI = ref.topk_inner_product(q, items, k)
    check(D.cpu().numpy(), I.cpu().numpy(), eD, eI, q, items)


def test_ties_order_by_index(dev):
    """integer-valued vectors: exact scores with many ties -> smaller index first, bit-exact scores."""
    from recamd import ops
    rng = np.random.default_rng(3)
    q = rng.integers(-2, 3, size=(40, 8)).astype(np.float32)
    items = rng.integers(-2, 3, size=(500, 8)).astype(np.float32)
    D, I = ops.topk_inner_product(torch.from_numpy(q).to(dev), torch.from_numpy(items).to(dev), 10)
    eD, eI = ref.topk_inner_product(q, items, 10)
    assert np.array_equal(D.cpu().numpy(), eD.astype(np.float32))
    assert np.array_equal(I.cpu().numpy(), eI)


def test_index_flat_ip_object(dev):
    """the faiss-shaped wrapper used like src/match/dssm/dssm_train.py:74-78 (two adds concatenate)."""
    from recamd.retrieval import IndexFlatIP
    rng = np.random.default_rng(5)
    items = rng.normal(size=(700, 32)).astype(np.float32)
    users = rng.normal(size=(90, 32)).astype(np.float32)
    index = IndexFlatIP(32)
    index.add(items[:300])
    index.add(items[300:])
    assert index.ntotal == 700
    D, I = index.search(np.ascontiguousarray(users), 10)
    eD, eI = ref.topk_inner_product(users, items, 10)
    check(D, I, eD, eI, users, items)


def test_empty_index_and_no_queries(dev):
    from recamd import ops
    q = torch.randn(3, 8, device=dev)
    D, I = ops.topk_inner_product(q, torch.empty((0, 8), device=dev), 4)
    assert torch.isinf(D).all() and (D < 0).all() and (I == -1).all()
    D, I = ops.topk_inner_product(torch.empty((0, 8), device=dev), torch.randn(5, 8, device=dev), 4)
    assert D.shape == (0, 4) and I.shape == (0, 4)


def test_split_n_ties_are_deterministic(dev):
    """few queries, many items (split-N + merge kernel), integer-valued data with many equal scores"""
    from recamd import ops
    rng = np.random.default_rng(8)
    q = rng.integers(-2, 3, size=(33, 16)).astype(np.float32)
    items = rng.integers(-2, 3, size=(30000, 16)).astype(np.float32)
    D, I = ops.topk_inner_product(torch.from_numpy(q).to(dev), torch.from_numpy(items).to(dev), 10)
    eD, eI = ref.topk_inner_product(q, items, 10)
    assert np.array_equal(D.cpu().numpy(), eD.astype(np.float32))
    assert np.array_equal(I.cpu().numpy(), eI)
