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


@pytest.mark.parametrize("N", [3, 200, 5000, 70000])
@pytest.mark.parametrize("d", [8, 64])
def test_topk_all_negative_scores(dev, N, d):
    """Every candidate scores < 0 (queries > 0, items < 0): the insertion path must compare against the live k-th
    entry, not against the (0.0, 0) a shuffle from an EXEC-disabled lane returns (round-1 ADVICE, topk.hip:351)."""
    from recamd import ops
    rng = np.random.default_rng(N + d)
    q = rng.uniform(0.1, 1.0, size=(33, d)).astype(np.float32)
    items = -rng.uniform(0.1, 1.0, size=(N, d)).astype(np.float32)
    k = 10
    D, I = ops.topk_inner_product(torch.from_numpy(q).to(dev), torch.from_numpy(items).to(dev), k)
    eD, eI = ref.topk_inner_product(q, items, k)
    check(D.cpu().numpy(), I.cpu().numpy(), eD, eI, q, items)


def test_topk_tiny_negative_kat(dev):
    """Q=1, N=3, scores -1, -2, -3 -> indices [0, 1, 2], then -1 padding."""
    from recamd import ops
    q = np.ones((1, 8), np.float32)
    items = np.zeros((3, 8), np.float32)
    items[:, 0] = [-1, -2, -3]
    D, I = ops.topk_inner_product(torch.from_numpy(q).to(dev), torch.from_numpy(items).to(dev), 10)
    I = I.cpu().numpy()
    D = D.cpu().numpy()
    assert I[0, :3].tolist() == [0, 1, 2] and np.all(I[0, 3:] == -1)
    assert D[0, :3].tolist() == [-1.0, -2.0, -3.0]


def test_topk_zero_score_ties(dev):
    """all scores exactly 0 (orthogonal vectors): ties resolve to the smallest indices, in order."""
    from recamd import ops
    q = np.zeros((5, 16), np.float32)
    q[:, 0] = 1
    items = np.zeros((300, 16), np.float32)
    items[:, 1] = np.arange(300)
    D, I = ops.topk_inner_product(torch.from_numpy(q).to(dev), torch.from_numpy(items).to(dev), 10)
    assert np.array_equal(I.cpu().numpy(), np.tile(np.arange(10), (5, 1)))
    assert np.array_equal(D.cpu().numpy(), np.zeros((5, 10), np.float32))


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
