"""The opt-in bf16x3 matrix-core variant of the fused gather + pairwise dot (REC_PAIRDOT_IMPL=gram,
csrc/pairwise_dot_gram.hip) against the fp64 oracle, tolerance 1e-5 (north_star), through the same entry point."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def run(dev, B, F, V, with_dense, ids_dtype=np.int32, oob=False, scale=0.3, seed=0):
    """scale 0.3: |dot| stays O(1), so the 1e-5 * max(1, |ref|) bound is not eaten by plain fp32 accumulation
    error (128 terms x 6e-8 x |term|), which the default VALU kernel has as well."""
    from recamd import ops
    rng = np.random.default_rng(seed)
    D = 128
    tables = [(rng.normal(size=(V, D)) * scale).astype(np.float32) for _ in range(F)]
    lo, hi = (-2, V + 2) if oob else (0, V)
    ids = rng.integers(lo, hi, size=(B, F))
    dense = (rng.normal(size=(B, D)) * scale).astype(np.float32) if with_dense else None
    g = ops.TableGroup([T(t, dev) for t in tables])
    got = ops.gather_pairwise_dot(g, T(ids.astype(ids_dtype), dev), None if dense is None else T(dense, dev))
    X = ref.gather_concat([t.astype(np.float64) for t in tables], ids).reshape(B, F, D)
    if with_dense:
        X = np.concatenate([X, dense[:, None, :].astype(np.float64)], axis=1)
    exp = ref.pairwise_dot(X)
    if with_dense:
        exp = np.concatenate([exp, dense.astype(np.float64)], axis=1)
    return got.cpu().numpy(), exp


@pytest.fixture
def gram(monkeypatch):
    monkeypatch.setenv("REC_PAIRDOT_IMPL", "gram")


@pytest.mark.parametrize("B,F,with_dense", [(1, 26, True), (5, 26, True), (300, 26, True), (4097, 26, True),
                                            (9000, 26, True), (257, 26, False), (64, 1, True), (64, 2, False),
                                            (130, 31, True), (130, 32, False), (33, 8, True)])
def test_gram_matches_oracle(dev, gram, B, F, with_dense):
    got, exp = run(dev, B, F, 50, with_dense, seed=B + F)
    assert got.shape == exp.shape
    assert close(got, exp, 1e-5)


def test_gram_float_ids_and_out_of_range(dev, gram):
    got, exp = run(dev, 777, 26, 40, True, ids_dtype=np.float32, oob=True, seed=3)
    assert close(got, exp, 1e-5)


def test_gram_small_values_keep_relative_accuracy(dev, gram):
    """embedding-scale inputs (|x| ~ 0.05): the three-term split must not lose the low bits."""
    got, exp = run(dev, 512, 26, 30, False, scale=0.05, seed=9)
    assert np.max(np.abs(got - exp)) <= 2e-7 * max(1.0, np.max(np.abs(exp)))


def test_gram_and_default_kernel_agree(dev, monkeypatch):
    monkeypatch.delenv("REC_PAIRDOT_IMPL", raising=False)
    base, _ = run(dev, 1000, 26, 64, True, seed=5)
    monkeypatch.setenv("REC_PAIRDOT_IMPL", "gram")
    got, _ = run(dev, 1000, 26, 64, True, seed=5)
    assert close(got, base, 1e-5)
