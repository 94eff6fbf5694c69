"""K5 parity: DLRM pairwise-dot (standalone and fused with the gather) vs the fp64 numpy oracle.

Tolerance (BASELINE north_star): |a-b| <= 1e-5 * max(1, |b|) on fp32 values; the kernel splits the
k-sum over lanes and tree-reduces, so it is not bit-identical to a k-ordered fp32 dot."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref

pytestmark = pytest.mark.gpu


def close(a, b, tol=1e-5):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("n,D", [(27, 128), (26, 128), (9, 128), (4, 128), (27, 64), (9, 64), (4, 64),
                                 (9, 32), (27, 16), (5, 16),      # register-tiled instantiations
                                 (3, 128), (7, 20), (13, 8), (2, 4), (1, 16)])  # generic kernel
@pytest.mark.parametrize("B", [1, 2, 3, 130])
def test_pairwise_dot_plain(dev, n, D, B):
    from recamd import ops
    rng = np.random.default_rng(n * 100 + D + B)
    x = rng.normal(size=(B, n, D)).astype(np.float32)
    out = ops.pairwise_dot(torch.from_numpy(x).to(dev)).cpu().numpy()
    exp = ref.pairwise_dot(x, np.float64)
    assert out.shape == (B, n * (n - 1) // 2)
    assert close(out, exp)


def test_pairwise_dot_order_kat(dev):
    """Known answer: X rows = scaled unit vectors -> Z[i][j] = 0 except duplicates; order (i,j) i>j."""
    from recamd import ops
    n, D = 4, 16
    x = np.zeros((1, n, D), np.float32)
    x[0, 0, 0] = 1
    x[0, 1, 0] = 2
    x[0, 2, 1] = 3
    x[0, 3, :2] = [5, 7]
    # pairs: (1,0)=2 (2,0)=0 (2,1)=0 (3,0)=5 (3,1)=10 (3,2)=21
    out = ops.pairwise_dot(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.array_equal(out, np.array([[2, 0, 0, 5, 10, 21]], np.float32))


@pytest.mark.parametrize("F,D,with_dense", [(26, 128, True), (26, 128, False), (8, 128, True), (3, 128, True),
                                            (26, 64, True), (8, 64, True), (8, 32, True), (26, 16, True),
                                            (4, 16, True)])
@pytest.mark.parametrize("B", [1, 2, 5, 257])
def test_gather_pairwise_dot_fused(dev, F, D, with_dense, B):
    from recamd import ops
    rng = np.random.default_rng(F * 1000 + D + B)
    vocabs = [int(v) for v in rng.integers(5, 500, size=F)]
    tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
    ids = np.stack([rng.integers(0, v, size=B) for v in vocabs], axis=1).astype(np.int32)
    dense = rng.normal(size=(B, D)).astype(np.float32) if with_dense else None
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    out = ops.gather_pairwise_dot(g, torch.from_numpy(ids).to(dev),
                                  None if dense is None else torch.from_numpy(dense).to(dev)).cpu().numpy()
    emb = ref.gather_concat(tables, ids).reshape(B, F, D)
    X = emb if dense is None else np.concatenate([emb, dense[:, None, :]], axis=1)
    exp = ref.pairwise_dot(X, np.float64)
    n = X.shape[1]
    P = n * (n - 1) // 2
    assert close(out[:, :P], exp)
    if dense is not None:
        assert out.shape == (B, P + D)
        assert np.array_equal(out[:, P:], dense)  # pass-through is a bit-exact copy


def test_gather_pairwise_dot_oob_and_float_ids(dev):
    from recamd import ops
    rng = np.random.default_rng(9)
    F, D, B = 26, 128, 64
    vocabs = [50] * F
    tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
    ids = rng.integers(0, 50, size=(B, F)).astype(np.int32)
    ids[3, 5] = -1
    ids[7, 0] = 50
    dense = rng.normal(size=(B, D)).astype(np.float32)
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    flag = ops.new_oob_flag(dev)
    idsf = torch.from_numpy(ids.astype(np.float32) + np.where(ids >= 0, 0.5, 0.0).astype(np.float32)).to(dev)
    out = ops.gather_pairwise_dot(g, idsf, torch.from_numpy(dense).to(dev), oob_flag=flag).cpu().numpy()
    emb = ref.gather_concat(tables, ids, oob="zero").reshape(B, F, D)
    X = np.concatenate([emb, dense[:, None, :]], axis=1)
    assert close(out[:, :351], ref.pairwise_dot(X))
    assert int(flag.item()) == 1


def test_fused_matches_unfused(dev):
    """gather_concat -> pairwise_dot (two launches) and the fused launch agree bit-for-bit
    (same per-lane arithmetic order)."""
    from recamd import ops
    rng = np.random.default_rng(21)
    F, D, B = 26, 128, 300
    tables = [torch.from_numpy(rng.normal(size=(100, D)).astype(np.float32)).to(dev) for _ in range(F)]
    ids = torch.from_numpy(rng.integers(0, 100, size=(B, F)).astype(np.int32)).to(dev)
    dense = torch.from_numpy(rng.normal(size=(B, D)).astype(np.float32)).to(dev)
    g = ops.TableGroup(tables)
    fused = ops.gather_pairwise_dot(g, ids, dense)
    X = torch.cat([ops.gather_concat(g, ids).view(B, F, D), dense[:, None, :]], dim=1).contiguous()
    unf = ops.pairwise_dot(X)
    assert torch.equal(fused[:, :351], unf)
