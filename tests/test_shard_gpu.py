"""C2 device helpers: stable owner bucketing + un-permute, bit-exact vs a numpy stable sort."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 5, 1024, 1025, 100_003])
@pytest.mark.parametrize("G", [1, 2, 8, 7])
def test_shard_bucket_matches_stable_sort(dev, n, G):
    from recamd import ops
    rng = np.random.default_rng(n + G)
    ids = rng.integers(0, 1_000_000, size=n).astype(np.int32)
    if n > 3:
        ids[1] = -5  # negative -> owner 0, local -1
    counts, perm, send_local = ops.shard_bucket(torch.from_numpy(ids).to(dev), G)
    counts, perm, send_local = counts.cpu().numpy(), perm.cpu().numpy(), send_local.cpu().numpy()
    owner = np.where(ids < 0, 0, ids % G)
    local = np.where(ids < 0, -1, ids // G)
    order = np.argsort(owner, kind="stable")
    exp_perm = np.empty(n, np.int64)
    exp_perm[order] = np.arange(n)
    assert np.array_equal(counts, np.bincount(owner, minlength=G))
    assert np.array_equal(perm, exp_perm)
    assert np.array_equal(send_local, local[order])


def test_unpermute_rows(dev):
    from recamd import ops
    rng = np.random.default_rng(0)
    for D in (128, 16, 6):
        rows = rng.normal(size=(500, D)).astype(np.float32)
        perm = rng.permutation(500).astype(np.int32)
        out = ops.unpermute_rows(torch.from_numpy(rows).to(dev), torch.from_numpy(perm).to(dev)).cpu().numpy()
        assert np.array_equal(out, rows[perm])
