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


def test_sharded_tables_world1_hip(dev):
    """ShardedTables on the HIP kernels with a single rank == plain gather+concat (bit-exact)."""
    from oracle import ref_numpy as ref
    from recamd.dist import ShardedTables
    rng = np.random.default_rng(1)
    vocabs, D, B = [100, 37, 64, 1000], 16, 333
    tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
    ids = np.stack([rng.integers(-1, v + 1, size=B) for v in vocabs], axis=1).astype(np.int32)
    st = ShardedTables([torch.from_numpy(t).to(dev) for t in tables], vocabs, 0, 1)
    out = st.lookup(torch.from_numpy(ids).to(dev)).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.gather_concat(tables, ids, oob="zero").view(np.uint32))


@pytest.mark.parametrize("G", [2, 8])
def test_sharded_lookup_simulated_ranks_hip(dev, G):
    """All G ranks simulated on ONE GPU with the real HIP kernels (bucket -> route -> owner gather
    -> route back -> un-permute): every rank's result equals the unsharded oracle bit-for-bit."""
    from oracle import ref_numpy as ref
    from recamd import ops
    from recamd.dist import ShardedTables, shard_table
    rng = np.random.default_rng(G)
    vocabs, D, B = [1000] * 6, 32, 200
    tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
    ranks = [ShardedTables([shard_table(torch.from_numpy(t), r, G).to(dev) for t in tables], vocabs, r, G) for r in range(G)]
    ids = [np.stack([rng.integers(0, v, size=B) for v in vocabs], axis=1).astype(np.int32) for _ in range(G)]
    # step 1: bucket on every rank
    buck = []
    for r in range(G):
        st = ranks[r]
        t = torch.from_numpy(ids[r]).to(dev)
        vids = (t + st._shift).reshape(-1).contiguous()
        counts, perm, send_local = ops.shard_bucket(vids, G)
        buck.append((counts.cpu().numpy(), perm, send_local))
    # step 2-4: route requests to owners, gather there, route rows back
    for r in range(G):
        counts, perm, send_local = buck[r]
        offs = np.concatenate([[0], np.cumsum(counts)])
        rows = torch.empty((B * len(vocabs), D), dtype=torch.float32, device=dev)
        for o in range(G):
            req = send_local[offs[o]:offs[o + 1]].contiguous()
            served = ranks[o].kernels.gather(ranks[o].arena, req)
            rows[offs[o]:offs[o + 1]] = served
        out = ops.unpermute_rows(rows, perm).view(B, -1).cpu().numpy()
        assert np.array_equal(out.view(np.uint32), ref.gather_concat(tables, ids[r]).view(np.uint32))
