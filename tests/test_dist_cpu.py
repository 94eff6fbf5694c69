"""N > 1 path of the row-sharded lookup, world_size 2 and 3 over gloo on the CPU.

The transport logic (owner bucketing contract, split sizes, the two all-to-alls, un-permute) is the
product code (recamd.dist.ShardedTables); the three DEVICE steps are replaced by a numpy-oracle
stand-in injected by the test (tests may use the oracle; the product has no CPU path).  The result
must be bit-identical to the single-device gather+concat of the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleKernels:
    """CPU stand-ins with the exact contract of the HIP kernels (stable bucketing, zero row on OOB)."""

    def bucket(self, vids, G):
        v = vids.numpy()
        owner = np.where(v < 0, 0, v % G)
        local = np.where(v < 0, -1, v // G).astype(np.int32)
        order = np.argsort(owner, kind="stable")
        perm = np.empty(len(v), np.int32)
        perm[order] = np.arange(len(v), dtype=np.int32)
        counts = np.bincount(owner, minlength=G).astype(np.int32)
        return torch.from_numpy(counts), torch.from_numpy(perm), torch.from_numpy(local[order])

    def gather(self, arena2d, local_rows, oob_flag=None):
        from oracle import ref_numpy as ref
        return torch.from_numpy(ref.embedding_lookup(arena2d.numpy(), local_rows.numpy(), oob="zero"))

    def unpermute(self, rows, perm, out):
        out.copy_(rows[perm.long()])
        return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, vocabs, D, B, seed, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recamd.dist import ShardedTables, shard_table
        rng = np.random.default_rng(seed)  # every rank builds the same global tables
        tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
        rng_i = np.random.default_rng(seed + 1 + rank)  # ... and its own batch
        ids = np.stack([rng_i.integers(-1, v + 1, size=B) for v in vocabs], axis=1).astype(np.int32)  # incl. OOB
        st = ShardedTables([shard_table(torch.from_numpy(t), rank, world) for t in tables], vocabs, rank, world,
                           kernels=OracleKernels())
        out = st.lookup(torch.from_numpy(ids)).numpy()
        from oracle import ref_numpy as ref
        exp = ref.gather_concat(tables, ids, oob="zero")
        ret[rank] = bool(np.array_equal(out.view(np.uint32), exp.view(np.uint32)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,vocabs,D,B", [(2, [10, 33, 7, 100], 8, 57), (3, [50, 5, 64], 4, 20), (2, [1000] * 26, 16, 128)])
def test_sharded_lookup_gloo(world, vocabs, D, B):
    import tests.conftest  # noqa: F401  (sys.path for spawned children comes from PYTHONPATH below)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "recommend-tf2.0_amd"),
                                                os.environ.get("PYTHONPATH", "")])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), vocabs, D, B, 123, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_helpers():
    from recamd.dist import local_rows_of, shard_table
    t = torch.arange(10 * 2, dtype=torch.float32).view(10, 2)
    for world in (1, 2, 3, 4):
        tot = 0
        for r in range(world):
            s = shard_table(t, r, world)
            assert s.shape[0] == local_rows_of(10, r, world)
            assert torch.equal(s, t[r::world])
            tot += s.shape[0]
        assert tot == 10
