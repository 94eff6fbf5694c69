"""N > 1 path of the row-sharded lookup and of the gradient merge, world_size 2 and 3 over gloo on the CPU.

The transport logic (virtual ids, the row space, split sizes from the gathered count matrix, the two all-to-alls,
reading rows through uidx, local-shard bypass, the prefetch pipeline's bookkeeping, the hot-row replica cache, the
reverse all-to-all of the backward, the all-reduce) is the product code (recamd.dist.ShardedTables, transport
'torch'); the DEVICE steps are replaced by the numpy stand-ins of tests/shard_oracle.py::OracleShardedTables (tests
may use the oracle; the product has no CPU path).  Results must be bit-identical to the single-device gather+concat
of the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(seed, rank, vocabs, B):
    rng_i = np.random.default_rng(seed + 1 + rank)
    ids = np.stack([rng_i.integers(-1, v + 1, size=B) for v in vocabs], axis=1).astype(np.int32)  # incl. OOB
    ids[B // 2:] = ids[: B - B // 2]                                                              # ... and duplicates
    return ids


def _worker(rank, world, port, vocabs, D, B, seed, dedup, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_numpy as ref
        from recamd.dist import shard_table
        from tests.shard_oracle import OracleShardedTables
        rng = np.random.default_rng(seed)  # every rank builds the same global tables
        tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
        ids = _batch(seed, rank, vocabs, B)
        st = OracleShardedTables([shard_table(torch.from_numpy(t), rank, world) for t in tables], vocabs, rank, world,
                                 dedup=dedup)
        flag = torch.zeros(1, dtype=torch.int32)
        t_ids = torch.from_numpy(ids)
        st.prefetch(t_ids)                                    # the pipelined form: plan first, look up later
        out, plan = st.lookup(t_ids, oob_flag=flag, keep_plan=True)
        exp = ref.gather_concat(tables, ids, oob="zero")
        ok = bool(np.array_equal(out.numpy().view(np.uint32), exp.view(np.uint32)))
        has_oob = bool(((ids < 0) | (ids >= np.asarray(vocabs)[None, :])).any())
        ok = ok and int(flag.item()) == int(has_oob) and st.stats["prefetch_hits"] == 1
        if dedup:
            ok = ok and st.stats["unique_sent"] < st.stats["ids"]
        own = ((ids >= 0) & (ids < np.asarray(vocabs)[None, :]) & (ids % world == rank)).sum()
        ok = ok and st.np_stats["local"] == int(own)           # rows this rank owns were read in place ...
        ok = ok and plan.send_splits[rank] == 0                # ... and never sent to itself

        # backward: every rank's dy, summed at the owners == the oracle's dense gradient over ALL ranks' lookups
        dys = [np.random.default_rng(seed + 100 + r).normal(size=(B, len(vocabs) * D)).astype(np.float32) for r in range(world)]
        grad_arena = torch.zeros_like(st.arena)
        st.backward(plan, torch.from_numpy(dys[rank]), grad_arena)
        full = [np.zeros((v, D), np.float64) for v in vocabs]
        for r in range(world):
            g = ref.embedding_grad(_batch(seed, r, vocabs, B), dys[r], vocabs, [D] * len(vocabs))
            for f in range(len(vocabs)):
                full[f] += g[f]
        for f, v in enumerate(vocabs):
            mine = grad_arena[f * st.rows_local: f * st.rows_local + len(range(rank, v, world))].numpy()
            ok = ok and bool(np.allclose(mine, full[f][rank::world], rtol=1e-5, atol=1e-6))

        # dense-parameter gradient merge
        t = torch.full((5,), float(rank + 1))
        st.allreduce_sum_(t)
        ok = ok and bool(torch.equal(t, torch.full((5,), float(world * (world + 1) // 2))))

        # the DLRM sparse stage through the exchange (second lookup of the same ids: plans on the spot)
        dense = torch.from_numpy(np.random.default_rng(seed + 7 + rank).normal(size=(B, D)).astype(np.float32))
        z = st.lookup_pairwise_dot(t_ids, dense).numpy()
        X = np.concatenate([exp.reshape(B, len(vocabs), D), dense.numpy()[:, None, :]], axis=1)
        ok = ok and bool(np.array_equal(z[:, :-D], ref.pairwise_dot(X).astype(np.float32))) and \
            bool(np.array_equal(z[:, -D:], dense.numpy()))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def _spawn(fn, world, *args):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "recommend-tf2.0_amd"),
                                                os.environ.get("PYTHONPATH", "")])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(fn, args=(world, _free_port()) + args + (ret,), nprocs=world, join=True)
    return dict(ret)


@pytest.mark.parametrize("world,vocabs,D,B,dedup", [(2, [10, 33, 7, 100], 8, 57, True), (3, [50, 5, 64], 4, 20, True),
                                                    (2, [1000] * 26, 16, 128, True), (2, [10, 33, 7, 100], 8, 57, False)])
def test_sharded_lookup_backward_allreduce_gloo(world, vocabs, D, B, dedup):
    ret = _spawn(_worker, world, vocabs, D, B, 123, dedup)
    assert all(ret.get(r) for r in range(world)), ret


def _sasrec_worker(rank, world, port, V, S, n_neg, B, ret):
    """BASELINE configs[4] through the exchange: the seq / pos / neg lookups of src/match/sasrec/model.py:75-79 travel
    as ONE sharded lookup (pad id 0 dropped before it); the rows read back through uidx feed the oracle's encoder and
    must give the logits of the oracle's sasrec_forward on the UNSHARDED tables."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_numpy as ref
        from recamd.dist import shard_table
        from tests.shard_oracle import OracleShardedTables
        d = 64
        rng = np.random.default_rng(7)                      # same tables and weights on every rank
        T = [rng.normal(size=(V, d)).astype(np.float32) * 0.3 for _ in range(3)]
        P = dict(Wq=rng.normal(size=(d, d)) * 0.1, bq=rng.normal(size=d) * 0.1, Wk=rng.normal(size=(d, d)) * 0.1,
                 bk=rng.normal(size=d) * 0.1, Wv=rng.normal(size=(d, d)) * 0.1, bv=rng.normal(size=d) * 0.1,
                 W1=rng.normal(size=(d, 128)) * 0.1, b1=rng.normal(size=128) * 0.1, W2=rng.normal(size=(128, d)) * 0.1,
                 b2=rng.normal(size=d) * 0.1, ln1_g=np.ones(d), ln1_b=np.zeros(d), ln2_g=np.ones(d), ln2_b=np.zeros(d))
        r2 = np.random.default_rng(100 + rank)              # every rank its own batch
        lens = r2.integers(0, S + 1, size=B)
        seq = r2.integers(1, V, size=(B, S))
        seq[np.arange(S)[None, :] < (S - lens)[:, None]] = 0
        seq, pos, neg = seq.astype(np.int32), r2.integers(0, V, size=(B, 1)).astype(np.int32), \
            r2.integers(0, V, size=(B, n_neg)).astype(np.int32)
        st = OracleShardedTables([shard_table(torch.from_numpy(t), rank, world) for t in T], [V] * 3, rank, world, dedup=True)
        ts, tp, tn = (torch.from_numpy(a) for a in (seq, pos, neg))
        vids = torch.cat([st.virtual_ids(0, ts, pad_id=0).reshape(-1), st.virtual_ids(1, tp).reshape(-1),
                          st.virtual_ids(2, tn).reshape(-1)])
        rows, uidx = st.lookup_rows(vids)
        rows, uidx = rows.numpy(), uidx.numpy()
        got = np.where((uidx >= 0)[:, None], rows[np.maximum(uidx, 0)], 0.0).astype(np.float32)
        seq_e = got[:B * S].reshape(B, S, d)
        pos_e = got[B * S:B * S + B].reshape(B, 1, d)
        neg_e = got[B * S + B:].reshape(B, n_neg, d)
        mask = (seq != 0).astype(np.float32)[..., None]
        ok = bool(np.array_equal(seq_e, T[0][seq] * mask)) and bool(np.array_equal(pos_e, T[1][pos])) and \
            bool(np.array_equal(neg_e, T[2][neg]))                      # bit-exact rows, pads as zero rows
        ok = ok and st.stats["unique_sent"] <= int((seq != 0).sum()) + B + B * n_neg      # pads were never sent
        x = ref.transformer_encoder(seq_e.astype(np.float64) * mask, mask, P, 1) * mask
        si = x[:, -1][:, None, :]
        logits = np.concatenate([np.sum(si * pos_e, -1), np.sum(si * neg_e, -1)], -1)
        exp, _ = ref.sasrec_forward(seq, pos, neg, T[0], T[1], T[2], [P], 1)
        ok = ok and bool(np.allclose(logits, exp, rtol=1e-12, atol=1e-12))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_sasrec_lookups_through_the_exchange_gloo():
    ret = _spawn(_sasrec_worker, 2, 700, 200, 20, 9)
    assert all(ret.get(r) for r in range(2)), ret


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


def test_dedup_bucket_contract():
    """the numpy specification itself: representatives are first occurrences, send order is stable by owner"""
    from tests.shard_oracle import dedup_bucket_np
    v = np.array([5, -1, 9, 5, 4, 9, 9, 2], np.int32)
    counts, uidx, send_local, first, perm = dedup_bucket_np(v, 2)
    # unique reps: i=0 (5, owner 1), 2 (9, owner 1), 4 (4, owner 0), 7 (2, owner 0) -> send order: 4, 2 | 5, 9
    assert counts.tolist() == [2, 2]
    assert send_local.tolist() == [2, 1, 2, 4]
    assert first.tolist() == [0, -1, 2, 0, 4, 2, 2, 7]
    assert uidx.tolist() == [2, -1, 3, 2, 0, 3, 3, 1]


def test_int64_ids_are_range_checked_before_narrowing():
    from tests.shard_oracle import OracleShardedTables
    st = OracleShardedTables([torch.zeros((10, 4))], [10], 0, 1)
    v = st._vids(torch.tensor([[3], [2 ** 32 + 3], [-7], [9]], dtype=torch.int64))
    assert v.tolist() == [3, -1, -1, 9]


def test_resolve_contract_local_cached_remote():
    """the numpy specification of rec_shard_resolve_i32: local rows in place, cached rows from the replica region,
    the rest de-duplicated into the send list and read at recv_base + position"""
    from tests.shard_oracle import resolve_np
    G, me = 2, 1
    v = np.array([5, -1, 8, 5, 4, 9, 8, 2, 9], np.int32)         # odd ids are local to rank 1
    cache_slot = np.full(16, -1, np.int32)
    cache_slot[8] = 3                                            # row 8 has a replica in cache entry 3
    hot = np.zeros(16, np.int32)
    counts, uidx, send_local, first, perm = resolve_np(v, G, me, True, cache_slot, cache_base=100, recv_base=1000,
                                                       hot_count=hot)
    # remote uncached: 4 (i=4), 2 (i=7) -> both owner 0, send order 4, 2 -> local rows 2, 1
    assert counts.tolist() == [2, 0] and send_local.tolist() == [2, 1]
    assert uidx.tolist() == [2, -1, 103, 2, 1000, 4, 103, 1001, 4]
    assert hot[8] == 2 and hot[4] == 1 and hot[2] == 1 and hot[5] == 0 and hot[9] == 0   # remote lookups only


def _pipeline_worker(rank, world, port, ret):
    """the prefetch pipeline as bench.py drives it: plan two batches ahead, ids + rows one batch ahead, consume"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_numpy as ref
        from recamd.dist import shard_table
        from tests.shard_oracle import OracleShardedTables
        vocabs, D, B, steps = [40, 17, 64], 8, 33, 7
        rng = np.random.default_rng(5)
        tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
        st = OracleShardedTables([shard_table(torch.from_numpy(t), rank, world) for t in tables], vocabs, rank, world,
                                 max_ids=B * len(vocabs), slots=3)
        batches = [torch.from_numpy(_batch(50 + k, rank, vocabs, B)) for k in range(steps + 2)]
        ok = True
        st.prefetch(batches[0], rows=True)
        st.prefetch(batches[1])
        for i in range(steps):
            st.prefetch(batches[i + 2])                       # plan (i+2)
            st.prefetch(batches[i + 1], rows=True)            # ids + rows of (i+1)
            out = st.lookup(batches[i])                       # consume (i)
            exp = ref.gather_concat(tables, batches[i].numpy(), oob="zero")
            ok = ok and bool(np.array_equal(out.numpy().view(np.uint32), exp.view(np.uint32)))
        ok = ok and st.stats["prefetch_hits"] == steps and st.stats["rows_prefetched"] == steps + 1
        extra = [torch.from_numpy(_batch(999 + k, rank, vocabs, B)) for k in range(2)]
        st.prefetch(extra[0])                                  # third lookup in flight: the last free slot
        try:                                                   # a 4th must fail loudly, not overwrite a slot
            st.prefetch(extra[1])
            ok = False
        except RuntimeError:
            pass
        ret[rank] = (ok, dict(st.stats))
    finally:
        dist.destroy_process_group()


def test_prefetch_pipeline_gloo():
    ret = _spawn(_pipeline_worker, 2)
    assert all(ret.get(r) and ret[r][0] for r in range(2)), ret


def _cache_worker(rank, world, port, ret):
    """hot-row replica cache under Zipf ids: unique remote rows per step drop >= 5x, results stay bit-exact, a weight
    write invalidates the replicas"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_numpy as ref
        from recamd.dist import shard_table
        from tests.shard_oracle import OracleShardedTables
        F, V, D, B = 4, 2000, 8, 512
        vocabs = [V] * F
        rng = np.random.default_rng(11)
        tables = [rng.normal(size=(V, D)).astype(np.float32) for _ in range(F)]

        def zipf_batch(k):
            z = np.random.default_rng(1000 * rank + k).zipf(1.4, size=(B, F))
            return torch.from_numpy(((z - 1) % V).astype(np.int32))

        def run(cache_rows):
            OracleShardedTables.generation = 0
            st = OracleShardedTables([shard_table(torch.from_numpy(t), rank, world) for t in tables], vocabs, rank, world,
                                     max_ids=B * F, cache_rows=cache_rows, cache_refresh_every=4 if cache_rows else 0)
            sent, ok = [], True
            for k in range(32):
                ids = zipf_batch(k)
                before = st.stats["unique_sent"]
                out = st.lookup(ids)
                sent.append(st.stats["unique_sent"] - before)
                ok = ok and bool(np.array_equal(out.numpy().view(np.uint32),
                                                ref.gather_concat(tables, ids.numpy(), oob="zero").view(np.uint32)))
            return st, sent, ok

        _, sent_plain, ok0 = run(0)
        st, sent_cached, ok1 = run(2048)
        ok = ok0 and ok1 and st.stats["cache_refreshes"] == 7
        steady_plain, steady_cached = sum(sent_plain[24:]), sum(sent_cached[24:])
        ok = ok and steady_cached * 5 <= steady_plain and st.np_stats["cached"] > 0
        # a weight write: the owners' rows change, the replicas must not be served any more
        for f in range(F):
            st.tables[f].mul_(2.0)
        OracleShardedTables.generation += 1
        ids = zipf_batch(99)
        out = st.lookup(ids)
        ok = ok and bool(np.array_equal(out.numpy(), 2.0 * ref.gather_concat(tables, ids.numpy(), oob="zero")))
        ret[rank] = (ok, steady_plain, steady_cached)
    finally:
        dist.destroy_process_group()


def test_hot_row_cache_gloo():
    ret = _spawn(_cache_worker, 2)
    assert all(ret.get(r) and ret[r][0] for r in range(2)), ret


def _sasrec_train_worker(rank, world, port, V, S, n_neg, B, ret):
    """TRAINING of the row-sharded SASRec (BASELINE configs[4]) through the exchange: the seq / pos / neg lookups travel as
    one sharded lookup kept for backward; the loss (src/match/sasrec/model.py:93-95, scaled by 1 / world on every rank)
    is differentiated by the fp64 autograd oracle down to the looked-up rows; ShardedTables.backward returns those row
    gradients to the owners.  Every rank's gradient arena must equal its rows of the oracle's gradient of the GLOBAL mean
    loss on the unsharded tables; the dense parameters' gradients merge by the all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_torch
        from recamd.dist import shard_table
        from tests.shard_oracle import OracleShardedTables
        d = 16
        rng = np.random.default_rng(3)                      # same tables and weights on every rank
        T = [rng.normal(size=(V, d)) * 0.3 for _ in range(3)]
        blk = dict(Wq=rng.normal(size=(d, d)) * 0.2, bq=rng.normal(size=d) * 0.1, Wk=rng.normal(size=(d, d)) * 0.2,
                   bk=rng.normal(size=d) * 0.1, Wv=rng.normal(size=(d, d)) * 0.2, bv=rng.normal(size=d) * 0.1,
                   W1=rng.normal(size=(d, 24)) * 0.2, b1=rng.normal(size=24) * 0.1, W2=rng.normal(size=(24, d)) * 0.2,
                   b2=rng.normal(size=d) * 0.1, ln1_g=np.ones(d), ln1_b=np.zeros(d), ln2_g=np.ones(d), ln2_b=np.zeros(d))

        def batch(r):
            r2 = np.random.default_rng(100 + r)
            lens = r2.integers(0, S + 1, size=B)
            seq = r2.integers(1, V, size=(B, S))
            seq[np.arange(S)[None, :] < (S - lens)[:, None]] = 0
            return seq.astype(np.int32), r2.integers(0, V, size=(B, 1)).astype(np.int32), \
                r2.integers(0, V, size=(B, n_neg)).astype(np.int32)

        # ---- the product path: one exchange forward, one back ----------------------------------------------------
        st = OracleShardedTables([shard_table(torch.from_numpy(t.astype(np.float32)), rank, world) for t in T], [V] * 3,
                                 rank, world)
        seq, pos, neg = batch(rank)
        ts, tp, tn = (torch.from_numpy(a) for a in (seq, pos, neg))
        vids = torch.cat([st.virtual_ids(0, ts, pad_id=0).reshape(-1), st.virtual_ids(1, tp).reshape(-1),
                          st.virtual_ids(2, tn).reshape(-1)])
        rows, uidx, plan = st.lookup_rows(vids, keep_plan=True)
        E = torch.where((uidx >= 0)[:, None], rows[uidx.clamp(min=0).long()], torch.zeros(1)).double().requires_grad_(True)
        Pw = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in blk.items()}
        seq_e = E[:B * S].view(B, S, d)
        mask = torch.from_numpy((seq != 0).astype(np.float64))
        x = ref_torch.encoder(seq_e * mask[..., None], mask, Pw, 1) * mask[..., None]
        si = x[:, -1][:, None, :]
        pos_e, neg_e = E[B * S:B * S + B].view(B, 1, d), E[B * S + B:].view(B, n_neg, d)
        pl, nl = (si * pos_e).sum(-1), (si * neg_e).sum(-1)
        loss = ((-torch.log(torch.sigmoid(pl)) - torch.log(1 - torch.sigmoid(nl))) / 2).mean()     # :93-95, (B,1)+(B,n)
        (loss / world).backward()
        grad_arena = torch.zeros_like(st.arena)
        st.backward(plan, E.grad.float(), grad_arena)
        dense = torch.cat([Pw[k].grad.reshape(-1) for k in sorted(Pw)]).float()
        st.allreduce_sum_(dense)

        # ---- the oracle: global mean loss on the unsharded tables ---------------------------------------------------
        Tt = [torch.tensor(t.astype(np.float32).astype(np.float64), requires_grad=True) for t in T]
        Pf = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in blk.items()}
        tot = 0.0
        for r in range(world):
            sq, ps, ng = batch(r)
            _, l_r = ref_torch.sasrec(sq, ps, ng, Tt[0], Tt[1], Tt[2], [Pf], 1)
            tot = tot + l_r / world
        tot.backward()
        ok = True
        for f in range(3):
            mine = grad_arena[f * st.rows_local: f * st.rows_local + len(range(rank, V, world))].numpy()
            ok = ok and bool(np.allclose(mine, Tt[f].grad.numpy()[rank::world], rtol=1e-5, atol=1e-7))
        edense = torch.cat([Pf[k].grad.reshape(-1) for k in sorted(Pf)]).numpy()
        ok = ok and bool(np.allclose(dense.numpy(), edense, rtol=1e-5, atol=1e-7))
        ok = ok and all(not s.busy for s in st._slots)                  # the kept plan released its receive slot
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_sasrec_training_through_the_exchange_gloo():
    ret = _spawn(_sasrec_train_worker, 2, 300, 12, 6, 10)
    assert all(ret.get(r) for r in range(2)), ret
