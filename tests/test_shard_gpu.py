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


def _resolve_hip(dev, vids, G, me, rep, cache_slot, hot, cache_base, recv_base, stat):
    from recamd._lib import C
    n = len(vids)
    t_v = torch.from_numpy(vids).to(dev)
    i32 = lambda m: torch.empty(max(1, m), dtype=torch.int32, device=dev)  # noqa: E731
    first, uniq, perm, uidx, send_local, counts = i32(n), i32(n), i32(n), i32(n), i32(n), i32(G)
    ws = torch.empty(max(1, C.shard_bucket_workspace_bytes(n, G)), dtype=torch.uint8, device=dev)
    ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    C.shard_resolve_i32(t_v.data_ptr(), n, G, me, ptr(rep), ptr(cache_slot), ptr(hot), cache_base, recv_base, ptr(stat),
                        first.data_ptr(), uniq.data_ptr(), perm.data_ptr(), uidx.data_ptr(), send_local.data_ptr(),
                        counts.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return counts.cpu().numpy(), uidx[:n].cpu().numpy(), send_local[:n].cpu().numpy()


@pytest.mark.parametrize("n", [0, 1, 7, 1024, 1025, 50_003])
@pytest.mark.parametrize("G", [1, 2, 8, 7])
@pytest.mark.parametrize("dedup", [True, False])
def test_dedup_bucket_matches_numpy_contract(dev, n, G, dedup):
    """rec_shard_resolve_i32 without a local shard / cache (= rec_shard_dedup_bucket_i32) == tests/shard_oracle.
    dedup_bucket_np, integer for integer, and the representative table is left clean (all INT32_MAX) for the next call."""
    from tests.shard_oracle import dedup_bucket_np
    rng = np.random.default_rng(n * 3 + G)
    R = 5000
    vids = rng.integers(0, R, size=n).astype(np.int32)          # many duplicates
    if n > 3:
        vids[1] = -1
        vids[n - 1] = vids[0]
    rep = torch.full((R,), 2 ** 31 - 1, dtype=torch.int32, device=dev) if dedup else None
    counts, uidx, send_local = _resolve_hip(dev, vids, G, -1, rep, None, None, 0, 0, None)
    e_counts, e_uidx, e_send, _, _ = dedup_bucket_np(vids, G, dedup)
    assert np.array_equal(counts, e_counts)
    assert np.array_equal(uidx, e_uidx)
    assert np.array_equal(send_local[:len(e_send)], e_send)
    if dedup:
        assert int((rep != 2 ** 31 - 1).sum().item()) == 0


@pytest.mark.parametrize("n", [1, 1000, 70_001])
@pytest.mark.parametrize("G,me", [(2, 0), (8, 5), (3, 2)])
@pytest.mark.parametrize("cached", [False, True])
def test_resolve_row_space_matches_numpy_contract(dev, n, G, me, cached):
    """rec_shard_resolve_i32 with a local shard (rows read in place, never sent to itself), a replica cache and the
    row-space offsets == tests/shard_oracle.resolve_np; hot counts and the local / cached statistics agree."""
    from tests.shard_oracle import resolve_np
    rng = np.random.default_rng(n + 13 * G + me)
    R = 4000
    vids = rng.integers(-1, R, size=n).astype(np.int32)
    cs_np = np.full(R, -1, np.int32)
    if cached:
        hot_rows = rng.choice(R, size=300, replace=False)
        cs_np[hot_rows] = np.arange(300, dtype=np.int32)
    rep = torch.full((R,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
    cache_slot = torch.from_numpy(cs_np).to(dev) if cached else None
    hot = torch.zeros(R, dtype=torch.int32, device=dev)
    stat = torch.zeros(2, dtype=torch.int64, device=dev)
    counts, uidx, send_local = _resolve_hip(dev, vids, G, me, rep, cache_slot, hot, 10_000, 20_000, stat)
    hot_np = np.zeros(R, np.int32)
    e_counts, e_uidx, e_send, _, _ = resolve_np(vids, G, me, True, cs_np if cached else None, 10_000, 20_000, hot_np)
    assert np.array_equal(counts, e_counts) and counts[me] == 0
    assert np.array_equal(uidx, e_uidx)
    assert np.array_equal(send_local[:len(e_send)], e_send)
    assert np.array_equal(hot.cpu().numpy(), hot_np)
    u = e_uidx.astype(np.int64)
    assert stat.tolist() == [int(((u >= 0) & (u < 10_000)).sum()), int(((u >= 10_000) & (u < 20_000)).sum())]
    assert int((rep != 2 ** 31 - 1).sum().item()) == 0


@pytest.mark.parametrize("cache_rows", [0, 512])
def test_pipeline_bypass_cache_simulated_ranks(dev, cache_rows):
    """The product's lookup pipeline on the real kernels and streams, G ranks simulated in one process
    (tests/shard_oracle.py::PeersShardedTables replaces only the transport): plan two batches ahead, ids + rows one
    batch ahead on the communication stream, consume on the compute stream — as bench.py drives it.  The fused gather +
    pairwise dot and the gather + concat read the row space (local shard in place | replica cache | receive slot) and
    must equal the unsharded kernels bit for bit."""
    from recamd import ops
    from recamd.dist import shard_table
    from tests.shard_oracle import PeersShardedTables
    G, F, D, B, V, steps = 4, 26, 128, 256, 3000, 9
    rng = np.random.default_rng(17)
    tables = [torch.from_numpy(rng.normal(size=(V, D)).astype(np.float32)).to(dev) for _ in range(F)]
    full = ops.TableGroup(tables)
    st = [PeersShardedTables([shard_table(t, r, G) for t in tables], [V] * F, r, G, max_ids=B * F, cache_rows=cache_rows,
                             cache_refresh_every=3 if cache_rows else 0) for r in range(G)]
    for s in st:
        s.link_peers(st)
    dense = torch.from_numpy(rng.normal(size=(B, D)).astype(np.float32)).to(dev)

    def batch(r, k):
        z = np.random.default_rng(100 * r + k).zipf(1.3, size=(B, F))
        ids = ((z - 1) % V).astype(np.int32)
        ids[0, 0] = -1 if k % 2 else V                    # an out-of-range id in every batch
        return torch.from_numpy(ids).to(dev)
    for r in range(G):
        batches = [batch(r, k) for k in range(steps + 2)]
        st[r].prefetch(batches[0], rows=True)
        st[r].prefetch(batches[1])
        for i in range(steps):
            st[r].prefetch(batches[i + 2])
            st[r].prefetch(batches[i + 1], rows=True)
            flag = ops.new_oob_flag(dev)
            if i % 2 == 0:
                got = st[r].lookup_pairwise_dot(batches[i], dense, oob_flag=flag)
                exp = ops.gather_pairwise_dot(full, batches[i], dense)
            else:
                got = st[r].lookup(batches[i], oob_flag=flag)
                exp = ops.gather_concat(full, batches[i])
            assert torch.equal(got.view(torch.int32), exp.view(torch.int32)), (r, i)
            assert int(flag.item()) == 1
        d = st[r].describe()
        assert d["prefetch_hits"] == steps and d["rows_prefetched"] == steps + 1 and d["pipelined"]
        assert d["local_lookups"] > 0 and d["unique_sent"] < d["ids"]
        if cache_rows:
            assert d["cache_refreshes"] >= 2 and d["cache_hits"] > 0


@pytest.mark.parametrize("G", [2, 8])
@pytest.mark.parametrize("dedup", [True, False])
def test_cabi_exchange_simulated_ranks(dev, G, dedup):
    """The C-ABI exchange (rec_shard_plan_* / rec_shard_exchange_* / rec_shard_serve_f32) with G simulated ranks on ONE
    GPU over the in-process transport (rec_comm_create_local): every phase is run for every rank before the next one.
    Forward: each rank's lookup == the unsharded oracle bit for bit, read through uidx by the real consumer kernels
    (gather+concat and the fused pairwise dot).  Backward: the reverse all-to-all + owner scatter-add == the oracle's
    dense gradient over all ranks' lookups.  All-reduce: sum over ranks."""
    from oracle import ref_numpy as ref
    from recamd import ops
    from recamd._lib import C
    from recamd.dist import ShardedTables, shard_table
    rng = np.random.default_rng(G)
    F, D, B, V = 6, 128, 150, 400
    vocabs = [V - f for f in range(F)]
    tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
    comms = C.comm_create_local(G)
    st = [ShardedTables([shard_table(torch.from_numpy(t), r, G).to(dev) for t in tables], vocabs, r, G,
                        transport="torch", dedup=dedup) for r in range(G)]   # used for arena / vids / rep only
    ids = [np.stack([rng.integers(-1, v + 1, size=B) for v in vocabs], axis=1).astype(np.int32) for _ in range(G)]
    for r in range(G):
        ids[r][B // 2:] = ids[r][:B - B // 2]
        if r % 2 == 0:
            ids[r][0, 0] = -3                                              # make sure some ranks do have a bad id
    stream = torch.cuda.current_stream().cuda_stream
    n = B * F
    plans, wss, vids = [], [], []
    for r in range(G):                                                     # phase 1: plan
        plans.append(C.shard_plan_create(comms[r], n))
        wss.append(torch.empty(C.shard_plan_workspace_bytes(n, G), dtype=torch.uint8, device=dev))
        vids.append(st[r]._vids(torch.from_numpy(ids[r]).to(dev)))
        C.shard_plan_ids(plans[r], vids[r].data_ptr(), n, st[r]._rep.data_ptr() if dedup else 0, wss[r].data_ptr(), stream)
    sizes = [C.shard_plan_finish(plans[r]) for r in range(G)]              # phase 2: host split sizes
    if dedup:
        assert all(nu < n for nu, _ in sizes)
    recv_local = [torch.empty(max(1, nr), dtype=torch.int32, device=dev) for _, nr in sizes]
    served = [torch.empty((max(1, nr), D), dtype=torch.float32, device=dev) for _, nr in sizes]
    rows = [torch.empty((max(1, nu), D), dtype=torch.float32, device=dev) for nu, _ in sizes]
    for r in range(G):                                                     # phase 3: all-to-all #1
        C.shard_exchange_ids(plans[r], recv_local[r].data_ptr(), stream)
    for r in range(G):                                                     # phase 4: owner-side gather
        C.shard_serve_f32(plans[r], st[r].arena.data_ptr(), st[r].arena.shape[0], D, recv_local[r].data_ptr(),
                          served[r].data_ptr(), 0, stream)
    for r in range(G):                                                     # phase 5: all-to-all #2
        C.shard_exchange_rows_f32(plans[r], served[r].data_ptr(), D, rows[r].data_ptr(), 0, stream)
    dys = [rng.normal(size=(B, F * D)).astype(np.float32) for _ in range(G)]
    d_rows, d_served, uidxs = [], [], []
    for r in range(G):                                                     # consumers + backward scatter by uidx
        off = C.shard_plan_uidx(plans[r]) - wss[r].data_ptr()
        uidx = wss[r][off: off + 4 * n].view(torch.int32)
        uidxs.append(uidx)
        nu = sizes[r][0]
        flag = ops.new_oob_flag(dev)
        g = ops.TableGroup([rows[r][:nu]] * F, out_cols=[f * D for f in range(F)])
        out = ops.gather_concat(g, uidx.view(B, F), oob_flag=flag).cpu().numpy()
        exp = ref.gather_concat(tables, ids[r], oob="zero")
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))
        has_oob = bool(((ids[r] < 0) | (ids[r] >= np.asarray(vocabs)[None, :])).any())
        assert int(flag.item()) == int(has_oob)                            # raised on the REQUESTING rank
        dr = torch.zeros((max(1, nu), D), dtype=torch.float32, device=dev)
        ops.embedding_grad(ops.TableGroup([dr[:nu]]), uidx.view(-1, 1), torch.from_numpy(dys[r]).to(dev).view(-1, D))
        d_rows.append(dr)
        d_served.append(torch.empty((max(1, sizes[r][1]), D), dtype=torch.float32, device=dev))
    for r in range(G):                                                     # phase 6: reverse all-to-all
        C.shard_exchange_rows_f32(plans[r], d_rows[r].data_ptr(), D, d_served[r].data_ptr(), 1, stream)
    full = [np.zeros((v, D), np.float64) for v in vocabs]
    for r in range(G):
        gr = ref.embedding_grad(ids[r], dys[r], vocabs, [D] * F)
        for f in range(F):
            full[f] += gr[f]
    for r in range(G):                                                     # owner-side scatter-add
        ga = torch.zeros_like(st[r].arena)
        nr = sizes[r][1]
        if nr:
            ops.embedding_grad(ops.TableGroup([ga]), recv_local[r][:nr].view(-1, 1), d_served[r][:nr])
        ga = ga.cpu().numpy()
        for f, v in enumerate(vocabs):
            mine = ga[f * st[r].rows_local: f * st[r].rows_local + len(range(r, v, G))]
            assert np.allclose(mine, full[f][r::G], rtol=1e-5, atol=1e-5)
    bufs = [torch.full((1000,), float(r + 1), device=dev) for r in range(G)]
    for r in range(G):                                                     # gradient merge of dense parameters
        C.comm_allreduce_sum_f32(comms[r], bufs[r].data_ptr(), 1000, stream)
    torch.cuda.synchronize()
    for r in range(G):
        assert torch.equal(bufs[r], torch.full((1000,), float(G * (G + 1) // 2), device=dev))
        C.shard_plan_destroy(plans[r])
        C.comm_destroy(comms[r])


def test_rccl_world1_executes(dev):
    """A real RCCL communicator (one rank: the only size a one-GPU box offers): unique id, ncclCommInitRank, the
    all-gather of the counts, grouped send/recv to self for both all-to-alls, the all-reduce — through the C ABI."""
    from oracle import ref_numpy as ref
    from recamd import ops
    from recamd._lib import C
    from recamd.dist import Comm
    comm = Comm(0, 1)
    assert C.comm_world(comm.handle) == 1 and C.comm_rank(comm.handle) == 0
    rng = np.random.default_rng(3)
    V, D, n = 300, 64, 1000
    table = rng.normal(size=(V, D)).astype(np.float32)
    vids = rng.integers(-1, V, size=n).astype(np.int32)
    t_tab, t_v = torch.from_numpy(table).to(dev), torch.from_numpy(vids).to(dev)
    rep = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
    plan = C.shard_plan_create(comm.handle, n)
    ws = torch.empty(C.shard_plan_workspace_bytes(n, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    C.shard_plan_ids(plan, t_v.data_ptr(), n, rep.data_ptr(), ws.data_ptr(), stream)
    nu, nr = C.shard_plan_finish(plan)
    assert nu == nr == len(np.unique(vids[vids >= 0]))
    recv_local = torch.empty(nr, dtype=torch.int32, device=dev)
    served = torch.empty((nr, D), dtype=torch.float32, device=dev)
    rows = torch.empty((nu, D), dtype=torch.float32, device=dev)
    C.shard_lookup_f32(plan, t_tab.data_ptr(), V, D, recv_local.data_ptr(), nr, served.data_ptr(), rows.data_ptr(), nu, 0,
                       stream)
    off = C.shard_plan_uidx(plan) - ws.data_ptr()
    uidx = ws[off: off + 4 * n].view(torch.int32)
    out = ops.gather_concat(ops.TableGroup([rows]), uidx.view(-1, 1)).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.embedding_lookup(table, vids, oob="zero").view(np.uint32))
    t = torch.arange(16, dtype=torch.float32, device=dev)
    comm.allreduce_sum_(t)
    assert torch.equal(t.cpu(), torch.arange(16, dtype=torch.float32))
    C.shard_plan_destroy(plan)
    comm.destroy()
