"""numpy restatement of the sharded lookup's device-step contracts (test infrastructure).

`resolve_np` is the bit-exact specification of rec_shard_resolve_i32 (`dedup_bucket_np` = its form without a local
shard, a cache or row-space offsets = rec_shard_dedup_bucket_i32).  `OracleShardedTables` is recamd.dist.ShardedTables
with its device steps replaced by numpy stand-ins, for the CPU / gloo tests (the product has no CPU path; the
transport logic, the row space, the pipeline bookkeeping and the cache policy under test are the product's).
`PeersShardedTables` (GPU tests) keeps the HIP device steps and replaces the TRANSPORT: one process holds every rank's
object and a rank's requests are served straight from the owner's shard — exactly what the two all-to-alls deliver —
so a whole model forward can run per simulated rank on a one-GPU box."""
import numpy as np
import torch

from oracle import ref_numpy as ref
from recamd.dist import ShardedTables


def resolve_np(vids, G, me=-1, dedup=True, cache_slot=None, cache_base=0, recv_base=0, hot_count=None):
    """-> counts[G], uidx[n], send_local[n_unique], first[n], perm[n]; hot_count (if given) is updated in place"""
    v = np.asarray(vids, np.int64)
    n = len(v)
    first = np.full(n, -1, np.int64)
    valid = v >= 0
    local = valid & ((v % G) == me) if me >= 0 else np.zeros(n, bool)
    first[local] = -2 - v[local] // G
    remote = valid & ~local
    if hot_count is not None:
        np.add.at(hot_count, v[remote], 1)
    cached = np.zeros(n, bool)
    if cache_slot is not None:
        cs = np.asarray(cache_slot)[np.where(remote, v, 0)]
        cached = remote & (cs >= 0)
        first[cached] = -2 - (cache_base + cs[cached])
    sent = np.nonzero(remote & ~cached)[0]
    if dedup and len(sent):
        _, idx, inv = np.unique(v[sent], return_index=True, return_inverse=True)
        first[sent] = sent[idx][inv]              # np.unique returns the FIRST occurrence
    else:
        first[sent] = sent
    reps = np.nonzero(first == np.arange(n))[0]   # ascending lookup index
    owner = v[reps] % G
    order = np.argsort(owner, kind="stable")
    perm = np.full(n, -1, np.int64)
    perm[reps[order]] = np.arange(len(reps))
    send_local = (v[reps] // G)[order]
    counts = np.bincount(owner, minlength=G)
    uidx = np.where(first >= 0, recv_base + perm[np.maximum(first, 0)], np.where(first == -1, -1, -2 - first))
    return (counts.astype(np.int32), uidx.astype(np.int32), send_local.astype(np.int32), first.astype(np.int32),
            perm.astype(np.int32))


def dedup_bucket_np(vids, G, dedup=True):
    return resolve_np(vids, G, -1, dedup)


class OracleShardedTables(ShardedTables):
    """ShardedTables on CPU tensors: the device steps in numpy."""
    generation = 0          # tests bump this to simulate recamd.ops.note_weights_written

    def _weight_generation(self):
        return OracleShardedTables.generation

    def _dev_resolve(self, vids, recv_base, use_cache):
        cs = self._cache_slot.numpy() if (use_cache and self.cache_rows) else None
        hot = self._hot.numpy() if self._hot is not None else None      # shares memory: updated in place
        counts, uidx, send_local, _, _ = resolve_np(
            vids.numpy(), self.world, self.rank if self.bypass_local else -1, self.dedup, cs,
            self.cache_base + self._cache_region * self.cache_rows, recv_base, hot)
        self.np_stats = getattr(self, "np_stats", {"local": 0, "cached": 0})
        u = uidx.astype(np.int64)
        self.np_stats["local"] += int(((u >= 0) & (u < self.arena_rows)).sum())
        self.np_stats["cached"] += int(((u >= self.cache_base) & (u < self.cache_base + 2 * self.cache_rows)).sum())
        return torch.from_numpy(counts), torch.from_numpy(uidx), torch.from_numpy(send_local)

    def _dev_gather_rows(self, table2d, rows, out=None, oob_flag=None):
        x = torch.from_numpy(ref.embedding_lookup(table2d.numpy(), rows.numpy(), oob="zero"))
        if out is None:
            return x
        out.copy_(x)
        return out

    def _dev_scatter_add_rows(self, table2d, rows, dy):
        r = rows.numpy()
        ok = (r >= 0) & (r < table2d.shape[0])
        np.add.at(table2d.numpy(), r[ok], dy.numpy()[ok])

    def _rows_of(self, uidx, oob_flag):
        u = uidx.numpy()
        out = np.zeros((len(u), self.D), np.float32)
        ok = u >= 0
        out[ok] = self.space.numpy()[u[ok]]
        if oob_flag is not None and not ok.all():
            oob_flag[0] = 1
        return out

    def _dev_consume_concat(self, uidx, B, out, oob_flag):
        x = torch.from_numpy(self._rows_of(uidx, oob_flag).reshape(B, self.F * self.D))
        if out is None:
            return x
        out.copy_(x)
        return out

    def _dev_consume_pairwise_dot(self, uidx, B, dense, out, oob_flag):
        D = self.D
        X = np.concatenate([self._rows_of(uidx, oob_flag).reshape(B, self.F, D), dense.numpy()[:, None, :]], axis=1)
        x = torch.from_numpy(np.concatenate([ref.pairwise_dot(X).astype(np.float32), dense.numpy()], axis=1))
        if out is None:
            return x
        out.copy_(x)
        return out


class PeersShardedTables(ShardedTables):
    """transport 'peers': the ShardedTables of ALL ranks live in this process (link_peers)."""

    def __init__(self, local_tables, vocabs, rank, world, **kw):
        kw.setdefault("transport", "peers" if world > 1 else None)
        super().__init__(local_tables, vocabs, rank, world, **kw)
        self.peers = None

    def _transports(self):
        return ("cabi", "torch", "peers")

    def link_peers(self, peers):
        if len(peers) != self.world:
            raise ValueError("link_peers: need one ShardedTables per rank")
        self.peers = list(peers)

    def _plan_into(self, p, use_cache):
        p.counts, p.uidx, p.send_local = self._dev_resolve(p.vids, p.recv_base, use_cache)
        p.matrix_host = p.event = None

    def _finish(self, p):
        p.send_splits = [int(c) for c in p.counts.tolist()]
        p.recv_splits = None
        p.n_unique, p.n_recv = sum(p.send_splits), 0
        self.stats["lookups"] += 1
        self.stats["ids"] += p.n
        self.stats["unique_sent"] += p.n_unique

    def _exchange(self, p, oob_flag=None):
        if self.peers is None:
            raise RuntimeError("transport 'peers': call link_peers() first")
        self._finish(p)
        with self._on_comm():
            off = 0
            for o, c in enumerate(p.send_splits):   # what owner o's gather + all-to-all #2 would hand back
                if c:
                    self._dev_gather_rows(self.peers[o].arena, p.send_local[off:off + c].contiguous(),
                                          out=self.space[p.recv_base + off: p.recv_base + off + c])
                off += c
            if self._comm_stream is not None:
                p.rows_ready = torch.cuda.Event()
                p.rows_ready.record()
        p.exchanged = True

    def backward(self, plan, dy, grad_arena):
        """the reverse all-to-all, in process: the gradient rows of the unique remote lookups are scatter-added straight
        into the owners' gradient arenas (`peer_grad_arena`, set by the test for every rank before the first backward)"""
        D = self.D
        dyr = dy.reshape(-1, D)
        uidx = plan.uidx
        neg = torch.full_like(uidx, -1)
        self._dev_scatter_add_rows(grad_arena, torch.where((uidx >= 0) & (uidx < self.arena_rows), uidx, neg), dyr)
        if plan.n_unique:
            d_rows = torch.zeros((plan.n_unique, D), dtype=torch.float32, device=dy.device)
            self._dev_scatter_add_rows(d_rows, torch.where(uidx >= plan.recv_base, uidx - plan.recv_base, neg), dyr)
            off = 0
            for o, c in enumerate(plan.send_splits):
                if c:
                    self._dev_scatter_add_rows(self.peers[o].peer_grad_arena, plan.send_local[off:off + c].contiguous(),
                                               d_rows[off:off + c])
                off += c
        self._release(plan)
