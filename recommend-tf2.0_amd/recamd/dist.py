"""Row-sharded embedding tables over the GPUs of one node: RCCL all-to-all over xGMI, plus the gradient merge.

The reference's only distribution mechanism is tf.distribute.MirroredStrategy (replicated variables + an NCCL
gradient all-reduce inside fit(), e.g. src/ctr/fm/train.py:43-45); its forward pass has no collective.  North-star
extension: tables are sharded row-wise, cyclically (owner = row % G, local row = row // G, which spreads hot rows),
one process per GPU.  All F tables of a model travel in ONE exchange: field f's ids are shifted by f * Vpad (Vpad a
multiple of G, so the owner is unchanged) into one virtual table whose local shard is this rank's (F, Vpad/G, D) arena.

One lookup (C ABI: include/recamd.h `rec_shard_*`, csrc/shard_exchange.cpp):

    plan      exact de-duplication of the virtual ids + stable bucketing of the unique ones by owner (HIP),
              all-gather of the send counts, counts to pinned host memory behind an event            [prefetch()]
    exchange  all-to-all #1: int32 local rows of the unique ids (~4 B each)  ->  owner-side gather from its shard
              (the K1 kernel)  ->  all-to-all #2: fp32 rows back (D*4 B each)
    consume   lookup i reads row uidx[i] of the returned buffer: the fused gather + pairwise-dot kernel (or the
              gather+concat kernel) runs with that buffer as its table and uidx as its ids — no un-permute pass.

`prefetch(next_ids)` issues the plan for the NEXT batch while the current one runs, so the host-side split sizes
that all-to-all(v) needs are in pinned memory before they are asked for: no host sync on the step's critical path.
With world == 1 every row is local and the consumers read the shard in place (no plan, no copy).

Transports (same algorithm, same results):
  'cabi'   the library's own RCCL communicator (rec_comm): grouped ncclSend/ncclRecv issued from C; torch.distributed
           only broadcasts the 128-byte communicator id.  Default when the process group's backend is nccl.
  'torch'  torch.distributed collectives (all_to_all_single / all_gather_into_tensor) driven from Python.  Default
           otherwise; with `kernels=` a numpy stand-in this is what the CPU/gloo tests run (the product has no CPU path).
  'peers'  test double for ONE process that holds every rank's ShardedTables (`link_peers`): a rank's requests are
           served straight from the owner's arena, which is exactly what the two all-to-alls deliver to the requester.
           Lets a whole model forward run per simulated rank on a one-GPU box.

Backward (training): the gradient of the lookup is a scatter-add of dy by uidx into one row per unique lookup,
the reverse all-to-all to the owners, and the owner's scatter-add into its gradient arena (`backward`); dense
parameters merge with `allreduce_sum_` (MirroredStrategy's all-reduce).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

SKIP = -2 ** 31


class HipKernels:
    """Device steps of the sharded lookup on the HIP kernels."""

    def __init__(self):
        from . import ops
        from ._lib import C
        self.ops, self.C = ops, C

    def dedup_bucket(self, vids: torch.Tensor, G: int, rep: Optional[torch.Tensor]):
        n, dev = vids.numel(), vids.device
        i32 = lambda m: torch.empty(max(1, m), dtype=torch.int32, device=dev)  # noqa: E731
        first, uniq, perm, uidx, send_local, counts = i32(n), i32(n), i32(n), i32(n), i32(n), i32(G)
        ws = torch.empty(max(1, self.C.shard_bucket_workspace_bytes(n, G)), dtype=torch.uint8, device=dev)
        self.C.shard_dedup_bucket_i32(vids.data_ptr(), n, G, 0 if rep is None else rep.data_ptr(), first.data_ptr(),
                                      uniq.data_ptr(), perm.data_ptr(), uidx.data_ptr(), send_local.data_ptr(),
                                      counts.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return counts, uidx[:n], send_local[:n]

    def gather_rows(self, table2d: torch.Tensor, rows: torch.Tensor, oob_flag=None) -> torch.Tensor:
        g = self.ops.TableGroup([table2d])
        return self.ops.gather_concat(g, rows.view(-1, 1), oob_flag=oob_flag)

    def scatter_add_rows(self, table2d: torch.Tensor, rows: torch.Tensor, dy: torch.Tensor) -> None:
        self.ops.embedding_grad(self.ops.TableGroup([table2d]), rows.view(-1, 1), dy)

    def consume_concat(self, rows: torch.Tensor, uidx: torch.Tensor, B: int, F: int, out, oob_flag):
        D = rows.shape[1]
        g = self.ops.TableGroup([rows] * F, out_cols=[f * D for f in range(F)])
        return self.ops.gather_concat(g, uidx.view(B, F), out=out, oob_flag=oob_flag)

    def consume_pairwise_dot(self, rows: torch.Tensor, uidx: torch.Tensor, B: int, F: int, dense, out, oob_flag):
        g = self.ops.TableGroup([rows] * F)
        return self.ops.gather_pairwise_dot(g, uidx.view(B, F), dense, out=out, oob_flag=oob_flag)


def local_rows_of(vocab: int, rank: int, world: int) -> int:
    """Number of rows of a `vocab`-row table owned by `rank` under cyclic sharding."""
    return (vocab + world - 1 - rank) // world


def shard_table(table: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows r of `table` with r % world == rank, in local order (row r -> local r // world)."""
    return table[rank::world].contiguous()


class Comm:
    """The library's own RCCL communicator (rec_comm).  torch.distributed is used once, to hand rank 0's 128-byte
    unique id to the other ranks (any backend)."""

    def __init__(self, rank: int, world: int, group=None, handle: Optional[int] = None):
        from ._lib import C
        self.C, self.rank, self.world = C, rank, world
        if handle is not None:
            self.handle = handle
            return
        if world == 1:
            uid = C.comm_unique_id()
        else:
            box = [C.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = box[0]
        self.handle = C.comm_init_rank(uid, world, rank)

    def allreduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("allreduce_sum_: expected a contiguous fp32 GPU tensor")
        self.C.comm_allreduce_sum_f32(self.handle, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        return t

    def destroy(self):
        if getattr(self, "handle", None):
            self.C.comm_destroy(self.handle)
            self.handle = None


class _Plan:
    """State of one lookup between prefetch() and its consumers."""
    __slots__ = ("key", "vids", "n", "B", "cplan", "ws", "uidx", "send_local", "counts", "matrix_host", "event",
                 "send_splits", "recv_splits", "n_unique", "n_recv", "recv_local", "rows")


class ShardedTables:
    """F same-width tables, row-sharded cyclically over `world` ranks.

    local_tables[f]: this rank's shard of table f, shape (local_rows_of(vocab[f]), D).  If they are consecutive
    views of one (F, rows_local, D) allocation that arena is used in place (`self.arena` aliases it: in-place
    updates of the tables are seen); otherwise they are COPIED into a new arena and `self.tables` (per-field views
    of it) become the source of truth — update those, not the originals."""

    def __init__(self, local_tables: Sequence[torch.Tensor], vocabs: Sequence[int], rank: int, world: int,
                 group=None, kernels=None, transport: Optional[str] = None, dedup: bool = True, comm: Optional[Comm] = None):
        self.rank, self.world, self.group = rank, world, group
        self.F = len(local_tables)
        self.vocabs = [int(v) for v in vocabs]
        self.D = int(local_tables[0].shape[1])
        self.kernels = kernels if kernels is not None else HipKernels()
        vmax = max(self.vocabs)
        self.vpad = (vmax + world - 1) // world * world
        self.rows_local = self.vpad // world
        if self.F * self.vpad >= 2 ** 31:
            raise ValueError("ShardedTables: F * padded vocab must stay below 2^31 (int32 ids)")
        dev = local_tables[0].device
        for f, t in enumerate(local_tables):  # validated on BOTH construction paths
            if t.dim() != 2 or t.shape[1] != self.D:
                raise ValueError("ShardedTables: all tables must be 2-D and share one embed_dim")
            want = local_rows_of(self.vocabs[f], rank, world)
            if t.shape[0] != want and t.shape[0] != self.rows_local:
                raise ValueError(f"table {f}: expected {want} local rows (or the padded {self.rows_local}), got {t.shape[0]}")
        first = local_tables[0]
        need = self.F * self.rows_local * self.D * 4
        st0 = first.untyped_storage()
        contiguous_arena = all(
            t.shape[0] == self.rows_local and t.is_contiguous() and
            t.untyped_storage().data_ptr() == st0.data_ptr() and      # views of ONE allocation ...
            t.data_ptr() == first.data_ptr() + f * self.rows_local * self.D * 4
            for f, t in enumerate(local_tables)) and \
            (first.data_ptr() - st0.data_ptr()) + need <= st0.nbytes()  # ... that really holds F shards
        if contiguous_arena:
            self.arena = torch.as_strided(first, (self.F * self.rows_local, self.D), (self.D, 1))
        else:
            self.arena = torch.zeros((self.F * self.rows_local, self.D), dtype=torch.float32, device=dev)
            for f, t in enumerate(local_tables):
                self.arena[f * self.rows_local: f * self.rows_local + t.shape[0]] = t
        self.aliases_inputs = bool(contiguous_arena)
        self.tables = [self.arena[f * self.rows_local:(f + 1) * self.rows_local] for f in range(self.F)]
        self._shift = (torch.arange(self.F, dtype=torch.int32, device=dev) * self.vpad)[None, :]
        self._vocab_t = torch.tensor(self.vocabs, dtype=torch.int32, device=dev)[None, :]
        if transport is None:
            import os
            transport = os.environ.get("REC_SHARD_TRANSPORT") or None      # 'cabi' | 'torch': overrides the default below
        if transport is None:
            is_nccl = world > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl"
            transport = "cabi" if (is_nccl and kernels is None and dev.type == "cuda") else "torch"
        if transport not in ("cabi", "torch", "peers"):
            raise ValueError("transport must be 'cabi', 'torch' or 'peers'")
        self.peers = None
        self.transport = transport
        self.dedup = bool(dedup)
        self._rep = None
        if self.dedup and world > 1 and dev.type == "cuda":
            self._rep = torch.full((self.F * self.vpad,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
        self.comm = comm
        if transport == "cabi" and world > 1 and self.comm is None:
            self.comm = Comm(rank, world, group)
        self._plans: List[_Plan] = []       # prefetched, not yet consumed
        self._cplans = {}                    # max_ids -> free C plan handles
        self._local_group = None
        self.stats = {"lookups": 0, "ids": 0, "unique_sent": 0, "prefetch_hits": 0}

    def link_peers(self, peers: Sequence["ShardedTables"]) -> None:
        """transport 'peers': the ShardedTables of ALL ranks, in rank order, living in this process"""
        if len(peers) != self.world:
            raise ValueError("link_peers: need one ShardedTables per rank")
        self.peers = list(peers)

    def virtual_ids(self, field: int, ids: torch.Tensor, pad_id: Optional[int] = None) -> torch.Tensor:
        """virtual row ids of `ids` (any shape, int32) looked up in table `field`; out-of-range ids and `pad_id`
        (e.g. SASRec's 0, whose row is multiplied by 0 anyway: src/match/sasrec/model.py:72,82) become -1 = not sent"""
        ok = (ids >= 0) & (ids < self.vocabs[field])
        if pad_id is not None:
            ok = ok & (ids != pad_id)
        return torch.where(ok, ids + field * self.vpad, torch.full_like(ids, -1))

    def lookup_rows(self, vids: torch.Tensor):
        """Generic form: flat virtual ids (virtual_ids(); -1 = skip) -> (rows, uidx): lookup i reads rows[uidx[i]]
        (uidx -1 = zero row).  The consumer kernels take `rows` as their table and `uidx` as their ids."""
        vids = vids.reshape(-1).contiguous()
        if self.world == 1:
            # every row is local: the arena IS the row buffer, local row = virtual row (Vpad = padded vocabulary)
            return self.arena, vids
        p = _Plan()
        p.key, p.B, p.vids, p.n = None, 0, vids, vids.numel()
        p.recv_local = p.rows = p.send_splits = None
        self._plan_into(p)
        rows = self._exchange(p)
        uidx = p.uidx
        self._release(p)
        return rows, uidx

    # ---- ids -> virtual rows -----------------------------------------------------------------------------
    def _vids(self, ids: torch.Tensor) -> torch.Tensor:
        if ids.dtype == torch.int64:  # range-check BEFORE narrowing: ids >= 2^31 must not wrap onto valid rows
            ok = (ids >= 0) & (ids < self._vocab_t.to(torch.int64))
            ids = torch.where(ok, ids, torch.full_like(ids, -1)).to(torch.int32)
        if ids.dtype != torch.int32:
            raise TypeError("ShardedTables: ids must be int32 (or int64, range-checked and narrowed)")
        ok = (ids >= 0) & (ids < self._vocab_t)
        return torch.where(ok, ids + self._shift, torch.full_like(ids, -1)).reshape(-1).contiguous()

    # ---- plan --------------------------------------------------------------------------------------------
    def _key(self, ids: torch.Tensor):
        return (ids.data_ptr(), tuple(ids.shape), ids._version)

    def prefetch(self, ids: torch.Tensor) -> None:
        """Issue the plan (de-duplication, bucketing, count exchange) for a batch that will be looked up later.
        Optional: lookup() plans on the spot if the batch was not prefetched."""
        if self.world == 1:
            return
        self._plans.append(self._plan(ids))

    def _plan(self, ids: torch.Tensor) -> _Plan:
        B, F = ids.shape
        if F != self.F:
            raise ValueError(f"ids has {F} columns, the sharded model has {self.F} tables")
        p = _Plan()
        p.key, p.B = self._key(ids), B
        p.vids = self._vids(ids)
        p.n = p.vids.numel()
        p.recv_local = p.rows = p.send_splits = None
        self._plan_into(p)
        return p

    def _plan_into(self, p: _Plan) -> None:
        G = self.world
        if self.transport == "cabi":
            C = self.comm.C
            free = self._cplans.setdefault(p.n, [])
            p.cplan = free.pop() if free else C.shard_plan_create(self.comm.handle, p.n)
            p.ws = torch.empty(max(1, C.shard_plan_workspace_bytes(p.n, G)), dtype=torch.uint8, device=p.vids.device)
            C.shard_plan_ids(p.cplan, p.vids.data_ptr(), p.n, 0 if self._rep is None else self._rep.data_ptr(),
                             p.ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        elif self.transport == "peers":
            p.cplan = None
            p.counts, p.uidx, p.send_local = self.kernels.dedup_bucket(p.vids, G, self._rep)
            p.matrix_host, p.event = None, None
        else:
            p.cplan = None
            p.counts, p.uidx, p.send_local = self.kernels.dedup_bucket(p.vids, G, self._rep)
            matrix = torch.empty(G * G, dtype=torch.int32, device=p.vids.device)
            dist.all_gather_into_tensor(matrix, p.counts, group=self.group)
            if matrix.is_cuda:
                p.matrix_host = torch.empty(G * G, dtype=torch.int32, pin_memory=True)
                p.matrix_host.copy_(matrix, non_blocking=True)
                p.event = torch.cuda.Event()
                p.event.record()
            else:
                p.matrix_host, p.event = matrix, None

    def _take_plan(self, ids: torch.Tensor) -> _Plan:
        key = self._key(ids)
        for i, p in enumerate(self._plans):
            if p.key == key:
                self.stats["prefetch_hits"] += 1
                return self._plans.pop(i)
        return self._plan(ids)

    def _finish(self, p: _Plan) -> None:
        G, me = self.world, self.rank
        if self.transport == "cabi":
            p.n_unique, p.n_recv = self.comm.C.shard_plan_finish(p.cplan)
        elif self.transport == "peers":
            p.send_splits = [int(c) for c in p.counts.tolist()]
            p.recv_splits = None
            p.n_unique, p.n_recv = sum(p.send_splits), 0
        else:
            if p.event is not None:
                p.event.synchronize()          # already complete when the plan was prefetched a step ahead
            m = p.matrix_host.view(G, G).tolist()
            p.send_splits = [int(m[me][q]) for q in range(G)]
            p.recv_splits = [int(m[q][me]) for q in range(G)]
            p.n_unique, p.n_recv = sum(p.send_splits), sum(p.recv_splits)
        self.stats["lookups"] += 1
        self.stats["ids"] += p.n
        self.stats["unique_sent"] += p.n_unique

    def _exchange(self, p: _Plan, oob_flag=None) -> torch.Tensor:
        """both all-to-alls and the owner-side gather; returns the (n_unique, D) rows in send order"""
        self._finish(p)
        dev, D = p.vids.device, self.D
        p.recv_local = torch.empty(max(1, p.n_recv), dtype=torch.int32, device=dev)[:p.n_recv]
        p.rows = torch.empty((max(1, p.n_unique), D), dtype=torch.float32, device=dev)[:p.n_unique]
        if self.transport == "cabi":
            C = self.comm.C
            served = torch.empty((max(1, p.n_recv), D), dtype=torch.float32, device=dev)
            C.shard_lookup_f32(p.cplan, self.arena.data_ptr(), self.arena.shape[0], D, p.recv_local.data_ptr(), p.n_recv,
                               served.data_ptr(), p.rows.data_ptr(), p.n_unique, 0 if oob_flag is None else oob_flag.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
            uidx_ptr = C.shard_plan_uidx(p.cplan)
            off = uidx_ptr - p.ws.data_ptr()
            p.uidx = p.ws[off: off + 4 * p.n].view(torch.int32)
        elif self.transport == "peers":
            if self.peers is None:
                raise RuntimeError("transport 'peers': call link_peers() first")
            off = 0
            for o, c in enumerate(p.send_splits):   # what owner o's gather + all-to-all #2 would hand back
                if c:
                    p.rows[off:off + c] = self.kernels.gather_rows(self.peers[o].arena, p.send_local[off:off + c].contiguous())
                off += c
        else:
            send = p.send_local[:p.n_unique].contiguous()
            dist.all_to_all_single(p.recv_local, send, p.recv_splits, p.send_splits, group=self.group)
            served = self.kernels.gather_rows(self.arena, p.recv_local, oob_flag) if p.n_recv else \
                torch.empty((0, D), dtype=torch.float32, device=dev)
            dist.all_to_all_single(p.rows, served, p.send_splits, p.recv_splits, group=self.group)
        return p.rows

    def _release(self, p: _Plan) -> None:
        if p.cplan is not None:
            self._cplans.setdefault(p.n, []).append(p.cplan)
            p.cplan = None

    # ---- forward -----------------------------------------------------------------------------------------
    def _local_table_group(self):
        if self._local_group is None:
            from . import ops
            # world == 1: every row is local; the descriptors carry the REAL vocabularies, so range checks match
            self._local_group = ops.TableGroup([self.tables[f][:self.vocabs[f]] for f in range(self.F)])
        return self._local_group

    def lookup(self, ids: torch.Tensor, out: Optional[torch.Tensor] = None, oob_flag=None, keep_plan: bool = False):
        """ids (B, F) int32 (global row ids) -> (B, F*D), identical to the unsharded gather+concat.  Out-of-range
        ids read as zero rows and raise `oob_flag` on THIS (the requesting) rank."""
        B, F = ids.shape
        if self.world == 1 and isinstance(self.kernels, HipKernels):
            return self.kernels.ops.gather_concat(self._local_table_group(), ids, out=out, oob_flag=oob_flag)
        p = self._take_plan(ids)
        rows = self._exchange(p)
        out = self.kernels.consume_concat(rows, p.uidx, B, F, out, oob_flag)
        if keep_plan:
            return out, p
        self._release(p)
        return out

    def lookup_pairwise_dot(self, ids: torch.Tensor, dense: torch.Tensor, out: Optional[torch.Tensor] = None,
                            oob_flag=None):
        """The DLRM sparse stage on sharded tables: (B, P + D) = [pairwise dots of the F rows + dense, dense]."""
        B, F = ids.shape
        if self.world == 1 and isinstance(self.kernels, HipKernels):
            return self.kernels.ops.gather_pairwise_dot(self._local_table_group(), ids, dense, out=out, oob_flag=oob_flag)
        p = self._take_plan(ids)
        rows = self._exchange(p)
        out = self.kernels.consume_pairwise_dot(rows, p.uidx, B, F, dense, out, oob_flag)
        self._release(p)
        return out

    # ---- backward ----------------------------------------------------------------------------------------
    def backward(self, plan: _Plan, dy: torch.Tensor, grad_arena: torch.Tensor) -> None:
        """Gradient of lookup(): dy (B, F*D) -> grad_arena (F*rows_local, D) += the rows this rank OWNS, summed over
        all ranks' lookups.  `plan` is the one lookup(..., keep_plan=True) returned."""
        D = self.D
        d_rows = torch.zeros((max(1, plan.n_unique), D), dtype=torch.float32, device=dy.device)[:plan.n_unique]
        if plan.n_unique:
            self.kernels.scatter_add_rows(d_rows, plan.uidx, dy.reshape(-1, D))     # duplicates + this batch's rows
        d_served = torch.empty((max(1, plan.n_recv), D), dtype=torch.float32, device=dy.device)[:plan.n_recv]
        if self.transport == "cabi":
            self.comm.C.shard_exchange_rows_f32(plan.cplan, d_rows.data_ptr(), D, d_served.data_ptr(), 1,
                                                torch.cuda.current_stream().cuda_stream)
        else:
            dist.all_to_all_single(d_served, d_rows, plan.recv_splits, plan.send_splits, group=self.group)
        if plan.n_recv:
            self.kernels.scatter_add_rows(grad_arena, plan.recv_local, d_served)
        self._release(plan)

    def allreduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        """in-place sum over ranks of a dense-parameter gradient (MirroredStrategy's all-reduce)"""
        if self.world == 1:
            return t
        if self.transport == "cabi":
            return self.comm.allreduce_sum_(t)
        dist.all_reduce(t, group=self.group)
        return t

    # ---- reporting ---------------------------------------------------------------------------------------
    def exchange_bytes(self, B: int) -> dict:
        """Expected xGMI traffic per rank and lookup for uniformly distributed ids."""
        n = B * self.F
        remote = n * (self.world - 1) / self.world
        return {"ids_out": remote * 4, "rows_in": remote * self.D * 4}

    def describe(self) -> dict:
        s = dict(self.stats)
        s.update(transport=self.transport, dedup=self.dedup, world=self.world,
                 unique_fraction=round(s["unique_sent"] / s["ids"], 4) if s["ids"] else None)
        return s
