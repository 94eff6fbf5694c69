"""Row-sharded embedding tables over the GPUs of one node: RCCL all-to-all over xGMI, plus the gradient merge.

The reference's only distribution mechanism is tf.distribute.MirroredStrategy (replicated variables + an NCCL
gradient all-reduce inside fit(), e.g. src/ctr/fm/train.py:43-45); its forward pass has no collective.  North-star
extension: tables are sharded row-wise, cyclically (owner = row % G, local row = row // G, which spreads hot rows),
one process per GPU.  All F tables of a model travel in ONE exchange: field f's ids are shifted by f * Vpad (Vpad a
multiple of G, so the owner is unchanged) into one virtual table whose local shard is this rank's (F, Vpad/G, D) arena.

ONE ROW SPACE.  The consumer kernels (fused gather + pairwise dot, gather + concat, SASRec's attention / dot scores)
read rows through a table descriptor + an index.  Everything a rank can read lives in one allocation

    space = [ this rank's shard (F * Vpad/G rows) | hot-row replica cache (2 regions) | S receive slots ]

so a single descriptor covers it and a lookup's index says where its row is:
    local    rows this rank owns are read IN PLACE (they never enter the exchange),
    cached   replicas of the hottest remote rows (exact copies, refilled by refresh_cache()),
    remote   rows returned by the exchange, in the receive slot of that lookup.

One lookup (C ABI: include/recamd.h `rec_shard_*`, csrc/shard_exchange.cpp, csrc/shard.hip):

    plan      resolve (classification above + exact de-duplication of the remote ids + stable bucketing of the
              unique ones by owner), all-gather of the send counts, counts to pinned host memory behind an event
    exchange  all-to-all #1: int32 local rows of the unique remote ids (~4 B each)  ->  owner-side gather from its
              shard (the K1 kernel)  ->  all-to-all #2: fp32 rows back (D*4 B each) into the lookup's receive slot
    consume   lookup i reads row uidx[i] of the row space — no un-permute pass, no concat copy.

PIPELINE.  plan and exchange run on a communication stream; events order it against the caller's (compute) stream:
`prefetch(ids, rows=True)` moves the ids AND rows of batch i+1 while batch i's consumer kernel runs, so a step costs
max(exchange, compute) instead of their sum; `prefetch(ids)` issues only the plan (its count matrix is in pinned
memory before the host asks for it: no host wait on the step).  The receive slots (default 3: one being consumed,
one being filled, one planned) make that safe; a slot is reused only after everything enqueued on the compute stream
at that moment has finished.  With world == 1 every row is local and the consumers read the shard in place.

Transports (same algorithm, same results):
  'cabi'   the library's own RCCL communicator (rec_comm): grouped ncclSend/ncclRecv issued from C; torch.distributed
           only broadcasts the 128-byte communicator id.  Default when the process group's backend is nccl.
  'torch'  torch.distributed collectives (all_to_all_single / all_gather_into_tensor) driven from Python: any backend.

Backward (training): the gradient of the lookup is a scatter-add of dy by uidx — local lookups straight into the
gradient arena, remote ones into one row per unique lookup, the reverse all-to-all to the owners, and the owner's
scatter-add (`backward`); dense parameters merge with `allreduce_sum_` (MirroredStrategy's all-reduce).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

INT32_MAX = 2 ** 31 - 1


def local_rows_of(vocab: int, rank: int, world: int) -> int:
    """Number of rows of a `vocab`-row table owned by `rank` under cyclic sharding."""
    return (vocab + world - 1 - rank) // world


def shard_table(table: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows r of `table` with r % world == rank, in local order (row r -> local r // world)."""
    return table[rank::world].contiguous()


def _space_rows(arena_rows: int, cache_rows: int, nslots: int, slot_cap: int) -> int:
    rows = arena_rows + 2 * cache_rows + nslots * slot_cap
    if rows >= 2 ** 31 - 2:
        raise ValueError("ShardedTables: the row space (shard + cache + receive slots) must stay below 2^31 rows")
    return rows


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

    def describe(self) -> dict:
        """what actually runs underneath: the rank count the communicator itself reports, and its kind"""
        return {"rccl_ranks": int(self.C.comm_world(self.handle)), "comm": self.C.comm_transport_name(self.handle)}

    def allreduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("allreduce_sum_: expected a contiguous fp32 GPU tensor")
        self.C.comm_allreduce_sum_f32(self.handle, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        return t

    def destroy(self):
        if getattr(self, "handle", None):
            self.C.comm_destroy(self.handle)
            self.handle = None


class _Slot:
    """One receive slot of the row space + the per-lookup buffers that live as long as the slot's lookup."""
    __slots__ = ("index", "base", "cplan", "ws", "busy")


class _Plan:
    """State of one lookup between prefetch() and its consumers."""
    __slots__ = ("ids", "version", "vids", "n", "B", "slot", "uidx", "send_local", "counts", "matrix_host", "event",
                 "send_splits", "recv_splits", "n_unique", "n_recv", "recv_local", "exchanged", "rows_ready",
                 "used_cache", "recv_base")


class ShardedTables:
    """F same-width tables, row-sharded cyclically over `world` ranks.

    local_tables[f]: this rank's shard of table f, shape (local_rows_of(vocab[f]), D).  With world == 1 and the
    tables being consecutive views of one (F, rows_local, D) allocation that arena is used in place; otherwise the
    shards are COPIED into the row space and `self.tables` (per-field views of it) become the source of truth —
    update those, not the originals.  `ShardedTables.empty(...)` allocates the row space directly (no copy): fill
    `st.arena` / `st.tables[f]` afterwards.

    max_ids      lookups per call (B * F) the receive slots are sized for; None = sized by the first lookup (the row
                 space is then rebuilt once, and `arena` / `tables` are re-pointed: take views after that, or pass it)
    slots        receive slots = lookups that may be in flight (consumed / exchanged / planned)
    bypass_local rows this rank owns are read in place instead of travelling to itself through the exchange
    cache_rows   replicas of the hottest remote rows kept on this rank (0 = off); `refresh_cache()` refills them from
                 the running per-row lookup counts — a COLLECTIVE, every rank must call it at the same point; with
                 cache_refresh_every = R it is called automatically before every R-th planned lookup (R >= slots).
                 Replicas are exact copies; `recamd.ops.note_weights_written` (every optimiser step calls it)
                 invalidates them."""

    def __init__(self, local_tables: Sequence[torch.Tensor], vocabs: Sequence[int], rank: int, world: int,
                 group=None, transport: Optional[str] = None, dedup: bool = True, comm: Optional[Comm] = None,
                 max_ids: Optional[int] = None, slots: int = 3, bypass_local: bool = True, cache_rows: int = 0,
                 cache_refresh_every: int = 0, _space: Optional[torch.Tensor] = None):
        self.rank, self.world, self.group = int(rank), int(world), group
        self.F = len(local_tables)
        self.vocabs = [int(v) for v in vocabs]
        self.D = int(local_tables[0].shape[1])
        vmax = max(self.vocabs)
        self.vpad = (vmax + world - 1) // world * world
        self.rows_local = self.vpad // world
        self.arena_rows = self.F * self.rows_local
        if self.F * self.vpad >= 2 ** 31:
            raise ValueError("ShardedTables: F * padded vocab must stay below 2^31 (int32 ids)")
        dev = local_tables[0].device
        self.device = dev
        for f, t in enumerate(local_tables):  # validated on every construction path
            if t.dim() != 2 or t.shape[1] != self.D:
                raise ValueError("ShardedTables: all tables must be 2-D and share one embed_dim")
            want = local_rows_of(self.vocabs[f], rank, world)
            if t.shape[0] != want and t.shape[0] != self.rows_local:
                raise ValueError(f"table {f}: expected {want} local rows (or the padded {self.rows_local}), got {t.shape[0]}")
        self.nslots = max(1, int(slots)) if world > 1 else 0
        self.cache_rows = int(cache_rows) if world > 1 else 0
        self.cache_refresh_every = int(cache_refresh_every)
        if self.cache_rows and self.cache_refresh_every and self.cache_refresh_every < self.nslots:
            raise ValueError("cache_refresh_every must be >= slots: a replica region is rewritten two refreshes later, "
                             "lookups planned before a refresh still read it")
        self.bypass_local = bool(bypass_local)
        self.dedup = bool(dedup)
        self.slot_cap = 0
        first = local_tables[0]
        if _space is not None:                                   # empty(): the row space exists already
            self.space = _space
            self.slot_cap = int(max_ids) if world > 1 else 0
            self.aliases_inputs = True
        else:
            need = self.arena_rows * self.D * 4
            st0 = first.untyped_storage()
            contiguous_arena = world == 1 and all(
                t.shape[0] == self.rows_local and t.is_contiguous() and
                t.untyped_storage().data_ptr() == st0.data_ptr() and      # views of ONE allocation ...
                t.data_ptr() == first.data_ptr() + f * self.rows_local * self.D * 4
                for f, t in enumerate(local_tables)) and \
                (first.data_ptr() - st0.data_ptr()) + need <= st0.nbytes()  # ... that really holds F shards
            if contiguous_arena:
                self.space = torch.as_strided(first, (self.arena_rows, self.D), (self.D, 1))
            else:
                self.slot_cap = int(max_ids) if (max_ids and world > 1) else 0
                self.space = torch.zeros((self._space_rows(self.slot_cap), self.D), dtype=torch.float32, device=dev)
                for f, t in enumerate(local_tables):
                    self.space[f * self.rows_local: f * self.rows_local + t.shape[0]] = t
            self.aliases_inputs = bool(contiguous_arena)
        self.on_rebuild = []       # callables run after the row space was re-allocated (holders of table views re-point)
        self._point_views()
        self._shift = (torch.arange(self.F, dtype=torch.int32, device=dev) * self.vpad)[None, :]
        self._vocab_t = torch.tensor(self.vocabs, dtype=torch.int32, device=dev)[None, :]
        if transport is None:
            is_nccl = world > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl"
            transport = "cabi" if (is_nccl and dev.type == "cuda") else "torch"
        if transport not in self._transports():
            raise ValueError(f"transport must be one of {self._transports()}")
        self.transport = transport
        self._rep = self._cache_slot = self._hot = self._stat = None
        if world > 1:
            nv = self.F * self.vpad
            if self.dedup:
                self._rep = torch.full((nv,), INT32_MAX, dtype=torch.int32, device=dev)
            if self.cache_rows:
                self._cache_slot = torch.full((nv + 1,), -1, dtype=torch.int32, device=dev)   # [nv] = scratch for "no row"
                self._hot = torch.zeros((nv,), dtype=torch.int32, device=dev)
            if dev.type == "cuda":
                self._stat = torch.zeros(2, dtype=torch.int64, device=dev)
        self._cached_vids: Optional[torch.Tensor] = None      # virtual ids of the live replicas (-1 = unused entry)
        self._cache_region = 0
        self._cache_gen = None
        self.comm = comm
        if transport == "cabi" and world > 1 and self.comm is None:
            self.comm = Comm(rank, world, group)
        self._comm_stream = torch.cuda.Stream(device=dev) if (world > 1 and dev.type == "cuda") else None
        self._slots: List[_Slot] = []
        self._plans: List[_Plan] = []       # prefetched, not yet consumed
        self._served = self._recv_local = None
        self._planned = 0
        self._local_group = None
        self._space_group = None
        self.stats = {"lookups": 0, "ids": 0, "unique_sent": 0, "prefetch_hits": 0, "rows_prefetched": 0,
                      "cache_refreshes": 0}
        if self.slot_cap:
            self._make_slots()

    @classmethod
    def empty(cls, F: int, vocabs: Sequence[int], D: int, rank: int, world: int, device, max_ids: int, **kw):
        """Allocate the row space directly (shard + cache + receive slots in one allocation, nothing copied); the
        caller fills `st.arena` (F * rows_local, D) or `st.tables[f]`."""
        vpad = (max(int(v) for v in vocabs) + world - 1) // world * world
        rows_local = vpad // world
        rows = _space_rows(F * rows_local, int(kw.get("cache_rows", 0)) if world > 1 else 0,
                           max(1, int(kw.get("slots", 3))) if world > 1 else 0, int(max_ids) if world > 1 else 0)
        space = torch.empty((rows, D), dtype=torch.float32, device=device)
        views = [space[f * rows_local: f * rows_local + local_rows_of(int(vocabs[f]), rank, world)] for f in range(F)]
        return cls(views, vocabs, rank, world, max_ids=max_ids, _space=space, **kw)

    # ---- the row space -----------------------------------------------------------------------------------
    def _transports(self):
        return ("cabi", "torch")

    def _space_rows(self, slot_cap: int) -> int:
        return _space_rows(self.arena_rows, self.cache_rows, self.nslots, slot_cap)

    def _point_views(self):
        self.arena = self.space[:self.arena_rows]
        self.tables = [self.arena[f * self.rows_local:(f + 1) * self.rows_local] for f in range(self.F)]
        self.cache_base = self.arena_rows
        self._local_group = self._space_group = None

    def _slot_base(self, s: int) -> int:
        return self.arena_rows + 2 * self.cache_rows + s * self.slot_cap

    def _ensure_slots(self, n: int) -> None:
        """receive slots for lookups of n ids; sized by the first lookup when max_ids was not given (rebuilds the
        row space once: shard and replicas are copied, `arena` / `tables` re-pointed)"""
        if self.world == 1 or n <= self.slot_cap:
            return
        if any(s.busy for s in self._slots):
            raise RuntimeError(f"ShardedTables: a lookup of {n} ids exceeds the receive slots ({self.slot_cap}) while "
                               "lookups are in flight; pass max_ids= at construction")
        self._sync_streams()
        old = self.space
        self.slot_cap = int(n)
        self.space = torch.empty((self._space_rows(self.slot_cap), self.D), dtype=torch.float32, device=self.device)
        keep = self.arena_rows + 2 * self.cache_rows
        self.space[:keep] = old[:keep]
        self._point_views()
        self._make_slots()
        for fn in self.on_rebuild:
            fn(self)

    def _make_slots(self) -> None:
        for s in self._slots:
            if s.cplan is not None:
                self.comm.C.shard_plan_destroy(s.cplan)
        self._slots = []
        for i in range(self.nslots):
            s = _Slot()
            s.index, s.base, s.busy, s.cplan, s.ws = i, self._slot_base(i), False, None, None
            if self.transport == "cabi":
                C = self.comm.C
                s.cplan = C.shard_plan_create(self.comm.handle, self.slot_cap)
                s.ws = torch.empty(max(1, C.shard_plan_workspace_bytes(self.slot_cap, self.world)), dtype=torch.uint8,
                                   device=self.device)
            self._slots.append(s)

    # ---- streams: all no-ops on CPU tensors (gloo tests) ----------------------------------------------------
    @contextlib.contextmanager
    def _on_comm(self):
        """run the enclosed calls on the communication stream, AFTER everything the caller's stream holds right now
        (the ids of the batch; the consumers of a receive slot or serve buffer about to be reused)"""
        if self._comm_stream is None:
            yield
            return
        self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._comm_stream):
            yield

    def _stream_handle(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _sync_streams(self) -> None:
        if self._comm_stream is not None:
            self._comm_stream.synchronize()
            torch.cuda.current_stream(self.device).synchronize()

    # ---- device steps (HIP; tests/shard_oracle.py overrides them with numpy stand-ins on the CPU) ----------------
    def _dev_resolve(self, vids: torch.Tensor, recv_base: int, use_cache: bool):
        """-> counts[G], uidx[n], send_local[n]  (csrc/shard.hip rec_shard_resolve_i32)"""
        from ._lib import C
        n, dev, G = vids.numel(), vids.device, self.world
        i32 = lambda m: torch.empty(max(1, m), dtype=torch.int32, device=dev)  # noqa: E731
        first, uniq, perm, uidx, send_local, counts = i32(n), i32(n), i32(n), i32(n), i32(n), i32(G)
        ws = torch.empty(max(1, C.shard_bucket_workspace_bytes(n, G)), dtype=torch.uint8, device=dev)
        cs, hot = (self._cache_slot, self._hot) if (use_cache and self.cache_rows) else (None, self._hot)
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        C.shard_resolve_i32(vids.data_ptr(), n, G, self.rank if self.bypass_local else -1, ptr(self._rep), ptr(cs), ptr(hot),
                            self.cache_base + self._cache_region * self.cache_rows, recv_base, ptr(self._stat),
                            first.data_ptr(), uniq.data_ptr(), perm.data_ptr(), uidx.data_ptr(), send_local.data_ptr(),
                            counts.data_ptr(), ws.data_ptr(), self._stream_handle())
        return counts, uidx[:n], send_local[:n]

    def _dev_gather_rows(self, table2d: torch.Tensor, rows: torch.Tensor, out=None, oob_flag=None) -> torch.Tensor:
        from . import ops
        return ops.gather_concat(ops.TableGroup([table2d]), rows.view(-1, 1), out=out, oob_flag=oob_flag)

    def _dev_scatter_add_rows(self, table2d: torch.Tensor, rows: torch.Tensor, dy: torch.Tensor) -> None:
        from . import ops
        ops.embedding_grad(ops.TableGroup([table2d]), rows.view(-1, 1), dy)

    def _space_table_group(self):
        from . import ops
        if self._space_group is None:
            D = self.D
            self._space_group = (ops.TableGroup([self.space] * self.F, out_cols=[f * D for f in range(self.F)]),
                                 ops.TableGroup([self.space] * self.F))
        return self._space_group

    def _dev_consume_concat(self, uidx: torch.Tensor, B: int, out, oob_flag):
        from . import ops
        return ops.gather_concat(self._space_table_group()[0], uidx.view(B, self.F), out=out, oob_flag=oob_flag)

    def _dev_consume_pairwise_dot(self, uidx: torch.Tensor, B: int, dense, out, oob_flag):
        from . import ops
        return ops.gather_pairwise_dot(self._space_table_group()[1], uidx.view(B, self.F), dense, out=out, oob_flag=oob_flag)

    def _weight_generation(self) -> int:
        from . import ops
        return ops.weight_generation()

    # ---- ids -> virtual rows -----------------------------------------------------------------------------
    def virtual_ids(self, field: int, ids: torch.Tensor, pad_id: Optional[int] = None) -> torch.Tensor:
        """virtual row ids of `ids` (any shape, int32) looked up in table `field`; out-of-range ids and `pad_id`
        (e.g. SASRec's 0, whose row is multiplied by 0 anyway: src/match/sasrec/model.py:72,82) become -1 = not sent"""
        ok = (ids >= 0) & (ids < self.vocabs[field])
        if pad_id is not None:
            ok = ok & (ids != pad_id)
        return torch.where(ok, ids + field * self.vpad, torch.full_like(ids, -1))

    def _vids(self, ids: torch.Tensor) -> torch.Tensor:
        if ids.dtype == torch.int64:  # range-check BEFORE narrowing: ids >= 2^31 must not wrap onto valid rows
            ok = (ids >= 0) & (ids < self._vocab_t.to(torch.int64))
            ids = torch.where(ok, ids, torch.full_like(ids, -1)).to(torch.int32)
        if ids.dtype != torch.int32:
            raise TypeError("ShardedTables: ids must be int32 (or int64, range-checked and narrowed)")
        ok = (ids >= 0) & (ids < self._vocab_t)
        return torch.where(ok, ids + self._shift, torch.full_like(ids, -1)).reshape(-1).contiguous()

    # ---- plan --------------------------------------------------------------------------------------------
    def prefetch(self, ids: torch.Tensor, rows: bool = False) -> None:
        """Start the lookup of a batch that will be asked for later, on the communication stream.
        rows=False: the plan (resolve, de-duplication, bucketing, count exchange) — the host-side split sizes are then
        in pinned memory before lookup() needs them.  rows=True: the plan (unless already prefetched) AND both
        all-to-alls: ids and rows of this batch travel while the caller's stream computes the previous one.
        Optional: lookup() does whatever has not been done yet.  Every rank must prefetch the same batches in the same
        order (the plan and the exchange are collectives).  Forward only: rows prefetched before an optimiser step
        are the pre-step rows."""
        if self.world == 1:
            return
        p = self._find_plan(ids)
        if p is None:
            p = self._plan(ids)
            self._plans.append(p)
        if rows and not p.exchanged:
            self._exchange(p)
            self.stats["rows_prefetched"] += 1

    def _find_plan(self, ids: torch.Tensor) -> Optional[_Plan]:
        for p in self._plans:   # identity + in-place version: a recycled address can never match a stale plan
            if p.ids is ids and ids._version == p.version:
                return p
        return None

    def _plan(self, ids: torch.Tensor, use_cache: bool = True) -> _Plan:
        B, F = ids.shape
        if F != self.F:
            raise ValueError(f"ids has {F} columns, the sharded model has {self.F} tables")
        with self._on_comm():
            vids = self._vids(ids)
        return self._plan_vids(vids, ids, B, use_cache)

    def _plan_vids(self, vids: torch.Tensor, ids, B: int, use_cache: bool = True) -> _Plan:
        n = vids.numel()
        self._ensure_slots(n)
        if self.cache_rows:
            gen = self._weight_generation()
            if self._cache_gen is not None and gen != self._cache_gen:
                self.invalidate_cache()
            if use_cache and self.cache_refresh_every and self._planned and self._planned % self.cache_refresh_every == 0:
                self.refresh_cache()
        self._planned += 1
        p = _Plan()
        p.ids, p.version, p.B, p.vids, p.n = ids, (ids._version if ids is not None else 0), B, vids, n
        p.recv_local = p.send_splits = p.rows_ready = None
        p.exchanged = False
        p.used_cache = bool(use_cache and self.cache_rows and self._cached_vids is not None)
        p.slot = self._acquire_slot()
        p.recv_base = p.slot.base
        with self._on_comm():                  # the slot's previous consumers and the ids precede the plan
            self._plan_into(p, use_cache)
        return p

    def _acquire_slot(self) -> _Slot:
        for s in self._slots:
            if not s.busy:
                s.busy = True
                return s
        raise RuntimeError(f"ShardedTables: more than {self.nslots} lookups in flight (prefetched or kept for backward); "
                           "raise slots= or consume / release them")

    def _plan_into(self, p: _Plan, use_cache: bool) -> None:
        G = self.world
        if self.transport == "cabi":
            C = self.comm.C
            ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
            cs = self._cache_slot if (use_cache and self.cache_rows) else None
            C.shard_plan_ids_ex(p.slot.cplan, p.vids.data_ptr(), p.n, ptr(self._rep), 1 if self.bypass_local else 0,
                                ptr(cs), ptr(self._hot), self.cache_base + self._cache_region * self.cache_rows,
                                p.recv_base, ptr(self._stat), p.slot.ws.data_ptr(), self._stream_handle())
            uidx_ptr = C.shard_plan_uidx(p.slot.cplan)
            off = uidx_ptr - p.slot.ws.data_ptr()
            p.uidx = p.slot.ws[off: off + 4 * p.n].view(torch.int32)
            p.matrix_host = p.event = None
        else:
            p.counts, p.uidx, p.send_local = self._dev_resolve(p.vids, p.recv_base, use_cache)
            matrix = torch.empty(G * G, dtype=torch.int32, device=p.vids.device)
            dist.all_gather_into_tensor(matrix, p.counts, group=self.group)
            if matrix.is_cuda:
                p.matrix_host = torch.empty(G * G, dtype=torch.int32, pin_memory=True)
                p.matrix_host.copy_(matrix, non_blocking=True)
                p.event = torch.cuda.Event()
                p.event.record()
            else:
                p.matrix_host, p.event = matrix, None

    def _take_plan(self, ids: torch.Tensor, use_cache: bool = True) -> _Plan:
        p = self._find_plan(ids)
        if p is not None:
            self._plans.remove(p)
            self.stats["prefetch_hits"] += 1
            if p.used_cache and not use_cache:
                raise RuntimeError("ShardedTables: this batch was prefetched against the replica cache; a lookup kept for "
                                   "backward (keep_plan=True) must not be (its gradient rows go to the owners)")
            return p
        return self._plan(ids, use_cache)

    def _finish(self, p: _Plan) -> None:
        G, me = self.world, self.rank
        if self.transport == "cabi":
            p.n_unique, p.n_recv = self.comm.C.shard_plan_finish(p.slot.cplan)
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

    def _serve_buffers(self, n_recv: int):
        """(recv_local, served): the rows this rank serves in one exchange; exchanges are serialised on the
        communication stream, so one pair (grown on demand) serves all slots"""
        if self._served is None or self._served.shape[0] < n_recv:
            cap = max(1, int(n_recv * 1.25))
            self._recv_local = torch.empty(cap, dtype=torch.int32, device=self.device)
            self._served = torch.empty((cap, self.D), dtype=torch.float32, device=self.device)
        return self._recv_local, self._served

    def _exchange(self, p: _Plan, oob_flag=None) -> None:
        """both all-to-alls and the owner-side gather, on the communication stream; the rows land in the plan's
        receive slot of the row space"""
        self._finish(p)
        if p.n_unique > self.slot_cap:
            raise RuntimeError("ShardedTables: internal error: more unique rows than the receive slot holds")
        D = self.D
        with self._on_comm():
            recv_local, served = self._serve_buffers(p.n_recv)
            p.recv_local = recv_local[:p.n_recv]
            rows = self.space[p.recv_base: p.recv_base + max(1, p.n_unique)]
            if self.transport == "cabi":
                C = self.comm.C
                C.shard_lookup_f32(p.slot.cplan, self.arena.data_ptr(), self.arena_rows, D, recv_local.data_ptr(),
                                   recv_local.shape[0], served.data_ptr(), rows.data_ptr(), self.slot_cap,
                                   0 if oob_flag is None else oob_flag.data_ptr(), self._stream_handle())
            else:
                send = p.send_local[:p.n_unique].contiguous()
                dist.all_to_all_single(p.recv_local, send, p.recv_splits, p.send_splits, group=self.group)
                if p.n_recv:
                    self._dev_gather_rows(self.arena, p.recv_local, out=served[:p.n_recv], oob_flag=oob_flag)
                dist.all_to_all_single(rows[:p.n_unique], served[:p.n_recv], p.send_splits, p.recv_splits, group=self.group)
            if self._comm_stream is not None:
                p.rows_ready = torch.cuda.Event()
                p.rows_ready.record()
        p.exchanged = True

    def _ready(self, p: _Plan) -> None:
        """exchange if not done yet, then make the caller's stream wait for the rows (and uidx)"""
        if not p.exchanged:
            self._exchange(p)
        if p.rows_ready is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(p.rows_ready)
            if p.uidx.untyped_storage().data_ptr() != (p.slot.ws.untyped_storage().data_ptr() if p.slot.ws is not None else 0):
                p.uidx.record_stream(cur)      # allocated on the communication stream, read by the caller's consumers

    def _release(self, p: _Plan) -> None:
        """the slot may be re-acquired; its next plan waits (on the communication stream) for everything the caller's
        stream holds at THAT moment, which includes the consumers of this lookup"""
        if p.slot is not None:
            p.slot.busy = False
            p.slot = None

    # ---- forward -----------------------------------------------------------------------------------------
    def _local_table_group(self):
        if self._local_group is None:
            from . import ops
            # world == 1: every row is local; the descriptors carry the REAL vocabularies, so range checks match
            self._local_group = ops.TableGroup([self.tables[f][:self.vocabs[f]] for f in range(self.F)])
        return self._local_group

    def lookup_rows(self, vids: torch.Tensor, keep_plan: bool = False):
        """Generic form: flat virtual ids (virtual_ids(); -1 = skip) -> (rows, uidx): lookup i reads rows[uidx[i]]
        (uidx -1 = zero row).  The consumer kernels take `rows` (the row space) as their table and `uidx` as their ids;
        the returned index stays valid until `slots` further lookups have been planned.  keep_plan=True returns
        (rows, uidx, plan) for backward(plan, dy (n_lookups, D), grad_arena); the plan holds its receive slot until then
        (world == 1: plan is None, the gradient of lookup i belongs to row uidx[i] of the arena)."""
        vids = vids.reshape(-1).contiguous()
        if self.world == 1:
            # every row is local: the arena IS the row space, local row = virtual row (Vpad = padded vocabulary)
            return (self.arena, vids, None) if keep_plan else (self.arena, vids)
        p = self._plan_vids(vids, None, 0, use_cache=not keep_plan)
        self._ready(p)
        uidx = p.uidx if self.transport != "cabi" else p.uidx.clone()    # the slot's workspace is reused; the index is small
        if keep_plan:
            p.uidx = uidx
            p.recv_local = None if p.recv_local is None else p.recv_local.clone()     # the shared serve buffer is rewritten by the next exchange
            return self.space, uidx, p
        self._release(p)
        return self.space, uidx

    def lookup(self, ids: torch.Tensor, out: Optional[torch.Tensor] = None, oob_flag=None, keep_plan: bool = False):
        """ids (B, F) int32 (global row ids) -> (B, F*D), identical to the unsharded gather+concat.  Out-of-range
        ids read as zero rows and raise `oob_flag` on THIS (the requesting) rank.  keep_plan=True returns
        (out, plan) for backward(): the plan holds its receive slot until then."""
        B, F = ids.shape
        if self.world == 1:
            from . import ops
            return ops.gather_concat(self._local_table_group(), ids, out=out, oob_flag=oob_flag)
        p = self._take_plan(ids, use_cache=not keep_plan)
        self._ready(p)
        out = self._dev_consume_concat(p.uidx, B, out, oob_flag)
        if keep_plan:
            p.recv_local = None if p.recv_local is None else p.recv_local.clone()     # the shared serve buffer is rewritten by the next exchange
            return out, p
        self._release(p)
        return out

    def lookup_pairwise_dot(self, ids: torch.Tensor, dense: torch.Tensor, out: Optional[torch.Tensor] = None,
                            oob_flag=None):
        """The DLRM sparse stage on sharded tables: (B, P + D) = [pairwise dots of the F rows + dense, dense]."""
        B, F = ids.shape
        if self.world == 1:
            from . import ops
            return ops.gather_pairwise_dot(self._local_table_group(), ids, dense, out=out, oob_flag=oob_flag)
        p = self._take_plan(ids)
        self._ready(p)
        out = self._dev_consume_pairwise_dot(p.uidx, B, dense, out, oob_flag)
        self._release(p)
        return out

    # ---- hot-row replica cache -------------------------------------------------------------------------------
    def refresh_cache(self) -> None:
        """COLLECTIVE.  Replicate this rank's `cache_rows` hottest remote rows (running lookup counts) into the
        inactive replica region, through the ordinary exchange; then point the per-row map at them.
        Lookups planned before the refresh keep reading the other region (rewritten only by the refresh after next)."""
        if self.world == 1 or not self.cache_rows:
            return
        K = min(self.cache_rows, self._hot.numel())
        nv = self._hot.numel()
        with self._on_comm():
            vals, vsel = torch.topk(self._hot, K)
            vsel = torch.where(vals > 0, vsel.to(torch.int32), torch.full((K,), -1, dtype=torch.int32, device=self.device))
        p = self._plan_vids(vsel, None, 0, use_cache=False)       # straight from the owners
        self._planned -= 1                                        # not a user lookup
        self._ready(p)                                            # (the caller's stream waits too: harmless)
        region = 1 - self._cache_region
        with self._on_comm():
            dst = self.space[self.cache_base + region * self.cache_rows: self.cache_base + region * self.cache_rows + K]
            self._dev_gather_rows(self.space, p.uidx[:K], out=dst)
            scratch = torch.full((1,), nv, dtype=torch.int64, device=self.device)
            if self._cached_vids is not None:                      # retire the previous replicas
                old = self._cached_vids.to(torch.int64)
                self._cache_slot.index_fill_(0, torch.where(old >= 0, old, scratch), -1)
            v64 = vsel.to(torch.int64)
            self._cache_slot.index_copy_(0, torch.where(v64 >= 0, v64, scratch),
                                         torch.arange(K, dtype=torch.int32, device=self.device))
            self._cache_slot[nv] = -1
            self._hot.clamp_(max=2 ** 30)          # running counts, kept clear of int32 overflow
        self._cached_vids = vsel
        self._cache_region = region
        self._cache_gen = self._weight_generation()
        self.stats["cache_refreshes"] += 1
        self.stats["lookups"] -= 1
        self.stats["ids"] -= p.n
        self.stats["unique_sent"] -= p.n_unique
        self._release(p)

    def invalidate_cache(self) -> None:
        """forget the replicas (the tables were written): lookups planned from now on fetch from the owners"""
        if self._cached_vids is not None:
            with self._on_comm():
                old = self._cached_vids.to(torch.int64)
                scratch = torch.full((1,), self._hot.numel(), dtype=torch.int64, device=self.device)
                self._cache_slot.index_fill_(0, torch.where(old >= 0, old, scratch), -1)
            self._cached_vids = None
        self._cache_gen = None

    # ---- backward ----------------------------------------------------------------------------------------
    def backward(self, plan: _Plan, dy: torch.Tensor, grad_arena: torch.Tensor) -> None:
        """Gradient of lookup(): dy (B, F*D) -> grad_arena (F*rows_local, D) += the rows this rank OWNS, summed over
        all ranks' lookups.  `plan` is the one lookup(..., keep_plan=True) returned."""
        D = self.D
        dyr = dy.reshape(-1, D)
        if plan is None:                       # world == 1 (lookup_rows): every lookup is a row of the arena
            raise ValueError("backward: world == 1 has no plan; scatter-add dy by the returned index into the gradient arena")
        uidx = plan.uidx
        neg = torch.full_like(uidx, -1)
        # lookups answered from this rank's shard: straight into the gradient arena
        self._dev_scatter_add_rows(grad_arena, torch.where((uidx >= 0) & (uidx < self.arena_rows), uidx, neg), dyr)
        # remote lookups: one gradient row per unique row, the reverse all-to-all, the owner's scatter-add
        d_rows = torch.zeros((max(1, plan.n_unique), D), dtype=torch.float32, device=dy.device)[:plan.n_unique]
        if plan.n_unique:
            self._dev_scatter_add_rows(d_rows, torch.where(uidx >= plan.recv_base, uidx - plan.recv_base, neg), dyr)
        d_served = torch.empty((max(1, plan.n_recv), D), dtype=torch.float32, device=dy.device)[:plan.n_recv]
        if self.transport == "cabi":
            self.comm.C.shard_exchange_rows_f32(plan.slot.cplan, d_rows.data_ptr(), D, d_served.data_ptr(), 1,
                                                self._stream_handle())
        else:
            dist.all_to_all_single(d_served, d_rows, plan.recv_splits, plan.send_splits, group=self.group)
        if plan.n_recv:
            self._dev_scatter_add_rows(grad_arena, plan.recv_local, d_served)
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
        """what ran: transport, the rank count the communicator itself reports, and where the lookups were answered"""
        s = dict(self.stats)
        s.update(transport=self.transport, dedup=self.dedup, world=self.world, bypass_local=self.bypass_local,
                 pipelined=self._comm_stream is not None, slots=self.nslots, cache_rows=self.cache_rows,
                 unique_fraction=round(s["unique_sent"] / s["ids"], 4) if s["ids"] else None)
        if self.comm is not None:
            s.update(self.comm.describe())
        else:
            s.update(rccl_ranks=0, comm=("torch.distributed/" + dist.get_backend(self.group)) if
                     (self.world > 1 and dist.is_initialized()) else "none")
        if self._stat is not None:
            loc, cached = (int(x) for x in self._stat.tolist())
            s.update(local_lookups=loc, cache_hits=cached)
        return s
