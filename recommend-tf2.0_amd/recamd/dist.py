"""Row-sharded embedding tables over the GPUs of one node: RCCL all-to-all over xGMI.

The reference's only distribution mechanism is tf.distribute.MirroredStrategy (replicated
variables, e.g. src/ctr/fm/train.py:43); its forward pass has no collective.  North-star
extension: tables too large to replicate are sharded row-wise, cyclically (owner = row % G, local
row = row // G, which spreads hot rows), one process per GPU, and a lookup is

    bucket ids by owner (HIP, stable)  ->  all-to-all #1: int32 local rows   (~4 B per lookup)
    -> local gather on the owner (HIP, the K1 kernel)
    ->  all-to-all #2: fp32 rows back  (D*4 B per lookup)  ->  un-permute (HIP)

All F tables of a model travel in ONE exchange: field f's ids are shifted by f * Vpad (Vpad a
multiple of G, so the owner is unchanged) into one virtual table whose local shard is the
(F, Vpad/G, D) arena of this rank.  The result is bit-identical to the single-device gather.

`torch.distributed` (backend "nccl" == RCCL on ROCm) is the transport; the three device steps come
from a `kernels` object — `HipKernels` (recamd.ops) in the product; the CPU/gloo tests inject a
numpy-oracle stand-in, the product has no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class HipKernels:
    """Device steps of the sharded lookup on the HIP kernels."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def bucket(self, vids: torch.Tensor, G: int):
        return self.ops.shard_bucket(vids, G)

    def gather(self, arena2d: torch.Tensor, local_rows: torch.Tensor, oob_flag=None) -> torch.Tensor:
        g = self.ops.TableGroup([arena2d])
        return self.ops.gather_concat(g, local_rows.view(-1, 1), oob_flag=oob_flag)

    def unpermute(self, rows: torch.Tensor, perm: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        return self.ops.unpermute_rows(rows, perm, out=out)


def local_rows_of(vocab: int, rank: int, world: int) -> int:
    """Number of rows of a `vocab`-row table owned by `rank` under cyclic sharding."""
    return (vocab + world - 1 - rank) // world


def shard_table(table: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows r of `table` with r % world == rank, in local order (row r -> local r // world)."""
    return table[rank::world].contiguous()


class ShardedTables:
    """F same-width tables, row-sharded cyclically over `world` ranks.

    local_tables[f]: this rank's shard of table f, shape (local_rows_of(vocab[f]), D)."""

    def __init__(self, local_tables: Sequence[torch.Tensor], vocabs: Sequence[int], rank: int, world: int,
                 group=None, kernels=None):
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
        # one arena (F, rows_local, D): virtual local row = f * rows_local + local
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
                if t.shape[1] != self.D:
                    raise ValueError("ShardedTables: all tables must share one embed_dim")
                if t.shape[0] != local_rows_of(self.vocabs[f], rank, world):
                    raise ValueError(f"table {f}: expected {local_rows_of(self.vocabs[f], rank, world)} local rows")
                self.arena[f * self.rows_local: f * self.rows_local + t.shape[0]] = t
        self._shift = (torch.arange(self.F, dtype=torch.int32, device=dev) * self.vpad)[None, :]
        self._vocab_t = torch.tensor(self.vocabs, dtype=torch.int32, device=dev)[None, :]

    def lookup(self, ids: torch.Tensor, out: Optional[torch.Tensor] = None, oob_flag=None) -> torch.Tensor:
        """ids (B, F) int32 (global row ids) -> (B, F*D), identical to the unsharded gather+concat."""
        B, F = ids.shape
        assert F == self.F and ids.dtype == torch.int32
        G = self.world
        ok = (ids >= 0) & (ids < self._vocab_t)
        vids = torch.where(ok, ids + self._shift, torch.full_like(ids, -1)).reshape(-1).contiguous()
        n = vids.numel()
        counts, perm, send_local = self.kernels.bucket(vids, G)
        if G == 1:
            rows = self.kernels.gather(self.arena, send_local, oob_flag)
        else:
            recv_counts = torch.empty_like(counts)
            dist.all_to_all_single(recv_counts, counts, group=self.group)
            send_splits = counts.tolist()      # host sync: all_to_all(v) needs host-side split sizes
            recv_splits = recv_counts.tolist()
            n_recv = int(sum(recv_splits))
            recv_local = torch.empty(n_recv, dtype=torch.int32, device=ids.device)
            dist.all_to_all_single(recv_local, send_local, recv_splits, send_splits, group=self.group)
            served = self.kernels.gather(self.arena, recv_local, oob_flag)          # (n_recv, D)
            rows = torch.empty((n, self.D), dtype=torch.float32, device=ids.device)
            dist.all_to_all_single(rows, served, send_splits, recv_splits, group=self.group)
        if out is None:
            out = torch.empty((B, F * self.D), dtype=torch.float32, device=ids.device)
        self.kernels.unpermute(rows, perm, out.view(n, self.D))
        return out

    def exchange_bytes(self, B: int) -> dict:
        """Expected xGMI traffic per rank and lookup for uniformly distributed ids."""
        n = B * self.F
        remote = n * (self.world - 1) / self.world
        return {"ids_out": remote * 4, "rows_in": remote * self.D * 4}
