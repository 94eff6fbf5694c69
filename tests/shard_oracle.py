"""numpy restatement of the sharded lookup's device-step contracts (test infrastructure).

`dedup_bucket_np` is the bit-exact specification of rec_shard_dedup_bucket_i32; `OracleKernels` plugs numpy stand-ins
with the HipKernels interface into recamd.dist.ShardedTables for the CPU / gloo tests (the product has no CPU path)."""
import numpy as np
import torch

from oracle import ref_numpy as ref


def dedup_bucket_np(vids, G, dedup=True):
    """-> counts[G], uidx[n], send_local[n_unique], first[n], perm[n]"""
    v = np.asarray(vids, np.int64)
    n = len(v)
    first = np.full(n, -1, np.int64)
    valid = np.nonzero(v >= 0)[0]
    if dedup and len(valid):
        _, idx, inv = np.unique(v[valid], return_index=True, return_inverse=True)
        first[valid] = valid[idx][inv]            # np.unique returns the FIRST occurrence
    else:
        first[valid] = valid
    reps = np.nonzero(first == np.arange(n))[0]   # ascending lookup index
    owner = v[reps] % G
    order = np.argsort(owner, kind="stable")
    perm = np.full(n, -1, np.int64)
    perm[reps[order]] = np.arange(len(reps))
    send_local = (v[reps] // G)[order]
    counts = np.bincount(owner, minlength=G)
    uidx = np.where(first >= 0, perm[np.maximum(first, 0)], -1)
    return (counts.astype(np.int32), uidx.astype(np.int32), send_local.astype(np.int32), first.astype(np.int32),
            perm.astype(np.int32))


class OracleKernels:
    def __init__(self, dedup=True):
        self.dedup = dedup

    def dedup_bucket(self, vids, G, rep):
        counts, uidx, send_local, _, _ = dedup_bucket_np(vids.numpy(), G, self.dedup)
        return torch.from_numpy(counts), torch.from_numpy(uidx), torch.from_numpy(send_local)

    def gather_rows(self, table2d, rows, oob_flag=None):
        return torch.from_numpy(ref.embedding_lookup(table2d.numpy(), rows.numpy(), oob="zero"))

    def scatter_add_rows(self, table2d, rows, dy):
        r = rows.numpy()
        ok = (r >= 0) & (r < table2d.shape[0])
        np.add.at(table2d.numpy(), r[ok], dy.numpy()[ok])

    def _rows_of(self, rows, uidx, oob_flag):
        u = uidx.numpy()
        out = np.zeros((len(u), rows.shape[1]), np.float32)
        ok = u >= 0
        out[ok] = rows.numpy()[u[ok]]
        if oob_flag is not None and not ok.all():
            oob_flag[0] = 1
        return out

    def consume_concat(self, rows, uidx, B, F, out, oob_flag):
        x = torch.from_numpy(self._rows_of(rows, uidx, oob_flag).reshape(B, F * rows.shape[1]))
        if out is None:
            return x
        out.copy_(x)
        return out

    def consume_pairwise_dot(self, rows, uidx, B, F, dense, out, oob_flag):
        D = rows.shape[1]
        X = np.concatenate([self._rows_of(rows, uidx, oob_flag).reshape(B, F, D), dense.numpy()[:, None, :]], axis=1)
        x = torch.from_numpy(np.concatenate([ref.pairwise_dot(X).astype(np.float32), dense.numpy()], axis=1))
        if out is None:
            return x
        out.copy_(x)
        return out
