"""Torch-tensor front end of the C ABI (include/recamd.h).

Every function validates shapes/dtypes/devices on the host, then enqueues exactly one (or two)
HIP kernels on torch's current stream through the pybind11 shim.  Nothing here computes on the
CPU; tensors that are not on a GPU raise.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from ._lib import C

IDS_I32, IDS_F32 = 0, 1
ACT = {None: 0, "none": 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "prelu": 4}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _ids_dtype(ids: torch.Tensor) -> int:
    if ids.dtype == torch.int32:
        return IDS_I32
    if ids.dtype == torch.float32:
        return IDS_F32
    raise TypeError(f"ids: expected int32 or float32 (Keras Embedding cast), got {ids.dtype}")


def _rows2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


class TableGroup:
    """Per-field embedding tables of one model (the reference's `embed_layers` dict).

    Holds the descriptor tuples the C ABI takes; `out_cols` default to the tf.concat offsets."""

    def __init__(self, tables: Sequence[torch.Tensor], out_cols: Optional[Sequence[int]] = None):
        self.tables = [_chk(t, f"tables[{i}]") for i, t in enumerate(tables)]
        for i, t in enumerate(self.tables):
            if t.dim() != 2 or not t.is_contiguous():
                raise ValueError(f"tables[{i}]: expected contiguous (vocab, dim)")
        self.dims = [int(t.shape[1]) for t in self.tables]
        if out_cols is None:
            out_cols, c = [], 0
            for d in self.dims:
                out_cols.append(c)
                c += d
        self.out_cols = [int(c) for c in out_cols]
        self.width = max(c + d for c, d in zip(self.out_cols, self.dims))
        self.descs = [(t.data_ptr(), int(t.shape[0]), int(t.shape[1]), c)
                      for t, c in zip(self.tables, self.out_cols)]

    def __len__(self):
        return len(self.tables)


def new_oob_flag(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def gather_concat(group: TableGroup, ids: torch.Tensor, out: Optional[torch.Tensor] = None,
                  oob_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b, out_col_f : out_col_f + D_f] = tables[f][ids[b, f]]   (rec_gather_concat_f32).

    ids: (B, F) int32 or float32 (truncated).  `out` may be a wider pre-allocated (B, W) buffer
    (fused concat with other features).  Out-of-range ids give zero rows and set oob_flag."""
    ids = _rows2d(_chk(ids, "ids", None), "ids")
    F = len(group)
    if ids.shape[1] != F:
        raise ValueError(f"ids has {ids.shape[1]} columns, model has {F} tables")
    B = ids.shape[0]
    if out is None:
        out = torch.empty((B, group.width), dtype=torch.float32, device=ids.device)
    else:
        _rows2d(_chk(out, "out"), "out")
        if out.shape[0] != B or out.shape[1] < group.width:
            raise ValueError("out: wrong shape")
    for lo in range(0, F, C.MAX_TABLES):  # wider models: one launch per 64 fields
        hi = min(F, lo + C.MAX_TABLES)
        sub = ids[:, lo:hi]
        C.gather_concat_f32(group.descs[lo:hi], sub.data_ptr(), _ids_dtype(ids), ids.stride(0), B,
                            out.data_ptr(), out.stride(0), _ptr(oob_flag), _stream())
    return out


def pairwise_dot(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DLRM dot interaction: x (B, n, D) -> (B, n(n-1)/2), pairs (i,j), i>j, row-major."""
    _chk(x, "x")
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("x: expected contiguous (B, n, D)")
    B, n, D = x.shape
    P = n * (n - 1) // 2
    if out is None:
        out = torch.empty((B, P), dtype=torch.float32, device=x.device)
    else:
        _rows2d(_chk(out, "out"), "out")
    C.pairwise_dot_f32(x.data_ptr(), B, n, D, out.data_ptr(), out.stride(0), _stream())
    return out


def gather_pairwise_dot(group: TableGroup, ids: torch.Tensor, dense: Optional[torch.Tensor] = None,
                        append_dense: bool = True, out: Optional[torch.Tensor] = None,
                        oob_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused K1+K5: gathers the F embedding rows of each sample (never materialised), appends
    `dense` (B, D) as vector F, and writes the strictly-lower-triangle dots [+ dense]."""
    ids = _rows2d(_chk(ids, "ids", None), "ids")
    F = len(group)
    if ids.shape[1] != F:
        raise ValueError(f"ids has {ids.shape[1]} columns, model has {F} tables")
    if len(set(group.dims)) != 1:
        raise ValueError("gather_pairwise_dot: all tables must share one embed_dim")
    D = group.dims[0]
    B = ids.shape[0]
    n = F + (1 if dense is not None else 0)
    P = n * (n - 1) // 2
    if dense is not None:
        _rows2d(_chk(dense, "dense"), "dense")
        if dense.shape != (B, D):
            raise ValueError(f"dense: expected {(B, D)}, got {tuple(dense.shape)}")
    width = P + (D if (dense is not None and append_dense) else 0)
    if out is None:
        out = torch.empty((B, width), dtype=torch.float32, device=ids.device)
    else:
        _rows2d(_chk(out, "out"), "out")
        if out.shape[0] != B or out.shape[1] < width:
            raise ValueError("out: wrong shape")
    C.gather_pairwise_dot_f32(group.descs, ids.data_ptr(), _ids_dtype(ids), ids.stride(0),
                              _ptr(dense), dense.stride(0) if dense is not None else 0, B,
                              out.data_ptr(), out.stride(0),
                              1 if (dense is not None and append_dense) else 0,
                              _ptr(oob_flag), _stream())
    return out
