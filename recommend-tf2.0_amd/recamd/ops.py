"""Torch-tensor front end of the C ABI (include/recamd.h).

Every function validates shapes/dtypes/devices on the host, then enqueues exactly one (or two)
HIP kernels on torch's current stream through the pybind11 shim.  Nothing here computes on the
CPU; tensors that are not on a GPU raise.
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Sequence

import os

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
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
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


def place_table_arena(F: int, V: int, D: int, device, candidates: int = 4, probe=None, probe_launches: int = 12,
                      probe_name: Optional[str] = None, first: Optional[torch.Tensor] = None):
    """(F, V, D) fp32 arena for F embedding tables, PLACED BY MEASUREMENT.

    The rate at which random rows of a multi-GB arena can be read depends on which physical memory the allocation
    received — same kernel, same ids, same buffers: 307-329 us for the BASELINE configs[1] gather and 165-175 us for the
    fused gather + pairwise dot on one box, stable per allocation over time, different again after a free +
    re-allocation at the same virtual address, and NOT the same ranking for the two kernels (tools/exp/arena_lottery.py,
    fused_lottery.py; profiles/r02c_*lottery*.txt).  Result, dense and id buffers do not matter (<= 1 %).  Tables are
    allocated once and read for the life of the model, and 288 GB of HBM leave room to choose: `candidates` arenas are
    allocated side by side (all alive, so they are distinct memory), `probe(group, i)` — one launch of the kernel the
    tables will serve, i = launch counter; or a list of such probes, whose normalised times are added — is timed on
    each, the fastest is kept and the rest are freed.  Default probe:
    the materialised gather over uniform ids.  Returns (arena, info) with the probe time of every candidate — callers
    report them (bench.py does).  candidates <= 1: one plain allocation, no probe.  `first`: an existing (F, V, D)
    arena that takes part as candidate 0 (its contents are kept).

    Why allocations differ is only partly understood (profiles/r03_placement_probe.txt): the allocation call does not
    matter (hipMalloc, contiguous, VMM chunks of 2 MiB .. 1 GiB show the same per-arena spread), the per-arena UTCL1
    miss / multi-miss counters do — the page-table fragments the driver could build for the physical blocks it had.
    There is no user-space control over that; this function measures instead, and bench.py keeps it out of its
    headline (a plain allocation) and reports the placed result beside it."""
    dev = torch.device(device)
    if candidates <= 1:
        return (first if first is not None else torch.empty((F, V, D), dtype=torch.float32, device=dev)), {"candidates": 1}
    if probe is None:
        gen = torch.Generator(device=dev).manual_seed(0)
        Bp = 65536
        pids = [torch.randint(0, V, (Bp, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(4)]
        pout = torch.empty((Bp, F * D), dtype=torch.float32, device=dev)
        probe = lambda g, i: gather_concat(g, pids[i % 4], out=pout)  # noqa: E731
        probe_name = probe_name or "rec_gather_concat_f32, %d x %d uniform ids" % (Bp, F)
    arenas = [] if first is None else [first]
    for _ in range(candidates - len(arenas)):
        try:
            a = torch.empty((F, V, D), dtype=torch.float32, device=dev)
        except torch.OutOfMemoryError:
            break
        a.zero_()                                   # touch every page: the probe must see the final mapping
        arenas.append(a)
    a = None
    groups = [TableGroup([t[f] for f in range(F)]) for t in arenas]
    probes = list(probe) if isinstance(probe, (list, tuple)) else [probe]     # several kernels: the sum of their
    for i in range(max(100, 8 * probe_launches)):   # clocks up before anything is compared   # normalised times decides
        probes[0](groups[0], i)                     # (the first ~100 ms after idle run up to 25 % slower)
    times = [[float("inf")] * len(arenas) for _ in probes]
    for _rep in range(2):                           # two interleaved passes, the faster one counts
        for k, g in enumerate(groups):
            for pi, pr in enumerate(probes):
                for i in range(2):
                    pr(g, i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(probe_launches):
                    pr(g, i)
                e1.record()
                e1.synchronize()
                times[pi][k] = min(times[pi][k], e0.elapsed_time(e1) / probe_launches * 1e3)
    g = None
    del groups
    score = [sum(t[k] / min(t) for t in times) for k in range(len(arenas))]
    best = min(range(len(arenas)), key=lambda k: score[k])
    arena = arenas[best]
    del arenas
    torch.cuda.empty_cache()
    rounded = [[round(v, 1) for v in t] for t in times]
    return arena, {"candidates": len(score), "probe": probe_name or "caller's kernel",
                   "probe_us": rounded[0] if len(rounded) == 1 else rounded, "chosen": best}


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
        out = torch.empty((B, (P + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :P]
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
        # row stride padded to a multiple of 4 floats: the kernel then stages each sample's results in
        # LDS and writes aligned 16-B vectors (the pad column holds zeros); consumers take the stride
        out = torch.empty((B, (width + 3) // 4 * 4), dtype=torch.float32, device=ids.device)[:, :width]
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


def _act_id(act) -> int:
    if act not in ACT:
        raise ValueError(f"unsupported activation {act!r} (the reference's default 'prelu' string is not a "
                         "valid Keras activation either; pass act='prelu' together with an alpha tensor)")
    return ACT[act]


_fm_ws = {}


def _fm_workspace(device, B: int) -> torch.Tensor:
    need = C.fm_layer_workspace_floats(B)
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _fm_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _fm_ws[key] = torch.empty(need, dtype=torch.float32, device=device)
    return ws


def fm_layer(first: torch.Tensor, second: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """FM layer of DeepFM (src/ctr/layers/modules.py:57-72): first (B,L1), second (B,M), w (L1[,1])
    -> (B,1).  The first-order term is ONE scalar summed over the whole batch (modules.py:65)."""
    _rows2d(_chk(first, "first"), "first")
    _rows2d(_chk(second, "second"), "second")
    w = _chk(w, "w").reshape(-1)
    B, L1 = first.shape
    if second.shape[0] != B or w.numel() != L1 or not w.is_contiguous():
        raise ValueError("fm_layer: inconsistent shapes")
    out = torch.empty((B, 1), dtype=torch.float32, device=first.device)
    ws = _fm_workspace(first.device, B)
    C.fm_layer_f32(first.data_ptr(), first.stride(0), L1, w.data_ptr(), second.data_ptr(), second.stride(0),
                   second.shape[1], B, out.data_ptr(), ws.data_ptr(), _stream())
    return out


def cross_network(x: torch.Tensor, W: torch.Tensor, Bv: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DCN CrossNetwork (src/ctr/layers/modules.py:105-112): x (B,dim); W, Bv (L,dim)."""
    _rows2d(_chk(x, "x"), "x")
    _chk(W, "W")
    _chk(Bv, "Bv")
    B, dim = x.shape
    L = W.shape[0]
    if W.shape != (L, dim) or Bv.shape != (L, dim) or not W.is_contiguous() or not Bv.is_contiguous():
        raise ValueError("cross_network: W and Bv must be contiguous (L, dim)")
    if out is None:
        out = torch.empty((B, dim), dtype=torch.float32, device=x.device)
    else:
        _rows2d(_chk(out, "out"), "out")
        if out.shape != (B, dim):
            raise ValueError("cross_network: out must be (B, dim)")
    C.cross_f32(x.data_ptr(), x.stride(0), dim, W.data_ptr(), Bv.data_ptr(), L, B, out.data_ptr(), out.stride(0),
                _stream())
    return out


def fm_onehot(dense: torch.Tensor, ids: torch.Tensor, vocab: Sequence[int], w0: torch.Tensor, w: torch.Tensor,
              V: torch.Tensor) -> torch.Tensor:
    """ctr FM model in gather form (src/ctr/fm/model.py:34-53): dense (B,nd), ids (B,F) int32,
    w0 (1,), w (L[,1]), V (k,L) -> sigmoid output (B,1)."""
    _rows2d(_chk(dense, "dense"), "dense")
    _rows2d(_chk(ids, "ids", torch.int32), "ids")
    w0, w, V = _chk(w0, "w0"), _chk(w, "w").reshape(-1), _chk(V, "V")
    B, nd = dense.shape
    F = ids.shape[1]
    L = nd + int(sum(vocab))
    if len(vocab) != F or w.numel() != L or V.dim() != 2 or V.shape[1] != L or not V.is_contiguous():
        raise ValueError("fm_onehot: inconsistent shapes")
    out = torch.empty((B, 1), dtype=torch.float32, device=dense.device)
    C.fm_onehot_f32(dense.data_ptr(), dense.stride(0), nd, ids.data_ptr(), ids.stride(0), [int(v) for v in vocab],
                    w0.data_ptr(), w.data_ptr(), V.data_ptr(), V.shape[0], B, out.data_ptr(), _stream())
    return out


# Pre-split weights for the large-layer Dense kernel (rec_dense_prepare_f32): one entry per live weight tensor,
# keyed by identity and invalidated by torch's in-place version counter; a weakref callback drops the entry with
# the tensor, so a recycled address can never hit a stale entry.
_prep_cache = {}


def _prepared_weights(W: torch.Tensor) -> torch.Tensor:
    """the prepared form of a weight matrix (rec_dense_prepare_f32), cached per tensor version"""
    key = id(W)
    ent = _prep_cache.get(key)
    if ent is not None and ent[0]() is W and ent[1] == W._version:
        return ent[2]
    K, N = W.shape
    buf = torch.empty(C.dense_prepared_bytes(K, N), dtype=torch.uint8, device=W.device)
    C.dense_prepare_f32(W.data_ptr(), K, N, buf.data_ptr(), _stream())
    _prep_cache[key] = (weakref.ref(W, lambda _r, k=key: _prep_cache.pop(k, None)), W._version, buf)
    return buf


def dense(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
          alpha: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
          row_absmax: Optional[torch.Tensor] = None, out_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Keras Dense on the last axis: act(x @ W + bias); x (..., K) with unit inner stride.
    Large layers run on the f16x2 kernel (rec_dense_prep_rs_f32), which scales every row of x by a power of two found from
    the row's largest magnitude.  row_absmax: those maxima (M floats), when the kernel that produced x delivered them — a
    chain of Dense layers passes them along and no layer re-reads its input to find them.  out_absmax: M ZEROED floats that
    receive the maxima of the output rows (the next layer's row_absmax)."""
    _chk(x, "x")
    _chk(W, "W")
    if W.dim() != 2 or not W.is_contiguous():
        raise ValueError("W: expected contiguous (K, N)")
    K, N = W.shape
    if x.shape[-1] != K:
        raise ValueError(f"dense: x last dim {x.shape[-1]} != K {K}")
    lead = x.shape[:-1]
    if x.dim() == 2 and x.stride(1) == 1:
        x2, xs = x, x.stride(0)
    else:
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        xs = K
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    else:
        _rows2d(_chk(out, "out"), "out")
    if bias is not None:
        _chk(bias, "bias")
    if alpha is not None:
        _chk(alpha, "alpha")
    for nm, t in (("row_absmax", row_absmax), ("out_absmax", out_absmax)):
        if t is not None and (_chk(t, nm).numel() != M or not t.is_contiguous()):
            raise ValueError(f"{nm}: expected M contiguous floats")
    prep = _prepared_weights(W) if (M >= 1024 and K * N >= 4096 and not (K <= 64 and N <= 64) and N > 8) else None
    scaled = prep is not None and K % 32 == 0 and xs % 4 == 0 and x2.data_ptr() % 16 == 0
    if scaled or out_absmax is not None:
        # one float per row of workspace: the f16x2 kernel (three MFMAs per product) scales every row by its own power of two
        ws = row_absmax if row_absmax is not None else (torch.empty(M, dtype=torch.float32, device=x.device) if scaled else None)
        C.dense_prep_rs_f32(x2.data_ptr(), xs, W.data_ptr(), _ptr(prep), _ptr(bias), _ptr(alpha), _act_id(act), M, K, N,
                            out.data_ptr(), out.stride(0), _ptr(ws), 1 if row_absmax is not None else 0, _ptr(out_absmax), _stream())
    else:
        C.dense_prep_f32(x2.data_ptr(), xs, W.data_ptr(), _ptr(prep), _ptr(bias), _ptr(alpha), _act_id(act), M, K, N,
                         out.data_ptr(), out.stride(0), _stream())
    return out.view(*lead, N) if out.is_contiguous() and out.shape[1] == N else out


def mha_ctr(xq: torch.Tensor, xk: torch.Tensor, xv: torch.Tensor, Wq, Wk, Wv, W0=None, head_num=1, head_size=None,
            act="relu") -> torch.Tensor:
    """ctr MultiHeadAttention (src/ctr/layers/modules.py:285-325) on (B, N, din) inputs."""
    for t, nm in ((xq, "xq"), (xk, "xk"), (xv, "xv"), (Wq, "Wq"), (Wk, "Wk"), (Wv, "Wv")):
        _chk(t, nm)
        if not t.is_contiguous():
            raise ValueError(f"{nm}: must be contiguous")
    B, N, din = xq.shape
    HS = Wq.shape[1]
    S = head_size if head_size is not None else HS // head_num
    if HS != head_num * S or Wq.shape[0] != din:
        raise ValueError("mha_ctr: weight shape mismatch")
    out = torch.empty((B, N, HS), dtype=torch.float32, device=xq.device)
    C.mha_ctr_f32(xq.data_ptr(), xk.data_ptr(), xv.data_ptr(), B, N, din, Wq.data_ptr(), Wk.data_ptr(),
                  Wv.data_ptr(), _ptr(W0), head_num, S, _act_id(act), out.data_ptr(), _stream())
    return out


def mha_ctr_stack(x: torch.Tensor, layers, head_num: int, head_size: int, act="relu") -> Optional[torch.Tensor]:
    """Stacked ctr MultiHeadAttention layers (one input tensor each) in ONE launch: layers = [(Wq, Wk, Wv, W0|None), ...].
    Returns None when the stack is not covered by the fused kernel (the caller then applies the layers one by one)."""
    _chk(x, "x")
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("x: expected contiguous (B, N, din)")
    B, N, din = x.shape
    HS = head_num * head_size
    L = len(layers)
    if head_size != 16 or din not in (16, 32) or head_num not in (1, 2) or N > 64 or not 1 <= L <= 4:
        return None
    has_res = [w[3] is not None for w in layers]
    for l, (wq, wk, wv, w0) in enumerate(layers):
        kin = din if l == 0 else HS
        for t in (wq, wk, wv) + ((w0,) if w0 is not None else ()):
            _chk(t, "W")
            if tuple(t.shape) != (kin, HS) or not t.is_contiguous():
                return None
    if any(has_res) and not all(has_res):
        w0s = [w[3].data_ptr() if w[3] is not None else 0 for w in layers]
    else:
        w0s = [w[3].data_ptr() for w in layers] if all(has_res) else []
    out = torch.empty((B, N, HS), dtype=torch.float32, device=x.device)
    C.mha_ctr_stack_f32(x.data_ptr(), B, N, din, [w[0].data_ptr() for w in layers], [w[1].data_ptr() for w in layers],
                        [w[2].data_ptr() for w in layers], w0s, head_num, head_size, _act_id(act), out.data_ptr(), _stream())
    return out


def autoint_forward(group: "TableGroup", ids: torch.Tensor, dense: torch.Tensor, dense_embed: torch.Tensor, layers,
                    head_num: int, head_size: int, act, head_w: torch.Tensor, head_b: Optional[torch.Tensor],
                    oob_flag=None) -> Optional[torch.Tensor]:
    """AutoInt.call (src/ctr/autoint/model.py:46-55, 3-D form) in ONE launch: fields = [embedding rows of `ids` | dense
    values x dense_embed rows], the stacked interacting layers in registers, Dense(1) + sigmoid on the flattened result.
    layers = [(Wq, Wk, Wv, W0|None), ...].  Returns probabilities (B, 1), or None when the fused kernel does not cover
    the configuration (the caller composes the separate ops)."""
    F = len(group)
    if F > C.MAX_TABLES or len(set(group.dims)) != 1:
        return None
    D = group.dims[0]
    HS = head_num * head_size
    L = len(layers)
    if head_size != 16 or D not in (16, 32) or head_num not in (1, 2) or not 1 <= L <= 4:
        return None
    if ids.dtype != torch.int32:
        return None
    _rows2d(_chk(ids, "ids", torch.int32), "ids")
    _rows2d(_chk(dense, "dense"), "dense")
    _chk(dense_embed, "dense_embed")
    B, nd = dense.shape
    N = F + nd
    if ids.shape != (B, F) or tuple(dense_embed.shape) != (nd, D) or not dense_embed.is_contiguous() or N > 64:
        return None
    has_res = [w[3] is not None for w in layers]
    if any(has_res) and not all(has_res):
        return None
    for l, (wq, wk, wv, w0) in enumerate(layers):
        kin = D if l == 0 else HS
        for t in (wq, wk, wv) + ((w0,) if w0 is not None else ()):
            _chk(t, "W")
            if tuple(t.shape) != (kin, HS) or not t.is_contiguous():
                return None
    head_w = _chk(head_w, "head_w").reshape(-1)
    if head_w.numel() != N * HS or not head_w.is_contiguous():
        return None
    out = torch.empty((B, 1), dtype=torch.float32, device=ids.device)
    C.autoint_forward_f32(group.descs, ids.data_ptr(), ids.stride(0), dense.data_ptr(), dense.stride(0), nd,
                          dense_embed.data_ptr(), D, [w[0].data_ptr() for w in layers], [w[1].data_ptr() for w in layers],
                          [w[2].data_ptr() for w in layers], [w[3].data_ptr() for w in layers] if all(has_res) else [],
                          head_num, head_size, _act_id(act), head_w.data_ptr(), _ptr(head_b), B, out.data_ptr(), 0,
                          _ptr(oob_flag), _stream())
    return out


def din_attention_pool(q, k, v, mask, W, bias, act="sigmoid", alpha=None) -> torch.Tensor:
    """DIN AttentionLayer (src/ctr/layers/modules.py:144-175), hidden_unit = 1.
    q (B,d); k,v (B,T,d); mask (B,T) float tensor or None (None => uniform, modules.py:164-165)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (W, "W"), (bias, "bias")):
        _chk(t, nm)
        if not t.is_contiguous():
            raise ValueError(f"{nm}: must be contiguous")
    B, T, d = k.shape
    if mask is not None:
        mask = _chk(mask, "mask").contiguous()
    W = W.reshape(-1)
    if W.numel() != 4 * d:
        raise ValueError("din_attention_pool: W must have 4*d elements (Dense(hidden_unit=1))")
    out = torch.empty((B, d), dtype=torch.float32, device=q.device)
    C.din_attn_pool_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(mask), W.data_ptr(), bias.data_ptr(),
                        _ptr(alpha), _act_id(act), B, T, d, out.data_ptr(), _stream())
    return out


def _row_strided(t: torch.Tensor, nm: str) -> int:
    """(B, S, dm) tensor whose rows may sit in a wider buffer: unit inner stride, batch stride = S * row stride."""
    if t.dim() != 3 or t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)):
        raise ValueError(f"{nm}: expected (B, S, dm) rows with unit inner stride (a contiguous tensor or a column "
                         "slice of one)")
    return t.stride(1)


def mha_rowmask(q, k, v, mask, num_heads) -> torch.Tensor:
    """match scaled_dot_product_attention + head split/merge (src/match/layers/modules.py:76-96,
    119-130) on projected q (B,Sq,dm), k/v (B,Sk,dm); mask (B,Sq) floats (0 = padded query).
    q / k / v may be column slices of a wider (B, S, W) buffer (fused projections)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (mask, "mask")):
        _chk(t, nm)
    if not mask.is_contiguous():
        raise ValueError("mask: must be contiguous")
    qs, ks, vs = _row_strided(q, "q"), _row_strided(k, "k"), _row_strided(v, "v")
    B, Sq, dm = q.shape
    Sk = k.shape[1]
    dk = dm // max(1, num_heads)
    if (qs != dm or ks != dm or vs != dm) and not (Sq <= 8 or (Sq >= 16 and dk in (32, 64))):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()   # shapes served by the contiguous-only kernels
        qs = ks = vs = dm
    out = torch.empty((B, Sq, dm), dtype=torch.float32, device=q.device)
    step = 65535
    for b0 in range(0, B, step):  # grid.z limit
        b1 = min(B, b0 + step)
        C.mha_rowmask_strided_f32(q[b0:b1].data_ptr(), qs, k[b0:b1].data_ptr(), ks, v[b0:b1].data_ptr(), vs,
                                  mask[b0:b1].data_ptr(), b1 - b0, Sq, Sk, dm, num_heads, out[b0:b1].data_ptr(),
                                  _stream())
    return out


def gather_mha_fewq(q, table, ids, mask, num_heads) -> torch.Tensor:
    """Attention of a few query rows q (B, Sq <= 8, dm) over keys = values = table[ids] (ids (B, Sk); out-of-range
    ids are zero rows), without materialising the (B, Sk, dm) sequence tensor.  mask (B, Sq): 0 = padded query."""
    _chk(q, "q")
    _chk(table, "table")
    _chk(mask, "mask")
    ids = _rows2d(_chk(ids, "ids", None), "ids")
    if not (table.is_contiguous() and mask.is_contiguous() and ids.is_contiguous()):
        raise ValueError("gather_mha_fewq: table, ids and mask must be contiguous")
    qs = _row_strided(q, "q")
    B, Sq, dm = q.shape
    if table.shape[1] != dm or ids.shape[0] != B:
        raise ValueError("gather_mha_fewq: inconsistent shapes")
    out = torch.empty((B, Sq, dm), dtype=torch.float32, device=q.device)
    C.gather_mha_fewq_f32(q.data_ptr(), qs, table.data_ptr(), table.shape[0], ids.data_ptr(), _ids_dtype(ids),
                          mask.data_ptr(), B, Sq, ids.shape[1], dm, num_heads, out.data_ptr(), _stream())
    return out


def layernorm_residual(x, r, gamma, beta, eps, row_mask=None) -> torch.Tensor:
    """LayerNormalization(x + r) over the last axis [* row_mask] (src/match/layers/modules.py:175,183)."""
    _chk(x, "x")
    x = x.contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    if r is not None:
        r = _chk(r, "r").contiguous()
    if row_mask is not None:
        row_mask = _chk(row_mask, "row_mask").contiguous()
    out = torch.empty_like(x)
    C.layernorm_residual_f32(x.data_ptr(), _ptr(r), _chk(gamma, "gamma").data_ptr(), _chk(beta, "beta").data_ptr(),
                             float(eps), _ptr(row_mask), rows, d, out.data_ptr(), _stream())
    return out


def gather_dot_scores(seq_info: torch.Tensor, table: torch.Tensor, ids: torch.Tensor,
                      out: Optional[torch.Tensor] = None, oob_flag=None) -> torch.Tensor:
    """out[b, j] = seq_info[b] . table[ids[b, j]]  (src/match/sasrec/model.py:90-91, fused gather+dot)."""
    _rows2d(_chk(seq_info, "seq_info"), "seq_info")
    _chk(table, "table")
    _rows2d(_chk(ids, "ids", torch.int32), "ids")
    B, n = ids.shape
    if out is None:
        out = torch.empty((B, n), dtype=torch.float32, device=ids.device)
    C.gather_dot_scores_f32(seq_info.data_ptr(), seq_info.stride(0),
                            (table.data_ptr(), int(table.shape[0]), int(table.shape[1]), 0), ids.data_ptr(),
                            ids.stride(0), n, B, out.data_ptr(), out.stride(0), _ptr(oob_flag), _stream())
    return out


def sasrec_last_row_supported(d_model: int, ffn_hidden: int, S: int, n_cand: int) -> bool:
    """Shapes served by the one-launch SASRec kernel (rec_sasrec_last_row_f32): d 64, ffn 64 / 128, id buffers that fit
    the LDS next to the block's weights."""
    return bool(C.sasrec_last_row_supported(int(d_model), int(ffn_hidden), int(S), int(n_cand)))


def sasrec_last_row(weights, eps1, eps2, seq_table, seq_ids, pad_id, mask_ids, mask_stride, pos_table, pos_ids, neg_table,
                    neg_ids, oob_flag=None):
    """SASRec with one encoder block / one head, last position only, in ONE launch (src/match/sasrec/model.py:72-96).
    weights = (wq, bq, wk, wv, bv, ln1_gamma, ln1_beta, w1, b1, w2, b2, ln2_gamma, ln2_beta), Keras layouts.
    seq_ids (B, S) int32: id == pad_id or out of range -> zero row.  mask_ids: an int32 tensor VIEW whose element
    [b * mask_stride] != 0 is sample b's query / output mask.  Returns (logits (B, n_pos + n_neg), seq_info (B, d))."""
    ws = [_chk(w, "weight").contiguous() for w in weights]
    if len(ws) != 13:
        raise ValueError("sasrec_last_row: 13 weight tensors expected")
    for tb in (seq_table, pos_table, neg_table):
        if not _chk(tb, "table").is_contiguous():
            raise ValueError("sasrec_last_row: tables must be contiguous")
    _rows2d(_chk(seq_ids, "seq_ids", torch.int32), "seq_ids")
    _rows2d(_chk(pos_ids, "pos_ids", torch.int32), "pos_ids")
    _rows2d(_chk(neg_ids, "neg_ids", torch.int32), "neg_ids")
    _chk(mask_ids, "mask_ids", torch.int32)
    B, S = seq_ids.shape
    d = seq_table.shape[1]
    fh = ws[7].shape[1]
    if pos_ids.shape[0] != B or neg_ids.shape[0] != B or pos_table.shape[1] != d or neg_table.shape[1] != d:
        raise ValueError("sasrec_last_row: inconsistent shapes")
    if tuple(ws[0].shape) != (d, d) or tuple(ws[2].shape) != (d, d) or tuple(ws[3].shape) != (d, d) or \
            tuple(ws[7].shape) != (d, fh) or tuple(ws[9].shape) != (fh, d):
        raise ValueError("sasrec_last_row: weight shapes do not match d_model / ffn_hidden")
    n_pos, n_neg = pos_ids.shape[1], neg_ids.shape[1]
    logits = torch.empty((B, n_pos + n_neg), dtype=torch.float32, device=seq_ids.device)
    seq_info = torch.empty((B, d), dtype=torch.float32, device=seq_ids.device)
    C.sasrec_last_row_f32([w.data_ptr() for w in ws], float(eps1), float(eps2), int(fh), seq_table.data_ptr(),
                          int(seq_table.shape[0]), seq_ids.data_ptr(), seq_ids.stride(0), S, int(pad_id), mask_ids.data_ptr(),
                          int(mask_stride), pos_table.data_ptr(), int(pos_table.shape[0]), pos_ids.data_ptr(),
                          pos_ids.stride(0), n_pos, neg_table.data_ptr(), int(neg_table.shape[0]), neg_ids.data_ptr(),
                          neg_ids.stride(0), n_neg, B, d, seq_info.data_ptr(), logits.data_ptr(), logits.stride(0),
                          _ptr(oob_flag), _stream())
    return logits, seq_info


def shard_bucket(ids_flat: torch.Tensor, G: int):
    """Stable bucketing of a flat int32 id list by owner = id % G.
    Returns (counts[G] int32, perm[n] int32, send_local[n] int32)."""
    _chk(ids_flat, "ids", torch.int32)
    if ids_flat.dim() != 1 or not ids_flat.is_contiguous():
        raise ValueError("ids: expected a contiguous 1-D tensor")
    n = ids_flat.numel()
    dev = ids_flat.device
    counts = torch.empty(G, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    send_local = torch.empty(n, dtype=torch.int32, device=dev)
    ws = torch.empty(max(1, C.shard_bucket_workspace_bytes(n, G)), dtype=torch.uint8, device=dev)
    C.shard_bucket_i32(ids_flat.data_ptr(), n, G, counts.data_ptr(), perm.data_ptr(), send_local.data_ptr(),
                       ws.data_ptr(), _stream())
    return counts, perm, send_local


def unpermute_rows(rows: torch.Tensor, perm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = rows[perm[i]]"""
    _rows2d(_chk(rows, "rows"), "rows")
    _chk(perm, "perm", torch.int32)
    n, D = perm.numel(), rows.shape[1]
    if out is None:
        out = torch.empty((n, D), dtype=torch.float32, device=rows.device)
    if not rows.is_contiguous():
        raise ValueError("rows: must be contiguous")
    C.unpermute_rows_f32(rows.data_ptr(), perm.data_ptr(), n, D, out.data_ptr(), out.stride(0), _stream())
    return out


def gather_dots(group: TableGroup, ids: torch.Tensor, Wd: torch.Tensor, out: Optional[torch.Tensor] = None,
                oob_flag: Optional[torch.Tensor] = None, row_absmax: Optional[torch.Tensor] = None):
    """gather_concat + dots[b, v] = <concat row b, Wd[v]> for up to 8 weight vectors Wd (nv, width), in one pass.
    Returns (concat rows (B, >= group.width), dots (B, nv))."""
    ids = _rows2d(_chk(ids, "ids", None), "ids")
    _rows2d(_chk(Wd, "Wd"), "Wd")
    F = len(group)
    B = ids.shape[0]
    nv, width = Wd.shape
    if ids.shape[1] != F or F > C.MAX_TABLES or not Wd.is_contiguous():
        raise ValueError("gather_dots: inconsistent shapes")
    if out is None:
        out = torch.empty((B, (group.width + 3) // 4 * 4), dtype=torch.float32, device=ids.device)[:, :group.width]
    dots = torch.empty((B, nv), dtype=torch.float32, device=ids.device)
    # row_absmax (B floats): also receive max |element| of every gathered row (ops.dense's row_absmax)
    C.gather_dots_absmax_f32(group.descs, ids.data_ptr(), _ids_dtype(ids), ids.stride(0), Wd.data_ptr(), nv, width, B,
                             out.data_ptr(), out.stride(0), dots.data_ptr(), _ptr(oob_flag), _ptr(row_absmax), _stream())
    return out, dots


def dcn_logit(dots: torch.Tensor, G: torch.Tensor, c: float, extra: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sigmoid(alpha_L * dots[:, L] + c + extra) with alpha from the closed-form cross recurrence (rec_dcn_logit_f32)."""
    dots = _chk(dots, "dots").contiguous()
    G = _chk(G, "G").contiguous()
    B, L1 = dots.shape
    if G.numel() != L1 - 1:
        raise ValueError("dcn_logit: G must have one entry per cross layer")
    if extra is not None:
        extra = _chk(extra, "extra").contiguous().reshape(-1)
        if extra.numel() != B:
            raise ValueError("dcn_logit: extra must have one value per sample")
    out = torch.empty((B, 1), dtype=torch.float32, device=dots.device)
    C.dcn_logit_f32(dots.data_ptr(), L1 - 1, G.data_ptr() if L1 > 1 else 0, float(c), _ptr(extra), B, out.data_ptr(),
                    _stream())
    return out


def add_sigmoid(a: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sigmoid(a + b), elementwise (final logits of DeepFM / DCN / DLRM)."""
    _chk(a, "a")
    a = a.contiguous()
    if b is not None:
        b = _chk(b, "b").contiguous()
        if b.shape != a.shape:
            raise ValueError("add_sigmoid: shape mismatch")
    out = torch.empty_like(a)
    C.add_sigmoid_f32(a.data_ptr(), _ptr(b), a.numel(), out.data_ptr(), _stream())
    return out


def axpby_act(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0, beta: float = 1.0, act=None) -> torch.Tensor:
    """act(alpha * a + beta * b), elementwise (Residual_Units' relu(x + inputs), Wide&Deep's 0.5/0.5 blend)."""
    a = _chk(a, "a").contiguous()
    b = _chk(b, "b").contiguous()
    if b.shape != a.shape:
        raise ValueError("axpby_act: shape mismatch")
    if act == 'prelu':
        raise ValueError("axpby_act: prelu is not supported here")
    out = torch.empty_like(a)
    C.axpby_act_f32(a.data_ptr(), float(alpha), b.data_ptr(), float(beta), a.numel(), _act_id(act), out.data_ptr(),
                    _stream())
    return out


def mul_act(a: torch.Tensor, b: torch.Tensor, act=None) -> torch.Tensor:
    """act(a * b), elementwise, same shapes (NCF's GMF vector, ESMM's pCTR * pCVR)."""
    a = _chk(a, "a").contiguous()
    b = _chk(b, "b").contiguous()
    if b.shape != a.shape:
        raise ValueError("mul_act: shape mismatch")
    out = torch.empty_like(a)
    C.mul_act_f32(a.data_ptr(), b.data_ptr(), a.numel(), _act_id(act), out.data_ptr(), _stream())
    return out


def cosine_flat(a: torch.Tensor, b: torch.Tensor, sigmoid: bool = False) -> torch.Tensor:
    """Cosine of the two tensors flattened to single vectors (Dssm.cosine_similarity) -> shape (1,)."""
    a = _chk(a, "a").contiguous()
    b = _chk(b, "b").contiguous()
    if a.numel() != b.numel() or a.numel() == 0:
        raise ValueError("cosine_flat: tensors must have the same non-zero number of elements")
    ws = torch.empty(C.cosine_flat_workspace_bytes(a.numel()), dtype=torch.uint8, device=a.device)
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    C.cosine_flat_f32(a.data_ptr(), b.data_ptr(), a.numel(), int(bool(sigmoid)), out.data_ptr(), ws.data_ptr(),
                      _stream())
    return out


def scale_rows(x: torch.Tensor, row_scale: torch.Tensor) -> torch.Tensor:
    """x[..., :] * row_scale[...]  (SASRec `att_outputs *= mask`)."""
    _chk(x, "x")
    x = x.contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    row_scale = _chk(row_scale, "row_scale").contiguous()
    if row_scale.numel() != rows:
        raise ValueError("scale_rows: row_scale must have one entry per row")
    out = torch.empty_like(x)
    C.scale_rows_f32(x.data_ptr(), row_scale.data_ptr(), rows, d, out.data_ptr(), _stream())
    return out


def copy_cols(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[:, :N] = src (a concat part written at a column offset: pass dst = buf[:, off:]); src fp32 or int32 (ids
    concatenated as floats, src/ctr/din/model.py:68)."""
    _chk(src, "src", None)
    _chk(dst, "dst")
    if src.dim() != 2 or dst.dim() != 2 or src.stride(1) != 1 or dst.stride(1) != 1 or dst.shape[0] != src.shape[0] or \
            dst.shape[1] < src.shape[1] or src.dtype not in (torch.float32, torch.int32):
        raise ValueError("copy_cols: expected 2-D fp32/int32 src and a 2-D fp32 dst view with at least as many columns")
    C.copy2d_f32(src.data_ptr(), src.stride(0), 1 if src.dtype == torch.float32 else 0, src.shape[0], src.shape[1],
                 dst.data_ptr(), dst.stride(0), _stream())
    return dst


def concat_cols(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """tf.concat(parts, axis=-1) of 2-D fp32 tensors as column-offset writes into one buffer (rec_copy2d_f32): where a
    producer kernel cannot write its columns in place, this is the copy — no ATen cat on any model path"""
    B = parts[0].shape[0]
    out = torch.empty((B, sum(int(t.shape[1]) for t in parts)), dtype=torch.float32, device=parts[0].device)
    c = 0
    for t in parts:
        w = int(t.shape[1])
        copy_cols(t if t.stride(1) == 1 else t.contiguous(), out[:, c:c + w])
        c += w
    return out


def scale_embed(x: torch.Tensor, E: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[b, j*D:(j+1)*D] = x[b, j] * E[j]  (out: a (B, >= nd*D) view at the column offset of the concat buffer)"""
    _rows2d(_chk(x, "x"), "x")
    _chk(E, "E")
    _chk(out, "out")
    nd, D = E.shape
    if x.shape[1] != nd or not E.is_contiguous() or out.shape[0] != x.shape[0] or out.shape[1] < nd * D or out.stride(1) != 1:
        raise ValueError("scale_embed: inconsistent shapes")
    C.scale_embed_f32(x.data_ptr(), x.stride(0), E.data_ptr(), x.shape[0], nd, D, out.data_ptr(), out.stride(0), _stream())
    return out


def dice(x: torch.Tensor, alpha: torch.Tensor, mean=None, var=None, eps: float = 1e-3) -> torch.Tensor:
    """Dice activation at inference (src/ctr/layers/modules.py:333-337)."""
    _chk(x, "x")
    x = x.contiguous()
    d = x.shape[-1]
    out = torch.empty_like(x)
    C.dice_f32(x.data_ptr(), _chk(alpha, "alpha").data_ptr(), _ptr(mean), _ptr(var), float(eps), x.numel() // d, d,
               out.data_ptr(), _stream())
    return out


def gather_din_attention_pool(q, group: TableGroup, ids, mask, W, bias, act="sigmoid", alpha=None,
                              mask_from_ids=False, oob_flag=None, out=None) -> torch.Tensor:
    """Fused history gather + DIN pooling: ids (B, T, n_tab) index `group`'s tables (one shared dim);
    k = v = the gathered (B, T, n_tab*Dt) history, never materialised.  mask: (B,T) float tensor, or
    None with mask_from_ids=True (slot real iff ids[b,t,0] != 0), or None (uniform, modules.py:164-165)."""
    _chk(q, "q")
    _chk(ids, "ids", None)
    if ids.dim() != 3 or not ids.is_contiguous() or ids.shape[2] != len(group):
        raise ValueError("ids: expected contiguous (B, T, n_tab)")
    B, T, n_tab = ids.shape
    d = sum(group.dims)
    if len(set(group.dims)) != 1 or q.shape != (B, d) or not q.is_contiguous():
        raise ValueError("gather_din_attention_pool: tables must share one dim and q must be contiguous (B, n_tab*Dt)")
    if mask is not None:
        mask = _chk(mask, "mask").contiguous()
    W = _chk(W, "W").reshape(-1)
    if out is None:
        out = torch.empty((B, d), dtype=torch.float32, device=q.device)
    elif _chk(out, "out").shape != (B, d) or not out.is_contiguous():
        raise ValueError("out: expected contiguous (B, n_tab*Dt)")
    C.gather_din_attn_pool_f32(q.data_ptr(), group.descs, ids.data_ptr(), _ids_dtype(ids), _ptr(mask),
                               1 if mask_from_ids else 0, W.data_ptr(), _chk(bias, "bias").data_ptr(), _ptr(alpha),
                               _act_id(act), B, T, out.data_ptr(), _ptr(oob_flag), _stream())
    return out


def gather_fm(group: TableGroup, ids: torch.Tensor, dense: Optional[torch.Tensor], w_padded: torch.Tensor, nd_padded: int,
              emb_out: torch.Tensor, oob_flag=None, row_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused K1+K3 (DeepFM): gathers into the concat buffer `emb_out` (group.out_cols) and returns the FM
    layer output (B,1) computed in the same pass.  `dense` is the (B, nd_padded) dense block AS STORED in
    the concat buffer (zero-padded to a multiple of 4 columns), `w_padded` = FM weights in the same
    padded concat order (pad entries multiply zeros)."""
    ids = _rows2d(_chk(ids, "ids", None), "ids")
    _rows2d(_chk(emb_out, "emb_out"), "emb_out")
    B = ids.shape[0]
    w_padded = _chk(w_padded, "w").reshape(-1)
    fm_out = torch.empty((B, 1), dtype=torch.float32, device=ids.device)
    ws = _fm_workspace(ids.device, B)
    # row_absmax (B floats): also receive max |element| of every sample's concat row (ops.dense's row_absmax)
    C.gather_fm_absmax_f32(group.descs, ids.data_ptr(), _ids_dtype(ids), ids.stride(0), _ptr(dense),
                           dense.stride(0) if dense is not None else 0, nd_padded, w_padded.data_ptr(), B, emb_out.data_ptr(),
                           emb_out.stride(0), fm_out.data_ptr(), ws.data_ptr(), _ptr(oob_flag), _ptr(row_absmax), _stream())
    return fm_out


def embedding_grad(grad_group: TableGroup, ids: torch.Tensor, dy: torch.Tensor) -> None:
    """Backward of gather_concat: grad_tables[f][ids[b,f]] += dy[b, out_col_f : +D_f] (duplicates summed;
    TF's IndexedSlices gradient of tf.gather).  `grad_group` wraps the (V_f, D_f) fp32 accumulators."""
    ids = _rows2d(_chk(ids, "ids", None), "ids")
    _rows2d(_chk(dy, "dy"), "dy")
    B, F = ids.shape
    if F != len(grad_group) or dy.shape[0] != B or dy.shape[1] < grad_group.width:
        raise ValueError("embedding_grad: inconsistent shapes")
    for lo in range(0, F, C.MAX_TABLES):
        hi = min(F, lo + C.MAX_TABLES)
        C.embedding_grad_f32(grad_group.descs[lo:hi], ids[:, lo:hi].data_ptr(), _ids_dtype(ids), ids.stride(0),
                             dy.data_ptr(), dy.stride(0), B, _stream())


# Raw-pointer writers (the optimiser kernels) change weights behind torch's back.  Every cache of derived weights
# must notice: `note_weights_written` bumps torch's own version counter of the tensor (the pre-split Dense cache keys
# on it) and a process-wide generation that recamd.nn.Layer._version includes (folded BatchNorm, fused QKV, ...).
_weight_generation = [0]


def weight_generation() -> int:
    return _weight_generation[0]


def note_weights_written(*tensors: torch.Tensor) -> None:
    _weight_generation[0] += 1
    for t in tensors:
        torch.autograd.graph.increment_version(t)


def adam_step(var: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad: torch.Tensor, step: int, lr: float = 1e-3,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7, l2: float = 0.0) -> None:
    """In-place dense Keras-Adam update (TF2 defaults) with the embeddings' l2(c) regulariser gradient 2 c var."""
    for t, nm in ((var, "var"), (m, "m"), (v, "v"), (grad, "grad")):
        _chk(t, nm)
        if not t.is_contiguous() or t.numel() != var.numel():
            raise ValueError(f"{nm}: must be contiguous and match var")
    C.adam_f32(var.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), var.numel(), float(lr), float(beta1),
               float(beta2), float(eps), int(step), float(l2), _stream())
    note_weights_written(var, m, v)


def topk_inner_product(queries: torch.Tensor, items: torch.Tensor, k: int):
    """Exact inner-product top-k (faiss IndexFlatIP.search): returns (scores (Q,k) descending, idx (Q,k) int64)."""
    _rows2d(_chk(queries, "queries"), "queries")
    _rows2d(_chk(items, "items"), "items")
    Q, d = queries.shape
    N = items.shape[0]
    if items.shape[1] != d:
        raise ValueError("topk_inner_product: queries and items differ in dim")
    scores = torch.empty((Q, k), dtype=torch.float32, device=queries.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=queries.device)
    nws = C.topk_ip_workspace_bytes(Q, N, int(k)) if (Q and N) else 0
    ws = torch.empty(nws, dtype=torch.uint8, device=queries.device) if nws else None
    C.topk_ip_ws_f32(queries.data_ptr(), queries.stride(0), Q, items.data_ptr() if N else 0, items.stride(0) if N else d,
                     N, d, int(k), scores.data_ptr(), idx.data_ptr(), _ptr(ws), _stream())
    return scores, idx


def _metric_args(y_true: torch.Tensor, y_pred: torch.Tensor):
    y_true = _chk(y_true, "y_true").contiguous().reshape(-1)
    y_pred = _chk(y_pred, "y_pred").contiguous().reshape(-1)
    if y_true.numel() != y_pred.numel() or y_true.numel() == 0:
        raise ValueError("y_true / y_pred must have the same non-zero number of elements")
    ws = torch.empty(C.metrics_workspace_bytes(y_true.numel()), dtype=torch.uint8, device=y_true.device)
    out = torch.empty(1, dtype=torch.float32, device=y_true.device)
    return y_true, y_pred, ws, out


def binary_crossentropy(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """Mean Keras binary_crossentropy over all samples (probabilities clipped to [1e-7, 1-1e-7]) -> shape (1,)."""
    y_true, y_pred, ws, out = _metric_args(y_true, y_pred)
    C.binary_crossentropy_f32(y_true.data_ptr(), y_pred.data_ptr(), y_true.numel(), out.data_ptr(), ws.data_ptr(),
                              _stream())
    return out


def pairwise_rank_loss(logits: torch.Tensor) -> torch.Tensor:
    """add_loss of SASRec / NCF (src/match/sasrec/model.py:93-95): logits (B, 1 + n), column 0 = the positive score ->
    mean(-log sigmoid(pos) - log(1 - sigmoid(neg))) / 2 with (B,1)+(B,n) broadcasting, shape (1,)."""
    _rows2d(_chk(logits, "logits"), "logits")
    B, w = logits.shape
    if w < 2:
        raise ValueError("pairwise_rank_loss: logits need a positive column and at least one negative column")
    ws = torch.empty(int(C.metrics_workspace_bytes(B * (w - 1))), dtype=torch.uint8, device=logits.device)
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    C.pairwise_rank_loss_f32(logits.data_ptr(), logits.stride(0), B, w - 1, out.data_ptr(), ws.data_ptr(), _stream())
    return out


def auc(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """tf.keras.metrics.AUC() with its defaults (200 thresholds, ROC, trapezoid) -> shape (1,)."""
    y_true, y_pred, ws, out = _metric_args(y_true, y_pred)
    C.auc_f32(y_true.data_ptr(), y_pred.data_ptr(), y_true.numel(), out.data_ptr(), ws.data_ptr(), _stream())
    return out
