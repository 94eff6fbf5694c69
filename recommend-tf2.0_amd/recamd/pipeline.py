"""The step before the path, on the device: raw columns -> what the models take.

  LabelEncoder   sklearn LabelEncoder of src/ctr/utils/data_process.py:66-68: `fit` builds the per-column sorted
                 vocabularies once on the host (string order: the missing token '-1' first, then the 8-digit hex
                 categories = numeric order), `transform` is the rec_label_encode_u32 kernel (binary search per id).
  hash_ids       id = hash(token) mod vocabulary size, when no vocabulary is kept (rec_hash_ids_u32).
  MinMaxScaler   the scaler of :76-78 on astype(int) values, per column (rec_minmax_fit_f32 / rec_minmax_scale_f32).
  pad_sequences  tf.keras pad_sequences(maxlen) of src/match/utils/data_process.py:138 on a ragged batch.
  BatchFeeder    pinned, double-buffered host -> device feeder: batch i+1 crosses PCIe on a copy stream and is
                 transformed on the device while batch i computes; the consumer waits on an event, never on the host.

The kernels have no CPU fallback; `fit` of the label encoder is host-side numpy by design (one-time vocabulary build)."""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from ._lib import C

TOKEN_MISSING = 0xFFFFFFFF


def _s():
    return torch.cuda.current_stream().cuda_stream


def hex_tokens(col) -> np.ndarray:
    """Criteo categorical column (8-digit hex strings; NaN / None / '' / '-1' = missing) -> uint32 tokens (host)."""
    out = np.empty(len(col), np.uint32)
    for i, s in enumerate(col):
        if s is None or s == "" or s == "-1" or (isinstance(s, float) and s != s):
            out[i] = TOKEN_MISSING
        else:
            out[i] = int(s, 16)
    return out


def _sort_key(tok: np.ndarray) -> np.ndarray:
    t = tok.astype(np.uint64)
    return np.where(tok == np.uint32(TOKEN_MISSING), np.uint64(0), t + np.uint64(1))


class LabelEncoder:
    """One sklearn LabelEncoder per categorical column."""

    def __init__(self, device=None):
        self.device = device
        self.vocab_host: List[np.ndarray] = []
        self.vocab_dev: List[torch.Tensor] = []

    def fit(self, tokens: np.ndarray) -> "LabelEncoder":
        """tokens (n, F) uint32 on the host: per column the sorted (string order) unique tokens"""
        tokens = np.asarray(tokens, np.uint32)
        self.vocab_host = []
        for f in range(tokens.shape[1]):
            u = np.unique(tokens[:, f])
            self.vocab_host.append(u[np.argsort(_sort_key(u), kind="stable")])
        dev = self.device or torch.device("cuda", torch.cuda.current_device())
        self.vocab_dev = [torch.from_numpy(v.view(np.int32)).to(dev) for v in self.vocab_host]
        return self

    @property
    def classes_(self):
        return [len(v) for v in self.vocab_host]

    def transform(self, tokens: torch.Tensor, out: Optional[torch.Tensor] = None, unseen_flag: Optional[torch.Tensor] = None):
        """tokens (B, F) uint32-as-int32 device tensor -> ids (B, F) int32; unseen tokens give -1 and set the flag"""
        if not tokens.is_cuda or tokens.dtype != torch.int32 or tokens.dim() != 2 or tokens.stride(1) != 1:
            raise ValueError("transform: expected a (B, F) int32 GPU tensor holding the uint32 tokens")
        B, F = tokens.shape
        if F != len(self.vocab_dev):
            raise ValueError(f"transform: {F} columns, encoder was fitted on {len(self.vocab_dev)}")
        if out is None:
            out = torch.empty((B, F), dtype=torch.int32, device=tokens.device)
        for lo in range(0, F, C.MAX_TABLES):
            hi = min(F, lo + C.MAX_TABLES)
            C.label_encode_u32([v.data_ptr() for v in self.vocab_dev[lo:hi]], [int(v.numel()) for v in self.vocab_dev[lo:hi]],
                               tokens[:, lo:hi].data_ptr(), tokens.stride(0), B, out[:, lo:hi].data_ptr(), out.stride(0),
                               0 if unseen_flag is None else unseen_flag.data_ptr(), _s())
        return out


def hash_ids(tokens: torch.Tensor, vocab_sizes: Sequence[int], seed: int = 0, out: Optional[torch.Tensor] = None):
    B, F = tokens.shape
    if out is None:
        out = torch.empty((B, F), dtype=torch.int32, device=tokens.device)
    C.hash_ids_u32(tokens.data_ptr(), tokens.stride(0), [int(v) for v in vocab_sizes], B, int(seed) & 0xFFFFFFFF,
                   out.data_ptr(), out.stride(0), _s())
    return out


class MinMaxScaler:
    """sklearn MinMaxScaler on astype(int) values, per column, on the device."""

    def __init__(self, truncate_to_int: bool = True):
        self.trunc = 1 if truncate_to_int else 0
        self.data_min_ = self.data_max_ = None

    def fit(self, x: torch.Tensor) -> "MinMaxScaler":
        M, N = x.shape
        self.data_min_ = torch.empty(N, dtype=torch.float32, device=x.device)
        self.data_max_ = torch.empty(N, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(1, C.minmax_workspace_bytes(M, N)), dtype=torch.uint8, device=x.device)
        C.minmax_fit_f32(x.data_ptr(), x.stride(0), M, N, self.trunc, self.data_min_.data_ptr(), self.data_max_.data_ptr(),
                         ws.data_ptr(), _s())
        return self

    def transform(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        M, N = x.shape
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=x.device)
        C.minmax_scale_f32(x.data_ptr(), x.stride(0), M, N, self.data_min_.data_ptr(), self.data_max_.data_ptr(), self.trunc,
                           out.data_ptr(), out.stride(0), _s())
        return out

    def fit_transform(self, x: torch.Tensor) -> torch.Tensor:
        return self.fit(x).transform(x)


def pad_sequences(values: torch.Tensor, offsets: torch.Tensor, maxlen: int, padding: str = "pre", truncating: str = "pre",
                  value: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ragged batch (values int32 (n,), offsets int64 (B+1,)) on the device -> (B, maxlen) int32"""
    B = offsets.numel() - 1
    if out is None:
        out = torch.empty((B, maxlen), dtype=torch.int32, device=values.device)
    C.pad_sequences_i32(values.data_ptr(), offsets.data_ptr(), B, maxlen, int(value), 1 if padding == "pre" else 0,
                        1 if truncating == "pre" else 0, out.data_ptr(), out.stride(0), _s())
    return out


def ragged(seqs) -> tuple:
    """host helper: list of lists -> (values int32, offsets int64) numpy arrays"""
    lens = np.fromiter((len(s) for s in seqs), np.int64, len(seqs))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = np.fromiter((v for s in seqs for v in s), np.int32, int(offsets[-1]))
    return values, offsets


class BatchFeeder:
    """Pinned, double-buffered host -> device feeder for Criteo-shaped batches.

    Host side: raw dense values (n, nd) fp32 and categorical tokens (n, F) uint32.  Per batch: copy the slice into a
    pinned staging buffer, async H2D on a private copy stream, label-encode + min-max on the device (same stream),
    record an event.  `for dense, ids in feeder:` yields device tensors whose producer event the consumer's stream
    has been told to wait for; two buffers rotate, so batch i+1 crosses PCIe while batch i computes."""

    def __init__(self, dense, tokens, batch_size: int, encoder: Optional[LabelEncoder] = None,
                 scaler: Optional[MinMaxScaler] = None, hash_vocab: Optional[Sequence[int]] = None, device=None, depth: int = 2):
        """dense / tokens: numpy arrays (staged through pinned buffers batch by batch), or PINNED torch tensors
        (fp32 / int32 holding the uint32 tokens): then the H2D copies read them in place, no staging copy."""
        self.pinned_src = isinstance(dense, torch.Tensor) and isinstance(tokens, torch.Tensor)
        if self.pinned_src:
            if not (dense.is_pinned() and tokens.is_pinned() and dense.dtype == torch.float32 and tokens.dtype == torch.int32):
                raise ValueError("BatchFeeder: torch inputs must be pinned fp32 / int32 host tensors")
            self.dense, self.tokens = dense, tokens
        else:
            self.dense = np.ascontiguousarray(dense, np.float32)
            self.tokens = np.ascontiguousarray(tokens).view(np.int32) if tokens.dtype == np.uint32 else np.ascontiguousarray(tokens, np.int32)
        self.n, self.bs = len(self.dense), int(batch_size)
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.encoder, self.scaler, self.hash_vocab = encoder, scaler, hash_vocab
        nd, F = self.dense.shape[1], self.tokens.shape[1]
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.slots = []
        for _ in range(depth):
            self.slots.append({
                "h_dense": torch.empty((self.bs, nd), dtype=torch.float32, pin_memory=True),
                "h_tok": torch.empty((self.bs, F), dtype=torch.int32, pin_memory=True),
                "d_dense_raw": torch.empty((self.bs, nd), dtype=torch.float32, device=self.dev),
                "d_tok": torch.empty((self.bs, F), dtype=torch.int32, device=self.dev),
                "d_dense": torch.empty((self.bs, nd), dtype=torch.float32, device=self.dev),
                "d_ids": torch.empty((self.bs, F), dtype=torch.int32, device=self.dev),
                "ready": torch.cuda.Event(), "free": torch.cuda.Event()})
        self.unseen = torch.zeros(1, dtype=torch.int32, device=self.dev)

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def _stage(self, slot, lo):
        hi = min(self.n, lo + self.bs)
        b = hi - lo
        slot["free"].synchronize()                          # the consumer of this slot's previous batch is done
        if self.pinned_src:
            src_dense, src_tok = self.dense[lo:hi], self.tokens[lo:hi]
        else:
            slot["h_dense"][:b].numpy()[...] = self.dense[lo:hi]
            slot["h_tok"][:b].numpy()[...] = self.tokens[lo:hi]
            src_dense, src_tok = slot["h_dense"][:b], slot["h_tok"][:b]
        with torch.cuda.stream(self.copy_stream):
            slot["d_dense_raw"][:b].copy_(src_dense, non_blocking=True)
            slot["d_tok"][:b].copy_(src_tok, non_blocking=True)
            if self.scaler is not None:
                self.scaler.transform(slot["d_dense_raw"][:b], out=slot["d_dense"][:b])
            else:
                slot["d_dense"][:b].copy_(slot["d_dense_raw"][:b], non_blocking=True)
            if self.encoder is not None:
                self.encoder.transform(slot["d_tok"][:b], out=slot["d_ids"][:b], unseen_flag=self.unseen)
            elif self.hash_vocab is not None:
                hash_ids(slot["d_tok"][:b], self.hash_vocab, out=slot["d_ids"][:b])
            else:
                slot["d_ids"][:b].copy_(slot["d_tok"][:b], non_blocking=True)
            slot["ready"].record(self.copy_stream)
        return b

    def __iter__(self) -> Iterator:
        """Order per batch i: [wait ready(i)] yield -> the consumer enqueues compute(i) -> record free(i) -> stage
        batch i+1 into the OTHER slot (host waits for free(i-1), i.e. for compute(i-1), while compute(i) is already
        queued: the GPU never idles on the host) -> its H2D + transforms overlap compute(i)."""
        nb = len(self)
        depth = len(self.slots)
        # the scaler's statistics / the encoder's vocabulary were produced on the caller's stream (MinMaxScaler.fit
        # launches asynchronously): the copy stream must not read them before they are written
        self.copy_stream.wait_stream(torch.cuda.current_stream())
        sizes = {0: self._stage(self.slots[0], 0)} if nb else {}
        for i in range(nb):
            slot = self.slots[i % depth]
            torch.cuda.current_stream().wait_event(slot["ready"])
            b = sizes.pop(i)
            yield slot["d_dense"][:b], slot["d_ids"][:b]
            slot["free"].record(torch.cuda.current_stream())
            if i + 1 < nb:
                sizes[i + 1] = self._stage(self.slots[(i + 1) % depth], (i + 1) * self.bs)
