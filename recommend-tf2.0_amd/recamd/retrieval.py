"""The retrieval step of the match train scripts on the GPU: a faiss.IndexFlatIP-shaped object
(src/match/dssm/dssm_train.py:74-78, src/match/fm/train.py:71-75):

    index = IndexFlatIP(item_embs.shape[1]); index.add(item_embs); D, I = index.search(user_embs, 10)
"""
import numpy as np
import torch

from . import ops
from .nn import default_device, to_device_f32


class IndexFlatIP:
    """Exact inner-product index.  `add` appends vectors (kept in HBM), `search` returns (D, I) as numpy arrays:
    D (Q, k) scores descending, I (Q, k) int64 positions in insertion order.  Where fewer than k vectors exist the
    label is -1 and the score is -FLT_MAX (-3.4028235e38), faiss' heap-neutral value for IndexFlatIP (the kernel
    itself pads with -inf: `recamd.ops.topk_inner_product`).  Limits of the kernel (it raises beyond them; faiss does
    not have them): k <= 32, d <= 128, ntotal <= 2^31 - 257.  NaN scores are never selected, and a +-inf component in
    a vector turns its scores into NaN for d <= 64 (bf16x3 split, csrc/bf16x3.h).  faiss is not importable here:
    parity with it is unpinned (ties resolve to the smaller index, which is also what faiss' heap yields for
    insertion-ordered equal scores in practice, not by contract)."""

    def __init__(self, d: int, device=None):
        self.d = int(d)
        self.device = device or default_device()
        self._items = torch.empty((0, self.d), dtype=torch.float32, device=self.device)

    @property
    def ntotal(self) -> int:
        return int(self._items.shape[0])

    def add(self, x) -> None:
        x = to_device_f32(np.asarray(x) if not isinstance(x, torch.Tensor) else x, self.device)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add: expected (n, {self.d}), got {tuple(x.shape)}")
        self._items = torch.cat([self._items, x.contiguous()], dim=0)

    def reset(self) -> None:
        self._items = torch.empty((0, self.d), dtype=torch.float32, device=self.device)

    def search(self, x, k: int):
        x = to_device_f32(np.asarray(x) if not isinstance(x, torch.Tensor) else x, self.device)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"search: expected (n, {self.d}), got {tuple(x.shape)}")
        D, I = ops.topk_inner_product(x.contiguous(), self._items, k)
        D, I = D.cpu().numpy(), I.cpu().numpy()
        D[I < 0] = np.float32(-3.4028235e38)
        return D, I
