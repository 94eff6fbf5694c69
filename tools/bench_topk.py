"""§8f-4 timing: exact inner-product top-10 (the faiss IndexFlatIP step) at retrieval sizes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "recommend-tf2.0_amd"))
from recamd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    res = {}
    for Q, N, d in ((6040, 3706, 32), (65536, 100_000, 64), (16384, 1_000_000, 64), (65536, 1_000_000, 32)):
        q = torch.randn(Q, d, device=dev)
        items = torch.randn(N, d, device=dev)
        ops.topk_inner_product(q, items, 10)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3
        a.record()
        for _ in range(n):
            ops.topk_inner_product(q, items, 10)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        res[f"Q{Q}_N{N}_d{d}"] = {"ms": round(ms, 3), "TFLOPs": round(2.0 * Q * N * d / ms / 1e9, 1),
                                   "queries_per_s": round(Q / ms * 1e3)}
        print(json.dumps({f"Q{Q}_N{N}_d{d}": res[f"Q{Q}_N{N}_d{d}"]}), flush=True)


if __name__ == "__main__":
    main()
