"""§8f-1 timing: embedding backward (scatter-add) and dense Adam at the DLRM shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "recommend-tf2.0_amd"))
from recamd import ops  # noqa: E402


def timeit(fn, n=10, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    B, F, D, V = 65536, 26, 128, 100_000
    res = {}
    for name, Vv in (("uniform_V100k", V), ("hot_V1k", 1000)):
        ids = torch.randint(0, Vv, (B, F), device=dev, dtype=torch.int32)
        dy = torch.randn(B, F * D, device=dev)
        g = torch.zeros(F, Vv, D, device=dev)
        grp = ops.TableGroup([g[f] for f in range(F)])
        ms = timeit(lambda: ops.embedding_grad(grp, ids, dy))
        res[f"embedding_grad_{name}"] = {"ms": ms, "added_GBps": B * F * D * 4 / ms / 1e6}
    n = F * V * D
    var, m, v, grad = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    ms = timeit(lambda: ops.adam_step(var, m, v, grad, 3, l2=1e-4), n=5, w=1)
    res["adam_dense_26x100kx128"] = {"ms": ms, "GBps": n * 28 / ms / 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
