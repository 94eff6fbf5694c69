"""DeepFM / DCN sparse stage at the Criteo shape (65 536 x 26 x dim 128, 1M rows per table):
fused gather+FM vs gather then FM layer; gather then cross."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
from recamd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, F, V, D, nd = 65536, 26, 1_000_000, 128, 13


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


arena = torch.empty((F, V, D), device=dev).uniform_(-0.05, 0.05)
pad = (-nd) % 4
cols = [pad + nd + f * D for f in range(F)]
g = ops.TableGroup([arena[f] for f in range(F)], out_cols=cols)
ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
buf = torch.zeros((B, pad + nd + F * D), device=dev)
buf[:, pad:pad + nd] = torch.rand((B, nd), device=dev)
w = torch.randn(nd + F * D, device=dev) * 0.05
wp = torch.zeros(pad + nd + F * D, device=dev)
wp[pad:] = w
embeds, sparse = buf[:, pad:], buf[:, pad + nd:]
t_g = timeit(lambda: ops.gather_concat(g, ids, out=buf))
t_fm = timeit(lambda: ops.fm_layer(embeds, sparse, w))
t_fused = timeit(lambda: ops.gather_fm(g, ids, buf[:, :pad + nd], wp, pad + nd, buf))
xs = torch.empty((B, F * D), device=dev)
g2 = ops.TableGroup([arena[f] for f in range(F)])
ops.gather_concat(g2, ids, out=xs)
W = torch.randn(3, F * D, device=dev) * 0.01
Bv = torch.randn(3, F * D, device=dev) * 0.01
t_cross = timeit(lambda: ops.cross_network(xs, W, Bv))
bytes_fused = B * (F * (2 * D * 4 + 4) + (nd + pad) * 4 + 4)
bytes_cross = B * F * D * 4 * 2
print(json.dumps({
    "shape": "65536 x 26 x dim128, V=1M",
    "gather_ms": round(t_g, 4), "fm_layer_ms": round(t_fm, 4), "gather_plus_fm_ms": round(t_g + t_fm, 4),
    "fused_gather_fm_ms": round(t_fused, 4), "fused_GBs": round(bytes_fused / t_fused / 1e6, 1),
    "fused_frac_of_8TBs": round(bytes_fused / t_fused / 1e6 / 8000, 4),
    "cross_3layers_ms": round(t_cross, 4), "cross_GBs": round(bytes_cross / t_cross / 1e6, 1),
    "cross_frac_of_8TBs": round(bytes_cross / t_cross / 1e6 / 8000, 4)}, indent=1))
