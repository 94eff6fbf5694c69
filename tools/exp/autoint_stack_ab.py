import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")):
    sys.path.insert(0, p)
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, N, H, S = 4096, 39, 2, 16
x = torch.randn((B, N, 16), device=dev) * 0.5
layers = []
for l in range(3):
    k = 16 if l == 0 else 32
    layers.append(tuple(torch.randn((k, 32), device=dev) / k ** 0.5 for _ in range(4)))
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def seq():
    h = x
    for (q, k, v, r) in layers:
        h = ops.mha_ctr(h, h, h, q, k, v, r, H, S, "relu")
    return h
print("stack us:", t(lambda: ops.mha_ctr_stack(x, layers, H, S, "relu")))
print("3 launches us:", t(seq))
print("one layer (din 16) us:", t(lambda: ops.mha_ctr(x, x, x, *layers[0], H, S, "relu")))
print("max diff:", (ops.mha_ctr_stack(x, layers, H, S, "relu") - seq()).abs().max().item())
