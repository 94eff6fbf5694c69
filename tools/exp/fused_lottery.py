"""The fused gather + pairwise-dot kernel runs at ~166 us in some processes and ~178 us in others on the same box.  One
process, one arena: does the time depend on which allocation holds the result buffer / the dense rows / the ids, or does it
move over time with everything fixed?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
gen = torch.Generator(device=dev).manual_seed(0)
def mk_ids():
    return [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
def mk_out():
    return torch.empty((B, 480), dtype=torch.float32, device=dev)[:, :479]
ids0, out0, dense0 = mk_ids(), mk_out(), torch.rand((B, D), device=dev, generator=gen)
arenas = [torch.empty((F, V, D), dtype=torch.float32, device=dev).uniform_(-0.05, 0.05, generator=gen) for _ in range(3)]
groups = [ops.TableGroup([a[f] for f in range(F)]) for a in arenas]
def timeit(g, ids, dense, out, n=100, warm=10):
    for i in range(warm):
        ops.gather_pairwise_dot(g, ids[i % 8], dense, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.gather_pairwise_dot(g, ids[i % 8], dense, out=out)
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(groups[0], ids0, dense0, out0, 20, 0)
print("time series, everything fixed (arena 0):", [timeit(groups[0], ids0, dense0, out0) for _ in range(12)], flush=True)
print("by arena:", [[timeit(g, ids0, dense0, out0) for g in groups] for _ in range(3)], flush=True)
outs = [mk_out() for _ in range(6)]
print("by result buffer:", [[timeit(groups[0], ids0, dense0, o) for o in outs] for _ in range(2)], flush=True)
denses = [torch.rand((B, D), device=dev, generator=gen) for _ in range(4)]
print("by dense buffer:", [[timeit(groups[0], ids0, d, out0) for d in denses] for _ in range(2)], flush=True)
idss = [mk_ids() for _ in range(3)]
print("by id buffers:", [[timeit(groups[0], i, dense0, out0) for i in idss] for _ in range(2)], flush=True)
print("time series again:", [timeit(groups[0], ids0, dense0, out0) for _ in range(12)], flush=True)
