"""bench.py's fused step ran at 175 us on an arena that the placement probe timed at 163 us.  Which step in between changes
it: burst vs sustained timing, freeing the other candidates, or the data the tables hold (zeros vs random)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
gen = torch.Generator(device=dev).manual_seed(0)
ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
out = torch.empty((B, 480), dtype=torch.float32, device=dev)[:, :479]
dense = torch.rand((B, D), device=dev, generator=gen)
def timeit(g, n, warm=2):
    for i in range(warm):
        ops.gather_pairwise_dot(g, ids[i % 8], dense, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.gather_pairwise_dot(g, ids[i % 8], dense, out=out)
    e1.record(); e1.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)
arenas = [torch.empty((F, V, D), dtype=torch.float32, device=dev).zero_() for _ in range(6)]
groups = [ops.TableGroup([a[f] for f in range(F)]) for a in arenas]
t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(groups[0], 20, 0)
print("zeros, burst 12   :", [timeit(g, 12) for g in groups], flush=True)
print("zeros, sustained 300:", [timeit(g, 300, 20) for g in groups], flush=True)
arenas[5].uniform_(-0.05, 0.05, generator=gen)
arenas[0].uniform_(-0.05, 0.05, generator=gen)
print("arenas 0 and 5 now random; burst 12:", [timeit(g, 12) for g in groups], flush=True)
print("sustained 300:", [timeit(g, 300, 20) for g in groups], flush=True)
for a in arenas[1:5]:
    a.uniform_(-0.05, 0.05, generator=gen)
print("all random; sustained 300:", [timeit(g, 300, 20) for g in groups], flush=True)
keep, g5 = arenas[5], groups[5]
del arenas, groups
torch.cuda.empty_cache()
print("others freed; arena 5 sustained 300 x3:", [timeit(g5, 300, 20) for _ in range(3)], flush=True)
