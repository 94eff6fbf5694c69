"""Why did bench.py's gather run at 327 us on an arena whose placement probe read 309 us?  One process: 6 arenas alive, then
(a) the probe as place_table_arena does it (12 launches each), (b) 200-launch timings on each arena, twice, (c) the bench's
own sequence on the best one: free the others, allocate fresh ids / out, time again."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
gen = torch.Generator(device=dev).manual_seed(0)
ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
out = torch.empty((B, F * D), dtype=torch.float32, device=dev)
def timeit(g, o, idl, n, warm=4):
    for i in range(warm):
        ops.gather_concat(g, idl[i % len(idl)], out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.gather_concat(g, idl[i % len(idl)], out=o)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
arenas = [torch.empty((F, V, D), dtype=torch.float32, device=dev).zero_() for _ in range(6)]
groups = [ops.TableGroup([a[f] for f in range(F)]) for a in arenas]
print("probe  (12 launches):", [round(timeit(g, out, ids[:4], 12), 1) for g in groups], flush=True)
print("probe2 (12 launches):", [round(timeit(g, out, ids[:4], 12), 1) for g in groups], flush=True)
for rep in range(2):
    print(f"long {rep} (200 launches):", [round(timeit(g, out, ids, 200, 20), 1) for g in groups], flush=True)
print("probe3 (12 launches):", [round(timeit(g, out, ids[:4], 12), 1) for g in groups], flush=True)
best = 4
keep = arenas[best]
del arenas, groups
torch.cuda.empty_cache()
g = ops.TableGroup([keep[f] for f in range(F)])
print("kept arena, same out/ids, 200:", round(timeit(g, out, ids, 200, 20), 1), flush=True)
ids2 = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
out2 = torch.empty((B, F * D), dtype=torch.float32, device=dev)
print("kept arena, fresh out/ids, 200:", round(timeit(g, out2, ids2, 200, 20), 1), flush=True)
keep.uniform_(-0.05, 0.05, generator=gen)
print("after uniform_ fill, 200:", round(timeit(g, out2, ids2, 200, 20), 1), flush=True)
print("zero_ again, 200:", round(timeit(g, out2, ids2, 200, 20) if keep.zero_() is not None else 0, 1), flush=True)
