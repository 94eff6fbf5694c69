"""Follow-up of gather_alloc_lottery.py: the time of the gather AND of the fused gather + pairwise-dot kernel depends on
WHICH allocation holds the 13.3 GB table arena (same kernel, ids, outputs: 308 vs 329 us).  N arenas kept alive in one
process, both kernels timed on each; then the arenas are freed and re-allocated in one block to see whether a fresh
allocation of the same memory behaves the same."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
N = int(os.environ.get("N_ARENAS", "8"))
gen = torch.Generator(device=dev).manual_seed(0)
ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
dense = torch.rand((B, D), device=dev, generator=gen)
out_g = torch.empty((B, F * D), dtype=torch.float32, device=dev)
out_f = torch.empty((B, 480), dtype=torch.float32, device=dev)[:, :479]
def timeit(fn, n=60):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def measure(tag, ar):
    g = ops.TableGroup([ar[f] for f in range(F)])
    tg = timeit(lambda i: ops.gather_concat(g, ids[i % 8], out=out_g))
    tf = timeit(lambda i: ops.gather_pairwise_dot(g, ids[i % 8], dense, out=out_f))
    print(f"{tag} @ {ar.data_ptr():#x}: gather {tg:6.1f} us ({1751646208 / tg / 8e6:.4f})  fused {tf:6.1f} us ({1038352384 / tf / 8e6:.4f})", flush=True)
arenas = []
for a in range(N):
    t = torch.empty((F, V, D), dtype=torch.float32, device=dev)
    t.uniform_(-0.05, 0.05, generator=gen)
    arenas.append(t)
t0 = time.time()
g0 = ops.TableGroup([arenas[0][f] for f in range(F)])
while time.time() - t0 < 0.5:
    timeit(lambda i: ops.gather_concat(g0, ids[i % 8], out=out_g), 10)
for rep in range(2):
    for a, ar in enumerate(arenas):
        measure(f"pass {rep} arena {a}", ar)
del arenas, g0, ar
torch.cuda.empty_cache()
for a in range(3):
    t = torch.empty((F, V, D), dtype=torch.float32, device=dev)
    t.uniform_(-0.05, 0.05, generator=gen)
    measure(f"after free, fresh arena {a}", t)
    del t
    torch.cuda.empty_cache()
