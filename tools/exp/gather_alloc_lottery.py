"""Is the materialised gather's run-to-run bimodality (307 vs 327 us in separate processes on ONE box) a property of the
allocation?  One process: several output buffers and several table arenas (all kept alive, so every one has its own
address range), the same ids, the gather timed into every (arena, out) pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
gen = torch.Generator(device=dev).manual_seed(0)
ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
arenas, outs = [], []
for a in range(3):
    t = torch.empty((F, V, D), dtype=torch.float32, device=dev)
    t.uniform_(-0.05, 0.05, generator=gen)
    arenas.append(t)
    for o in range(2):
        outs.append(torch.empty((B, F * D), dtype=torch.float32, device=dev))
    pad = torch.empty(B * F * D + (a + 1) * 3_700_000, dtype=torch.float32, device=dev)   # shifts the next allocations
    off = (a + 1) * 1_234_560                                                  # a 16-B aligned offset inside it
    outs.append(pad[off:off + B * F * D].view(B, F * D))
def timeit(group, out, n=60):
    for i in range(20):
        ops.gather_concat(group, ids[i % 8], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.gather_concat(group, ids[i % 8], out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t0 = time.time()
while time.time() - t0 < 0.5:
    timeit(ops.TableGroup([arenas[0][f] for f in range(F)]), outs[0], 10)
for ai, ar in enumerate(arenas):
    g = ops.TableGroup([ar[f] for f in range(F)])
    for oi, o in enumerate(outs):
        us = timeit(g, o)
        print(f"arena {ai} @ {ar.data_ptr():#x}  out {oi} @ {o.data_ptr():#x} (mod 2MiB {o.data_ptr() % (2 << 20):#x}): {us:6.1f} us  frac {1751646208 / us / 1e3 / 8000:.4f}", flush=True)
