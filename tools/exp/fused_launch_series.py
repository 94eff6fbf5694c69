"""per-launch times of the fused kernel over the 8 rotating id batches: is the p10-p90 spread of some processes (167-182 us)
tied to particular id batches (period 8), to time, or random?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
import torch
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
gen = torch.Generator(device=dev).manual_seed(1)
ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
dense = torch.rand((B, D), device=dev, generator=gen)
out = torch.empty((B, 480), dtype=torch.float32, device=dev)[:, :479]
probe = lambda g, i: ops.gather_pairwise_dot(g, ids[i % 8], dense, out=out)
arena, info = ops.place_table_arena(F, V, D, dev, candidates=6, probe=probe)
print(info, flush=True)
arena.uniform_(-0.05, 0.05, generator=gen)
g = ops.TableGroup([arena[f] for f in range(F)])
t0 = time.time()
i = 0
while time.time() - t0 < 0.4:
    probe(g, i); i += 1
torch.cuda.synchronize()
for rep in range(3):
    n = 64
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        probe(g, i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    print("rep", rep, "by id batch (mean over 8 visits):", [round(sum(us[k::8]) / 8, 1) for k in range(8)], flush=True)
    print("   series:", [round(u) for u in us], flush=True)
# the same 64 launches on ONE id batch
for k in (0, 3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(33)]
    ev[0].record()
    for i in range(32):
        ops.gather_pairwise_dot(g, ids[k], dense, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    print(f"only id batch {k}:", [round(ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(32)], flush=True)
