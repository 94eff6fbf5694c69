"""Dev microbenchmark: K1 gather at the BASELINE config-2 shape (65536 x 26 x 128, V=1M)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
from recamd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=65536)
ap.add_argument("--F", type=int, default=26)
ap.add_argument("--V", type=int, default=1_000_000)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--zipf", type=float, default=0.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
arena = torch.empty((a.F, a.V, a.D), dtype=torch.float32, device=dev).uniform_(-0.05, 0.05)
g = ops.TableGroup([arena[f] for f in range(a.F)])
gen = torch.Generator(device=dev).manual_seed(1)
if a.zipf > 0:
    import numpy as np
    z = np.random.default_rng(1).zipf(a.zipf, size=(a.B, a.F))
    ids = torch.from_numpy(((z - 1) % a.V).astype("int32")).to(dev)
else:
    ids = torch.randint(0, a.V, (a.B, a.F), device=dev, dtype=torch.int32, generator=gen)
out = torch.empty((a.B, a.F * a.D), dtype=torch.float32, device=dev)
for _ in range(5):
    ops.gather_concat(g, ids, out=out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    ops.gather_concat(g, ids, out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
med = ts[len(ts) // 2]
byts = a.B * a.F * (2 * a.D * 4 + 4)
print(f"gather B={a.B} F={a.F} V={a.V} D={a.D} zipf={a.zipf}: median {med*1e3:.1f} us  min {ts[0]*1e3:.1f} us  "
      f"{byts/med/1e9:.3f} TB/s  ({byts/med/1e9/8*100:.1f}% of 8 TB/s)  {a.B/med/1e3:.1f} M samples/s")
ref = torch.cat([arena[f][ids[:, f].long()] for f in range(a.F)], dim=1)
print("bit-exact vs torch index:", bool(torch.equal(ref, out)))
