"""Dev microbenchmark: fused gather + pairwise-dot at the BASELINE config-2 shape."""
import argparse
import os
import sys

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
a = ap.parse_args()
dev = torch.device("cuda:0")
arena = torch.empty((a.F, a.V, a.D), dtype=torch.float32, device=dev).uniform_(-0.05, 0.05)
g = ops.TableGroup([arena[f] for f in range(a.F)])
gen = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(0, a.V, (a.B, a.F), device=dev, dtype=torch.int32, generator=gen)
dense = torch.rand((a.B, a.D), device=dev)
n = a.F + 1
P = n * (n - 1) // 2
out = torch.empty((a.B, (P + a.D + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :P + a.D]


def timeit(fn, name, byts):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
    med = ts[len(ts) // 2]
    print(f"{name}: median {med*1e3:.1f} us min {ts[0]*1e3:.1f} us  {byts/med/1e9:.3f} TB/s "
          f"({byts/med/1e9/8*100:.1f}% of 8 TB/s)  {a.B/med/1e3:.1f} M samples/s")


fused_bytes = a.B * (a.F * a.D * 4 + a.F * 4 + a.D * 4 + (P + a.D) * 4)
timeit(lambda: ops.gather_pairwise_dot(g, ids, dense, out=out), "fused gather+dot", fused_bytes)
X = torch.cat([ops.gather_concat(g, ids).view(a.B, a.F, a.D), dense[:, None, :]], dim=1).contiguous()
out2 = torch.empty((a.B, (P + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :P]
timeit(lambda: ops.pairwise_dot(X, out=out2), "plain pairwise_dot", a.B * (n * a.D * 4 + P * 4))
ref = torch.bmm(X[:4096], X[:4096].transpose(1, 2))
li, lj = zip(*[(i, j) for i in range(n) for j in range(i)])
refz = ref[:, list(li), list(lj)]
print("max abs err vs torch.bmm (first 4096):", float((refz - out[:4096, :P]).abs().max()))
# A/B in one process: staged (16-B aligned row stride) vs direct (tight 479-float rows) output path
out_tight = torch.empty((a.B, P + a.D), dtype=torch.float32, device=dev)
for rep in range(2):
    timeit(lambda: ops.gather_pairwise_dot(g, ids, dense, out=out), "fused, staged stores (stride 480)", fused_bytes)
    timeit(lambda: ops.gather_pairwise_dot(g, ids, dense, out=out_tight), "fused, direct stores (stride 479)", fused_bytes)
print("staged == direct:", bool(torch.equal(out, out_tight)))
