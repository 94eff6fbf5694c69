import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from recamd import ops
dev = torch.device("cuda:0")
G = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
rng = np.random.default_rng(3)
B, Nf, d, H, S = 64, 39, 16, 2, 16
x = (rng.normal(size=(B, Nf, d)) * 0.3).astype(np.float32)
Ws = [G(rng.normal(size=(d, H * S)) * 0.2) for _ in range(4)]
c1 = ops.mha_ctr(G(x), G(x), G(x), *Ws, H, S, "relu").cpu().numpy()
c2 = ops.mha_ctr(G(x), G(x), G(x), *Ws, H, S, "relu").cpu().numpy()
print("deterministic:", np.array_equal(c1, c2))
xb = x.copy(); xb[5, 3, 2] = np.inf
t = G(xb)
g = ops.mha_ctr(t, t, t, *Ws, H, S, "relu").cpu().numpy()
diff = [b for b in range(B) if not np.array_equal(g[b].view(np.uint32), c1[b].view(np.uint32))]
print("samples that differ:", diff)
for b in diff[:3]:
    print(b, "finite:", np.isfinite(g[b]).all(), "max abs diff", np.nanmax(np.abs(g[b] - c1[b])))
# separate-tensor call (xq, xk, xv distinct tensors) goes to a different kernel?
t1, t2, t3 = G(x), G(x), G(x)
print("same-tensor vs distinct-tensor identical:", np.array_equal(ops.mha_ctr(t1, t1, t1, *Ws, H, S, "relu").cpu().numpy(), c1))
