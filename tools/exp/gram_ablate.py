import os, sys, subprocess
# each config in a fresh process (env read at dispatch, but keep it simple)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
from recamd import ops
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
arena = torch.empty((F, V, D), device=dev).uniform_(-0.05, 0.05)
g = ops.TableGroup([arena[f] for f in range(F)])
ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
dense = torch.rand((B, D), device=dev)
out = torch.empty((B, 480), device=dev)[:, :479]
def t(name):
    for _ in range(5): ops.gather_pairwise_dot(g, ids, dense, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): ops.gather_pairwise_dot(g, ids, dense, out=out)
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b)/30*1e3:.1f} us", flush=True)
os.environ["REC_PAIRDOT_IMPL"] = "gram"
for cfg in ("os", "22"):
    os.environ["REC_GRAM_CFG"] = cfg
    for dbg, nm in ((0, "full"),):
        os.environ["REC_GRAM_DBG"] = str(dbg)
        t(f"cfg {cfg} dbg {dbg} ({nm})")
os.environ["REC_PAIRDOT_IMPL"] = ""
t("valu kernel")
