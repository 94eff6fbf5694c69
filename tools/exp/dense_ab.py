import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from recamd import ops
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (M, K, N) in [(65536, 512, 256), (65536, 1024, 512), (65536, 1024, 1024), (65536, 3456, 1024), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.dense(x, W, b, "relu", out=out))
    print(f"M={M} K={K} N={N}: {ms:.3f} ms ({2.0*M*K*N/ms/1e9:.1f} TF)", flush=True)
