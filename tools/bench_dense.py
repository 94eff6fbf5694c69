import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from recamd import ops
from recamd._lib import C
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (M, K, N) in [(1638400, 64, 64), (1638400, 64, 128), (1638400, 128, 64), (65536, 128, 64), (65536, 13, 512), (65536, 512, 256), (65536, 479, 1024), (65536, 1024, 512), (65536, 3456, 128), (65536, 3341, 256), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.dense(x, W, b, "relu", out=out))
    C.debug_force("dense", "t")
    ms_t = t(lambda: ops.dense(x, W, b, "relu", out=out))
    C.debug_force("dense", "b")
    ms_b = t(lambda: ops.dense(x, W, b, "relu", out=out))
    C.debug_force("dense", None)
    ms_torch = t(lambda: torch.relu(torch.addmm(b, x, W)))
    fl = 2.0 * M * K * N
    by = 4.0 * (M * K + M * N + K * N)
    print(f"M={M} K={K} N={N}: ours {ms:.3f} ms ({fl/ms/1e9:.1f} TF, {by/ms/1e6:.0f} GB/s) fp32-MFMA tiled {ms_t:.3f} ms  bf16x3 {ms_b:.3f} ms ({fl/ms_b/1e9:.1f} TF)  torch {ms_torch:.3f} ms")
