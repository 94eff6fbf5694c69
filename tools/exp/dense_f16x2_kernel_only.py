"""The f16x2 Dense kernel with and without its pass over x (row maxima handed over = what a Dense -> Dense chain does),
against the hand-counted bf16x3 kernel, interleaved on one box."""
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
torch.manual_seed(0)
for (M, K, N) in [(65536, 512, 256), (65536, 480, 1024), (65536, 1024, 1024), (65536, 1024, 512), (65536, 256, 128), (65536, 2048, 256), (65536, 3360, 256),
                  (65536, 3456, 128), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    am = x.abs().amax(dim=1).contiguous()
    res = {}
    for rep in range(2):
        C.debug_force("dense_pipe", "h")
        a = t(lambda: ops.dense(x, W, b, "relu", out=out))
        v = t(lambda: ops.dense(x, W, b, "relu", out=out, row_absmax=am))
        oam = torch.zeros(M, device=dev)
        w = t(lambda: ops.dense(x, W, b, "relu", out=out, row_absmax=am, out_absmax=oam))
        C.debug_force("dense_pipe", "s")
        s_ = t(lambda: ops.dense(x, W, b, "relu", out=out))
        C.debug_force("dense_pipe", None)
        for k, val in (("f16x2 + pass", a), ("f16x2, maxima handed over", v), ("... and delivering the output's", w), ("bf16x3 hand-counted", s_)):
            res[k] = min(res.get(k, 1e9), val)
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N}: " + "  ".join(f"{k}: {v:.4f} ms ({fl / v / 1e9:.0f} TF)" for k, v in res.items()), flush=True)
