"""The Dense kernels for aligned x / prepared W / K % 32 == 0 side by side on one box, interleaved:
default = f16x2 (three f16 MFMAs per product, row / column power-of-two scales, incl. its pass over x for the row maxima),
rec_debug_force("dense_pipe", "0") = the compiler-scheduled bf16x3 kernel, "s" = hand-counted bf16x3 (W planes by LDS-DMA,
x by whole lines), "d" = the same with the fragments one step ahead in registers.  The bf16x3 arms must be bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from recamd import ops
from recamd._lib import C
dev = torch.device("cuda:0")
ARMS = os.environ.get("ARMS", "h,0,s,d").split(",")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
torch.manual_seed(0)
for (M, K, N) in [(300, 64, 130), (4099, 96, 256), (65536, 512, 256), (65536, 480, 1024), (65536, 1024, 1024), (65536, 1024, 512),
                  (65536, 2048, 256), (65536, 3456, 128), (65536, 3360, 256), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    ref64 = torch.relu(x.double() @ W.double() + b.double()) if M * N <= 65536 * 1024 else None
    outs = {}
    for a in ARMS:
        C.debug_force("dense_pipe", None if a == "h" else a)
        outs[a] = ops.dense(x, W, b, "relu")
    torch.cuda.synchronize()
    b3 = [a for a in ARMS if a != "h"]
    same = all(bool(torch.equal(outs[b3[0]], outs[a])) for a in b3[1:]) if b3 else True
    errs = "" if ref64 is None else "  max err vs fp64: " + " ".join(f"{a}={float((outs[a].double() - ref64).abs().max()):.2e}" for a in ARMS)
    ms = {a: [] for a in ARMS}
    for rep in range(2):
        for a in ARMS:
            C.debug_force("dense_pipe", None if a == "h" else a)
            ms[a].append(t(lambda: ops.dense(x, W, b, "relu", out=outs[a])))
    C.debug_force("dense_pipe", None)
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N}: bf16x3 arms identical={same}  " + "  ".join(f"{a}: {min(ms[a]):.4f} ms ({fl / min(ms[a]) / 1e9:.1f} TF)" for a in ARMS) + errs, flush=True)
