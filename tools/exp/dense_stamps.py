"""round 3: where does a k-step of the hand-counted bf16x3 Dense kernel go?  Builds dense_bf16x3.hip with -DREC_DENSE_STAMPS
(s_memtime at the phase boundaries with scheduling barriers — the phases are kept apart, the product build overlaps them —
per-wave sums written over the first floats of each output tile) on the box, runs three layer shapes and prints the
per-wave phase shares.  Rebuilds the plain library afterwards."""
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CSRC = os.path.join(ROOT, "recommend-tf2.0_amd/csrc")


def build(flags):
    subprocess.check_call(["touch", os.path.join(CSRC, "dense_bf16x3.hip")])
    subprocess.check_call(["make", "-C", CSRC, "EXTRA_HIPFLAGS=" + flags], stdout=subprocess.DEVNULL)


build("-DREC_DENSE_STAMPS " + os.environ.get("DENSE_FLAGS", ""))
try:
    sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
    import numpy as np
    import torch
    from recamd import ops
    dev = torch.device("cuda:0")
    from recamd._lib import C
    C.debug_force("dense_pipe", "s")                      # the stamps live in the standard hand-counted kernel
    names = ["barrier -> operands read", "MFMA issue", "wait x pieces", "split + LDS writes", "wait W planes (DMA)", "barrier wait"]
    NP = len(names)
    for (M, K, N) in [(65536, 1024, 512), (65536, 512, 256), (8192, 4096, 4096)]:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(K, N, device=dev)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        for _ in range(5):
            ops.dense(x, W, b, "relu", out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.dense(x, W, b, "relu", out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        o = out.view(M // 128, 128, N // 128, 128)[:, :4, :, :NP + 1].permute(0, 2, 1, 3).reshape(-1, NP + 1).cpu().numpy()   # (tiles*4 waves, 6)
        tot = o[:, NP]
        nk = K // 16
        print(f"M={M} K={K} N={N}: {ms:.4f} ms with stamps; per wave: entry->done mean {tot.mean():.0f} ticks "
              f"({tot.mean() / nk:.0f} per k-step; 24 MFMAs = 768 matrix-pipe cycles)")
        for i, nm in enumerate(names):
            print(f"   {nm:28s} {o[:, i].mean() / nk:7.1f} ticks / k-step  ({100 * o[:, i].mean() / tot.mean():5.1f} %)   p90 {np.quantile(o[:, i], .9) / nk:7.1f}")
        print(f"   {'(epilogue + prologue)':28s} {(tot - o[:, :NP].sum(1)).mean():7.0f} ticks total")
finally:
    build(os.environ.get("DENSE_FLAGS", ""))
