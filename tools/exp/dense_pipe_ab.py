"""bf16x3 Dense: hand-counted load pipeline (default) vs the compiler-scheduled kernel (REC_DENSE_PIPE=0, child process)."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))


def run():
    import torch
    from recamd import ops
    dev = torch.device("cuda:0")

    def t(fn, it=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    for (M, K, N) in [(65536, 512, 256), (65536, 256, 128), (65536, 1024, 1024), (65536, 1024, 512), (65536, 3456, 1024),
                      (8192, 4096, 4096), (4096, 1024, 1024)]:
        x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        ms = t(lambda: ops.dense(x, W, b, "relu", out=out))
        ref = torch.relu(torch.addmm(b.double(), x.double(), W.double())).float() if M * K <= 2 ** 26 else None
        err = float((out - ref).abs().max() / ref.abs().max()) if ref is not None else -1
        print(f"M={M} K={K} N={N}: {ms:.3f} ms {2.0*M*K*N/ms/1e9:.1f} TF  rel err {err:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for v in ("1", "0"):
            print("REC_DENSE_PIPE=" + v, flush=True)
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, REC_DENSE_PIPE=v))
