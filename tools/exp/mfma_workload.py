"""Matrix-core kernels at their BASELINE sizes, for the MFMA-utilisation counter pass (tools/profile_mfma.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
from recamd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
# Dense layers of the DLRM top MLP (pipelined bf16x3 kernel) and a square GEMM
for (M, K, N) in [(65536, 1024, 1024), (65536, 1024, 512), (65536, 480, 1024), (65536, 3360, 256), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(12):
        ops.dense(x, W, b, "relu", out=out)
# SASRec full-block attention at config 5 size (8192 x S=200 x d=64, one head)
B, S, d = 8192, 200, 64
q = torch.randn(B, S, d, device=dev); k = torch.randn(B, S, d, device=dev); v = torch.randn(B, S, d, device=dev)
m = torch.ones(B, S, device=dev)
for _ in range(12):
    ops.mha_rowmask(q, k, v, m, 1)
# AutoInt interacting stack at config 3 size
Bc, N, H = 4096, 39, 2
xa = torch.randn((Bc, N, 16), device=dev) * 0.5
layers = []
for l in range(3):
    kk = 16 if l == 0 else 32
    layers.append(tuple(torch.randn((kk, 32), device=dev) / kk ** 0.5 for _ in range(4)))
for _ in range(50):
    ops.mha_ctr_stack(xa, layers, H, 16, "relu")
torch.cuda.synchronize()
