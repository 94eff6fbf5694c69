#!/bin/bash
# A/B of the global-load prefetch distance of the bf16x3 Dense kernel (rebuilds dense_bf16x3.hip per arm on the box)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for pd in ${ARMS:-1 2 3 4}; do
  touch recommend-tf2.0_amd/csrc/dense_bf16x3.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS=-DREC_DENSE_PD=$pd > gpurun_out/dpd_build_$pd.log 2>&1
  echo "== PD=$pd"
  timeout -k 10 300 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "recommend-tf2.0_amd"))
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
for (M, K, N) in [(65536, 512, 256), (65536, 479, 1024), (65536, 1024, 1024), (65536, 1024, 512), (65536, 3456, 1024), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.dense(x, W, b, "relu", out=out))
    print(f"M={M} K={K} N={N}: {ms:.3f} ms {2.0*M*K*N/ms/1e9:.1f} TF")
PY
done
timeout -k 10 300 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -2
