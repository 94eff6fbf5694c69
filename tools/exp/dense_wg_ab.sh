#!/bin/bash
# A/B of the pipelined bf16x3 Dense kernel's register budget: 2 (172 VGPRs) vs 3 (168 VGPRs + 1 spilled) workgroups per CU
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-2 3 2 3}; do
  touch recommend-tf2.0_amd/csrc/dense_bf16x3.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_DENSE_PIPE_WG=$k" > gpurun_out/dwg_build_$k.log 2>&1
  echo "== REC_DENSE_PIPE_WG=$k"
  timeout -k 10 300 python tools/bench_dense.py 2>/dev/null | grep -E "K=512 N=256|K=1024 N=512|K=3456 N=128|K=4096 N=4096|K=128 N=64" | cut -c1-90
done
timeout -k 10 300 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -2
