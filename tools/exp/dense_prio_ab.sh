#!/bin/bash
# round 3: bf16x3 Dense kernels, the MFMA cluster of a k-step at wave priority REC_DENSE_PRIO (0 = no priority changes)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-0 1 3 0 1}; do
  touch recommend-tf2.0_amd/csrc/dense_bf16x3.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_DENSE_PRIO=$k" > gpurun_out/dp_build_$k.log 2>&1
  echo "== REC_DENSE_PRIO=$k"
  timeout -k 10 300 python tools/bench_dense.py 2>/dev/null | grep -E "K=512 N=256|K=1024 N=512|K=3456 N=128|K=4096 N=4096" | cut -c1-70
done
timeout -k 10 300 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -2
