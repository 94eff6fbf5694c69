#!/bin/bash
# A/B of compile-time variants of the hand-counted bf16x3 Dense kernel: ARMS="flagsA|flagsB|..." (interleaved REPS times)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
IFS='|' read -ra ARMS_A <<< "${ARMS:-|-DREC_DENSE_PIPE_WG=2}"
for rep in $(seq 1 ${REPS:-2}); do
  for k in "${ARMS_A[@]}"; do
    touch recommend-tf2.0_amd/csrc/dense_bf16x3.hip
    make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="$k" > gpurun_out/dflags_build.log 2>&1
    echo "== flags '$k' rep $rep"
    timeout -k 10 300 python tools/bench_dense.py 2>/dev/null | grep -E "K=512 N=256|K=1024 N=512|K=3456 N=128|K=4096 N=4096" | cut -c1-60
  done
done
touch recommend-tf2.0_amd/csrc/dense_bf16x3.hip
make -C recommend-tf2.0_amd/csrc > gpurun_out/dflags_build.log 2>&1
timeout -k 10 300 python -m pytest tests/test_dense_gpu.py -x -q 2>&1 | tail -2
