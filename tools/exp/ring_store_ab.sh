#!/bin/bash
# A/B of the result-store cache policies of the fused gather + pairwise-dot ring kernel (experiment build on the box)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
touch recommend-tf2.0_amd/csrc/pairwise_dot_ring.hip
make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS=-DREC_RING_EXPERIMENTS > gpurun_out/ring_store_build.log 2>&1
RING_AB_REPS=2 timeout -k 10 500 python tools/exp/ring_ab.py child 0 43 51 52 53 54 43 > gpurun_out/ring_store_ab.txt 2>&1
cat gpurun_out/ring_store_ab.txt
