#!/bin/bash
# A/B of the sequence-row batch size of the one-launch SASRec kernel (rebuilds only sasrec_fused.hip per arm)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-6 8 4}; do
  touch recommend-tf2.0_amd/csrc/sasrec_fused.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS=-DREC_SASREC_KU=$k > gpurun_out/kus_build_$k.log 2>&1
  timeout -k 10 200 python -m pytest tests/test_sasrec_fused_gpu.py -x -q > gpurun_out/kus_test_$k.log 2>&1 || { tail -20 gpurun_out/kus_test_$k.log; exit 1; }
  timeout -k 10 200 python bench.py --workload sasrec > gpurun_out/kus_bench_$k.json 2> gpurun_out/kus_bench_$k.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/kus_bench_$k.json").read().strip().splitlines()[-1])
print("KUS=$k ms_per_step", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"])
PY
done
