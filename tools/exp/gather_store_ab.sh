#!/bin/bash
# A/B of the output-store cache policy of the materialised gather (rebuilds gather.hip per arm on the box)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-0 1 2 3 0 1}; do
  touch recommend-tf2.0_amd/csrc/gather.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS=-DREC_GATHER_STORE=$k > gpurun_out/gst_build_$k.log 2>&1
  timeout -k 10 200 python bench.py --workload gather --cpu-seconds 0 > gpurun_out/gst_bench_$k.json 2> gpurun_out/gst_bench_$k.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/gst_bench_$k.json").read().strip().splitlines()[-1])
print("GATHER_STORE=$k ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], "frac_p50", r["roofline"]["frac_p50"])
PY
done
timeout -k 10 300 python -m pytest tests/test_gather_gpu.py -x -q 2>&1 | tail -2
