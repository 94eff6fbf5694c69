#!/bin/bash
# A/B of the row-load cache policy of the materialised gather, uniform and Zipf ids (rebuilds gather.hip per arm)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-0 1 0 1}; do
  touch recommend-tf2.0_amd/csrc/gather.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS=-DREC_GATHER_LOAD_NT=$k > gpurun_out/gld_build_$k.log 2>&1
  for ids in uniform zipf; do
    timeout -k 10 200 python bench.py --workload gather --ids $ids --cpu-seconds 0 > gpurun_out/gld_${ids}_$k.json 2> gpurun_out/gld_${ids}_$k.err
    python - <<PY
import json
r = json.loads(open("gpurun_out/gld_${ids}_$k.json").read().strip().splitlines()[-1])
print("LOAD_NT=$k ids=$ids ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], "frac_p50", r["roofline"]["frac_p50"], flush=True)
PY
  done
done
timeout -k 10 300 python -m pytest tests/test_gather_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
