#!/bin/bash
# A/B of the DIN pooling kernel's row loads: REC_DIN_SKIP (lane groups without a slot issue nothing) x REC_DIN_NT
# (streaming policy); rebuilds attention.hip per arm on the box
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for cfg in ${ARMS:-00 02 00 02}; do
  s=${cfg:0:1}; n=${cfg:1:1}
  touch recommend-tf2.0_amd/csrc/attention.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_DIN_SKIP=$s -DREC_DIN_NT=$n" > gpurun_out/dinab_build_$cfg.log 2>&1
  timeout -k 10 200 python bench.py --workload din --cpu-seconds 0 > gpurun_out/dinab_$cfg.json 2> gpurun_out/dinab_$cfg.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/dinab_$cfg.json").read().strip().splitlines()[-1])
print("SKIP=$s NT=$n din ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
done
timeout -k 10 400 python -m pytest tests/test_attention_gpu.py tests/test_models_gpu.py tests/test_edge_cases_gpu.py tests/test_nonfinite_gpu.py -x -q 2>&1 | tail -2
