#!/bin/bash
# A/B of the AutoInt one-launch kernel's occupancy: workgroups per CU the register budget is cut for (and that are launched)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-2 3 2 3}; do
  touch recommend-tf2.0_amd/csrc/attention_ctr.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_AUTOINT_MINWG=$k" > gpurun_out/aiwg_build_$k.log 2>&1
  timeout -k 10 200 python bench.py --workload autoint --cpu-seconds 0 > gpurun_out/aiwg_$k.json 2> gpurun_out/aiwg_$k.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/aiwg_$k.json").read().strip().splitlines()[-1])
print("MINWG=$k autoint ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
done
timeout -k 10 300 python -m pytest tests/test_attention_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -2
