#!/bin/bash
# A/B of the DIN pooling kernel's batch depth U (4 U slots per batch; VGPRs 94 / 120 -> 5 / 4 waves per SIMD)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for u in ${ARMS:-2 1 2 1}; do
  touch recommend-tf2.0_amd/csrc/attention.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_DIN_U=$u" > gpurun_out/dinu_build_$u.log 2>&1
  timeout -k 10 200 python bench.py --workload din --cpu-seconds 0 > gpurun_out/dinu_$u.json 2> gpurun_out/dinu_$u.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/dinu_$u.json").read().strip().splitlines()[-1])
print("U=$u din ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
done
timeout -k 10 300 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -2
