#!/bin/bash
# A/B of the row-load cache policy of the fused SASRec kernel (the DIN arm lost: see csrc/common.h row_load) (rebuilds the two files per arm on the box)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${ARMS:-0 1 0 1}; do
  touch recommend-tf2.0_amd/csrc/attention.hip recommend-tf2.0_amd/csrc/sasrec_fused.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS=-DREC_SASREC_ROWS_NT=$k > gpurun_out/rnt_build_$k.log 2>&1
  for wl in sasrec; do
    timeout -k 10 200 python bench.py --workload $wl --cpu-seconds 0 > gpurun_out/rnt_${wl}_$k.json 2> gpurun_out/rnt_${wl}_$k.err
    python - <<PY
import json
r = json.loads(open("gpurun_out/rnt_${wl}_$k.json").read().strip().splitlines()[-1])
print("ROWS_NT=$k $wl ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
  done
done
timeout -k 10 400 python -m pytest tests/test_attention_gpu.py tests/test_sasrec_fused_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -2
