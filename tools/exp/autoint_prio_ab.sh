#!/bin/bash
# round 3: AutoInt one-launch kernel — wave priority by phase (REC_AUTOINT_PRIO: 0 none, 1 VALU phases high, 2 matrix
# bursts high) x workgroups per CU the register budget is cut for x query tiles batched per phase (REC_AUTOINT_BATCHQT); rebuilds
# attention_ctr.hip per arm on the box.  ARMS = "prio:minwg:batchqt ..."  (round-3 arms with the third field = next-sample
# prefetch, since removed: profiles/r03_autoint_prio_ab2.txt)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in ${ARMS:-0:3:0 1:3:0 2:3:0 0:3:0 1:3:0}; do
  IFS=: read pr wg pf <<< "$cfg"
  touch recommend-tf2.0_amd/csrc/attention_ctr.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_AUTOINT_PRIO=$pr -DREC_AUTOINT_MINWG=$wg " > gpurun_out/ap_build.log 2>&1
  timeout -k 10 200 python bench.py --workload autoint --steps 200 --warmup 20 --cpu-seconds 0 --no-side > gpurun_out/ap.json 2> gpurun_out/ap.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/ap.json").read().strip().splitlines()[-1])
print("prio=$pr minwg=$wg batchqt=$pf: us", round(r["roofline"]["ms_per_launch"] * 1e3, 2), "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
done
timeout -k 10 300 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -2
