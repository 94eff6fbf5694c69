#!/bin/bash
# round 3: landing buffers per wave of the SASRec one-launch kernel — attention phase x candidate phase (16 rows each);
# rebuilds sasrec_fused.hip per arm on the box, same box for all arms; the first arm is repeated at the end
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
# an arm is att:cand, att:cand:ku (row-load instructions per buffer, 4 rows each; default 4) 
for cfg in ${ARMS:-2:2 2:3 2:4 3:3 3:4 4:4 2:2}; do
  IFS=: read a c ku <<< "$cfg"
  ku=${ku:-4}
  touch recommend-tf2.0_amd/csrc/sasrec_fused.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_SASREC_ATT_BUFS=$a -DREC_SASREC_CAND_BUFS=$c -DREC_SASREC_KU=$ku" > gpurun_out/sb_build_${a}_$c.log 2>&1
  timeout -k 10 200 python bench.py --workload sasrec --cpu-seconds 0 --no-side > gpurun_out/sb_${a}_$c.json 2> gpurun_out/sb_${a}_$c.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/sb_${a}_$c.json").read().strip().splitlines()[-1])
print("att=$a cand=$c ku=$ku: ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
  # every arm is checked against the oracle (an arm that skips loads is fast and wrong)
  timeout -k 10 300 python -m pytest tests/test_sasrec_fused_gpu.py -x -q 2>&1 | tail -1
done
