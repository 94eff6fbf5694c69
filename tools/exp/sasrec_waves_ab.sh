#!/bin/bash
# A/B of the SASRec one-launch kernel's geometry: waves per workgroup (one workgroup per CU) x row-load instructions per
# landing buffer (bytes in flight per CU = waves x 2 x kU KiB); rebuilds sasrec_fused.hip per arm
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for cfg in ${ARMS:-16:4 12:6 8:8 8:12 8:16 16:4}; do
  w=${cfg%%:*}; u=${cfg##*:}
  touch recommend-tf2.0_amd/csrc/sasrec_fused.hip
  make -C recommend-tf2.0_amd/csrc EXTRA_HIPFLAGS="-DREC_SASREC_WAVES=$w -DREC_SASREC_KU=$u" > gpurun_out/sw_build_$w_$u.log 2>&1
  timeout -k 10 200 python bench.py --workload sasrec --cpu-seconds 0 > gpurun_out/sw_${w}_$u.json 2> gpurun_out/sw_${w}_$u.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/sw_${w}_$u.json").read().strip().splitlines()[-1])
print("waves=$w kU=$u (in flight per CU %d KiB): ms" % ($w * 2 * $u), r["ms_per_step"], "frac", r["roofline"]["frac"], "p50", r["roofline"]["launch_us"]["p50"], flush=True)
PY
done
timeout -k 10 300 python -m pytest tests/test_sasrec_fused_gpu.py -x -q 2>&1 | tail -2
