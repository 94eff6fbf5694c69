#!/bin/bash
# Round-N evidence for bench.py on the GPU box: rocprofv3 kernel stats + two separate PMC passes (FETCH_SIZE,
# WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md § rocprofv3 PMC slots) for the headline workload and for the
# DIN / SASRec workloads.  Writes under gpurun_out/prof_<tag>[_<workload>]/; tools/pmc_summary.py turns the CSVs into
# profiles/<tag>_pmc_traffic[_<workload>].json + profiles/<tag>_*kernel_stats*.csv.
#   usage: tools/profile_bench.sh r03 [workloads...]        (default: dlrm_fused din sasrec autoint)
set -e
TAG=${1:-r03}; shift || true
WLS=${@:-dlrm_fused din sasrec autoint}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$wl
  mkdir -p "$OUT"
  ARGS="--workload $wl --steps 50 --warmup 10 --cpu-seconds 0 --no-side"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
  if [ "$wl" != "autoint" ]; then
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
  fi
  echo "== $wl"; f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -d, -f1-4 | cut -c1-150
done
# the materialised gather beside the headline (same shape): its own three passes
OUT=$ROOT/gpurun_out/prof_${TAG}_gather; mkdir -p "$OUT"
ARGS="--workload gather --steps 50 --warmup 10 --cpu-seconds 0 --no-side"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
