#!/bin/bash
# Round-N evidence for bench.py on the GPU box: rocprofv3 kernel stats + two separate PMC passes (FETCH_SIZE,
# WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md § rocprofv3 PMC slots).  Writes under gpurun_out/prof_<tag>/;
# tools/pmc_summary.py turns the CSVs into profiles/<tag>_pmc_traffic.json.
#   usage: tools/profile_bench.sh r02 [bench.py args...]
set -e
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 10 --cpu-seconds 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o bench -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
find "$OUT" -name "*.csv" | head -20
