#!/bin/bash
# rocprofv3 per-kernel stats of the other BASELINE workloads (bench.py --workload autoint|din|sasrec)
#   usage: tools/profile_workloads.sh r02 ; summaries land in gpurun_out/prof_<tag>_<workload>/
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for wl in autoint din sasrec; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$wl
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --workload $wl --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/stats.err"
  f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$ROOT/gpurun_out/${TAG}_kernel_stats_$wl.csv"
  cp "$OUT/bench.json" "$ROOT/gpurun_out/${TAG}_profiled_run_$wl.json"
  echo "== $wl"; head -12 "$f" | cut -d, -f1-4 | cut -c1-160
done
