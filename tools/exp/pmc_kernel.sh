#!/bin/bash
# per-kernel SQ counters of a python script: tools/exp/pmc_kernel.sh <tag> <script.py> [args]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/a" -o p -- python3 "$ROOT/$@" > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o p -- python3 "$ROOT/$@" > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f[0])):
        a = agg[r["Kernel_Name"][:60]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, cs in agg.items():
        if "rec::" not in k: continue
        print(k, {c: round(v[0] / v[1]) for c, v in cs.items()})
PY
