#!/bin/bash
# MFMA utilisation of the matrix-core kernels (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE), its own counter pass.
#   usage: tools/profile_mfma.sh r02   -> gpurun_out/<tag>_pmc_mfma_util.json (+ raw csv)
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_mfma
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o w -- python3 "$ROOT/tools/exp/mfma_workload.py" > "$OUT/run.log" 2>&1
# calibration: the same counters over the pure-MFMA probe (tools/exp/mfma_peak: registers only / + LDS reads / + full LDS traffic)
if [ -x "$ROOT/tools/exp/mfma_peak" ]; then
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_probe" -o w -- "$ROOT/tools/exp/mfma_peak" 3 512 > "$OUT/probe.log" 2>&1 || echo "probe pass failed"
fi
python3 - "$OUT" "$ROOT/gpurun_out/${TAG}_pmc_mfma_util.json" <<'PY'
import collections, csv, glob, json, sys
out, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for sub in ("pmc", "pmc_probe"):
    for cc in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(cc)):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for kt in glob.glob(out + "/" + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(kt)):
            d = dur[r["Kernel_Name"]]
            d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
res = {}
for k, cs in agg.items():
    if ("rec::" not in k and "probe<" not in k) or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs:
        continue
    busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / cs["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    act = cs["GRBM_GUI_ACTIVE"][0] / cs["GRBM_GUI_ACTIVE"][1]
    if busy <= 0:
        continue
    name = k.split("(")[0].replace("void ", "")
    res[name] = {"dispatches": cs["GRBM_GUI_ACTIVE"][1], "SQ_VALU_MFMA_BUSY_CYCLES_avg": busy, "GRBM_GUI_ACTIVE_avg": act,
                 "duration_us_avg_under_counters": round(dur[k][0] / max(1, dur[k][1]), 1),
                 "mfma_utilisation": round(busy / (act / 8 * 1024), 4)}
json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/exp/mfma_workload.py "
                   "(MI355X). MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): fraction of "
                   "SIMD-cycles with the matrix pipe busy.", "kernels": res}, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
