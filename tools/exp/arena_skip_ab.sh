#!/bin/bash
# round 3: does keeping the table arena out of the low end of a fresh process's device memory (a spacer allocated first,
# freed afterwards) remove the slow first allocation?  Fresh process per run, arms interleaved.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_skip; mkdir -p $OUT
for rep in 1 2 3; do
  for skip in 0 16 32 64; do
    timeout -k 10 120 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-side --arena-skip-gb $skip > $OUT/skip${skip}_$rep.json 2>/dev/null
    python - <<P
import json
d=json.load(open("$OUT/skip${skip}_$rep.json"))
print("skip $skip rep $rep: fused %.1f us frac %.4f p50 %.1f" % (d["roofline"]["ms_per_launch"]*1e3, d["roofline"]["frac"], d["roofline"]["launch_us"]["p50"]))
P
  done
done
for skip in 0 32; do
  timeout -k 10 120 python bench.py --workload gather --steps 100 --warmup 10 --cpu-seconds 0 --no-side --arena-skip-gb $skip > $OUT/gather_skip${skip}.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/gather_skip${skip}.json')); print('gather skip $skip: %.1f us frac %.4f' % (d['roofline']['ms_per_launch']*1e3, d['roofline']['frac']))"
done
