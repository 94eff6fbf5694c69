#!/bin/bash
# round 3: 8 fresh processes per arm, interleaved: plain allocation vs a 32-GiB spacer held while the tables are allocated
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_skip2; mkdir -p $OUT
for rep in 1 2 3 4 5 6 7 8; do
  for skip in 0 32; do
    timeout -k 10 120 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-side --arena-skip-gb $skip > $OUT/skip${skip}_$rep.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/skip${skip}_$rep.json')); print('skip $skip rep $rep: fused %.1f us frac %.4f p50 %.1f' % (d['roofline']['ms_per_launch']*1e3, d['roofline']['frac'], d['roofline']['launch_us']['p50']))"
  done
done
