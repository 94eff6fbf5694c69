#!/bin/bash
# round 3: AutoInt one-launch kernel A/B on one box: stagger of the workgroups that share a CU (x 1024 cycles)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for st in 0 2 4 6 10 16; do
  timeout -k 10 120 python bench.py --workload autoint --steps 200 --warmup 20 --no-side --cpu-seconds 0 --force autoint_stagger=$st > /tmp/ai_$st.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ai_$st.json')); print('stagger $st rep $rep: %.2f us frac %.4f p50 %.1f' % (d['roofline']['ms_per_launch']*1e3, d['roofline']['frac'], d['roofline']['launch_us']['p50']))"
done
done
