#!/bin/bash
# round 3: 8 plain hipMalloc arenas side by side, per-arena UTCL1 / latency / DRAM-request counters (dispatch order = print order)
cd "$(dirname "$0")"
OUT=$(pwd)/../../gpurun_out/r03_place2
mkdir -p $OUT
export TMPDIR=/tmp
P=$(pwd)/placement_probe
cd /tmp
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$tag -o p -- $P malloc 8 4 > $OUT/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
