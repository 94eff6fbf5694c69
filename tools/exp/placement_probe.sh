#!/bin/bash
# round 3: allocation-mechanism A/B for the table arena + UTCL1 (TLB) / latency counters per arena
set -e
cd "$(dirname "$0")"
OUT=../../gpurun_out/r03_place
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 ./placement_probe malloc,contig,vmm1024,vmm64,vmm2,frag+malloc 3 > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
cd /tmp
P=$OLDPWD/placement_probe
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $OLDPWD/$OUT/pmc_$tag -o p -- $P malloc,vmm2,vmm1024 2 4 > $OLDPWD/$OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
ls -R $OLDPWD/$OUT | head -50
