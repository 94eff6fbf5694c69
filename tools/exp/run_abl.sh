#!/bin/bash
# swap ablation builds of librecamd.so in and time the gram kernel with each
L="recommend-tf2.0_amd/recamd/librecamd.so"
cp $L /tmp/lib_orig.so
for v in 0 1 2 3 5 7; do
  cp tools/exp/lib_abl$v.so $L
  echo "== ABL=$v (1 no global stores, 2 no split/MFMA, 4 no staging)"
  timeout -k 10 120 python tools/exp/gram_ablate.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_orig.so $L
