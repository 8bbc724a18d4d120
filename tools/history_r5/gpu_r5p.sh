#!/bin/bash
# Round 5, pass p: lane slots per bin of the histogram workgroups that see <= 64 KiB (batches of small elements): 4 / 8 (ships) / 16.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "16384 8192" "8192 16384" "4096 32768"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 base v_hs4.so v_hs16.so > $O/r5p_ab_hist_small_slots_bf16_$1x$2.txt 2>&1
  cut -c1-220 $O/r5p_ab_hist_small_slots_bf16_$1x$2.txt | tail -6
done
