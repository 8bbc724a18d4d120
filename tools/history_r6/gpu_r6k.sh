#!/bin/bash
# Round 6, pass k: look-back pause by chain length (base) against always-maximum and always-doubling.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
export DGPU_FUSED=0
for shape in "16 8388608" "1 134217728" "64 2097152" "256 524288"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 3 bf16 base v_alwaysmax.so v_alwaysdbl.so > $O/r6k_ab_poll_pause_bf16_$1x$2.txt 2>&1
  cut -c1-150 $O/r6k_ab_poll_pause_bf16_$1x$2.txt | tail -12
done
