#!/bin/bash
# Round 6, pass o: float histogram with 8 / 16 loads of 16 bytes in flight per lane instead of 4.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "1 16777216" "1 1048576" "256 524288" "16384 8192" "1 134217728"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 3 bf16 base v_deep8.so v_deep16.so > $O/r6o_ab_hist_deep_loads_bf16_$1x$2.txt 2>&1
  cut -c1-175 $O/r6o_ab_hist_deep_loads_bf16_$1x$2.txt | head -3; tail -3 $O/r6o_ab_hist_deep_loads_bf16_$1x$2.txt
done
