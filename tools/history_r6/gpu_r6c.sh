#!/bin/bash
# Round 6, pass c: look-back polling back-off, longer sleeps and the shapes with short chains.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "1 134217728" "16 8388608" "64 2097152" "256 524288"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 2 bf16 base v_bo127.so v_bo254.so v_bo508.so > $O/r6c_ab_lookback_backoff_bf16_$1x$2.txt 2>&1
  cut -c1-200 $O/r6c_ab_lookback_backoff_bf16_$1x$2.txt | head -8 | tail -4; tail -4 $O/r6c_ab_lookback_backoff_bf16_$1x$2.txt
done
