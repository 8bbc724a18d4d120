#!/bin/bash
# Round 5, pass g: why is the float encoder 40 % slower on few large elements?  Timing ablation (WRONG archives by design,
# DGPU_BENCH_ABLATION=1 skips bench.py's checks): v_abl_nolookback.so never waits for a predecessor's descriptor.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
export DGPU_BENCH_ABLATION=1
for shape in "256 524288" "16 8388608" "1 134217728"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 base v_abl_nolookback.so > $O/r5g_ablation_lookback_bf16_$1x$2.txt 2>&1
  cut -c1-260 $O/r5g_ablation_lookback_bf16_$1x$2.txt | tail -6
done
