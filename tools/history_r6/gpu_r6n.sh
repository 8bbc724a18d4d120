#!/bin/bash
# Round 6, pass n: two-level look-back (groups of 64 tiles) for elements of more than 64 tiles: parity, then on / off.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "lookback or absent_workgroups or one_gi or baseline_config or dispatch_modes or capacity" 2>&1 | tail -6 ) > $O/r6n_pytest_lookback.txt
cat $O/r6n_pytest_lookback.txt
for shape in "1 134217728" "16 8388608" "1 16777216" "4 16777216"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 3 bf16 base@DGPU_TWO_LEVEL_LOOKBACK=0 base@DGPU_TWO_LEVEL_LOOKBACK=1 > $O/r6n_ab_two_level_lookback_bf16_$1x$2.txt 2>&1
  cut -c1-175 $O/r6n_ab_two_level_lookback_bf16_$1x$2.txt | tail -8
done
for m in 0 1; do echo "== DGPU_TWO_LEVEL_LOOKBACK=$m"; DGPU_TWO_LEVEL_LOOKBACK=$m python tools/small_call_probe.py --sizes 4,8,16,32,64,128 --reps 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['mega_floats'], 'compress us one by one / back to back', d['compress']['one_by_one_median_us'], d['compress']['back_to_back_us'])"; done > $O/r6n_small_calls_two_level_lookback.txt
cat $O/r6n_small_calls_two_level_lookback.txt
