#!/bin/bash
# Round 6, pass v: look-back pause chosen by the tiles of an element in flight (base) against the round-5 library, Zipf bytes and
# the shapes the pause matters for; look-back tests.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -n 4 -k "lookback or absent_workgroups or dispatch_modes or baseline_config" 2>&1 | tail -2 ) > $O/r6v_pytest.txt; tail -1 $O/r6v_pytest.txt
AB_STEPS=60 timeout 900 tools/ab.sh 4 u8 base v_r5.so > $O/r6v_ab_round6_vs_round5_u8.txt 2>&1
grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6v_ab_round6_vs_round5_u8.txt | paste - - - | head -4; tail -2 $O/r6v_ab_round6_vs_round5_u8.txt
for shape in "256 524288" "64 2097152" "16 8388608" "1 134217728"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 900 tools/ab.sh 2 bf16 base v_r5.so > $O/r6v_ab_round6_vs_round5_bf16_$1x$2.txt 2>&1
  grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6v_ab_round6_vs_round5_bf16_$1x$2.txt | paste - - - | head -2; tail -2 $O/r6v_ab_round6_vs_round5_bf16_$1x$2.txt
done
