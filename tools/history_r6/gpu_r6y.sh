#!/bin/bash
# Round 6, pass y: the look-back walk in place with short pauses whenever few tiles of an element are in flight, for every input
# type (base) against the function form for floats (v_prev = the commit before) and the round-5 library (v_r5); look-back tests.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -n 4 -k "lookback or absent_workgroups or dispatch_modes or baseline_config or size_classes" 2>&1 | tail -2 ) > $O/r6y_pytest.txt; tail -1 $O/r6y_pytest.txt
for w in bf16 fp16 fp32 u8; do
  AB_STEPS=60 timeout 1200 tools/ab.sh 4 $w base v_prev.so v_r5.so > $O/r6y_ab_inplace_walk_$w.txt 2>&1
  grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6y_ab_inplace_walk_$w.txt | paste - - - | head -3; tail -3 $O/r6y_ab_inplace_walk_$w.txt | sed "s/^/$w /"
done
for shape in "2048 65536" "16384 8192"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 900 tools/ab.sh 3 bf16 base v_prev.so v_r5.so > $O/r6y_ab_inplace_walk_bf16_$1x$2.txt 2>&1; tail -3 $O/r6y_ab_inplace_walk_bf16_$1x$2.txt | sed "s/^/$1x$2 /"
done
