#!/bin/bash
# Round 5, pass x: pass w's decoder with the rows of the top group that neither half has skipped (base) against pass
# w's (v_tail3.so) and the scalar paths (v_pre_tail.so): the GPU parity tests, then batches of small ragged elements.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5x_pytest.txt
tail -3 $O/r5x_pytest.txt
for w in "bf16 32768 4000" "bf16 32768 1000" "fp32 20000 3000" "fp16 32768 3500" "bf16 20000 6000"; do
  set -- $w
  AB_ARGS="--batch $2 --elems $3" AB_STEPS=50 timeout 300 tools/ab.sh 2 $1 v_pre_tail.so v_tail3.so base > $O/r5x_ab_partial_blocks_$1_$2x$3.txt 2>&1
  cut -c1-230 $O/r5x_ab_partial_blocks_$1_$2x$3.txt | tail -6
done
