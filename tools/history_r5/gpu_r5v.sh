#!/bin/bash
# Round 5, pass v: the decoder's partial-block path with the top rows' non-compressed bytes requested eight rows at a
# time (base) against one row at a time (v_tail1.so) and against the scalar paths (v_pre_tail.so): the GPU parity
# tests, then the shapes of pass u.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5v_pytest.txt
tail -6 $O/r5v_pytest.txt
for shape in "256 530000" "40 3355440" "32768 4000" "20000 6000" "8192 15000" "4096 70000"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_pre_tail.so v_tail1.so base > $O/r5v_ab_partial_blocks_bf16_$1x$2.txt 2>&1
  cut -c1-230 $O/r5v_ab_partial_blocks_bf16_$1x$2.txt | tail -6
done
AB_ARGS="--batch 256 --elems 1060000" AB_STEPS=50 timeout 300 tools/ab.sh 2 u8 v_pre_tail.so v_tail1.so base > $O/r5v_ab_partial_blocks_u8_256x1060000.txt 2>&1
cut -c1-230 $O/r5v_ab_partial_blocks_u8_256x1060000.txt | tail -6
