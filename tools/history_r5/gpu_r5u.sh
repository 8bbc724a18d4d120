#!/bin/bash
# Round 5, pass u: blocks that are not full on the chunked / straight-line paths of BOTH coders (encodeRows kTail,
# decodeBlock kTail; base) against the scalar paths (v_pre_tail.so): whole GPU suite, then element sizes that are not
# whole tiles and batches of small ragged elements.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5u_pytest.txt
tail -6 $O/r5u_pytest.txt
for shape in "256 524288" "256 530000" "40 3355440" "32768 4000" "20000 6000" "8192 15000"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_pre_tail.so base > $O/r5u_ab_partial_blocks_bf16_$1x$2.txt 2>&1
  cut -c1-230 $O/r5u_ab_partial_blocks_bf16_$1x$2.txt | tail -4
done
AB_ARGS="--batch 256 --elems 530000" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp16 v_pre_tail.so base > $O/r5u_ab_partial_blocks_fp16_256x530000.txt 2>&1
cut -c1-230 $O/r5u_ab_partial_blocks_fp16_256x530000.txt | tail -3
AB_ARGS="--batch 256 --elems 265000" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp32 v_pre_tail.so base > $O/r5u_ab_partial_blocks_fp32_256x265000.txt 2>&1
cut -c1-230 $O/r5u_ab_partial_blocks_fp32_256x265000.txt | tail -3
