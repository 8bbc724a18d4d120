#!/bin/bash
# Round 5, pass w: the decoder's partial-block path with the groups above the whole ones run by the predicated form of
# the SAME pipelined group loop (base), against the top rows eight at a time without cross-group prefetch
# (v_tail2.so) and the scalar paths (v_pre_tail.so): the GPU parity tests, then the shapes of passes u and v.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5w_pytest.txt
tail -6 $O/r5w_pytest.txt
for shape in "256 524288" "256 530000" "32768 4000" "20000 6000" "8192 15000" "4096 70000"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_pre_tail.so v_tail2.so base > $O/r5w_ab_partial_blocks_bf16_$1x$2.txt 2>&1
  cut -c1-230 $O/r5w_ab_partial_blocks_bf16_$1x$2.txt | tail -6
done
AB_ARGS="--batch 20000 --elems 12000" AB_STEPS=50 timeout 300 tools/ab.sh 2 u8 v_pre_tail.so v_tail2.so base > $O/r5w_ab_partial_blocks_u8_20000x12000.txt 2>&1
cut -c1-230 $O/r5w_ab_partial_blocks_u8_20000x12000.txt | tail -6
AB_ARGS="--batch 20000 --elems 3000" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp32 v_pre_tail.so v_tail2.so base > $O/r5w_ab_partial_blocks_fp32_20000x3000.txt 2>&1
cut -c1-230 $O/r5w_ab_partial_blocks_fp32_20000x3000.txt | tail -6
