#!/bin/bash
# Round 5, pass j: float batches of small elements (2- and 4-block tiles), k_ans_encode as one workgroup per tile
# (v_smallhw.so with DGPU_ENC_DISPATCH=1: experiment with oversized static spill slots) against the persistent grid.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "16384 8192" "8192 16384" "4096 32768" "2048 65536"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_smallhw.so@DGPU_ENC_DISPATCH=0 v_smallhw.so@DGPU_ENC_DISPATCH=1 > $O/r5j_ab_small_tiles_hw_dispatch_bf16_$1x$2.txt 2>&1
  cut -c1-270 $O/r5j_ab_small_tiles_hw_dispatch_bf16_$1x$2.txt | tail -4
done
