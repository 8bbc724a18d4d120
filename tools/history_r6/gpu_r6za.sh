#!/bin/bash
# Round 6, pass za: the 1280-word stage for EVERY bf16 kernel (2- and 4-block tiles, pairs: v_st1280all) against the tree (1024
# there, 1280 for persistent 8-block tiles of few-tile elements).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "16384 8192" "8192 16384" "32768 4096"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 900 tools/ab.sh 3 bf16 base v_st1280all.so > $O/r6za_ab_wide_stage_small_tiles_bf16_$1x$2.txt 2>&1
  grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode[a-z_]*': [0-9.]*" $O/r6za_ab_wide_stage_small_tiles_bf16_$1x$2.txt | paste - - - | head -4; tail -2 $O/r6za_ab_wide_stage_small_tiles_bf16_$1x$2.txt
done
