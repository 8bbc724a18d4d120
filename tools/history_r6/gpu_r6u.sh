#!/bin/bash
# Round 6, pass u: the final round-6 library against the round-5 library (v_r5 = commit e794afa), BASELINE configs 2, 3, 4, fp32
# and the shapes of DESIGN.md's table, interleaved on one box.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for w in u8 bf16 fp16 fp32; do
  AB_STEPS=60 timeout 900 tools/ab.sh 3 $w base v_r5.so > $O/r6u_ab_round6_vs_round5_$w.txt 2>&1; tail -2 $O/r6u_ab_round6_vs_round5_$w.txt | sed "s/^/$w /"
done
for shape in "32768 4096" "16384 8192" "8192 16384" "2048 65536" "64 2097152" "16 8388608" "1 134217728"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 900 tools/ab.sh 2 bf16 base v_r5.so > $O/r6u_ab_round6_vs_round5_bf16_$1x$2.txt 2>&1; tail -2 $O/r6u_ab_round6_vs_round5_bf16_$1x$2.txt | sed "s/^/$1x$2 /"
done
