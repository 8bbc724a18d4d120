#!/bin/bash
# Round 5, pass n: float encoder, wavefront priority by position in the round (v_prio: the first quarter of the persistent grid
# runs at s_setprio 3, the last at 0; v_priorev: the reverse) -- does a deliberate skew between a tile and its predecessors
# shorten the look-back wait of few large elements?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "1 134217728" "16 8388608" "256 524288"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 base v_prio.so v_priorev.so > $O/r5n_ab_encoder_priority_bf16_$1x$2.txt 2>&1
  cut -c1-250 $O/r5n_ab_encoder_priority_bf16_$1x$2.txt | tail -6
done
