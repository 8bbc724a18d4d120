#!/bin/bash
# Round 5, pass r: (1) the batch size from which the decoder's workgroups go per XCD: 32 x 4 Mi and 48 x 2.67 Mi under order 0 / 2;
# (2) histogram workgroups per element now that parts are contiguous: 512 (ships) / 768 / 1024 target workgroups.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "32 4194304" "48 2796200" "40 3355440"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 base@DGPU_DEC_ORDER=0 base@DGPU_DEC_ORDER=2 > $O/r5r_ab_dec_order_bf16_$1x$2.txt 2>&1
  cut -c1-200 $O/r5r_ab_dec_order_bf16_$1x$2.txt | tail -4
done
AB_STEPS=100 timeout 300 tools/ab.sh 3 bf16 base v_hw768.so v_hw1024.so > $O/r5r_ab_hist_target_wgs_bf16.txt 2>&1
grep -o "^[^ ]* *cold [0-9.]*\|'float_histogram': [0-9.]*" $O/r5r_ab_hist_target_wgs_bf16.txt | paste - - - ; tail -3 $O/r5r_ab_hist_target_wgs_bf16.txt
