#!/bin/bash
# Round 5, pass q: as pass p with 32 lane slots too, plus the shapes around the 64 KiB-per-workgroup switch.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "16384 8192" "8192 16384" "4096 32768" "32768 6000"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 base v_hs16.so v_hs32.so > $O/r5q_ab_hist_small_slots_bf16_$1x$2.txt 2>&1
  grep -o "^[^ ]* *cold [0-9.]*\|'float_histogram': [0-9.]*" $O/r5q_ab_hist_small_slots_bf16_$1x$2.txt | paste - - - | head -6
done
AB_ARGS="--batch 16384 --elems 8192" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp16 base v_hs16.so v_hs32.so > $O/r5q_ab_hist_small_slots_fp16_16384x8192.txt 2>&1
grep -o "^[^ ]* *cold [0-9.]*\|'float_histogram': [0-9.]*" $O/r5q_ab_hist_small_slots_fp16_16384x8192.txt | paste - - - | head -6
