#!/bin/bash
# Round 6, pass r: the float encoder at five workgroups per CU: LDS padded by 4 KiB (v_occ5), or the stage at 1280 words for
# every float type (v_st1280; fp16 has it already) -- against six (base).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for shape in "256 524288" "64 2097152" "16 8388608" "2048 65536"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=60 timeout 900 tools/ab.sh 4 bf16 base v_occ5.so v_st1280.so > $O/r6r_ab_encoder_five_per_cu_bf16_$1x$2.txt 2>&1
  grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6r_ab_encoder_five_per_cu_bf16_$1x$2.txt | paste - - - | head -12; tail -3 $O/r6r_ab_encoder_five_per_cu_bf16_$1x$2.txt
done
AB_STEPS=60 timeout 900 tools/ab.sh 3 fp32 base v_occ5.so v_st1280.so > $O/r6r_ab_encoder_five_per_cu_fp32.txt 2>&1; tail -3 $O/r6r_ab_encoder_five_per_cu_fp32.txt
