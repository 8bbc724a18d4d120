#!/bin/bash
# Round 6, pass z: fp16 stage of 1536 words (four workgroups per CU, v_fp1536) against 1280 (five, base): BASELINE config 4
# (50 % zeros, probBits 11) and the same at probBits 10.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_STEPS=60 timeout 900 tools/ab.sh 4 fp16 base v_fp1536.so > $O/r6z_ab_fp16_stage_p11.txt 2>&1
grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*\|'ans_decode': [0-9.]*" $O/r6z_ab_fp16_stage_p11.txt | paste - - - - - | head -4; tail -2 $O/r6z_ab_fp16_stage_p11.txt
AB_ARGS="--prob-bits 10" AB_STEPS=60 timeout 900 tools/ab.sh 3 fp16 base v_fp1536.so > $O/r6z_ab_fp16_stage_p10.txt 2>&1; tail -2 $O/r6z_ab_fp16_stage_p10.txt
