#!/bin/bash
# Round 6, pass w: where the raw-byte encoder lost 3 us against round 5: without the two-level code (v_no2l), with one s_sleep 1
# between polls as in round 5 (v_sl1), the round-5 library (v_r5).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_STEPS=60 timeout 900 tools/ab.sh 3 u8 base v_no2l.so v_sl1.so v_r5.so > $O/r6w_ab_raw_encoder_regression_u8.txt 2>&1
grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6w_ab_raw_encoder_regression_u8.txt | paste - - - | head -12
