#!/bin/bash
# Round 6, pass t: the 16-block decoder at two workgroups per CU (LDS padded by 20 KiB) against three.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for w in bf16 u8; do
  AB_STEPS=60 timeout 900 tools/ab.sh 3 $w base v_dec2.so > $O/r6t_ab_decoder_two_per_cu_$w.txt 2>&1
  grep -o "^[a-z_0-9.]*so\|^base\|'ans_decode': [0-9.]*" $O/r6t_ab_decoder_two_per_cu_$w.txt | paste - - - | head -4; tail -2 $O/r6t_ab_decoder_two_per_cu_$w.txt
done
