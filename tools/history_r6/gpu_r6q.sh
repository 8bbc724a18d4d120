#!/bin/bash
# Round 6, pass q: the raw-byte encoder (k_ans_encode<10, 0, false, 8, false>, hardware dispatch) with its LDS padded to 2 and 1
# workgroups per CU (3 = base): how much of its time is the dependent chain of a wavefront?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_STEPS=30 timeout 900 tools/ab.sh 2 u8 base v_rawocc2.so v_rawocc1.so > $O/r6q_ab_raw_encoder_residency_u8.txt 2>&1
cut -c1-200 $O/r6q_ab_raw_encoder_residency_u8.txt | tail -9
