#!/bin/bash
# Round 6, pass x: which commit cost the raw-byte encoder its 3 us: the libraries of e794afa (round 5), 030931f (look-back loop
# generalised), 5f0e594 (poll pauses, lookBackExclusive), 787b8be (two-level look-back), and the tree (base).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_STEPS=60 timeout 1200 tools/ab.sh 3 u8 base v_r5.so v_c1.so v_c2.so v_c3.so > $O/r6x_ab_raw_encoder_by_commit_u8.txt 2>&1
grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6x_ab_raw_encoder_by_commit_u8.txt | paste - - - | head -15
