#!/bin/bash
# Round 5, pass l: raw-byte encoder with table entries / symbols fetched further ahead (v_raw48: 4 / 8 rows, v_raw36: 3 / 6,
# v_raw24: as shipped, 2 / 4) and the register budget of 3 wavefronts per SIMD (what its LDS allows) instead of 6.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_STEPS=50 timeout 400 tools/ab.sh 3 u8 base v_raw24.so v_raw36.so v_raw48.so > $O/r5l_ab_raw_encoder_prefetch_depth.txt 2>&1
cut -c1-230 $O/r5l_ab_raw_encoder_prefetch_depth.txt | tail -16
