#!/bin/bash
# Round 5, pass ad: the decoder's non-compressed bytes requested THREE groups ahead (v_ncdepth3.so) against two (base).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_STEPS=100 timeout 400 tools/ab.sh 4 bf16 base v_ncdepth3.so > $O/r5ad_ab_decoder_nc_depth_bf16.txt 2>&1
cut -c1-230 $O/r5ad_ab_decoder_nc_depth_bf16.txt | tail -10
AB_STEPS=100 timeout 300 tools/ab.sh 2 fp16 base v_ncdepth3.so > $O/r5ad_ab_decoder_nc_depth_fp16.txt 2>&1
cut -c1-230 $O/r5ad_ab_decoder_nc_depth_fp16.txt | tail -2
