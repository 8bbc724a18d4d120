#!/bin/bash
# Round 5, pass i: k_ans_encode_pair one workgroup per pair with pooled spill slots (base) against the persistent grid
# (v_r5e.so): whole GPU suite, then the A/B on batches of single-block elements.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -25 > $O/r5i_pytest.txt
tail -4 $O/r5i_pytest.txt
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 300 tools/ab.sh 3 bf16 v_r5e.so base > $O/r5i_ab_pair_encoder_hw_dispatch_bf16_32768x4096.txt 2>&1
cut -c1-250 $O/r5i_ab_pair_encoder_hw_dispatch_bf16_32768x4096.txt | tail -4
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp16 v_r5e.so base > $O/r5i_ab_pair_encoder_hw_dispatch_fp16_32768x4096.txt 2>&1
cut -c1-250 $O/r5i_ab_pair_encoder_hw_dispatch_fp16_32768x4096.txt | tail -3
AB_ARGS="--batch 65535 --elems 2048" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_r5e.so base > $O/r5i_ab_pair_encoder_hw_dispatch_bf16_65535x2048.txt 2>&1
cut -c1-250 $O/r5i_ab_pair_encoder_hw_dispatch_bf16_65535x2048.txt | tail -3
