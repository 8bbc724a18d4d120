#!/bin/bash
# Issue-model experiment (GPU box): does an extra instruction per row cost time in the decoder's row loop?
# Builds the library three times (as is / + 4 s_nop per row / + 4 independent v_mov per row) and interleaves them
# on the Zipf-byte workload.  Result on MI355X (round 2): +0.6 % and +2.5 % -- the loop is bound by the latency
# of its dependent chain, not by instruction issue or VALU throughput.
# Usage: gpurun --timeout 900 -- bash tools/issue_model.sh
set -e
L=dietgpu_amd/lib; C=dietgpu_amd/csrc/capi.hip; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
hipcc $F -DDGPU_DEC_RAW_WIDE16=1 -o $L/v_plain.so $C &
hipcc $F -DDGPU_DEC_RAW_WIDE16=1 -DDGPU_DEC_PAD_SNOP=4 -o $L/v_snop4.so $C &
hipcc $F -DDGPU_DEC_RAW_WIDE16=1 -DDGPU_DEC_PAD_VALU=4 -o $L/v_valu4.so $C &
wait
AB_STEPS=100 bash tools/ab.sh 2 u8 v_plain.so v_snop4.so v_valu4.so
