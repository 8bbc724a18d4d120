#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "capped_stride or histogram_load_policy" 2>&1 | tail -3 ) | tee gpurun_out/r3j_pytest.txt
PROFILE_ARGS="--rotate 1" tools/gpu_profile.sh r03one bf16 > /dev/null 2>&1
PROFILE_ARGS="--rotate 1" tools/gpu_profile.sh r03one u8 > /dev/null 2>&1
head -7 gpurun_out/rocprof_r03one_bf16.txt | cut -c1-140; head -7 gpurun_out/rocprof_r03one_u8.txt | cut -c1-140
