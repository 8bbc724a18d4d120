#!/bin/bash
mkdir -p gpurun_out
for w in bf16 u8; do PROFILE_ARGS="--rotate 1" tools/gpu_profile.sh r03one $w > /dev/null 2>&1; done
head -7 gpurun_out/rocprof_r03one_bf16.txt | cut -c1-140; head -7 gpurun_out/rocprof_r03one_u8.txt | cut -c1-140
