#!/bin/bash
# round 4, pass e: decoder start-up chain (descriptor / states / pdf / words / first nc bytes requested before the LUT
# build) + non-compressed bytes two groups ahead: parity, then A/B against the previous commit on the cold loop
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi.py tests/test_golden.py tests/test_torch_ops.py -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/r4e_pytest.txt
tail -3 gpurun_out/r4e_pytest.txt
AB_STEPS=100 bash tools/ab.sh 3 bf16 v_prev.so base > gpurun_out/r4e_ab_bf16.txt 2>&1
AB_STEPS=60 bash tools/ab.sh 2 fp16 v_prev.so base > gpurun_out/r4e_ab_fp16.txt 2>&1
AB_STEPS=60 bash tools/ab.sh 2 u8 v_prev.so base > gpurun_out/r4e_ab_u8.txt 2>&1
AB_STEPS=60 bash tools/ab.sh 2 fp32 v_prev.so base > gpurun_out/r4e_ab_fp32.txt 2>&1
AB_STEPS=60 AB_ARGS="--batch 2048 --elems 65536" bash tools/ab.sh 2 bf16 v_prev.so base > gpurun_out/r4e_ab_bf16_2048x64k.txt 2>&1
for f in gpurun_out/r4e_ab_*.txt; do echo $f; grep -v amdgpu $f | cut -c1-260; done
