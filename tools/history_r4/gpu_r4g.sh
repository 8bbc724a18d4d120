#!/bin/bash
# round 4, pass g: k_ans_decode_lanes (one lane per block) -- parity, then A/B against the previous commit on raw bytes
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_cabi.py -m gpu -q -x -k lanes 2>&1 | tail -15 ) > gpurun_out/r4g_pytest_lanes.txt
tail -15 gpurun_out/r4g_pytest_lanes.txt
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "not torch_ops" 2>&1 | tail -6 ) > gpurun_out/r4g_pytest.txt
tail -4 gpurun_out/r4g_pytest.txt
AB_STEPS=60 bash tools/ab.sh 2 u8 v_prev.so base > gpurun_out/r4g_ab_u8.txt 2>&1
grep -v amdgpu gpurun_out/r4g_ab_u8.txt | cut -c1-250
