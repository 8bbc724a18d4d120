#!/bin/bash
mkdir -p gpurun_out
( DGPU_LIB=$PWD/dietgpu_amd/lib/v_enc8x.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ans_ or config2 or worst or incompress" 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee gpurun_out/r3x_pytest.txt
AB_ARGS="--rotate 1" AB_STEPS=150 bash tools/ab.sh 3 u8 base v_enc8x.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tee gpurun_out/r3x_ab_entry8_exec.txt
