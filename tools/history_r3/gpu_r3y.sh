#!/bin/bash
mkdir -p gpurun_out
( DGPU_LIB=$PWD/dietgpu_amd/lib/v_spos.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not one_gi and not stream_state and not mismatch_reports" 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee gpurun_out/r3y_pytest.txt
for w in u8 bf16; do AB_ARGS="--rotate 1" AB_STEPS=150 bash tools/ab.sh 3 $w base v_spos.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"; done | tee gpurun_out/r3y_ab_scalar_pos_exec.txt
