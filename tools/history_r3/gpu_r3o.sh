#!/bin/bash
# decoder: word read + renormalisation under the ballot as exec mask (no branch per row): parity, then interleaved A/B
mkdir -p gpurun_out
( DGPU_LIB=$PWD/dietgpu_amd/lib/v_decx.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not one_gi and not stream_state" 2>&1 | grep -v amdgpu.ids | tail -4 ) | tee gpurun_out/r3o_pytest.txt
for w in u8 bf16 fp16 fp32; do AB_ARGS="--rotate 1" AB_STEPS=200 bash tools/ab.sh 3 $w base v_decx.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"; done | tee gpurun_out/r3o_ab_dec_exec_read.txt
