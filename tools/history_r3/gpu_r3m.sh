#!/bin/bash
# exec-masked stage write (+ renormalisation shift in the same exec window): parity, then interleaved A/B
mkdir -p gpurun_out
for v in v_encx2.so; do ( DGPU_LIB=$PWD/dietgpu_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not one_gi" 2>&1 | tail -2 ); done | tee gpurun_out/r3m_pytest.txt
for w in u8 bf16 fp16 fp32; do AB_ARGS="--rotate 1" AB_STEPS=200 bash tools/ab.sh 3 $w base v_encx1.so v_encx2.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"; done | tee gpurun_out/r3m_ab_exec_write.txt
