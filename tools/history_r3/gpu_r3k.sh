#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not one_gi" 2>&1 | tail -3 ) | tee gpurun_out/r3k_pytest.txt
for v in v_late1a3.so v_late1s6.so; do ( DGPU_LIB=$PWD/dietgpu_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ans_ or float_batch or config2 or config3 or incompressible or worst" 2>&1 | tail -2 ); done | tee -a gpurun_out/r3k_pytest.txt
for w in u8 bf16 fp16 fp32; do AB_ARGS="--rotate 1" AB_STEPS=200 bash tools/ab.sh 2 $w v_late0.so base v_late1a3.so v_late1s6.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"; done | tee gpurun_out/r3k_ab_late_addr.txt
for s in "8192 16384" "32768 4096" "1 134217728"; do set -- $s; AB_ARGS="--rotate 1 --batch $1 --elems $2" AB_STEPS=100 bash tools/ab.sh 1 bf16 v_late0.so base 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | head -2 | sed "s/^/$1x$2 /"; done | tee -a gpurun_out/r3k_ab_late_addr.txt
