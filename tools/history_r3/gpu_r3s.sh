#!/bin/bash
# deferred copy-out of the encoder (few long elements): parity with it forced on everywhere, then timing on / off per shape
mkdir -p gpurun_out
( DGPU_ENC_DEFER=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -m gpu -x -q -k "not one_gi" 2>&1 | grep -v amdgpu.ids | tail -5 ) | tee gpurun_out/r3s_pytest_defer_on.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_gi or absent or float" 2>&1 | grep -v amdgpu.ids | tail -3 ) | tee -a gpurun_out/r3s_pytest_defer_on.txt
for s in "1 134217728" "16 8388608" "64 2097152" "256 524288"; do set -- $s
  AB_ARGS="--rotate 1 --batch $1 --elems $2" AB_STEPS=100 bash tools/ab.sh 2 bf16 base@DGPU_ENC_DEFER=0 base@DGPU_ENC_DEFER=1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | sed "s/^/$1x$2 /"
done | tee gpurun_out/r3s_ab_defer.txt
