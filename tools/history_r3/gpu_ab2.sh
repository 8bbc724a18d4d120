#!/bin/bash
# Parity tests with the default build, then base-vs-default on every workload and on the small-element shapes.
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/ab2_pytest.txt
tail -3 gpurun_out/ab2_pytest.txt
for w in bf16 u8 fp16 fp32; do AB_STEPS=100 bash tools/ab.sh 1 $w base libdietgpu_amd.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | head -2; done | tee gpurun_out/ab2_workloads.txt
for s in "2048 65536" "8192 16384" "16384 8192" "32768 4096" "1 134217728" "16 8388608"; do set -- $s
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=60 bash tools/ab.sh 1 bf16 base libdietgpu_amd.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | head -2 | sed "s/^/$1x$2 /"; done | tee gpurun_out/ab2_shapes.txt
