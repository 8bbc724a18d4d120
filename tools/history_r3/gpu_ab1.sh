#!/bin/bash
# One GPU-box pass for a kernel change: parity tests with the default build, then an interleaved A/B of builds.
# Usage: gpurun --timeout 1100 -- bash tools/gpu_ab1.sh "<variants for bf16>" "<variants for u8>"
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/ab1_pytest.txt
tail -5 gpurun_out/ab1_pytest.txt
AB_STEPS=100 bash tools/ab.sh 2 bf16 $1 2>&1 | tee gpurun_out/ab1_bf16.txt | tail -8
AB_STEPS=100 bash tools/ab.sh 1 u8 $2 2>&1 | tee gpurun_out/ab1_u8.txt | tail -6
AB_STEPS=100 bash tools/ab.sh 1 fp16 $2 2>&1 | tee gpurun_out/ab1_fp16.txt | tail -6
