#!/bin/bash
# Round 4, pass k: single-block batches -- four input chunks in flight in the pair encoder (first ones requested before
# the table build), wave minimum instead of a bisection in the normalisation's full trips.  v_c1.so = commit 8f0bc92.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "single_block or ragged or small or whole_block or normalize" 2>&1 | tail -6 > gpurun_out/k_pytest_focus.txt
tail -3 gpurun_out/k_pytest_focus.txt
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/k_pytest.txt
grep -a "passed\|failed" gpurun_out/k_pytest.txt
AB_ARGS="--batch 32768 --elems 4096" timeout 300 tools/ab.sh 2 bf16 v_c1.so base > gpurun_out/k_ab_bf16_32768x4096.txt 2>&1
tail -6 gpurun_out/k_ab_bf16_32768x4096.txt
AB_ARGS="--batch 32768 --elems 4096" timeout 200 tools/ab.sh 1 fp32 v_c1.so base > gpurun_out/k_ab_fp32_32768x2048.txt 2>&1
tail -4 gpurun_out/k_ab_fp32_32768x2048.txt
AB_ARGS="--batch 32768 --elems 4096" timeout 200 tools/ab.sh 1 fp16 v_c1.so base > gpurun_out/k_ab_fp16_32768x4096.txt 2>&1
tail -4 gpurun_out/k_ab_fp16_32768x4096.txt
