#!/bin/bash
# Round 4, pass m: 16-bit floats at probBits 10 in 16-block decode tiles -- compact LUT + half transposition buffer
# (40 KiB: FOUR workgroups per CU, 58 registers) against 8-byte LUT entries + full buffer (48 KiB: three).
# v_c3.so = the committed tree before the change.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/m_pytest.txt
grep -a "passed\|failed" gpurun_out/m_pytest.txt
timeout 300 tools/ab.sh 3 bf16 v_c3.so base > gpurun_out/m_ab_bf16.txt 2>&1
tail -8 gpurun_out/m_ab_bf16.txt
AB_ARGS="--prob-bits 10" timeout 200 tools/ab.sh 1 fp16 v_c3.so base > gpurun_out/m_ab_fp16_p10.txt 2>&1
tail -4 gpurun_out/m_ab_fp16_p10.txt
AB_ARGS="--batch 2048 --elems 65536" timeout 200 tools/ab.sh 1 bf16 v_c3.so base > gpurun_out/m_ab_bf16_2048x65536.txt 2>&1
tail -4 gpurun_out/m_ab_bf16_2048x65536.txt
AB_ARGS="--batch 16 --elems 8388608" timeout 200 tools/ab.sh 1 bf16 v_c3.so base > gpurun_out/m_ab_bf16_16x8388608.txt 2>&1
tail -4 gpurun_out/m_ab_bf16_16x8388608.txt
