#!/bin/bash
# Round 4, pass l: decoders request pdf (+ for float archives: block descriptor and lane states) in the same round trip
# as the ANS header; pair decoder's early loads only where the bytes are known to exist.  v_c2.so = commit 6000131.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/l_pytest.txt
grep -a "passed\|failed" gpurun_out/l_pytest.txt
timeout 300 tools/ab.sh 3 bf16 v_c2.so base > gpurun_out/l_ab_bf16.txt 2>&1
tail -8 gpurun_out/l_ab_bf16.txt
timeout 200 tools/ab.sh 1 fp16 v_c2.so base > gpurun_out/l_ab_fp16.txt 2>&1
tail -4 gpurun_out/l_ab_fp16.txt
timeout 200 tools/ab.sh 1 u8 v_c2.so base > gpurun_out/l_ab_u8.txt 2>&1
tail -4 gpurun_out/l_ab_u8.txt
AB_ARGS="--batch 8192 --elems 16384" timeout 200 tools/ab.sh 1 bf16 v_c2.so base > gpurun_out/l_ab_bf16_8192x16384.txt 2>&1
tail -4 gpurun_out/l_ab_bf16_8192x16384.txt
AB_ARGS="--batch 32768 --elems 4096" timeout 200 tools/ab.sh 1 bf16 v_c2.so base > gpurun_out/l_ab_bf16_32768x4096.txt 2>&1
tail -4 gpurun_out/l_ab_bf16_32768x4096.txt
