#!/bin/bash
# Round 4, pass o: k_ans_decode_pair persistent and software-pipelined over a wavefront's pairs (header group of pair
# k + 1 and float header of pair k + 2 in flight during the rows of pair k).  v_c3.so = the committed tree before it.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/o_pytest.txt
grep -a "passed\|failed" gpurun_out/o_pytest.txt
for n in 32 4096; do
  AB_ARGS="--batch 32768 --elems $n" AB_STEPS=50 timeout 200 tools/ab.sh 2 bf16 v_c3.so base > gpurun_out/o_ab_bf16_32768x$n.txt 2>&1
  tail -6 gpurun_out/o_ab_bf16_32768x$n.txt
done
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 200 tools/ab.sh 1 fp16 v_c3.so base > gpurun_out/o_ab_fp16_32768x4096.txt 2>&1
tail -4 gpurun_out/o_ab_fp16_32768x4096.txt
