#!/bin/bash
# Round 5, pass d: order of k_ans_decode's workgroups (DGPU_DEC_ORDER: 0 element-major, 1 tile-major, 2 per-XCD element-major):
# parity under every order, then the A/B over shapes.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for o in 1 2; do
  DGPU_DEC_ORDER=$o timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi.py -m gpu -q -n 4 2>&1 | tail -4 > $O/r5d_pytest_order$o.txt
  tail -1 $O/r5d_pytest_order$o.txt
done
V="base@DGPU_DEC_ORDER=0 base@DGPU_DEC_ORDER=1 base@DGPU_DEC_ORDER=2"
AB_STEPS=100 timeout 300 tools/ab.sh 3 bf16 $V > $O/r5d_ab_dec_order_bf16.txt 2>&1; tail -3 $O/r5d_ab_dec_order_bf16.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 fp16 $V > $O/r5d_ab_dec_order_fp16.txt 2>&1; tail -3 $O/r5d_ab_dec_order_fp16.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 u8 $V > $O/r5d_ab_dec_order_u8.txt 2>&1; tail -3 $O/r5d_ab_dec_order_u8.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 fp32 $V > $O/r5d_ab_dec_order_fp32.txt 2>&1; tail -3 $O/r5d_ab_dec_order_fp32.txt
for shape in "16 8388608" "1 134217728" "2048 65536" "64 2097152"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 $V > $O/r5d_ab_dec_order_bf16_$1x$2.txt 2>&1
  tail -3 $O/r5d_ab_dec_order_bf16_$1x$2.txt
done
