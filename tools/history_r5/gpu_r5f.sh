#!/bin/bash
# Round 5, pass f: histogram workgroups stream one contiguous part of their element (base) against parts interleaved at
# 4 KiB (v_r5e.so = the tree before): statistics parity, then the A/B over shapes.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi.py tests/test_reference_pin.py -m gpu -q -n 4 2>&1 | tail -3 > $O/r5f_pytest.txt
tail -1 $O/r5f_pytest.txt
AB_STEPS=100 timeout 300 tools/ab.sh 3 bf16 v_r5e.so base > $O/r5f_ab_hist_contiguous_bf16.txt 2>&1; tail -2 $O/r5f_ab_hist_contiguous_bf16.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 u8 v_r5e.so base > $O/r5f_ab_hist_contiguous_u8.txt 2>&1; tail -2 $O/r5f_ab_hist_contiguous_u8.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 fp32 v_r5e.so base > $O/r5f_ab_hist_contiguous_fp32.txt 2>&1; tail -2 $O/r5f_ab_hist_contiguous_fp32.txt
for shape in "16 8388608" "1 134217728" "64 2097152" "4 33554432" "2048 65536"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_r5e.so base > $O/r5f_ab_hist_contiguous_bf16_$1x$2.txt 2>&1
  tail -2 $O/r5f_ab_hist_contiguous_bf16_$1x$2.txt
done
grep -h "float_histogram\|'histogram'" $O/r5f_ab_hist_contiguous_*.txt | awk '{print FILENAME, $1, $0}' | grep -o "^[^ ]* [^ ]*\|'float_histogram': [0-9.]*\|'histogram': [0-9.]*" | paste - - - | head -60
