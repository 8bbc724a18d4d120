#!/bin/bash
# Round 6, pass h: one-kernel compress with the arrival words on lines of their own and a doubling pause between polls.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -5 ) > $O/r6h_pytest_fused.txt
cat $O/r6h_pytest_fused.txt
bash tools/history_r6/gpu_r6g.sh | tail -15 | sed 's/^/r6h /'
for shape in "256 524288" "2048 65536"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 1 bf16 base@DGPU_FUSED=0 base@DGPU_FUSED=1 > $O/r6h_ab_fused_bf16_$1x$2.txt 2>&1
  cut -c1-250 $O/r6h_ab_fused_bf16_$1x$2.txt | head -2
done
