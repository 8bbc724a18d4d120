#!/bin/bash
# Round 5, pass k: float tiles of 2 / 4 blocks under hardware dispatch with pooled spill slots (base; forced persistent
# with DGPU_ENC_DISPATCH=0): whole GPU suite, then the A/B.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -25 > $O/r5k_pytest.txt
tail -4 $O/r5k_pytest.txt
for shape in "16384 8192" "8192 16384" "20000 6000"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 base@DGPU_ENC_DISPATCH=0 base > $O/r5k_ab_small_tiles_hw_dispatch_bf16_$1x$2.txt 2>&1
  cut -c1-250 $O/r5k_ab_small_tiles_hw_dispatch_bf16_$1x$2.txt | tail -4
done
AB_ARGS="--batch 8192 --elems 16384" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp16 base@DGPU_ENC_DISPATCH=0 base > $O/r5k_ab_small_tiles_hw_dispatch_fp16_8192x16384.txt 2>&1
cut -c1-250 $O/r5k_ab_small_tiles_hw_dispatch_fp16_8192x16384.txt | tail -3
