#!/bin/bash
# Round 5, pass z: vector paths for elements that are not 16-byte aligned (histogram head peel, encoder loads at any word
# alignment with the straddling part loaded as the 16 bytes that end at the element's end, decoder wide stores at any word
# alignment; base) against the 16-byte-aligned-only forms (v_aligned_only.so): the new alignment sweep, the GPU suite, then
# rows of [B, n] matrices whose row length is not a multiple of 16 bytes.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "every_word_alignment or unaligned or split_size or partial_last" 2>&1 | tail -8 > $O/r5z_pytest_alignment.txt
tail -4 $O/r5z_pytest_alignment.txt
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5z_pytest.txt
tail -3 $O/r5z_pytest.txt
for w in "bf16 256 524288" "bf16 256 530001" "bf16 256 530004" "bf16 32768 4001" "fp32 256 530002" "fp16 32768 3500"; do
  set -- $w
  AB_ARGS="--batch $2 --elems $3" AB_STEPS=50 timeout 300 tools/ab.sh 2 $1 v_aligned_only.so base > $O/r5z_ab_unaligned_elements_$1_$2x$3.txt 2>&1
  cut -c1-230 $O/r5z_ab_unaligned_elements_$1_$2x$3.txt | tail -4
done
