#!/bin/bash
# Round 6, pass p: k_stats_single with 2 (base) / 1 / 4 elements per wavefront, all their loads in flight up front.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -n 4 -k "small_elements or single_block or size_classes or ragged or partial_last or word_alignment or unaligned or fuzz" 2>&1 | tail -4 ) > $O/r6p_pytest.txt
cat $O/r6p_pytest.txt
for shape in "32768 4096" "32768 4000" "65535 1000"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 3 bf16 base v_spw1.so v_spw4.so > $O/r6p_ab_stats_single_per_wave_bf16_$1x$2.txt 2>&1
  cut -c1-200 $O/r6p_ab_stats_single_per_wave_bf16_$1x$2.txt | head -3; tail -3 $O/r6p_ab_stats_single_per_wave_bf16_$1x$2.txt
done
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 600 tools/ab.sh 2 fp16 base v_spw1.so v_spw4.so > $O/r6p_ab_stats_single_per_wave_fp16_32768x4096.txt 2>&1
head -3 $O/r6p_ab_stats_single_per_wave_fp16_32768x4096.txt | cut -c1-200
