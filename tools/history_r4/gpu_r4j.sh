#!/bin/bash
# Round 4, pass j: single-block batches -- k_stats_single (one wavefront counts + normalises an element) and the pair
# encoder's prefetch of the next pair's pdf table.  v_c1.so = commit 8f0bc92 (pair decoder / table-from-pdf already in).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "single_block or ragged or small or whole_block" 2>&1 | tail -6 > gpurun_out/j_pytest_focus.txt
tail -3 gpurun_out/j_pytest_focus.txt
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/j_pytest.txt
tail -3 gpurun_out/j_pytest.txt
AB_ARGS="--batch 32768 --elems 4096" timeout 300 tools/ab.sh 2 bf16 v_c1.so base > gpurun_out/j_ab_bf16_32768x4096.txt 2>&1
tail -6 gpurun_out/j_ab_bf16_32768x4096.txt
AB_ARGS="--batch 32768 --elems 4096" timeout 200 tools/ab.sh 1 fp32 v_c1.so base > gpurun_out/j_ab_fp32_32768x2048.txt 2>&1
tail -4 gpurun_out/j_ab_fp32_32768x2048.txt
timeout 120 python bench.py --quick --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/j_bench_bf16.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/j_bench_bf16.json')); print('headline', d['ms_per_step'], d['ms_per_step_one_buffer_set'], d['roofline']['frac'], d['roofline'].get('traffic'))"
