#!/bin/bash
# Round 4, pass n: fixed cost per ELEMENT of the single-block kernels -- 32768 elements of 32 / 256 / 1024 / 4096 bf16
# words (the first three take the predicated partial-block path, one to 32 rows; only 4096 takes the straight-line one).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
for n in 32 256 1024 4096; do
  timeout 120 python bench.py --quick --no-cpu-baseline --steps 50 --warmup 10 --batch 32768 --elems $n > /tmp/o.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/o.json'))
print('32768 x %5d bf16: cold step %.4f ms, one buffer set %.4f; kernels (cold) %s' % ($n, d['ms_per_step'], d['ms_per_step_one_buffer_set'], {k[2:]: v['avg_us'] for k, v in d['kernels'].items()}))"
done | tee gpurun_out/n_single_block_fixed_cost.txt
