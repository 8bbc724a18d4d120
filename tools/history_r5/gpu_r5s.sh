#!/bin/bash
# Round 5, pass s: what a partial last tile costs: 256 elements of 524288 (whole tiles) / 525288 / 530000 / 540000 words.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for n in 524288 525288 530000 540672 541000; do
  python bench.py --quick --no-cpu-baseline --steps 100 --warmup 10 --elems $n > $O/r5s_bench_bf16_256x$n.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/r5s_bench_bf16_256x$n.json')); print($n, d['ms_per_step'], {k[2:]:v['avg_us'] for k,v in d['kernels'].items()}, d['ms_compress_only'], d['ms_decompress_only'])"
done
