#!/bin/bash
# Round 5, pass y: what elements that are not 16-byte aligned cost (rows of a [B, n] matrix with n * 2 bytes not a multiple of
# 16: every row but each eighth starts between vector boundaries): 256 x 530000 (aligned) / 530004 (8-byte) / 530001 (2-byte).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for w in "bf16 256 530000" "bf16 256 530004" "bf16 256 530001" "bf16 32768 4001" "fp32 256 530002"; do
  set -- $w
  python bench.py --quick --no-cpu-baseline --workload $1 --steps 50 --warmup 10 --batch $2 --elems $3 > $O/r5y_bench_$1_$2x$3.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/r5y_bench_$1_$2x$3.json')); print('$1', $2, $3, d['ms_per_step'], {k[2:]:v['avg_us'] for k,v in d['kernels'].items()}, d['ms_compress_only'], d['ms_decompress_only'])"
done
