#!/bin/bash
# Occupancy scaling of the 16-block decoder: extra dynamic LDS per workgroup = fewer workgroups per CU.
# raw bytes: 40 KiB -> 4 / 3 / 2 workgroups per CU; bf16: 48 KiB -> 3 / 2.
mkdir -p gpurun_out
run() { DGPU_DEC_LDS_PAD=$2 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --workload $1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 pad $2: step', d['ms_per_step'], 'dec', d['decode_ms'], {k[6:]: v['avg_us'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do run u8 0; run u8 8192; run u8 24576; run bf16 0; run bf16 16384; run fp32 0; run fp32 8192; run fp32 24576; done 2>&1 | tee gpurun_out/occupancy_scaling.txt
