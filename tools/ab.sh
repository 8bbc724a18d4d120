#!/bin/bash
# A/B harness: tools/ab.sh <reps> <workload> <lib-or-"base"> ...   (libs relative to dietgpu_amd/lib/)
REPS=$1; WL=$2; shift 2
for rep in $(seq $REPS); do
for v in "$@"; do
  lib=""; [ "$v" != "base" ] && lib=$PWD/dietgpu_amd/lib/$v
  DGPU_LIB=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --workload $WL > /tmp/o.json 2>/tmp/e.txt || tail -3 /tmp/e.txt
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('%-14s' % '$v', d['ms_per_step'], 'enc', d['encode_ms'], 'dec', d['decode_ms'], {k[6:]: v['avg_us'] for k,v in d['kernels'].items()})"
done; done
