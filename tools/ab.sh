#!/bin/bash
# A/B harness: tools/ab.sh <reps> <workload> <variant> ...
#   variant = "base" | <lib relative to dietgpu_amd/lib/> , optionally followed by @ENV=VALUE[@ENV=VALUE...]
#   (e.g. base@DGPU_DEC_MT=0  v_enc8.so).  Variants are interleaved rep by rep; the last lines give the median.
REPS=$1; WL=$2; shift 2
rm -f /tmp/ab_*.txt
for rep in $(seq $REPS); do
for v in "$@"; do
  name=${v%%@*}; envs=""
  [ "$v" != "$name" ] && envs=$(echo "${v#*@}" | tr '@' ' ')
  lib=""; [ "$name" != "base" ] && lib=$PWD/dietgpu_amd/lib/$name
  env DGPU_LIB=$lib $envs python bench.py --steps ${AB_STEPS:-40} --warmup 5 --no-cpu-baseline --workload $WL $AB_ARGS > /tmp/o.json 2>/tmp/e.txt || tail -3 /tmp/e.txt
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('%-28s' % '$v', d['ms_per_step'], 'enc', d['encode_ms'], 'dec', d['decode_ms'], {k[6:]: v['avg_us'] for k,v in d['kernels'].items()})
open('/tmp/ab_$(echo $v | tr '/@=' '___').txt','a').write(str(d['ms_per_step'])+'\n')"
done; done
for v in "$@"; do python -c "
import statistics as s; x=[float(l) for l in open('/tmp/ab_$(echo $v | tr '/@=' '___').txt')]; print('%-28s median step %.4f ms  min %.4f  (n=%d)' % ('$v', s.median(x), min(x), len(x)))"; done
