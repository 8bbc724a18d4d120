#!/bin/bash
# A/B harness: tools/ab.sh <reps> <workload> <variant> ...
#   variant = "base" | <lib relative to dietgpu_amd/lib/> , optionally followed by @ENV=VALUE[@ENV=VALUE...]
#   (tools/build_variant.py makes dietgpu_amd/lib/v_<name>.so).  Variants are interleaved rep by rep -- the same build
#   has placement-dependent modes between processes (docs/HISTORY.md section 3) -- the last lines give the medians.
#   Columns: cold = the headline loop (rotating buffer sets), warm = one buffer set; kernels in us (cold | warm).
REPS=$1; WL=$2; shift 2
rm -f /tmp/ab_*.txt
for rep in $(seq $REPS); do
for v in "$@"; do
  name=${v%%@*}; envs=""
  [ "$v" != "$name" ] && envs=$(echo "${v#*@}" | tr '@' ' ')
  lib=""; [ "$name" != "base" ] && lib=$PWD/dietgpu_amd/lib/$name
  env DGPU_LIB=$lib $envs python bench.py --quick --steps ${AB_STEPS:-100} --warmup 10 --no-cpu-baseline --workload $WL $AB_ARGS > /tmp/o.json 2>/tmp/e.txt || tail -3 /tmp/e.txt
  python -c "
import json; d=json.load(open('/tmp/o.json'))
k=lambda t: {n[2:]: v['avg_us'] for n,v in (t or {}).items()}
print('%-24s' % '$v', 'cold', d['ms_per_step'], 'warm', d['ms_per_step_one_buffer_set'], 'enc-only', d.get('ms_compress_only'), 'dec-only', d.get('ms_decompress_only'), k(d['kernels']), '|', k(d.get('kernels_one_buffer_set')))
open('/tmp/ab_$(echo $v | tr '/@=' '___').txt','a').write('%s %s\n' % (d['ms_per_step'], d['ms_per_step_one_buffer_set']))"
done; done
for v in "$@"; do python -c "
import statistics as s; x=[[float(t) for t in l.split()] for l in open('/tmp/ab_$(echo $v | tr '/@=' '___').txt')]
print('%-24s median cold %.4f ms  warm %.4f ms  (n=%d)' % ('$v', s.median(r[0] for r in x), s.median(r[1] for r in x), len(x)))"; done
