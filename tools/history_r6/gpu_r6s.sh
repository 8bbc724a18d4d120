#!/bin/bash
# Round 6, pass s: the wide stage for bf16 / fp32 batches of elements of <= 32 tiles as shipped (base) against none (v_nowide);
# the GPU suite.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3 ) > $O/r6s_pytest.txt; tail -1 $O/r6s_pytest.txt
for shape in "256 524288" "2048 65536" "64 1048576"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=60 timeout 900 tools/ab.sh 4 bf16 base v_nowide.so > $O/r6s_ab_wide_stage_bf16_$1x$2.txt 2>&1
  grep -o "^[a-z_0-9.]*so\|^base\|'ans_encode': [0-9.]*" $O/r6s_ab_wide_stage_bf16_$1x$2.txt | paste - - - | head -4; tail -2 $O/r6s_ab_wide_stage_bf16_$1x$2.txt
done
AB_STEPS=60 timeout 900 tools/ab.sh 4 fp32 base v_nowide.so > $O/r6s_ab_wide_stage_fp32.txt 2>&1; tail -2 $O/r6s_ab_wide_stage_fp32.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('driver protocol', d['ms_per_step'], d['ms_per_step_steady_state'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
