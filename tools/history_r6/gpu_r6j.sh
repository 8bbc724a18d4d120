#!/bin/bash
# Round 6, pass j: the histogram's atomic counters of one or two large elements in 16 / 8 sets (base) against one set;
# the look-back pause by chain length; the whole GPU suite on the tree.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
export DGPU_FUSED=0
for shape in "1 134217728" "1 16777216" "2 8388608"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 2 bf16 base v_acc1.so > $O/r6j_ab_hist_acc_sets_bf16_$1x$2.txt 2>&1
  cut -c1-220 $O/r6j_ab_hist_acc_sets_bf16_$1x$2.txt | tail -6
done
for shape in "1 134217728" "16 8388608" "256 524288"; do
  set -- $shape
  python bench.py --quick --steps 50 --warmup 10 --no-cpu-baseline --batch $1 --elems $2 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 x $2', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
done
unset DGPU_FUSED
( timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -8 ) > $O/r6j_pytest.txt
tail -4 $O/r6j_pytest.txt
