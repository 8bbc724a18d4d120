#!/bin/bash
# Round 6, pass i: the pause between two polls of an unpublished look-back descriptor: doubling from 0.2 us (base), doubling
# from 0.85 us, fixed 3.4 us -- few large tensors, the headline shape, and small calls (two-kernel path throughout).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
export DGPU_FUSED=0
for shape in "1 134217728" "16 8388608" "256 524288"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 2 bf16 base v_from4.so v_fixed127.so > $O/r6i_ab_poll_pause_bf16_$1x$2.txt 2>&1
  tail -3 $O/r6i_ab_poll_pause_bf16_$1x$2.txt
done
for v in "" v_from4.so v_fixed127.so; do
  echo "== ${v:-base}"
  DGPU_LIB=${v:+$PWD/dietgpu_amd/lib/$v} python tools/small_call_probe.py --sizes 1,4,16,64 --reps 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['mega_floats'], 'compress', d['compress']['one_by_one_median_us'], d['compress']['back_to_back_us'])"
done > $O/r6i_small_calls_poll_pause.txt
cat $O/r6i_small_calls_poll_pause.txt
