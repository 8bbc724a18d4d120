#!/bin/bash
# Round 6, pass d: the one-kernel float compress (k_float_compress_fused): parity tests, then fused against the
# two-kernel path (DGPU_FUSED=1 / 0) on the headline shape, few large tensors and small calls.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15 ) > $O/r6d_pytest_fused.txt
cat $O/r6d_pytest_fused.txt
for shape in "256 524288" "64 2097152" "16 8388608" "2048 65536"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 600 tools/ab.sh 2 bf16 base@DGPU_FUSED=0 base@DGPU_FUSED=1 > $O/r6d_ab_fused_bf16_$1x$2.txt 2>&1
  cut -c1-250 $O/r6d_ab_fused_bf16_$1x$2.txt | tail -6
done
for w in fp16 fp32; do
  AB_STEPS=50 timeout 600 tools/ab.sh 2 $w base@DGPU_FUSED=0 base@DGPU_FUSED=1 > $O/r6d_ab_fused_$w.txt 2>&1
  cut -c1-250 $O/r6d_ab_fused_$w.txt | tail -6
done
DGPU_FUSED=0 python tools/small_call_probe.py --sizes 1,4,16,64 --reps 200 > $O/r6d_small_call_rates_two_kernels.txt 2>/dev/null
DGPU_FUSED=1 python tools/small_call_probe.py --sizes 1,4,16,64 --reps 200 > $O/r6d_small_call_rates_fused.txt 2>/dev/null
python - <<'PY'
import json
for f in ("two_kernels", "fused"):
    for l in open(f"gpurun_out/r6d_small_call_rates_{f}.txt"):
        d = json.loads(l)
        print(f, d["mega_floats"], "compress one-by-one", d["compress"]["one_by_one_median_us"], "back-to-back", d["compress"]["back_to_back_us"])
PY
