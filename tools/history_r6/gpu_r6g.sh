#!/bin/bash
# Round 6, pass g: where the one-kernel compress wins: single tensors of 0.5 .. 32 Mi words, and small batches.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for mode in 0 1; do
  DGPU_FUSED=$mode python tools/small_call_probe.py --sizes 0.5,1,2,4,8,16 --reps 200 > $O/r6g_rates_b1_fused$mode.txt 2>/dev/null
  DGPU_FUSED=$mode python tools/small_call_probe.py --batch 4 --sizes 0.25,0.5,1,2,4 --reps 200 > $O/r6g_rates_b4_fused$mode.txt 2>/dev/null
  DGPU_FUSED=$mode python tools/small_call_probe.py --batch 16 --sizes 0.125,0.25,0.5,1 --reps 200 > $O/r6g_rates_b16_fused$mode.txt 2>/dev/null
done
python - <<'PY'
import json
for b in (1, 4, 16):
    rows = {}
    for mode in (0, 1):
        for l in open(f"gpurun_out/r6g_rates_b{b}_fused{mode}.txt"):
            d = json.loads(l)
            rows.setdefault(d["mega_floats"], {})[mode] = d["compress"]
    for m, r in rows.items():
        print(f"batch {b:2d} x {m:6.3f} Mi ({b * m * 32:6.0f} tiles): two kernels {r[0]['one_by_one_median_us']:7.2f} / {r[0]['back_to_back_us']:7.2f}   one kernel {r[1]['one_by_one_median_us']:7.2f} / {r[1]['back_to_back_us']:7.2f}  us (one by one / back to back)")
PY
