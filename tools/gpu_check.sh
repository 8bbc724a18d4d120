#!/bin/bash
# One GPU-box pass: parity tests, then the bf16 and u8 bench lines.  Usage: gpurun -- tools/gpu_check.sh [tag]
TAG=${1:-run}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_bf16_$TAG.json 2> gpurun_out/bench_bf16_$TAG.err || tail -5 gpurun_out/bench_bf16_$TAG.err
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload u8 > gpurun_out/bench_u8_$TAG.json 2> gpurun_out/bench_u8_$TAG.err || tail -5 gpurun_out/bench_u8_$TAG.err
python - <<PY
import json
for w in ("bf16", "u8"):
    try:
        d = json.load(open(f"gpurun_out/bench_{w}_$TAG.json"))
        print(w, "value", d["value"], "GB/s  step", d["ms_per_step"], "ms  enc", d["encode_ms"], "dec", d["decode_ms"], " frac", d["step_frac_of_hbm_peak"])
        print("   ", {k: v["avg_us"] for k, v in d["kernels"].items()})
    except Exception as e:
        print(w, "FAILED", e)
PY
