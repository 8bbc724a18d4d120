#!/bin/bash
# Round 3, fourth GPU pass: where the collective's time goes at world 1, its bench lines, and the rotating-buffer
# per-kernel figures.
mkdir -p gpurun_out
python tools/collective_breakdown.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tee gpurun_out/r3d_collective_breakdown.txt
for c in 1 2; do python bench.py --collective --no-cpu-baseline --chunks $c 2>/dev/null | tee gpurun_out/r3d_collective_bf16_c$c.json; done
python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r3d_bench_bf16.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3d_bench_bf16.json"))
print("bf16", d["ms_per_step"], "no_preroll", d["ms_per_step_no_preroll"], "rotating", d["ms_per_step_rotating"], d["kernels_rotating_avg_us"], {k[6:]: v["avg_us"] for k, v in d["kernels"].items()})
PY
( timeout 600 python -m pytest tests/test_gpu_collective.py -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/r3d_pytest_collective.txt
