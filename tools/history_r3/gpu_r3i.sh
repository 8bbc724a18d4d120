#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram_load_policy or float_batch or baseline_config3" 2>&1 | tail -3 ) | tee gpurun_out/r3i_pytest.txt
python tools/rotating_phases.py 2>/dev/null | tee gpurun_out/r3i_rotating_phases.txt
python bench.py --no-cpu-baseline > gpurun_out/r3i_bench_bf16.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3i_bench_bf16.json"))
for k in ("ms_per_step", "ms_per_step_no_preroll", "ms_per_step_rotating", "ms_compress_only_rotating", "ms_decompress_only_rotating",
          "ms_per_step_rotating_cached_histogram_loads", "ms_compress_only_rotating_cached_histogram_loads", "kernels_rotating_avg_us"):
    print(k, d.get(k))
PY
