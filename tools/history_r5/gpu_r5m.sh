#!/bin/bash
# Round 5, pass m: compressed exchange with the host read of step k behind the launch of step k + 1 (plan depth 2):
# collective tests, then bench.py --collective at world 1.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_collective.py tests/test_gpu_bench.py -m gpu -q 2>&1 | tail -5
python bench.py --collective --no-cpu-baseline --chunks 1 --steps 100 --warmup 10 2>/dev/null | grep "^{" > $O/r5m_bench_collective_world1.json
python -c "
import json; d=json.load(open('gpurun_out/r5m_bench_collective_world1.json')); print({k: d[k] for k in ('ms_plain','ms_compressed','ms_compressed_pipelined','speedup_vs_plain')})"
python bench.py --collective --no-cpu-baseline --chunks 1 --steps 100 --warmup 10 --workload fp16 2>/dev/null | grep "^{" > $O/r5m_bench_collective_world1_fp16.json
python -c "
import json; d=json.load(open('gpurun_out/r5m_bench_collective_world1_fp16.json')); print({k: d[k] for k in ('ms_plain','ms_compressed','ms_compressed_pipelined','speedup_vs_plain')})"
