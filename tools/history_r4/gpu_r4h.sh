#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_cabi.py -m gpu -q -x -k lanes 2>&1 | tail -3 ) > gpurun_out/r4h_pytest_lanes.txt; tail -2 gpurun_out/r4h_pytest_lanes.txt
python bench.py --quick --steps 60 --warmup 10 --no-cpu-baseline --workload u8 > gpurun_out/r4h_u8.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r4h_u8.json')); print(d['ms_per_step'], d['ms_per_step_one_buffer_set'], d['ms_decompress_only'], d['kernels'], d['kernels_one_buffer_set'])"
