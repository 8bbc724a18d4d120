#!/bin/bash
# round 4, pass b: new C-ABI tests (histogram_dev, split-size decode, info, hooks), graph/ADVICE fixes, ops.py routed
# through torch.ops, bench tests, the default bench line (cold headline + roofline_by_direction + pooled cpu_baseline)
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r4b_pytest.txt
tail -6 gpurun_out/r4b_pytest.txt
python bench.py > gpurun_out/r4b_bench_bf16.json 2> gpurun_out/r4b_bench_bf16.err; tail -2 gpurun_out/r4b_bench_bf16.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b_bench_bf16_driver_protocol.json 2>/dev/null
python tools/api_rate.py > gpurun_out/r4b_api_rate.txt 2>&1
tail -5 gpurun_out/r4b_api_rate.txt
