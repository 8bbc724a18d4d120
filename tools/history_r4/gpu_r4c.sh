#!/bin/bash
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r4c_pytest.txt
tail -8 gpurun_out/r4c_pytest.txt
python tools/cpu_scaling_probe.py > gpurun_out/r4c_cpu_scaling_probe.txt 2>&1
cat gpurun_out/r4c_cpu_scaling_probe.txt
