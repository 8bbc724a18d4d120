#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/r3r_mode_probe2.txt
for i in 1 2; do timeout 300 python tools/mode_probe2.py bf16 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/r3r_mode_probe2.txt; done
