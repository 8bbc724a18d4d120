#!/bin/bash
mkdir -p gpurun_out
timeout 400 python tools/placement_probe.py bf16 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r3q_placement_probe.txt
