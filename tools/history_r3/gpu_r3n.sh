#!/bin/bash
mkdir -p gpurun_out
( DGPU_LIB=$PWD/dietgpu_amd/lib/v_encx2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not one_gi" 2>&1 | grep -v amdgpu.ids | tail -60 ) | tee gpurun_out/r3n_pytest.txt
