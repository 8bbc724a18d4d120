#!/bin/bash
# round 4, pass d: (1) does a windowed compress pipeline pay? slices x streams x histogram load policy with the shipped
# kernels; (2) decoder archive loads non-temporal, judged on the cold loop; (3) the bench tests again
mkdir -p gpurun_out
python tools/slice_streams_experiment.py --steps 100 > gpurun_out/r4d_slice_streams.txt 2>&1
grep -v amdgpu gpurun_out/r4d_slice_streams.txt | head -30
AB_STEPS=100 bash tools/ab.sh 3 bf16 base v_ntdec.so > gpurun_out/r4d_ab_ntdec_bf16.txt 2>&1
AB_STEPS=50 bash tools/ab.sh 2 u8 base v_ntdec.so > gpurun_out/r4d_ab_ntdec_u8.txt 2>&1
AB_STEPS=50 bash tools/ab.sh 2 fp16 base v_ntdec.so > gpurun_out/r4d_ab_ntdec_fp16.txt 2>&1
tail -n 3 gpurun_out/r4d_ab_ntdec_*.txt
( timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r4d_pytest_bench.txt; tail -3 gpurun_out/r4d_pytest_bench.txt
