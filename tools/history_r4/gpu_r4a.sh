#!/bin/bash
# round 4, pass a: the stripped kernels (no compile-time knobs, exec saved/restored, overrun -> failed element):
# GPU suite (archives must be unchanged: golden hashes + oracle), A/B of the exec save/restore against plain s_mov.
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r4a_pytest.txt
tail -4 gpurun_out/r4a_pytest.txt
AB_STEPS=200 bash tools/ab.sh 3 bf16 base v_smov.so > gpurun_out/r4a_ab_saveexec_bf16.txt 2>&1
AB_STEPS=200 bash tools/ab.sh 3 u8 base v_smov.so > gpurun_out/r4a_ab_saveexec_u8.txt 2>&1
AB_STEPS=100 bash tools/ab.sh 2 fp16 base v_smov.so > gpurun_out/r4a_ab_saveexec_fp16.txt 2>&1
tail -2 gpurun_out/r4a_ab_saveexec_*.txt
