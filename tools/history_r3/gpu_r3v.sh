#!/bin/bash
# ticket order of the encoder: W elements x all tiles per group (DGPU_ENC_GROUP_ELEMS); W = B (0) is the shipped order
mkdir -p gpurun_out
( DGPU_ENC_GROUP_ELEMS=32 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not one_gi" 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee gpurun_out/r3v_pytest_group32.txt
for w in bf16 u8; do AB_ARGS="--rotate 1" AB_STEPS=150 bash tools/ab.sh 3 $w base@DGPU_ENC_GROUP_ELEMS=0 base@DGPU_ENC_GROUP_ELEMS=128 base@DGPU_ENC_GROUP_ELEMS=64 base@DGPU_ENC_GROUP_ELEMS=32 base@DGPU_ENC_GROUP_ELEMS=16 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"; done | tee gpurun_out/r3v_ab_group_elems.txt
