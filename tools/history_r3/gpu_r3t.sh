#!/bin/bash
mkdir -p gpurun_out
for s in "1 134217728" "16 8388608" "64 2097152"; do set -- $s
  AB_ARGS="--rotate 1 --batch $1 --elems $2" AB_STEPS=100 bash tools/ab.sh 2 bf16 base v_sched0.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | sed "s/^/$1x$2 /"
done | tee gpurun_out/r3t_ab_schedule_small_batches.txt
