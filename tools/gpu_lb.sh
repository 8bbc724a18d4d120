#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q -k "lookback or large_single or absent or baseline_config3" 2>&1 | tail -4 ) | tee gpurun_out/lb_pytest.txt
for s in "1 134217728" "16 8388608" "256 524288"; do set -- $s
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=60 bash tools/ab.sh 2 bf16 v_lb1.so libdietgpu_amd.so v_lb8.so 2>&1 | grep "median" | sed "s/^/$1x$2 /"; done | tee gpurun_out/lb_shapes.txt
export DGPU_PT_LIB=$PWD/dietgpu_amd/lib/dbg_phase.so
DGPU_PT_BATCH=1 DGPU_PT_ELEMS=134217728 python tools/phase_timing.py bf16 2>&1 | grep "lookback\|rows\|total\|per-WG span" | tee gpurun_out/lb_phase.txt
