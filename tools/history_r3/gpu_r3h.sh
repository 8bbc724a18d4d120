#!/bin/bash
mkdir -p gpurun_out
for v in base v_nth1.so v_ntd0.so; do
  lib=""; [ "$v" != "base" ] && lib=$PWD/dietgpu_amd/lib/$v
  echo "== $v"; DGPU_LIB=$lib python tools/rotating_phases.py 2>/dev/null
done | tee gpurun_out/r3h_rotating_phases.txt
