#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r3p_mode_probe.txt; : > $out
for i in 1 2 3 4 5; do timeout 200 python tools/mode_probe.py bf16 4 2>&1 | grep "^pid" >> $out; done
cat $out
