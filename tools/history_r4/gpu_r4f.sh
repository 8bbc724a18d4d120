#!/bin/bash
# round 4, pass f: cache policy of the non-compressed plane alone (decoder loads / encoder stores non-temporal) and a
# balanced persistent encoder grid (tickets spread evenly over the rounds), on the cold loop
mkdir -p gpurun_out
AB_STEPS=100 bash tools/ab.sh 3 bf16 base v_ncld.so v_ncst.so v_ncboth.so v_balgrid.so > gpurun_out/r4f_ab_bf16.txt 2>&1
AB_STEPS=60 bash tools/ab.sh 2 fp16 base v_ncld.so v_balgrid.so > gpurun_out/r4f_ab_fp16.txt 2>&1
AB_STEPS=60 bash tools/ab.sh 2 u8 base v_balgrid.so > gpurun_out/r4f_ab_u8.txt 2>&1
for f in gpurun_out/r4f_ab_*.txt; do echo $f; grep -v amdgpu $f | cut -c1-250; done
