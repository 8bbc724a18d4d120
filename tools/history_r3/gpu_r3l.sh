#!/bin/bash
# encoder ablations (timing only; archives are wrong): which component is the u8 / bf16 encode sensitive to
mkdir -p gpurun_out; out=gpurun_out/r3l_enc_ablation.txt; : > $out
for rep in 1 2; do
for v in "" late0 abl1 abl2 abl3 abl4 abl5 abl6; do
  for wl in u8 bf16; do
    if [ -z "$v" ]; then timeout 120 python tools/enc_ablation.py $wl >> $out 2>&1
    else DGPU_LIB=$PWD/dietgpu_amd/lib/v_$v.so timeout 120 python tools/enc_ablation.py $wl >> $out 2>&1; fi
  done
done; done
cat $out | grep -v -i rccl
