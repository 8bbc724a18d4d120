#!/bin/bash
# Round 3, second GPU pass: decoder geometry variants (DGPU_DEC_MT = 0..3, unconditional word read) and the raw
# encoder at 4 workgroups per CU (1920-word stage + spill slots).
mkdir -p gpurun_out
rm -f gpurun_out/r3b_pytest_variants.txt
SUB="ans_ or config2 or lookback_windows_raw or decode_mt or worst or staging or fuzz or malformed or rejects or hip_archives"
run() { # lib env
  echo "== $1 $2" >> gpurun_out/r3b_pytest_variants.txt
  lib=""; [ "$1" != "base" ] && lib=$PWD/dietgpu_amd/lib/$1
  ( env DGPU_LIB=$lib $2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SUB" 2>&1 | tail -4 ) >> gpurun_out/r3b_pytest_variants.txt
}
run base DGPU_DEC_MT=2; run base DGPU_DEC_MT=3
run v_dmu.so DGPU_DEC_MT=1; run v_dmu.so DGPU_DEC_MT=2; run v_dmu.so DGPU_DEC_MT=3
run v_rs1920.so DGPU_DEC_MT=0; run v_rs1920x.so DGPU_DEC_MT=0
cat gpurun_out/r3b_pytest_variants.txt
AB_STEPS=200 bash tools/ab.sh 2 u8 base@DGPU_DEC_MT=0 base@DGPU_DEC_MT=1 base@DGPU_DEC_MT=2 base@DGPU_DEC_MT=3 \
  v_dmu.so@DGPU_DEC_MT=1 v_dmu.so@DGPU_DEC_MT=2 v_dmu.so@DGPU_DEC_MT=3 v_rs1920.so@DGPU_DEC_MT=0 v_rs1920x.so@DGPU_DEC_MT=0 2>&1 \
  | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/r3b_ab_u8.txt
