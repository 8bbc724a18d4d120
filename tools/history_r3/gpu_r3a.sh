#!/bin/bash
# Round 3, first GPU pass: parity of the new raw decoder (k_ans_decode_mt) and the encoder variants, then A/B on u8.
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3a_pytest.txt
tail -3 gpurun_out/r3a_pytest.txt
SUB="ans_ or config2 or lookback_windows_raw or decode_mt or worst or staging or fuzz or hip_archives"
for v in v_enc8.so v_encx.so v_enc8x.so; do
  echo "== $v" >> gpurun_out/r3a_pytest_variants.txt
  ( DGPU_LIB=$PWD/dietgpu_amd/lib/$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SUB" 2>&1 | tail -4 ) >> gpurun_out/r3a_pytest_variants.txt
done
cat gpurun_out/r3a_pytest_variants.txt
AB_STEPS=200 bash tools/ab.sh 2 u8 base@DGPU_DEC_MT=0 base v_enc8.so v_encx.so v_enc8x.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/r3a_ab_u8.txt
AB_STEPS=200 bash tools/ab.sh 1 bf16 base 2>&1 | tail -2 | tee gpurun_out/r3a_ab_bf16.txt
tools/gpu_pmc.sh r3a u8 > /dev/null 2>&1
grep -A30 "k_ans_decode" gpurun_out/pmc_r3a_u8.txt | head -70
