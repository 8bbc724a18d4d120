#!/bin/bash
# Round 5, pass a: GPU suite on the tree with the hardware-dispatched encoder (auto mode + forced modes), the
# reference's own Python tests (staged in oracle/_ref/), then the A/B batch:
#   encoder dispatch (persistent vs one workgroup per tile) on 4 shapes, v_r4.so = the round-4 tree;
#   single-block elements: pair encoder table build, k_stats_single bins (8/16/32 slots, 1 or 4 waves per workgroup,
#   hot symbol in a register); small tiles: the decoder's scan LUT build.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -15 > $O/r5a_pytest.txt
grep -a "passed\|failed\|error" $O/r5a_pytest.txt | tail -3
# parity of the variant builds on the tests that cover what they change
DGPU_LIB=$PWD/dietgpu_amd/lib/v_pairtab.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "ragged_batches_of_small or statistics_of_single or whole_block_elements or histogram_that_does_not" 2>&1 | tail -3 > $O/r5a_pytest_pairtab.txt
DGPU_LIB=$PWD/dietgpu_amd/lib/v_smalllut.so timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "small or ragged or fuzz or whole_block or staging or decode_ring" 2>&1 | tail -3 > $O/r5a_pytest_smalllut.txt
for v in stat16 stat16w1 stat8w1 stat32w1 stathot; do
  DGPU_LIB=$PWD/dietgpu_amd/lib/v_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 4 -k "ragged_batches_of_small or statistics_of_single" 2>&1 | tail -3 > $O/r5a_pytest_$v.txt
done
tail -n 1 $O/r5a_pytest_*.txt
# ---- A/B: encoder dispatch
AB_STEPS=100 timeout 300 tools/ab.sh 3 bf16 v_r4.so base@DGPU_ENC_DISPATCH=0 base@DGPU_ENC_DISPATCH=1 > $O/r5a_ab_dispatch_bf16.txt 2>&1
tail -4 $O/r5a_ab_dispatch_bf16.txt
AB_ARGS="--batch 16 --elems 8388608" AB_STEPS=50 timeout 200 tools/ab.sh 2 bf16 base@DGPU_ENC_DISPATCH=0 base@DGPU_ENC_DISPATCH=1 > $O/r5a_ab_dispatch_bf16_16x8388608.txt 2>&1
tail -3 $O/r5a_ab_dispatch_bf16_16x8388608.txt
AB_ARGS="--batch 1 --elems 134217728" AB_STEPS=50 timeout 200 tools/ab.sh 2 bf16 base@DGPU_ENC_DISPATCH=0 base@DGPU_ENC_DISPATCH=1 > $O/r5a_ab_dispatch_bf16_1x134217728.txt 2>&1
tail -3 $O/r5a_ab_dispatch_bf16_1x134217728.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 u8 base@DGPU_ENC_DISPATCH=0 base@DGPU_ENC_DISPATCH=1 > $O/r5a_ab_dispatch_u8.txt 2>&1
tail -3 $O/r5a_ab_dispatch_u8.txt
AB_STEPS=50 timeout 200 tools/ab.sh 2 fp16 base@DGPU_ENC_DISPATCH=0 base@DGPU_ENC_DISPATCH=1 > $O/r5a_ab_dispatch_fp16.txt 2>&1
tail -3 $O/r5a_ab_dispatch_fp16.txt
# ---- A/B: batches of single-block elements
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 500 tools/ab.sh 2 bf16 base v_pairtab.so v_stat16.so v_stat16w1.so v_stat8w1.so v_stat32w1.so v_stathot.so > $O/r5a_ab_single_block_bf16_32768x4096.txt 2>&1
tail -8 $O/r5a_ab_single_block_bf16_32768x4096.txt
# ---- A/B: small tiles, the decoder's LUT build
for shape in "8192 16384" "16384 8192"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 200 tools/ab.sh 2 bf16 base v_smalllut.so > $O/r5a_ab_smalllut_bf16_$1x$2.txt 2>&1
  tail -3 $O/r5a_ab_smalllut_bf16_$1x$2.txt
done
