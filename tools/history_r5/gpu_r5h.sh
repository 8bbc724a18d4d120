#!/bin/bash
# Round 5, pass h: k_ans_encode_pair launched one workgroup per pair (v_pairhw.so; its spill slots then come from an
# oversized temp region: experiment only) against the persistent grid; kernel trace of a 1 Mi-float compress + decompress.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 300 tools/ab.sh 3 bf16 base v_pairhw.so > $O/r5h_ab_pair_encoder_hw_dispatch_bf16.txt 2>&1
cut -c1-250 $O/r5h_ab_pair_encoder_hw_dispatch_bf16.txt | tail -8
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 300 tools/ab.sh 2 fp16 base v_pairhw.so > $O/r5h_ab_pair_encoder_hw_dispatch_fp16.txt 2>&1
cut -c1-250 $O/r5h_ab_pair_encoder_hw_dispatch_fp16.txt | tail -6
R=$PWD
cd /tmp && rm -rf /tmp/small && rocprofv3 --kernel-trace --stats -d /tmp/small -o t -- python $R/bench.py --reference-protocol --ref-sizes 1 > /tmp/small_bench.json 2>/dev/null
python $R/tools/rocpd_summary.py stats /tmp/small/t_results.db 2>/dev/null | grep -v "at::native\|rocclr\|elementwise" | head -12 > $R/$O/r5h_small_call_kernel_stats.txt
cat /tmp/small_bench.json | tail -1 >> $R/$O/r5h_small_call_kernel_stats.txt
cat $R/$O/r5h_small_call_kernel_stats.txt
