#!/bin/bash
# Round 4, pass p: conflict-free LUT build in k_ans_decode_pair (128 slots per step, four consecutive slots per lane).
# v_c3.so = the committed tree before it.  Then the re-stamp of the HBM traffic file for the new source hash.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/p_pytest.txt
grep -a "passed\|failed" gpurun_out/p_pytest.txt
for n in 32 4096; do
  AB_ARGS="--batch 32768 --elems $n" AB_STEPS=50 timeout 120 tools/ab.sh 2 bf16 v_c3.so base > gpurun_out/p_ab_bf16_32768x$n.txt 2>&1
  tail -6 gpurun_out/p_ab_bf16_32768x$n.txt
done
AB_ARGS="--batch 32768 --elems 4096" AB_STEPS=50 timeout 100 tools/ab.sh 1 fp16 v_c3.so base > gpurun_out/p_ab_fp16_32768x4096.txt 2>&1
tail -4 gpurun_out/p_ab_fp16_32768x4096.txt
T=r04
tools/gpu_pmc.sh $T bf16 > /dev/null 2>&1; tools/gpu_pmc.sh $T u8 > /dev/null 2>&1; tools/gpu_pmc.sh $T fp16 > /dev/null 2>&1
python tools/make_traffic_json.py $T bf16=gpurun_out/pmc_${T}_bf16.txt u8=gpurun_out/pmc_${T}_u8.txt fp16=gpurun_out/pmc_${T}_fp16.txt > gpurun_out/${T}_traffic_summary.txt 2>&1
cp profiles/${T}_hbm_traffic.json gpurun_out/${T}_hbm_traffic.json
python bench.py --no-cpu-baseline --steps 100 --warmup 20 --quick > gpurun_out/${T}_bench_bf16_after_pmc.json 2>/dev/null
python bench.py --quick --no-cpu-baseline --steps 100 --warmup 10 --batch 32768 --elems 4096 > gpurun_out/${T}_bench_bf16_32768x4096.json 2>/dev/null
python -c "
import json
for f in ('gpurun_out/r04_bench_bf16_after_pmc.json','gpurun_out/r04_bench_bf16_32768x4096.json'):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['step_frac_of_hbm_peak'], d['ms_per_step_one_buffer_set'], {k[2:]:v['avg_us'] for k,v in d['kernels'].items()}, d['roofline'].get('traffic'))"
cat gpurun_out/${T}_traffic_summary.txt | tail -9
