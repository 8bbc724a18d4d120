#!/bin/bash
# Round 6, pass b: the transient after load begins (copy kernel vs codec step); the driver's command under a kernel trace
# again with the steady-state window chosen correctly; the timeline of small calls; look-back polling back-off.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
python tools/transient_probe.py > $O/r6b_transient_probe.txt 2>/tmp/e0.txt || tail -3 /tmp/e0.txt
cat $O/r6b_transient_probe.txt
python tools/clock_sampler.py --seconds 300 -o /tmp/clk_trace.txt & S=$!
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt/out -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r6b_driver_cmd_traced.json 2>/tmp/e2.txt || tail -3 /tmp/e2.txt )
kill $S; wait $S 2>/dev/null
python tools/protocol_trace.py /tmp/pt/out/bench_results.db /tmp/clk_trace.txt 5 20 > $O/r6b_driver_protocol_trace.txt 2>&1
grep "^#" $O/r6b_driver_protocol_trace.txt | cut -c1-200 | head -12
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/sc/out -o sc -- python $R/tools/small_call_probe.py --sizes 1,16 --trace /tmp/sc_host.jsonl > /dev/null 2>/tmp/e4.txt || tail -5 /tmp/e4.txt )
ls /tmp/sc/out
python tools/small_call_probe.py --merge $(find /tmp/sc/out -name "*.db" | head -1) /tmp/sc_host.jsonl > $O/r6b_small_call_timeline.txt 2>&1
cat $O/r6b_small_call_timeline.txt | head -70
for shape in "1 134217728" "16 8388608"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 400 tools/ab.sh 2 bf16 base v_bo8.so v_bo32.so v_bo127.so > $O/r6b_ab_lookback_backoff_bf16_$1x$2.txt 2>&1
  cut -c1-230 $O/r6b_ab_lookback_backoff_bf16_$1x$2.txt | tail -4
done
