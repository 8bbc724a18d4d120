#!/bin/bash
# Round 6, pass a (exploration): what the box exposes of its clocks; the driver's bench command under a kernel trace with
# the clocks sampled beside it; where a 2-128 MiB call's time goes; look-back windows of 256 / 512 predecessors per round
# trip; the float encoder at 5 / 4 / 3 workgroups per CU.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out
python tools/clock_sampler.py --probe > $O/r6a_clock_probe.txt 2>&1
( which amd-smi rocm-smi; timeout 20 rocm-smi --showclocks 2>&1 | head -30 ) >> $O/r6a_clock_probe.txt 2>&1
head -c 3000 $O/r6a_clock_probe.txt

# --- the driver's command, plain (bench line + clock log), then under rocprofv3 --kernel-trace with the clock log
python tools/clock_sampler.py --seconds 300 -o /tmp/clk_plain.txt & S=$!
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r6a_driver_cmd_plain.json 2>/tmp/e1.txt || tail -3 /tmp/e1.txt
kill $S; wait $S 2>/dev/null
python tools/clock_sampler.py --seconds 300 -o /tmp/clk_trace.txt & S=$!
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/pt/out -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r6a_driver_cmd_traced.json 2>/tmp/e2.txt || tail -3 /tmp/e2.txt )
kill $S; wait $S 2>/dev/null
python tools/protocol_trace.py /tmp/pt/out/bench_results.db /tmp/clk_trace.txt 5 20 > $O/r6a_driver_protocol_trace.txt 2>&1
python - <<'PY' >> gpurun_out/r6a_driver_protocol_trace.txt
import json
for f in ("plain", "traced"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r6a_driver_cmd_{f}.json") if l.startswith("{")][-1])
        print(f"# bench.py --gpus 1 --steps 20 --warmup 5 ({f}): ms_per_step {d['ms_per_step']}  steady state {d.get('ms_per_step_steady_state')}  "
              f"one buffer set {d.get('ms_per_step_one_buffer_set')}  kernels {({k[2:]: v['avg_us'] for k, v in d['kernels'].items()})}")
    except Exception as e:
        print("#", f, "unreadable", e)
PY
cp /tmp/clk_plain.txt $O/r6a_clock_log_plain.txt; cp /tmp/clk_trace.txt $O/r6a_clock_log_traced.txt
tail -12 $O/r6a_driver_protocol_trace.txt | cut -c1-220

# --- small calls
python tools/small_call_probe.py --sizes 1,4,16,64 --reps 200 > $O/r6a_small_call_rates.txt 2>/tmp/e3.txt || tail -3 /tmp/e3.txt
( cd /tmp && rocprofv3 --kernel-trace -d /tmp/sc/out -o sc -- python $R/tools/small_call_probe.py --sizes 1,16 --trace /tmp/sc_host.jsonl > /dev/null 2>/tmp/e4.txt || tail -3 /tmp/e4.txt )
python tools/small_call_probe.py --merge /tmp/sc/out/sc_results.db /tmp/sc_host.jsonl > $O/r6a_small_call_timeline.txt 2>&1
cat $O/r6a_small_call_rates.txt | cut -c1-400; cat $O/r6a_small_call_timeline.txt | head -60

# --- look-back windows
for shape in "1 134217728" "16 8388608" "256 524288"; do
  set -- $shape
  AB_ARGS="--batch $1 --elems $2" AB_STEPS=50 timeout 400 tools/ab.sh 2 bf16 base v_lb4.so v_lb8.so > $O/r6a_ab_lookback_window_bf16_$1x$2.txt 2>&1
  cut -c1-230 $O/r6a_ab_lookback_window_bf16_$1x$2.txt | tail -9
done
# --- float encoder residency series (LDS padded to 5 / 4 / 3 workgroups per CU)
AB_STEPS=50 timeout 400 tools/ab.sh 2 bf16 base v_occ5.so v_occ4.so v_occ3.so > $O/r6a_ab_encoder_residency_bf16.txt 2>&1
cut -c1-230 $O/r6a_ab_encoder_residency_bf16.txt | tail -12
