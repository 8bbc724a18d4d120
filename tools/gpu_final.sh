#!/bin/bash
# Final evidence run of a round (GPU box): the GPU suite, bench lines, rocprof kernel stats, PMC passes (+ the traffic
# file stamped with the kernel source hash), timeline.
# Usage: gpurun --timeout 2400 -- bash tools/gpu_final.sh <tag>      (outputs under gpurun_out/, copy into profiles/)
T=${1:-r03}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/${T}_pytest.txt
tail -3 gpurun_out/${T}_pytest.txt
python bench.py > gpurun_out/${T}_bench_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${T}_bench_bf16_driver_protocol.json 2>/dev/null
for w in u8 fp16 fp32; do python bench.py --no-cpu-baseline --workload $w > gpurun_out/${T}_bench_$w.json 2>/dev/null; done
for s in "8192 16384" "16384 8192" "32768 4096" "1 134217728" "16 8388608" "2048 65536"; do set -- $s
  python bench.py --no-cpu-baseline --batch $1 --elems $2 > gpurun_out/${T}_bench_bf16_$1x$2.json 2>/dev/null; done
python bench.py --collective --no-cpu-baseline --chunks 1 > gpurun_out/${T}_bench_collective_world1.json 2>/dev/null
python bench.py --collective --no-cpu-baseline --chunks 1 --workload fp16 > gpurun_out/${T}_bench_collective_world1_fp16.json 2>/dev/null
python bench.py --reference-protocol > gpurun_out/${T}_reference_protocol.json 2>/dev/null
python tools/rotating_phases.py > gpurun_out/${T}_rotating_phases.txt 2>/dev/null
python tools/graph_rate.py > gpurun_out/${T}_graph_rate.txt 2>/dev/null
python tools/api_rate.py > gpurun_out/${T}_api_rate.txt 2>/dev/null
for w in bf16 u8 fp16 fp32; do tools/gpu_profile.sh $T $w > /dev/null 2>&1; done
# ... and of the one-buffer-set loop alone (what `roofline` in the bench line measures)
for w in bf16 u8; do PROFILE_ARGS="--rotate 1" tools/gpu_profile.sh ${T}one $w > /dev/null 2>&1; done
tools/gpu_pmc.sh $T bf16 > /dev/null 2>&1; tools/gpu_pmc.sh $T u8 > /dev/null 2>&1; tools/gpu_pmc.sh $T fp16 > /dev/null 2>&1
python tools/make_traffic_json.py $T bf16=gpurun_out/pmc_${T}_bf16.txt u8=gpurun_out/pmc_${T}_u8.txt fp16=gpurun_out/pmc_${T}_fp16.txt > gpurun_out/${T}_traffic_summary.txt 2>&1
cp profiles/${T}_hbm_traffic.json gpurun_out/${T}_hbm_traffic.json
tools/gpu_timeline.sh $T bf16 > /dev/null 2>&1
python bench.py --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/${T}_bench_bf16_after_pmc.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    try:
        d = json.load(open(f))
        if "ms_per_step" in d:
            print(f.split("/")[-1], d["ms_per_step"], "no_preroll", d.get("ms_per_step_no_preroll"), "rotating", d.get("ms_per_step_rotating"), d["value"], d.get("step_frac_of_hbm_peak"),
                  d.get("step_frac_of_hbm_peak_rotating"), d.get("round_trip_bit_exact"), {k[6:]: v["avg_us"] for k, v in d.get("kernels", {}).items()}, d.get("kernels_rotating_avg_us"),
                  "one-direction", d.get("ms_compress_only_rotating"), d.get("ms_decompress_only_rotating"), "cached-hist", d.get("ms_per_step_rotating_cached_histogram_loads"),
                  "traffic", (d.get("roofline") or {}).get("traffic"))
        else:
            print(f.split("/")[-1], d.get("ms_compressed"), d.get("ms_plain"), d.get("config"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat gpurun_out/${T}_graph_rate.txt gpurun_out/${T}_api_rate.txt gpurun_out/${T}_traffic_summary.txt
head -8 gpurun_out/rocprof_${T}_bf16.txt
