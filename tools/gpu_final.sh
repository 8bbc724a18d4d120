#!/bin/bash
# Evidence run of a round (GPU box): GPU suite, bench lines (all workloads and shapes), reference protocol, rocprofv3
# kernel stats of the headline loop alone and of the one-buffer-set loop alone, PMC passes (+ the traffic file stamped
# with the kernel source hash), timeline, microbenchmarks.
# Usage: gpurun --timeout 2700 -- bash tools/gpu_final.sh <tag>      (outputs under gpurun_out/; tools/collect_profiles.sh <tag> copies them)
T=${1:-r06}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -8 ) > gpurun_out/${T}_pytest.txt
tail -3 gpurun_out/${T}_pytest.txt
python bench.py > gpurun_out/${T}_bench_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${T}_bench_bf16_driver_protocol.json 2>/dev/null
for w in u8 fp16 fp32; do python bench.py --no-cpu-baseline --steps 200 --warmup 20 --workload $w > gpurun_out/${T}_bench_$w.json 2>/dev/null; done
for s in "8192 16384" "16384 8192" "32768 4096" "1 134217728" "16 8388608" "64 2097152" "2048 65536"; do set -- $s
  python bench.py --quick --no-cpu-baseline --steps 100 --warmup 10 --batch $1 --elems $2 > gpurun_out/${T}_bench_bf16_$1x$2.json 2>/dev/null; done
python bench.py --collective --no-cpu-baseline --chunks 1 --steps 100 --warmup 10 > gpurun_out/${T}_bench_collective_world1.json 2>/dev/null
python bench.py --reference-protocol > gpurun_out/${T}_reference_protocol.json 2>/dev/null
python tools/graph_rate.py > gpurun_out/${T}_graph_rate.txt 2>/dev/null
python tools/api_rate.py > gpurun_out/${T}_api_rate.txt 2>/dev/null
# rocprofv3 kernel stats: the headline (rotating) loop alone, and the one-buffer-set loop alone
for w in bf16 u8 fp16; do tools/gpu_profile.sh $T $w > /dev/null 2>&1; done
for w in bf16 u8; do PROFILE_ARGS="--rotate 1" tools/gpu_profile.sh ${T}one $w > /dev/null 2>&1; done
tools/gpu_pmc.sh $T bf16 > /dev/null 2>&1; tools/gpu_pmc.sh $T u8 > /dev/null 2>&1; tools/gpu_pmc.sh $T fp16 > /dev/null 2>&1
python tools/make_traffic_json.py $T bf16=gpurun_out/pmc_${T}_bf16.txt u8=gpurun_out/pmc_${T}_u8.txt fp16=gpurun_out/pmc_${T}_fp16.txt > gpurun_out/${T}_traffic_summary.txt 2>&1
cp profiles/${T}_hbm_traffic.json gpurun_out/${T}_hbm_traffic.json
tools/gpu_timeline.sh $T bf16 > /dev/null 2>&1
python bench.py --no-cpu-baseline --steps 100 --warmup 20 --quick > gpurun_out/${T}_bench_bf16_after_pmc.json 2>/dev/null
( cd tools/microbench && hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate valu_rate.hip 2>/dev/null && /tmp/valu_rate | head -9 ) > gpurun_out/${T}_valu_rate.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        if "ms_per_step" in d:
            k = lambda t: {n[2:]: v["avg_us"] for n, v in (t or {}).items()}
            rd = d.get("roofline_by_direction") or {}
            print(f.split("/")[-1], "cold", d["ms_per_step"], d.get("step_frac_of_hbm_peak"), "warm", d.get("ms_per_step_one_buffer_set"), "steady", d.get("ms_per_step_steady_state"),
                  "alone", d.get("ms_compress_only"), d.get("ms_decompress_only"), k(d.get("kernels")), "|", k(d.get("kernels_one_buffer_set")),
                  "roofline", (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), "traffic", (d.get("roofline") or {}).get("traffic"),
                  "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
        else:
            print(f.split("/")[-1], d.get("ms_compressed"), d.get("ms_plain"), d.get("config"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat gpurun_out/${T}_graph_rate.txt gpurun_out/${T}_api_rate.txt gpurun_out/${T}_traffic_summary.txt gpurun_out/${T}_valu_rate.txt | grep -v amdgpu
head -8 gpurun_out/rocprof_${T}_bf16.txt
