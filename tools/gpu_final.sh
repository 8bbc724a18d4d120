# Final evidence run of a round (GPU box): bench lines, rocprof kernel stats, PMC passes, timeline.
# Usage: gpurun --timeout 1500 -- bash tools/gpu_final.sh <tag>      (outputs under gpurun_out/, copy into profiles/)
T=${1:-r02e}
mkdir -p gpurun_out
python bench.py > gpurun_out/${T}_bench_bf16.json 2>/dev/null
for w in u8 fp16 fp32; do python bench.py --no-cpu-baseline --workload $w > gpurun_out/${T}_bench_$w.json 2>/dev/null; done
for s in "8192 16384" "16384 8192" "32768 4096" "1 134217728"; do set -- $s
  python bench.py --no-cpu-baseline --batch $1 --elems $2 > gpurun_out/${T}_bench_bf16_$1x$2.json 2>/dev/null; done
python bench.py --collective --no-cpu-baseline > gpurun_out/${T}_bench_collective_world1.json 2>/dev/null
python tools/graph_rate.py > gpurun_out/${T}_graph_rate.txt 2>/dev/null
for w in bf16 u8 fp16 fp32; do tools/gpu_profile.sh $T $w > /dev/null 2>&1; done
tools/gpu_pmc.sh $T bf16 > /dev/null 2>&1; tools/gpu_pmc.sh $T u8 > /dev/null 2>&1; tools/gpu_pmc.sh $T fp16 > /dev/null 2>&1
PMC_ARGS="--batch 32768 --elems 4096" tools/gpu_pmc.sh ${T}_32768x4096 bf16 > /dev/null 2>&1
tools/gpu_timeline.sh $T bf16 > /dev/null 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${T}_bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d.get("step_frac_of_hbm_peak"), d.get("round_trip_bit_exact", d.get("bit_exact")),
              {k[6:]: v["avg_us"] for k, v in d.get("kernels", {}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
cat gpurun_out/${T}_graph_rate.txt
head -8 gpurun_out/rocprof_${T}_bf16.txt
