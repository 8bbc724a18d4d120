mkdir -p gpurun_out
python bench.py > gpurun_out/r02d_bench_bf16.json 2>/dev/null
for w in u8 fp16 fp32; do python bench.py --no-cpu-baseline --workload $w > gpurun_out/r02d_bench_$w.json 2>/dev/null; done
DGPU_FUSED=1 python bench.py --no-cpu-baseline > gpurun_out/r02d_bench_bf16_fused.json 2>/dev/null
for w in bf16 u8 fp16 fp32; do tools/gpu_profile.sh r02d $w > /dev/null 2>&1; done
DGPU_FUSED=1 tools/gpu_profile.sh r02d_fused bf16 > /dev/null 2>&1
tools/gpu_pmc.sh r02d bf16 > /dev/null 2>&1; tools/gpu_pmc.sh r02d u8 > /dev/null 2>&1; DGPU_FUSED=1 tools/gpu_pmc.sh r02d_fused bf16 > /dev/null 2>&1
tools/gpu_timeline.sh r02d bf16 > /dev/null 2>&1
python - <<'PY'
import json
for w in ("bf16","u8","fp16","fp32","bf16_fused"):
    d=json.load(open(f"gpurun_out/r02d_bench_{w}.json"))
    print(w, d["ms_per_step"], d["value"], d["step_frac_of_hbm_peak"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], {k[6:]:v["avg_us"] for k,v in d["kernels"].items()}, d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("single_thread",{}).get("value"))
PY
head -5 gpurun_out/rocprof_r02d_bf16.txt
