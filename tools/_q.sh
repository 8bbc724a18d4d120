mkdir -p gpurun_out
python bench.py > gpurun_out/r02c_bench_bf16.json 2>/dev/null
for w in u8 fp16 fp32; do python bench.py --no-cpu-baseline --workload $w > gpurun_out/r02c_bench_$w.json 2>/dev/null; done
DGPU_FUSED=1 python bench.py --no-cpu-baseline > gpurun_out/r02c_bench_bf16_fused.json 2>/dev/null
for w in bf16 u8 fp16 fp32; do tools/gpu_profile.sh r02c $w > /dev/null 2>&1; done
DGPU_FUSED=1 tools/gpu_profile.sh r02c_fused bf16 > /dev/null 2>&1
tools/gpu_timeline.sh r02c bf16 > /dev/null 2>&1
for shape in "1 134217728" "16 8388608" "2048 65536" "8192 16384" "32768 4096"; do
  set -- $shape
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --batch $1 --elems $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shape $shape', d['ms_per_step'], 'enc', d['encode_ms'], 'dec', d['decode_ms'], d['value'])"
done
python - <<'PY'
import json
for w in ("bf16","u8","fp16","fp32","bf16_fused"):
    d=json.load(open(f"gpurun_out/r02c_bench_{w}.json"))
    print(w, d["ms_per_step"], d["value"], d["step_frac_of_hbm_peak"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], {k[6:]:v["avg_us"] for k,v in d["kernels"].items()}, d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("single_thread",{}).get("value"))
PY
head -6 gpurun_out/rocprof_r02c_bf16.txt
