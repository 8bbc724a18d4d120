#!/bin/bash
# Round 3, third GPU pass: the full GPU suite on both tensor surfaces (+ the fused / decode_mt builds on their own
# tests), the new bench fields, the reference-protocol sweep, the collective plan.
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3c_pytest.txt
tail -4 gpurun_out/r3c_pytest.txt
( DGPU_LIB=$PWD/dietgpu_amd/lib/v_fused.so timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r3c_pytest_fused.txt
( DGPU_LIB=$PWD/dietgpu_amd/lib/v_mt.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decode_mt or ans_ or config2 or fuzz or staging" 2>&1 | tail -3 ) > gpurun_out/r3c_pytest_mt.txt
cat gpurun_out/r3c_pytest_fused.txt gpurun_out/r3c_pytest_mt.txt
python bench.py --no-cpu-baseline > gpurun_out/r3c_bench_bf16.json 2> gpurun_out/r3c_bench_bf16.err; tail -c 3000 gpurun_out/r3c_bench_bf16.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3c_bench_bf16_driver.json 2>/dev/null
python bench.py --no-cpu-baseline --workload u8 > gpurun_out/r3c_bench_u8.json 2>/dev/null
python bench.py --reference-protocol > gpurun_out/r3c_reference_protocol.json 2> gpurun_out/r3c_reference_protocol.err; tail -c 2500 gpurun_out/r3c_reference_protocol.json; tail -3 gpurun_out/r3c_reference_protocol.err
for c in 1 4; do python bench.py --collective --no-cpu-baseline --chunks $c > gpurun_out/r3c_collective_bf16_c$c.json 2>gpurun_out/r3c_collective.err; python bench.py --collective --no-cpu-baseline --workload fp16 --chunks $c > gpurun_out/r3c_collective_fp16_c$c.json 2>>gpurun_out/r3c_collective.err; done
cat gpurun_out/r3c_collective_*.json; tail -3 gpurun_out/r3c_collective.err
python - <<'PY'
import json
for f in ("r3c_bench_bf16", "r3c_bench_bf16_driver", "r3c_bench_u8"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["ms_per_step"], "no_preroll", d["ms_per_step_no_preroll"], "rotating", d["ms_per_step_rotating"], d["rotating_sets"], d["rotating_footprint_bytes"],
              "frac", d["step_frac_of_hbm_peak"], d["step_frac_of_hbm_peak_rotating"], {k[6:]: v["avg_us"] for k, v in d["kernels"].items()}, d["roofline"]["traffic"])
    except Exception as e:
        print(f, "unreadable", e)
PY
