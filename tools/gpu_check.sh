#!/bin/bash
# The tree as the driver will see it, in one command a driver can run blind:
#   tools/gpu_check.sh [N]     N = GPUs of the node to use for the multi-GPU legs (default: all visible; 1 = skip them)
# smoke, the GPU suite, the default bench line, and -- with N > 1 -- the N-rank bench line (one rank per GPU over RCCL,
# weak scaling: BASELINE config 5 at N = 8) and the compressed all-gather over xGMI with its payload check.
N=${1:-$(python -c "import torch; print(torch.cuda.device_count())")}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|rror" | tail -5 ) | tee gpurun_out/check_pytest.txt
summ='
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print({k: d.get(k) for k in ("metric","value","unit","n_gpus","world_size_seen_by_backend","ms_per_step","per_rank_ms_per_step","ms_per_step_one_buffer_set","rccl_version","vs_baseline","dtype","speedup_vs_plain","ms_compressed","ms_plain","bit_exact")})
for k in ("roofline","roofline_by_direction","cpu_baseline","rank_binding","config"):
    if d.get(k): print(k, d[k])'
python bench.py 2>/dev/null | tee gpurun_out/check_bench.json | python -c "$summ"
if [ "$N" -gt 1 ]; then
  for n in 2 4 8; do
    [ "$n" -le "$N" ] || continue
    timeout 900 python bench.py --gpus $n --no-cpu-baseline 2>gpurun_out/check_bench_${n}gpu.err | tee gpurun_out/check_bench_${n}gpu.json | python -c "$summ"
  done
  timeout 900 python bench.py --gpus $N --collective --no-cpu-baseline --steps 50 --warmup 5 2>gpurun_out/check_collective_${N}gpu.err | tee gpurun_out/check_collective_${N}gpu.json | python -c "$summ"
fi
