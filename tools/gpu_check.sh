#!/bin/bash
# The tree as the driver will see it, in one command a driver can run blind:
#   tools/gpu_check.sh [N]     N = GPUs of the node to use for the multi-GPU legs (default: all visible; 1 = skip them)
# smoke, the GPU suite, the default bench line, and -- with N > 1 -- the N-rank bench line (one rank per GPU over RCCL,
# weak scaling: BASELINE config 5 at N = 8) and the compressed all-gather over xGMI with its payload check.
# Every leg ends in ONE line "PASS <leg> ..." or "FAIL <leg> ..."; the exit status is the number of FAILs.
N=${1:-$(python -c "import torch; print(torch.cuda.device_count())")}
mkdir -p gpurun_out
FAILS=0
verdict() {  # verdict <leg> <status> <detail>
  if [ "$2" -eq 0 ]; then echo "PASS $1 $3"; else echo "FAIL $1 $3"; FAILS=$((FAILS + 1)); fi
}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check_smoke.txt 2>&1; verdict smoke $? "$(tail -1 gpurun_out/check_smoke.txt)"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/check_pytest.txt 2>&1; verdict gpu-suite $? "$(grep -a 'passed\|failed' gpurun_out/check_pytest.txt | tail -1)"
summ='
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print({k: d.get(k) for k in ("metric","value","unit","n_gpus","world_size_seen_by_backend","ms_per_step","per_rank_ms_per_step","ms_per_step_one_buffer_set","rccl_version","vs_baseline","dtype","speedup_vs_plain","ms_compressed","ms_plain","bit_exact")})
for k in ("roofline","roofline_by_direction","cpu_baseline","rank_binding","config"):
    if d.get(k): print(k, d[k])'
# a bench leg passes when it exits 0 and its line says what was asked for: n_gpus ranks seen by the backend, bit-exact
bench_leg() {  # bench_leg <leg> <expected ranks> <json file> <command...>
  leg=$1; want=$2; out=$3; shift 3
  "$@" 2> ${out%.json}.err > $out
  rc=$?
  python -c "$summ" < $out
  detail=$(python - $out $want <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    ok = d.get("n_gpus") == int(sys.argv[2]) and d.get("world_size_seen_by_backend", d.get("n_gpus")) == int(sys.argv[2]) and \
        (d.get("round_trip_bit_exact") or d.get("bit_exact"))
    print(("ok " if ok else "BAD ") + f"n_gpus={d.get('n_gpus')} value={d.get('value')} {d.get('unit')} ms={d.get('ms_per_step', d.get('ms_compressed'))}")
except Exception as e:  # noqa: BLE001
    print(f"BAD no JSON line ({e})")
P
)
  case "$detail" in ok*) ;; *) rc=1;; esac
  verdict "$leg" $rc "$detail"
}
bench_leg bench-1gpu 1 gpurun_out/check_bench.json python bench.py
if [ "$N" -gt 1 ]; then
  for n in 2 4 8; do
    [ "$n" -le "$N" ] || continue
    bench_leg bench-${n}gpu $n gpurun_out/check_bench_${n}gpu.json timeout 900 python bench.py --gpus $n --no-cpu-baseline
  done
  bench_leg collective-${N}gpu $N gpurun_out/check_collective_${N}gpu.json timeout 900 python bench.py --gpus $N --collective --no-cpu-baseline --steps 50 --warmup 5
else
  # one GPU: the N = 8 launch paths with eight gloo ranks on this device (what tests/test_gpu_bench.py also covers)
  DGPU_BENCH_ONE_DEVICE=1 bench_leg bench-8ranks-one-device 8 gpurun_out/check_bench_8ranks.json timeout 900 python bench.py --gpus 8 --dist-backend gloo --steps 4 --warmup 1 --batch 32 --rotate 2 --quick --no-cpu-baseline
  DGPU_BENCH_ONE_DEVICE=1 bench_leg collective-8ranks-one-device 8 gpurun_out/check_collective_8ranks.json timeout 900 python bench.py --gpus 8 --dist-backend gloo --collective --steps 3 --warmup 1 --batch 16 --no-cpu-baseline
fi
echo "$FAILS leg(s) failed"
exit $FAILS
