#!/bin/bash
# Round 3, fifth GPU pass: cache policy of the three kernels judged on the ROTATING-buffer step (cold inputs,
# archives and outputs), and the collective after the statistics fusion.
mkdir -p gpurun_out
rm -f gpurun_out/r3e_rotating_ab.txt
for rep in 1 2; do
for v in base v_ntd0.so v_nth0.so v_nte0.so v_ntall0.so; do
  lib=""; [ "$v" != "base" ] && lib=$PWD/dietgpu_amd/lib/$v
  DGPU_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 30 > /tmp/o.json 2>/dev/null
  python - "$v" <<'PY' | tee -a gpurun_out/r3e_rotating_ab.txt
import json, sys
d = json.load(open("/tmp/o.json"))
print("%-14s step %.4f  rotating %.4f  warm kernels %s  rotating kernels %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_rotating"],
      {k[6:]: v["avg_us"] for k, v in d["kernels"].items()}, {k[6:]: v for k, v in d["kernels_rotating_avg_us"].items()}))
PY
done; done
for w in fp16 fp32; do for v in base v_ntd0.so; do
  lib=""; [ "$v" != "base" ] && lib=$PWD/dietgpu_amd/lib/$v
  DGPU_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 30 --workload $w > /tmp/o.json 2>/dev/null
  python - "$w $v" <<'PY' | tee -a gpurun_out/r3e_rotating_ab.txt
import json, sys
d = json.load(open("/tmp/o.json"))
print("%-14s step %.4f  rotating %.4f  warm kernels %s  rotating kernels %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_rotating"],
      {k[6:]: v["avg_us"] for k, v in d["kernels"].items()}, {k[6:]: v for k, v in d["kernels_rotating_avg_us"].items()}))
PY
done; done
python tools/collective_breakdown.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tee gpurun_out/r3e_collective_breakdown.txt
