#!/bin/bash
# cache-policy A/B, third batch: cacheable histogram loads on the other workloads and shapes
mkdir -p gpurun_out
rm -f gpurun_out/r3g_rotating_ab.txt
for args in "--workload fp16" "--workload fp32" "--workload u8" "--batch 1 --elems 134217728" "--batch 16 --elems 8388608" "--batch 2048 --elems 65536" "--batch 8192 --elems 16384"; do
for v in base v_nth0.so; do
  lib=""; [ "$v" != "base" ] && lib=$PWD/dietgpu_amd/lib/$v
  DGPU_LIB=$lib python bench.py --no-cpu-baseline --steps 200 --warmup 30 $args > /tmp/o.json 2>/dev/null
  python - "$args $v" <<'PY' | tee -a gpurun_out/r3g_rotating_ab.txt
import json, sys
d = json.load(open("/tmp/o.json"))
print("%-44s step %.4f  rotating %.4f  warm kernels %s  rotating kernels %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_rotating"],
      {k[6:]: v["avg_us"] for k, v in d["kernels"].items()}, {k[6:]: v for k, v in d["kernels_rotating_avg_us"].items()}))
PY
done; done
