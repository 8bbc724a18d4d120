#!/bin/bash
# cache-policy A/B, second batch (rotating-buffer step): archive stores non-temporal, decoder loads non-temporal
mkdir -p gpurun_out
rm -f gpurun_out/r3f_rotating_ab.txt
for rep in 1 2; do
for v in base v_nth0.so v_ntes1.so v_ntdl1.so v_ntes1dl1.so v_nth0dl1.so; do
  lib=""; [ "$v" != "base" ] && lib=$PWD/dietgpu_amd/lib/$v
  DGPU_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 30 > /tmp/o.json 2>/dev/null
  python - "$v" <<'PY' | tee -a gpurun_out/r3f_rotating_ab.txt
import json, sys
d = json.load(open("/tmp/o.json"))
print("%-14s step %.4f  rotating %.4f  warm kernels %s  rotating kernels %s" % (sys.argv[1], d["ms_per_step"], d["ms_per_step_rotating"],
      {k[6:]: v["avg_us"] for k, v in d["kernels"].items()}, {k[6:]: v for k, v in d["kernels_rotating_avg_us"].items()}))
PY
done; done
