#!/bin/bash
# Copies the outputs of tools/gpu_final.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked).
T=${1:-r06}
cd "$(dirname "$0")/.."
strip() { grep -a -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" "$1"; }
for f in gpurun_out/${T}_bench_*.json gpurun_out/${T}_reference_protocol.json; do
  [ -f "$f" ] && strip "$f" | grep -a "^{" | tail -1 > profiles/$(basename $f)
done
for f in api_rate graph_rate pytest valu_rate traffic_summary; do
  [ -f gpurun_out/${T}_$f.txt ] && strip gpurun_out/${T}_$f.txt > profiles/${T}_$f.txt
done
for w in bf16 u8 fp16 fp32; do  # (fp32 has bench lines only)
  [ -f gpurun_out/rocprof_${T}_$w.txt ] && strip gpurun_out/rocprof_${T}_$w.txt > profiles/${T}_rocprof_stats_$w.txt
  [ -f gpurun_out/rocprof_${T}one_$w.txt ] && strip gpurun_out/rocprof_${T}one_$w.txt > profiles/${T}_rocprof_stats_${w}_one_buffer_set.txt
  [ -f gpurun_out/pmc_${T}_$w.txt ] && strip gpurun_out/pmc_${T}_$w.txt > profiles/${T}_pmc_$w.txt
done
[ -f gpurun_out/timeline_${T}_bf16.txt ] && strip gpurun_out/timeline_${T}_bf16.txt > profiles/${T}_timeline_bf16.txt
[ -f gpurun_out/${T}_hbm_traffic.json ] && cp gpurun_out/${T}_hbm_traffic.json profiles/${T}_hbm_traffic.json
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/${T}_hbm_traffic.json"))
print("traffic file stamp", d.get("kernel_source_hash"), "tree", bench.kernel_source_hash())
PY
