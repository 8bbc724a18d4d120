#!/bin/bash
# Kernel + memory-copy timeline of the last bench steps (gaps between dependent launches).
# Usage: gpurun -- tools/gpu_timeline.sh <tag> [workload]
TAG=${1:-run}; WL=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out /tmp/tl_$TAG
CMD="python $R/bench.py --steps 10 --warmup 2 --timeline --workload $WL"
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_$TAG/out -o bench -- $CMD > /tmp/tl_$TAG/bench.json 2> /tmp/tl_$TAG/err.log
python $R/tools/rocpd_summary.py timeline /tmp/tl_$TAG/out/bench_results.db 36 > $R/gpurun_out/timeline_${TAG}_$WL.txt
tail -3 /tmp/tl_$TAG/err.log
cat /tmp/tl_$TAG/bench.json | head -c 300; echo
cat $R/gpurun_out/timeline_${TAG}_$WL.txt
