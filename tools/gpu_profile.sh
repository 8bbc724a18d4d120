#!/bin/bash
# rocprofv3 kernel-trace stats of THE HEADLINE LOOP ALONE (bench.py --timeline: the W warm-up + K timed steps over the
# rotating buffer sets and nothing else, so that roofline.frac can be recomputed from a file that holds no other loop);
# PROFILE_ARGS="--rotate 1" gives the one-buffer-set loop alone.  Writes text summaries under gpurun_out/ (copy the ones
# to keep into profiles/).  Usage: gpurun -- tools/gpu_profile.sh <tag> [workload]
TAG=${1:-run}; WL=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out /tmp/prof_$TAG
CMD="python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --timeline --workload $WL $PROFILE_ARGS"
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/stats -o bench -- $CMD > /tmp/prof_$TAG/bench.json 2> /tmp/prof_$TAG/err.log
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --timeline --workload $WL $PROFILE_ARGS"
  python $R/tools/rocpd_summary.py stats /tmp/prof_$TAG/stats/bench_results.db | grep -v "at::native\|rocclr\|elementwise"
  echo
  echo "# bench line printed by the same (profiled) command"
  cat /tmp/prof_$TAG/bench.json
} > $R/gpurun_out/rocprof_${TAG}_$WL.txt
cat $R/gpurun_out/rocprof_${TAG}_$WL.txt | head -12
