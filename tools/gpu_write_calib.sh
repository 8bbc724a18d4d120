#!/bin/bash
# WRITE_SIZE calibration for the decoder's store shapes (tools/microbench/write_calib.hip).
# Usage: gpurun -- tools/gpu_write_calib.sh <tag>
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/write_calib $R/tools/microbench/write_calib.hip || exit 1
{
  echo "# /tmp/write_calib (every kernel writes 256 MiB = 262144 KiB)"
  /tmp/write_calib
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/wc_$c
    rocprofv3 --pmc $c -d /tmp/wc_$c -o wc -- /tmp/write_calib > /dev/null 2>&1
  done
  echo "# rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes), mean per dispatch, KiB"
  python $R/tools/rocpd_summary.py pmcrows --only=wc_ /tmp/wc_*/wc_results.db
} > $R/gpurun_out/write_calib_$TAG.txt 2>&1
cat $R/gpurun_out/write_calib_$TAG.txt
