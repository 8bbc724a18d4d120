#!/bin/bash
# Round 5, pass ab: from how much emptiness in the rectangle do the work lists pay?  tools/ragged_probe.py's batches of
# merely varying sizes under the rectangles (0), the lists forced (1) and the policy (default).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
for m in 0 1 default; do
  echo "## DGPU_WORK_LISTS=$m"
  if [ $m = default ]; then timeout 250 python tools/ragged_probe.py 2>&1 | grep "256 tensors"; else DGPU_WORK_LISTS=$m timeout 250 python tools/ragged_probe.py 2>&1 | grep "256 tensors"; fi
done > $O/r5ab_varying_sizes.txt
cat $O/r5ab_varying_sizes.txt
