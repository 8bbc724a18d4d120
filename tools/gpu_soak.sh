#!/bin/bash
# Soak of the paths with cross-workgroup hand-offs (look-back in one and two levels, absent workgroups, both dispatch
# forms, spill pools, work lists, size classes): the same tests six times over, four processes sharing the GPU.
# Usage: gpurun -- bash tools/gpu_soak.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -n 4 -p no:cacheprovider \
    -k "lookback or absent_workgroups or size_classes or widely_different or dispatch_modes or ragged or spill" 2>&1 | tail -2
done
