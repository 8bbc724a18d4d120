#!/bin/bash
# Round 5, pass ac: the encoder's work list element by element (descriptors and claim words for the listed tiles only):
# the list tests, the GPU suite, tools/ragged_probe.py under the policy, the rectangles (0) and the lists forced (1).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "widely_different or one_large_tensor" 2>&1 | tail -15 > $O/r5ac_pytest_lists.txt
tail -3 $O/r5ac_pytest_lists.txt
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5ac_pytest.txt
tail -3 $O/r5ac_pytest.txt
for m in default 0 1; do
  echo "## DGPU_WORK_LISTS=$m"
  if [ $m = default ]; then timeout 250 python tools/ragged_probe.py 2>&1 | grep -v amdgpu.ids; else DGPU_WORK_LISTS=$m timeout 250 python tools/ragged_probe.py 2>&1 | grep -v amdgpu.ids; fi
done > $O/r5ac_ragged_probe.txt
cat $O/r5ac_ragged_probe.txt
