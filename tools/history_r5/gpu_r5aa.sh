#!/bin/bash
# Round 5, pass aa: work lists for batches whose elements differ widely in size (capi.hip, RaggedPlan): the new parity
# tests, the GPU suite, then tools/ragged_probe.py with the lists (default) and with the rectangles (DGPU_WORK_LISTS=0).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "widely_different or one_large_tensor" 2>&1 | tail -15 > $O/r5aa_pytest_lists.txt
tail -6 $O/r5aa_pytest_lists.txt
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > $O/r5aa_pytest.txt
tail -3 $O/r5aa_pytest.txt
{ echo "## work lists (default policy)"; timeout 250 python tools/ragged_probe.py 2>&1 | grep -v amdgpu.ids
  echo "## rectangles (DGPU_WORK_LISTS=0)"; DGPU_WORK_LISTS=0 timeout 250 python tools/ragged_probe.py 2>&1 | grep -v amdgpu.ids; } > $O/r5aa_ragged_probe.txt
cat $O/r5aa_ragged_probe.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 bf16 v_rectangles.so base > $O/r5aa_ab_headline.txt 2>&1
cut -c1-230 $O/r5aa_ab_headline.txt | tail -3
