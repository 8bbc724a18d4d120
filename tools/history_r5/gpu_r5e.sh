#!/bin/bash
# Round 5, pass e: the tree with the decoder order policy, precision-aware package ops and the 8-rank one-device tests:
# whole GPU suite, then tools/gpu_check.sh (PASS / FAIL per leg), then where the world-1 collective's time goes.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -25 > $O/r5e_pytest.txt
tail -4 $O/r5e_pytest.txt
timeout 1500 bash tools/gpu_check.sh 1 > $O/r5e_gpu_check.txt 2>&1
grep -a "^PASS\|^FAIL\|leg(s)" $O/r5e_gpu_check.txt
timeout 300 python tools/collective_breakdown.py > $O/r5e_collective_breakdown_world1.txt 2>&1
grep -v "c10d" $O/r5e_collective_breakdown_world1.txt | tail -10
