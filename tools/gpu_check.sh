#!/bin/bash
# Last look at the tree as the driver will see it: build check, smoke, the GPU suite, the default bench line.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|rror" | tail -5 ) | tee gpurun_out/check_pytest.txt
python bench.py 2>/dev/null | tee gpurun_out/check_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('metric','value','unit','ms_per_step','ms_per_step_no_preroll','ms_per_step_rotating','vs_baseline','dtype')}); print(d['roofline']); print(d['cpu_baseline'])"
