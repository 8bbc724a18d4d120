#!/bin/bash
# Round 5, pass b: the tree after pass a (raw-byte encoder dispatched by the hardware, pair encoder table build, small-tile
# LUT build): whole GPU suite (incl. the reference's own Python tests from oracle/_ref/reference_python_tests), smoke,
# the driver's bench command with the compact lines of BASELINE configs 2 and 4.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -25 > $O/r5b_pytest.txt
tail -5 $O/r5b_pytest.txt
timeout 600 python -m pytest tests/test_reference_python_tests.py -m gpu -q -rs 2>&1 | tail -8 > $O/r5b_pytest_reference_python_tests.txt
cat $O/r5b_pytest_reference_python_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5b_bench_driver_protocol.json 2> $O/r5b_bench_driver_protocol.err ) 2>&1 | grep real
python - <<'P'
import json
d = json.load(open('gpurun_out/r5b_bench_driver_protocol.json'))
print(d['ms_per_step'], d['step_frac_of_hbm_peak'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
for k, v in d.get('other_configs', {}).items():
    print(k, v['ms_per_step'], v['step_frac_of_hbm_peak'], v['dominant_kernel'], v['dominant_kernel_frac'], v['kernels_us'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P
