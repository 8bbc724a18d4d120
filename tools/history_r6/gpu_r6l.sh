#!/bin/bash
# Round 6, pass l: size classes inside one batch: the new parity test, the tests that go through the work lists, the probe.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "size_classes or widely_different or one_large_tensor or ragged" 2>&1 | tail -12 ) > $O/r6l_pytest_classes.txt
cat $O/r6l_pytest_classes.txt
python tools/ragged_probe.py --classes > $O/r6l_size_classes_probe.txt 2>/tmp/e.txt || tail -5 /tmp/e.txt
python tools/ragged_probe.py --classes --raw >> $O/r6l_size_classes_probe.txt 2>/tmp/e.txt || tail -5 /tmp/e.txt
cat $O/r6l_size_classes_probe.txt
