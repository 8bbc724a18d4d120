#!/bin/bash
# Round 6, pass m: the whole GPU suite on the tree (size classes, one-stream collective steps, RCCL helpers at world 1);
# the compressed all-gather at world 1 with one chunk and with four.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -8 ) > $O/r6m_pytest.txt
tail -4 $O/r6m_pytest.txt
for c in 1 4; do
  python bench.py --collective --no-cpu-baseline --chunks $c --steps 100 --warmup 10 > $O/r6m_bench_collective_world1_chunks$c.json 2>/dev/null
  python -c "
import json; d = json.loads([l for l in open('$O/r6m_bench_collective_world1_chunks$c.json') if l.startswith('{')][-1])
print('chunks $c', {k: v for k, v in d.items() if k.startswith('ms_') or k in ('value', 'unit')})"
done
python tools/collective_breakdown.py > $O/r6m_collective_breakdown_world1.txt 2>&1; grep -v "amdgpu.ids\|socket.cpp" $O/r6m_collective_breakdown_world1.txt | head -12
