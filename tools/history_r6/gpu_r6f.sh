#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
R=$PWD
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p)
print({k: getattr(p, k) for k in dir(p) if "shared" in k.lower() or "multi" in k.lower()})
PY
( cd /tmp && DGPU_FUSED=1 rocprofv3 --kernel-trace -d /tmp/kt/out -o kt -- python $R/bench.py --steps 3 --warmup 1 --timeline --no-cpu-baseline > /dev/null 2>/tmp/e.txt || tail -3 /tmp/e.txt )
python tools/kernel_dispatch_info.py $(find /tmp/kt/out -name "*.db" | head -1)
