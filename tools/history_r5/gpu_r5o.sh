#!/bin/bash
# Round 5, pass o: whole GPU suite once more (refreshes the evidence run's pytest file), PMC passes of the shapes whose kernels
# changed this round (single-block pairs, small tiles, few large elements) and the digest over all of them.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -8 ) > $O/r05_pytest.txt
tail -2 $O/r05_pytest.txt
PMC_ARGS="--batch 32768 --elems 4096" tools/gpu_pmc.sh r05_32768x4096 bf16 > /dev/null 2>&1
PMC_ARGS="--batch 16384 --elems 8192" tools/gpu_pmc.sh r05_16384x8192 bf16 > /dev/null 2>&1
PMC_ARGS="--batch 16 --elems 8388608" tools/gpu_pmc.sh r05_16x8388608 bf16 > /dev/null 2>&1
ls $O | grep pmc_r05
python tools/pmc_digest.py profiles/r05_pmc_bf16.txt profiles/r05_pmc_fp16.txt profiles/r05_pmc_u8.txt $O/pmc_r05_32768x4096_bf16.txt $O/pmc_r05_16384x8192_bf16.txt $O/pmc_r05_16x8388608_bf16.txt > $O/r05_pmc_digest.txt 2>&1
cat $O/r05_pmc_digest.txt
