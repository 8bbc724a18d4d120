#!/bin/bash
# PMC passes for our kernels on the GPU box (separate --pmc runs, no trace domains).
# Usage: gpurun -- tools/gpu_pmc.sh <tag> [workload]     (PMC_ARGS="--batch 32768 --elems 4096" for other shapes)
TAG=${1:-run}; WL=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out /tmp/pmc_$TAG
BENCH="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --timeline --workload $WL $PMC_ARGS"
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_$TAG/p$i -o bench -- $BENCH > /dev/null 2>&1
done
python $R/tools/rocpd_summary.py pmcrows /tmp/pmc_$TAG/p*/bench_results.db > $R/gpurun_out/pmc_${TAG}_$WL.txt 2>&1
cat $R/gpurun_out/pmc_${TAG}_$WL.txt
