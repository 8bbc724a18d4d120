#!/bin/bash
# Round 4, pass i: batches of single-block elements -- the pair encoder builds its table from the pdf in the archive
# header (no [B][256] x 16-byte table through HBM), the pair decoder requests header + pdf + descriptor + states in one
# round trip and words + non-compressed bytes before the LUT build (whole-block staging).  v_head.so = the previous commit.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/i_pytest.txt
for wl in bf16 fp32; do  # (not u8: that workload ignores --elems and would generate 32768 x 1 MiB on the host)
  AB_ARGS="--batch 32768 --elems 4096" tools/ab.sh 3 $wl v_head.so base > gpurun_out/i_ab_${wl}_32768x4096.txt 2>&1
done
AB_ARGS="--batch 8192 --elems 16384" tools/ab.sh 2 bf16 v_head.so base > gpurun_out/i_ab_bf16_8192x16384.txt 2>&1
tail -4 gpurun_out/i_pytest.txt; tail -3 gpurun_out/i_ab_*.txt
