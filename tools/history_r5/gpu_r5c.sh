#!/bin/bash
# Round 5, pass c: (1) raw bytes: slices of the compress call on two streams, so that histogram(k+1) overlaps encode(k);
# (2) decoder grid in tile-major order (v_dectm.so: the order the encoder writes the archives in) against element-major.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 600 python tools/slice_streams_u8_experiment.py > $O/r5c_slice_streams_u8.txt 2>&1
tail -10 $O/r5c_slice_streams_u8.txt
AB_STEPS=100 timeout 300 tools/ab.sh 3 bf16 base v_dectm.so > $O/r5c_ab_decoder_tile_major_bf16.txt 2>&1
tail -3 $O/r5c_ab_decoder_tile_major_bf16.txt
AB_STEPS=50 timeout 300 tools/ab.sh 2 u8 base v_dectm.so > $O/r5c_ab_decoder_tile_major_u8.txt 2>&1
tail -3 $O/r5c_ab_decoder_tile_major_u8.txt
AB_ARGS="--batch 16 --elems 8388608" AB_STEPS=50 timeout 200 tools/ab.sh 2 bf16 base v_dectm.so > $O/r5c_ab_decoder_tile_major_bf16_16x8388608.txt 2>&1
tail -3 $O/r5c_ab_decoder_tile_major_bf16_16x8388608.txt
