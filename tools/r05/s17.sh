#!/bin/bash
# fused 3-tap weight gradient: 32-deep stages in a 4-deep LDS-DMA ring (EEGLDM_WG3_DEEP=1) against the 64-deep double buffer
# NOTE: the switch this script toggles was taken out of the dispatch after the measurement; the numbers are in gpurun_out of that run and in DESIGN.
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s17; rm -rf $OUT; mkdir -p $OUT
EEGLDM_WG3_DEEP=1 timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_unet.py tests/test_gpu_fp16.py -x -q -k "conv or wgrad or weight or unet" > $OUT/t.log 2>&1; tail -3 $OUT/t.log
for i in 1 2; do
  python tools/debug/gemm_bench.py bf16 2>&1 | grep -E "TOTAL" | sed "s/^/base /" | tee -a $OUT/gb.log
  EEGLDM_WG3_DEEP=1 python tools/debug/gemm_bench.py bf16 2>&1 | grep -E "wgrad|TOTAL" | sed "s/^/deep /" | tee -a $OUT/gb.log
done
for i in 1 2; do
  python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/base /" | tee -a $OUT/qb.log
  EEGLDM_WG3_DEEP=1 python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/deep /" | tee -a $OUT/qb.log
done
