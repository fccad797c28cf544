#!/bin/bash
# fused 3-tap weight gradient: shifted tap fragments by VALU (default build) vs three transposed reads (tools/debug/libeegldm_ab.so)
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s16; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fp16.py -x -q -k "conv or wgrad or weight" > $OUT/t.log 2>&1; tail -3 $OUT/t.log
for i in 1 2; do
  python tools/debug/gemm_bench.py bf16 2>&1 | grep -E "wgrad|TOTAL" | sed "s/^/new /" | tee -a $OUT/gb.log
  EEGLDM_LIB=tools/debug/libeegldm_ab.so python tools/debug/gemm_bench.py bf16 2>&1 | grep -E "TOTAL" | sed "s/^/old /" | tee -a $OUT/gb.log
done
for i in 1 2; do
  python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/new /" | tee -a $OUT/qb.log
  EEGLDM_LIB=tools/debug/libeegldm_ab.so python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/old /" | tee -a $OUT/qb.log
done
