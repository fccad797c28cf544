#!/bin/bash
# transposed-read swizzle for 128-byte tile rows (no 2-way bank conflict between rows r and r+8): default build vs the previous one (tools/debug/libeegldm_ab.so)
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s19; rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_unet.py tests/test_gpu_fp16.py tests/test_gpu_fullsize_parity.py -x -q > $OUT/t.log 2>&1; tail -3 $OUT/t.log
for i in 1 2; do
  python tools/debug/gemm_bench.py bf16 2>&1 | grep -E "wgrad|TOTAL" | sed "s/^/new /" | tee -a $OUT/gb.log
  EEGLDM_LIB=tools/debug/libeegldm_ab.so python tools/debug/gemm_bench.py bf16 2>&1 | grep -E "TOTAL" | sed "s/^/old /" | tee -a $OUT/gb.log
done
for i in 1 2 3; do
  python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/new /" | tee -a $OUT/qb.log
  EEGLDM_LIB=tools/debug/libeegldm_ab.so python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/old /" | tee -a $OUT/qb.log
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/g -o pmc -- python tools/debug/gemm_bench.py bf16 one > $OUT/g.log 2>&1
find $OUT/g -name "*counter_collection.csv" -exec cp {} $OUT/cc.csv \; ; rm -rf $OUT/g
