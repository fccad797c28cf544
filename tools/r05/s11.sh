#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s11; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_conv_skinny.py tests/test_gpu_samplers.py tests/test_gpu_sampling.py -x -q -s > $OUT/t.log 2>&1; tail -12 $OUT/t.log
for SW in 0 1 0 1; do
  if [ $SW = 1 ]; then export EEGLDM_NO_FUSED_SKIP=1; else unset EEGLDM_NO_FUSED_SKIP; fi
  python tools/debug/b1_trace.py 1 2>&1 | tail -1 | sed "s/^/nofuse=$SW /"
done | tee $OUT/b1.log
unset EEGLDM_NO_FUSED_SKIP
python tools/debug/quick_bench.py float16 256 768 5 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/fp16 /" | tee $OUT/fp16.log
python tools/debug/quick_bench.py bfloat16 256 768 5 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/bf16 /" | tee -a $OUT/fp16.log
EEGLDM_NO_GEMM_BIG=1 EEGLDM_NO_CONV_WS=1 EEGLDM_NO_FUSED_ATTENTION=1 EEGLDM_GN_NO_PIPE=1 python tools/debug/quick_bench.py bfloat16 256 768 5 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/bf16-general-kernels /" | tee -a $OUT/fp16.log
