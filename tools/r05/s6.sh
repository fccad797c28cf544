#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s6; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_unet.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py -x -q > $OUT/t.log 2>&1; tail -8 $OUT/t.log
for SW in 0 1 0 1; do
  if [ $SW = 1 ]; then export EEGLDM_ATTN_NO_FUSED_DV=1; else unset EEGLDM_ATTN_NO_FUSED_DV; fi
  python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/nodv=$SW /"
done | tee $OUT/ab.log
