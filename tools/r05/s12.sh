#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s12; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_conv_skinny.py tests/test_gpu_samplers.py -x -q > $OUT/t.log 2>&1; tail -5 $OUT/t.log
for SW in 0 1 0 1; do
  if [ $SW = 1 ]; then export EEGLDM_ATTN_NO_DEEP_RING=1; else unset EEGLDM_ATTN_NO_DEEP_RING; fi
  python tools/debug/b1_trace.py 1 2>&1 | tail -1 | sed "s/^/nodeep=$SW /"
done | tee $OUT/b1.log
unset EEGLDM_ATTN_NO_DEEP_RING
EEGLDM_ATTN_STAMPS=1 python tools/debug/b1_trace.py 1 2>&1 | grep attn_chain | sort | uniq -c | sort -rn | head -3
