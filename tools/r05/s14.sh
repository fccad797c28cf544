#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s14; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_primitives.py tests/test_gpu_unet.py tests/test_gpu_fullsize_parity.py -x -q > $OUT/t.log 2>&1; tail -5 $OUT/t.log
python tools/debug/quick_bench.py float16 256 768 6 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/fp16 /" | tee $OUT/fp16.log
python tools/debug/quick_bench.py bfloat16 256 768 6 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/bf16 /" | tee -a $OUT/fp16.log
