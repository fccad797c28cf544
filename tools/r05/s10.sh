#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s10; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fp16.py -q -s > $OUT/t_fp16.log 2>&1; tail -60 $OUT/t_fp16.log
