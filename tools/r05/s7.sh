#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s7; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_aekl_primitives.py -x -q > $OUT/t_prim.log 2>&1; tail -12 $OUT/t_prim.log
timeout 900 python -m pytest tests/test_gpu_aekl.py -x -q -k "one_group or golden" > $OUT/t_twin.log 2>&1; tail -12 $OUT/t_twin.log
