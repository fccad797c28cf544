#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s8; rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_zz_convergence.py -x -q -s -k "deterministic_mode or warm_up" > $OUT/t_new.log 2>&1; tail -25 $OUT/t_new.log
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/t_full.log 2>&1; tail -8 $OUT/t_full.log
