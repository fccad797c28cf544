#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s5; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gn_qstats.py tests/test_gpu_concurrency.py -x -q > $OUT/t_q.log 2>&1; tail -5 $OUT/t_q.log
EEGLDM_DBG_GROUPS=1 python tools/debug/quick_bench.py bfloat16 256 768 1 2>&1 | grep "wgrad group" | sort | uniq -c > $OUT/groups_default.txt; cat $OUT/groups_default.txt
for CB in 0.004 0.02 0.05 0.1 0.2; do for CF in 8 16; do
  EEGLDM_WGRAD_COST_BLOCK=$CB EEGLDM_WGRAD_COST_FIXED=$CF python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step" | sed "s/^/cb=$CB cf=$CF /"
done; done | tee $OUT/sweep.log
