#!/bin/bash
# per-kernel comparison of the fp16 and bf16 LDM steps (same box): which bf16-only fast path is slower than the general kernel fp16 falls back to?
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s15; rm -rf $OUT; mkdir -p $OUT
export EEGLDM_NO_SIDE_STREAM=1
for D in float16 bfloat16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$D -o k -- python tools/debug/quick_bench.py $D 256 768 6 > $OUT/$D.txt 2> $OUT/$D.log
  find $OUT/tr_$D -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$D.csv \;
  rm -rf $OUT/tr_$D
  grep -E "ms/step|fwd only" $OUT/$D.txt
done
