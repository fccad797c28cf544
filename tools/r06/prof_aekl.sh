#!/bin/bash
# round 6: kernel table of the AEKL / GAN step (B = 256, bf16) -> gpurun_out/r06_prof_aekl/<tag>.txt
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp EEGLDM_PROFILE_ROUND=r06 EEGLDM_NO_SIDE_STREAM=1
TAG=${1:-a}
OUT=gpurun_out/r06_prof_aekl; mkdir -p $OUT; rm -rf $OUT/trace_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -o aekl -- python tools/debug/aekl_bench.py 256 bfloat16 > $OUT/aekl_$TAG.txt 2> $OUT/trace_$TAG.log
python - <<PY
import sys, os
sys.path.insert(0, "tools")
import pmc_traffic as P
P.kernel_table("$OUT/trace_$TAG", "$OUT/kernels_$TAG.txt", "# EEGLDM_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -- python tools/debug/aekl_bench.py 256 bfloat16   (8 steps)")
PY
rm -rf $OUT/trace_$TAG
head -50 $OUT/kernels_$TAG.txt
