#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s2; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm_big.py -x -q -k "skip" > $OUT/t_skip.log 2>&1; tail -15 $OUT/t_skip.log
timeout 900 python -m pytest tests/test_gpu_gemm_big.py tests/test_gpu_unet.py tests/test_gpu_fullsize_parity.py tests/test_gpu_comm_fake.py -x -q > $OUT/t_more.log 2>&1; tail -8 $OUT/t_more.log
for SW in 0 1 0 1; do
  if [ $SW = 1 ]; then export EEGLDM_NO_FUSED_SKIP=1; else unset EEGLDM_NO_FUSED_SKIP; fi
  python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/nofuse=$SW /"
done | tee $OUT/ab.log
unset EEGLDM_NO_FUSED_SKIP
python tools/debug/step_shapes.py 2>&1 | head -45 > $OUT/shapes.txt; head -30 $OUT/shapes.txt
