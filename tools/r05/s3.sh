#!/bin/bash
set -u
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r05_s3; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gn_qstats.py tests/test_gpu_comm_fake.py -x -q > $OUT/t_q.log 2>&1; tail -15 $OUT/t_q.log
timeout 1200 python -m pytest tests/test_gpu_gemm_big.py tests/test_gpu_unet.py tests/test_gpu_fullsize_parity.py tests/test_gpu_sampling.py tests/test_gpu_samplers.py -x -q > $OUT/t_more.log 2>&1; tail -8 $OUT/t_more.log
for SW in 0 1 0 1; do
  if [ $SW = 1 ]; then export EEGLDM_GN_NO_QSTATS=1; else unset EEGLDM_GN_NO_QSTATS; fi
  python tools/debug/quick_bench.py bfloat16 256 768 8 2>&1 | grep -E "ms/step|fwd only" | sed "s/^/noq=$SW /"
done | tee $OUT/ab.log
unset EEGLDM_GN_NO_QSTATS
export EEGLDM_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_ldm -o ldm -- python tools/debug/quick_bench.py bfloat16 256 768 3 > $OUT/quick.log 2>&1
F=$(find $OUT/trace_ldm -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py "$F" adam_kernel > $OUT/seq_ldm.txt 2>&1; head -40 $OUT/seq_ldm.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_aekl -o aekl -- python tools/debug/aekl_bench.py 256 bfloat16 > $OUT/aekl.log 2>&1
F=$(find $OUT/trace_aekl -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py "$F" thin_fwd_kernel --seq > $OUT/seq_aekl.txt 2>&1
rm -rf $OUT/trace_ldm $OUT/trace_aekl
